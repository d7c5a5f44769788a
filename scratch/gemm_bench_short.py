import sys, os
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py")).read().split("import os\nprint")[0])
run(0, 4096, 4096, 4096); run(0, 6400, 1024, 3072); run(1, 6400, 3072, 1024); run(2, 1024, 3072, 6400, mode=2, ks=2)
