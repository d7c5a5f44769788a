import sys, os
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py")).read().split("import os\nprint")[0])
run(0, 4096, 4096, 4096, iters=30)
