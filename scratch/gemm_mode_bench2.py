import sys, os
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py")).read().split("import os\nprint")[0])
for mode in (0, 2):
    run(1, 6400, 3072, 1024, mode=mode)
