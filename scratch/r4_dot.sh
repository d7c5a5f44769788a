#!/bin/bash
# dot2 form of the bf16x3 split: accuracy (GEMM tests on the variant library) and speed (A/B in one call)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ASTK_LIB_PATH=$PWD/scratch/libastk_dot.so timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm and bf16x3 or heavy_tailed" > gpurun_out/r4_dot_tests.log 2>&1 || { tail -n 30 gpurun_out/r4_dot_tests.log; exit 1; }
tail -n 2 gpurun_out/r4_dot_tests.log
bash scratch/ab_lib.sh scratch/libastk_dot.so --no-also
