#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/r5_ab2_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_ab2_tests.log
[ $rc -ne 0 ] && exit $rc
bash scratch/r5_ab_prev.sh
bash scratch/r5_ab_table.sh | tail -18
