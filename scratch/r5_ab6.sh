#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_golden.py -x -q -m gpu -k "decoder or golden or train_step or predict" > gpurun_out/r5_ab6_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_ab6_tests.log
[ $rc -ne 0 ] && exit $rc
bash scratch/r5_ab_prev.sh
