#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_options.py -x -q -m gpu -k "rng or random or dropout or drop or train_step or noise or option" > gpurun_out/r5_small5_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r5_small5_tests.log
[ $rc -ne 0 ] && exit $rc
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_cfg1.txt
grep "k_fill_random\|step span" gpurun_out/r5_small_trace_cfg1.txt | cut -c1-100
