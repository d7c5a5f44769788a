#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_small6_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r5_small6_tests.log
[ $rc -ne 0 ] && exit $rc
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_cfg1.txt
tail -1 gpurun_out/r5_small_trace_cfg1.txt
PROF_ARGS="--model es_en_20h --no-also --no-alt-precisions" bash scratch/trace_step.sh && tail -1 gpurun_out/trace_step.txt
bash scratch/r5_ab_prev.sh
