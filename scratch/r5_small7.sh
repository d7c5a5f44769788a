#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_small7_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_small7_tests.log
[ $rc -ne 0 ] && exit $rc
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_cfg1.txt
sed -n 22,40p gpurun_out/r5_small_trace_cfg1.txt | cut -c1-110; tail -1 gpurun_out/r5_small_trace_cfg1.txt
for i in 1 2 3; do python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('cfg1', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], 'es', [a.get('ms_per_step') for a in d.get('also',[])])"; done
