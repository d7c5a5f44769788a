#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_ab5_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_ab5_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for t in prev new; do
    D=$R
    [ $t = prev ] && D=$R/scratch/prev_tree
    cd $D
    timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$t rep $rep cfg1', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], 'es', [a.get('ms_per_step') for a in d.get('also',[])])"
  done
done
cd $R
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_cfg1.txt
tail -1 gpurun_out/r5_small_trace_cfg1.txt
