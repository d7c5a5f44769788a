#!/bin/bash
# same-box A/B inside ONE gpurun call: ASTK_SIDE_STREAM=0 (everything in line) against 1 (side-stream work beside the recurrences)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${1:-r6_ab_side}.txt
: > $OUT
cd $R
for rep in 1 2 3; do
  for side in 0 1; do
    ASTK_SIDE_STREAM=$side timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions ${BENCH_ARGS} 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('side $side rep $rep cfg1', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], 'es_en_20h', [a.get('ms_per_step') for a in d.get('also',[])], d['paths'].get('side_stream'), d['paths'].get('free_cus_beside_recurrences'))" >> $OUT || exit 1
  done
done
cat $OUT
