#!/bin/bash
# same-box A/B of tuning-knob sets inside ONE gpurun call.  usage: r6_ab_knobs.sh OUTNAME "name1:k=v,k=v" "name2:..." ...   (ASTK_SIDE_STREAM via env SIDE=0/1 per set: "name:SIDE=0")
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/$1.txt; shift
: > $OUT
cd $R
for rep in 1 2; do
  for set in "$@"; do
    name=${set%%:*}; knobs=${set#*:}
    ASTK_BENCH_KNOBS="$knobs" timeout -k 10 300 python3 scratch/bench_knobs.py --steps 40 --warmup 10 --no-cpu-baseline --no-alt-precisions --histogram none ${BENCH_ARGS} 2>gpurun_out/ab_err.log | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d.get('kernels',{})
print('$name rep $rep cfg1', d['ms_per_step'], 'es_en_20h', [a.get('ms_per_step') for a in d.get('also',[])], 'free', d['paths'].get('free_cus_beside_recurrences'), 'side', d['paths'].get('side_stream'))" >> $OUT || { tail -5 gpurun_out/ab_err.log >> $OUT; }
  done
done
cat $OUT
