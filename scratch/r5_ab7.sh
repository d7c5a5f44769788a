#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for t in prev eager lazy; do
    D=$R; E="X=1"
    [ $t = prev ] && D=$R/scratch/prev_tree
    [ $t = eager ] && E="ASTK_LAZY_CLEARGRADS=0"
    cd $D
    env $E timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$t rep $rep cfg1', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], 'es', [a.get('ms_per_step') for a in d.get('also',[])])"
  done
done
