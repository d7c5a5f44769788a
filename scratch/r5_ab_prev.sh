#!/bin/bash
# same-box A/B: an earlier tree against the working tree, alternating inside ONE gpurun call (boxes differ by 3-4 %: only this kind of
# comparison means anything).  Set-up (scratch/prev_tree is git-ignored, its built .so files travel with the snapshot):
#   git worktree add -f scratch/prev_tree d914766 && bash scratch/prev_tree/ast_amd/csrc/build.sh
# (d914766 = the tree in front of this round's launch / address work)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
mkdir -p $R/gpurun_out
: > $R/gpurun_out/r5_ab_prev.txt
for rep in 1 2 3; do
  for t in prev new; do
    if [ $t = prev ]; then D=$R/scratch/prev_tree; else D=$R; fi
    cd $D
    timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$t rep $rep cfg1', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], 'es_en_20h', [a.get('ms_per_step') for a in d.get('also',[])])" >> $R/gpurun_out/r5_ab_prev.txt || exit 1
  done
done
cat $R/gpurun_out/r5_ab_prev.txt
