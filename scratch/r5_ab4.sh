#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd $R
ASTK_LIB_PATH=$R/scratch/libastk_sc1.so timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_split_tiles or gemm_hybrid or twelve" > gpurun_out/r5_ab4_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_ab4_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for t in prev zero sc1 nofence; do
    D=$R; E="X=1"
    [ $t = prev ] && D=$R/scratch/prev_tree
    [ $t = nofence ] && E="ASTK_LIB_PATH=$R/scratch/libastk_nofence.so"
    [ $t = sc1 ] && E="ASTK_LIB_PATH=$R/scratch/libastk_sc1.so"
    [ $t = zero ] && E="ASTK_GEMM_TICKET=0"
    cd $D
    env $E timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$t rep $rep cfg1', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'])"
  done
done
