#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd $R
ASTK_LIB_PATH=$R/scratch/libastk_fence.so timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/r5_ab8_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_ab8_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for t in cur fence; do
    E="X=1"
    [ $t = fence ] && E="ASTK_LIB_PATH=$R/scratch/libastk_fence.so"
    env $E timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$t rep $rep cfg1', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'])"
  done
done
for t in cur fence; do
  E="X=1"
  [ $t = fence ] && E="ASTK_LIB_PATH=$R/scratch/libastk_fence.so"
  env $E bash scratch/gemm_step_table.sh > gpurun_out/r5_ab8_table_$t.txt 2>&1
done
paste -d'|' <(cut -c1-22 gpurun_out/r5_ab8_table_cur.txt) <(cut -c1-120 gpurun_out/r5_ab8_table_fence.txt)
