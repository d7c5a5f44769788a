#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_roles.txt; : > $O
for sh in "0 4096 4096 4096 0 30" "0 38400 512 1152 0 30" "2 1024 3072 6400 2 30"; do
  for v in ${VARIANTS:-product nostage nomult prio0 prio1 ring4}; do
    L=$PWD/scratch/libastk_$v.so; [ $v = product ] && L=$PWD/ast_amd/libastk.so
    echo -n "tile=256 $v: " >> $O; ASTK_GEMM_TILE=256 ASTK_LIB_PATH=$L python3 scratch/gemm_one.py $sh 2>&1 | tail -3 | tr '\n' ' ' >> $O; echo >> $O
  done
done
cat $O
