#!/bin/bash
# A/B of two builds of libastk.so inside ONE gpurun call (two boxes of the pool differ by up to 10 %): scratch/ab_lib.sh <other.so> [bench args]
cd "$GRAFT_REPO_ROOT"
OTHER=$1; shift
for i in 1 2 3; do
  for lib in ast_amd/libastk.so $OTHER; do
    ASTK_LIB_PATH=$PWD/$lib python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$lib', d['ms_per_step'], 'gemm', r['ms_per_step'], d['kernels'])"
  done
done
