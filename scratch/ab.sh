#!/bin/bash
# A/B of two library builds inside ONE gpurun call (boxes differ by 10-20 %): alternate runs
for rep in 1 2; do
  for v in ${AB_VARIANTS:-base loc}; do
    echo -n "== $v (rep $rep): "
    ASTK_LIB_PATH=$PWD/scratch/libastk_$v.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline $AB_ARGS 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], d['kernels'])"
  done
done
