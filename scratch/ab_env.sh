#!/bin/bash
# A/B of one build under different ASTK_GEMM_X3_BELOW thresholds, inside one gpurun call
for rep in 1 2; do
  for t in 1.5e9 3e9 6e9 12e9; do
    echo -n "== below $t (rep $rep): "
    ASTK_GEMM_X3_BELOW=$t python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'])"
  done
done
