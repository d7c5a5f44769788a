#!/bin/bash
# A/B of the eight-wave row-panel kernels inside one box: cfg5 (H = 1024 per-launch decoder) and cfg1 with the per-launch decoder forced
for v in 0 2048 1024; do
  echo "ASTK_ROW_LONGK=$v"
  ASTK_ROW_LONGK=$v python3 bench.py --model cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precisions --profile-steps 0 2>&1 | tail -1 | cut -c1-160
  ASTK_ROW_LONGK=$v ASTK_DEC_PERSIST=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precisions --profile-steps 0 2>&1 | tail -1 | cut -c1-160
done
