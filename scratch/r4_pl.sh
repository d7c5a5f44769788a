#!/bin/bash
# plane operands: lstm_stack op tests (K9 forward reads planes), then the step with / without planes (A/B in one call) and the GEMM table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "lstm_stack and bf16x3 or (fullsize_golden and bf16x3) or (train_step_parity and bf16x3)" > gpurun_out/r4_pl_tests.log 2>&1 || { tail -n 40 gpurun_out/r4_pl_tests.log; exit 1; }
tail -n 2 gpurun_out/r4_pl_tests.log
for i in 1 2; do
  for pl in 1 0; do
    ASTK_GEMM_PLANES=$pl python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('planes=$pl', d['precision'], d['ms_per_step'], 'gemm', r['ms_per_step'], 'launches', r['launches_per_step'])"
  done
done | tee gpurun_out/r4_pl_ab.log
bash scratch/gemm_step_table.sh > gpurun_out/r4_gemm_table_pl.log 2>&1
tail -n 22 gpurun_out/r4_gemm_table_pl.log | cut -c1-230
