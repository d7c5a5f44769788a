#!/bin/bash
# bf16x3 recurrences: op tests + goldens under bf16x3, then the bench with and without them (A/B in one call), then the GEMM table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "lstm_stack or fullsize_golden or train_step_parity or thirty_update" > gpurun_out/r4_x3_tests.log 2>&1 || { tail -n 40 gpurun_out/r4_x3_tests.log; exit 1; }
tail -n 3 gpurun_out/r4_x3_tests.log
for i in 1 2; do
  for x3 in 1 0; do
    ASTK_LSTM_X3=$x3 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('x3=$x3', d['precision'], d['ms_per_step'], 'gemm', r['ms_per_step'], d['kernels'])"
  done
done | tee gpurun_out/r4_x3_ab.log
bash scratch/gemm_step_table.sh > gpurun_out/r4_gemm_table_bf16x3.log 2>&1
tail -n 25 gpurun_out/r4_gemm_table_bf16x3.log
