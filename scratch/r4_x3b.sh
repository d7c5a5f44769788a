#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "lstm_stack and bf16x3" > gpurun_out/r4_x3_lstm.log 2>&1
grep -n "^FAILED\|passed\|failed\|max abs err" gpurun_out/r4_x3_lstm.log | cut -c1-160 | head -40
