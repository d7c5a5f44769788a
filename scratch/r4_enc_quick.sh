#!/bin/bash
# quick check of an encoder-recurrence change: lstm op tests + goldens + train-step parity, then three bench lines
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "lstm or (fullsize_golden and bf16x3) or (train_step_parity and bf16x3) or timeout_raises or bounded_spins" > gpurun_out/r4_enc_tests.log 2>&1 || { tail -n 30 gpurun_out/r4_enc_tests.log; exit 1; }
tail -n 2 gpurun_out/r4_enc_tests.log
bash scratch/r4_bench3.sh
