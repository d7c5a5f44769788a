#!/bin/bash
# quick check of the small-launch fusions (state bridge, bias gradients inside the encoder's backward recurrence, last-layer BatchNorm backward from
# the sequence layout): the op / model tests that touch them under the default arithmetic, then the bench lines and the ordered step trace
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "${KEXPR:-cnn or lstm or (fullsize_golden and bf16x3) or (train_step_parity and bf16x3) or sync_batchnorm or permutation or predict or checkpoints}" > gpurun_out/r4_small_tests.log 2>&1 || { tail -n 40 gpurun_out/r4_small_tests.log; exit 1; }
tail -n 2 gpurun_out/r4_small_tests.log
MODELS="${MODELS:-cfg1 cfg1 es_en_20h}" bash scratch/r4_bench3.sh
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/trace_step_cfg1.txt && tail -n 1 gpurun_out/trace_step_cfg1.txt
