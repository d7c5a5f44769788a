#!/bin/bash
# which launches are better off on the f32-input MFMA kernel: per-launch GEMM times of one train step for ASTK_GEMM_F32_BELOW thresholds
cd "$GRAFT_REPO_ROOT"
for b in 0 1e9 3e9 7e9; do
  export ASTK_GEMM_F32_BELOW=$b
  PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh
  echo "f32 below $b"; grep 'gemm_f32_kernel<64\|step span' gpurun_out/trace_step.txt | cut -c1-75
  grep 'gemm_f32_kernel<128' gpurun_out/trace_step.txt | awk '$1 < 130' | cut -c1-75
done
