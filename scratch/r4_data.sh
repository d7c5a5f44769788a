#!/bin/bash
# does the GEMM rate depend on the operand DATA (matrix-pipe power -> clock)?  same shapes, same kernel, three distributions
cd "$GRAFT_REPO_ROOT"
for d in rand randn relu; do for p in bf16x3 fp16x2 f32; do echo "== data $d prec $p"; DATA=$d ASTK_GEMM_PREC=$p python3 scratch/gemm_nosplit_bench.py 2>/dev/null; done; done > gpurun_out/r4_gemm_data.log 2>&1
grep "==\|sum\|4096" gpurun_out/r4_gemm_data.log
