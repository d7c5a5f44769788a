#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 2e10 5e9 2e9; do ASTK_GEMM_T256_ABOVE=$t bash scratch/gemm_step_table.sh > gpurun_out/r5_t256_$t.txt 2>&1; echo "== above $t"; grep "6400x512x512\|512x1024x1248\|total" gpurun_out/r5_t256_$t.txt | cut -c1-120; done
