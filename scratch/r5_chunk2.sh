#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 4 2 1; do ASTK_GEMM_CHUNK_DIV=$d bash scratch/gemm_step_table.sh > gpurun_out/r5_chunk_table_d$d.txt 2>&1; echo "== div $d"; grep "TN 1024x3072\|TN 512x1152\|total" gpurun_out/r5_chunk_table_d$d.txt | cut -c1-110; done
