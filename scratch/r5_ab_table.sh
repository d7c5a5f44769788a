#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd $R && bash scratch/gemm_step_table.sh > $R/gpurun_out/r5_ab_table_new.txt 2>&1
export GRAFT_REPO_ROOT=$R/scratch/prev_tree
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT && bash scratch/gemm_step_table.sh > $R/gpurun_out/r5_ab_table_prev.txt 2>&1
paste -d'|' <(cut -c1-22 $R/gpurun_out/r5_ab_table_prev.txt) <(cut -c1-130 $R/gpurun_out/r5_ab_table_new.txt)
