#!/bin/bash
# grid-size A/B of the last layer's sequence-layout BatchNorm backward kernels: kernel times from the ordered step trace
cd "$GRAFT_REPO_ROOT"
for b in "512 1024" "256 4096" "2048 8192"; do
  set -- $b
  export ASTK_SEQ_STATS_BLOCKS=$1 ASTK_SEQ_APPLY_BLOCKS=$2
  PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh
  echo "blocks $b"; grep 'colstats\|bn_bwd_stats\|bn_bwd_apply\|step span' gpurun_out/trace_step.txt | cut -c1-60
done
