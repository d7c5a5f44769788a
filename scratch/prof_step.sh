#!/bin/bash
# rocprofv3 kernel-trace summary of a short bench run -> gpurun_out/prof_step/
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $PROF_ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_step.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_step -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_step_kernel_stats.csv
find $GRAFT_REPO_ROOT/gpurun_out/prof_step -type f ! -name "*stats*" -delete
