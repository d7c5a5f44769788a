#!/bin/bash
# kernel-trace statistics of the default bench step: scratch/kstats_run.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=$1; shift
O=gpurun_out/ks_$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions "$@" > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 scratch/kstat.py $O k_bn k_repack k_phase k_absmax k_zero k_colstats k_seq k_im2col
