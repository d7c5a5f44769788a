#!/bin/bash
# scratch/kstats_any.sh <tag> <steps+warmup> [bench args]: per-step kernel table of one bench configuration
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=$1; N=$2; shift; shift
O=gpurun_out/ks_$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps $((N-3)) --warmup 3 --no-cpu-baseline --no-alt-precisions --profile-steps 0 "$@" > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 scratch/kstat2.py $O $N 60 > $O/table.txt
tail -1 $O/bench.log | cut -c1-200
