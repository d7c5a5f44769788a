#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ks_cfg5; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --model cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precisions --profile-steps 0 > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 scratch/kstat2.py $O 13 60 > $O/table.txt
tail -2 $O/bench.log | cut -c1-400
