#!/bin/bash
# longer repeatability soak on the round's final binary (hybrid schedule, 12-wave kernel, fused random fills, side-stream flag upload)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_soak_long.log
for w in cfg1 es_en_20h; do timeout -k 10 500 python3 scratch/soak.py $w 3000 2>&1 | tail -n 3 >> gpurun_out/r5_soak_long.log; echo "== $w done" >> gpurun_out/r5_soak_long.log; done
timeout -k 10 300 python3 scratch/soak.py cfg5 400 2>&1 | tail -n 3 >> gpurun_out/r5_soak_long.log
cat gpurun_out/r5_soak_long.log
