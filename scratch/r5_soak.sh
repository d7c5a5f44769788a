#!/bin/bash
# round 5: full GPU suite on the current build, then the repeatability soak (hybrid GEMM schedule, 12-wave kernel, fused random fills)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_gputests.log 2>&1 || { tail -30 gpurun_out/r5_gputests.log; exit 1; }
tail -3 gpurun_out/r5_gputests.log
for w in cfg1 es_en_20h; do timeout -k 10 400 python3 scratch/soak.py $w 1000 2>&1 | tail -n 4; done > gpurun_out/r5_soak.log 2>&1
timeout -k 10 300 python3 scratch/soak.py cfg5 200 2>&1 | tail -n 4 >> gpurun_out/r5_soak.log
cat gpurun_out/r5_soak.log
