#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r5_long_soak.log
timeout -k 10 500 python3 scratch/soak.py cfg1 10000 2>&1 | grep -v amdgpu.ids | tail -n 2 >> gpurun_out/r5_long_soak.log
timeout -k 10 500 python3 scratch/soak.py es_en_20h 5000 2>&1 | grep -v amdgpu.ids | tail -n 2 >> gpurun_out/r5_long_soak.log
cat gpurun_out/r5_long_soak.log
