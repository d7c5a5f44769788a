#!/bin/bash
# full GPU suite + repeatability soak on the binary with the ticket protocol and the merged small launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_small4_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r5_small4_tests.log
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/r5_small4_soak.log
timeout -k 10 500 python3 scratch/soak.py cfg1 3000 2>&1 | tail -n 3 >> gpurun_out/r5_small4_soak.log
timeout -k 10 500 python3 scratch/soak.py es_en_20h 1500 2>&1 | tail -n 3 >> gpurun_out/r5_small4_soak.log
timeout -k 10 300 python3 scratch/soak.py cfg5 300 2>&1 | tail -n 3 >> gpurun_out/r5_small4_soak.log
cat gpurun_out/r5_small4_soak.log
