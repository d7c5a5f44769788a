#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_gputests.log 2>&1; rc=$?
tail -15 gpurun_out/r5_gputests.log
exit $rc
