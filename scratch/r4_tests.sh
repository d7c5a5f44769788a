#!/bin/bash
# round 4: smoke + the whole GPU suite, log under gpurun_out/
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1 || { tail -n 30 gpurun_out/r4_smoke.log; exit 1; }
tail -n 2 gpurun_out/r4_smoke.log
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/r4_tests.log 2>&1
rc=$?
tail -n 40 gpurun_out/r4_tests.log
exit $rc
