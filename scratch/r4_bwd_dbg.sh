#!/bin/bash
# per-phase wait / work timers of the decoder persistent kernels (NL == 1): ASTK_PERSIST_DBG=1
cd "$GRAFT_REPO_ROOT"
ASTK_PERSIST_DBG=1 ASTK_BENCH_TEACH=${TEACH:-0.8} timeout -k 10 300 python3 bench.py --model cfg1 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also > gpurun_out/r4_bwd_dbg.log 2>&1
grep -a "pdec" gpurun_out/r4_bwd_dbg.log | tail -n 24
