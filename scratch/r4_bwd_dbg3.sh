#!/bin/bash
# per-phase timers of the 3-layer decoder kernels: needs a library built with ASTK_EXTRA_FLAGS=-DASTK_PDEC_TIMING_ALL=1
cd "$GRAFT_REPO_ROOT"
ASTK_PERSIST_DBG=1 ASTK_BENCH_TEACH=${TEACH:-0.8} timeout -k 10 300 python3 bench.py --model es_en_20h --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also > gpurun_out/r4_bwd_dbg3.log 2>&1
grep -a "pdec" gpurun_out/r4_bwd_dbg3.log | tail -n 16
