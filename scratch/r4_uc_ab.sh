#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for teach in 1.0 0.8; do for uc in 0 1; do for i in 1 2; do
ASTK_BENCH_TEACH=$teach ASTK_DEC_ENCC=$uc python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('teach $teach uc $uc', d['ms_per_step'], d['kernels']['decoder_us_per_decoder_step'], d['kernels']['decoder_persistent_ms_per_step'])"
done; done; done
