#!/bin/bash
# repeatability soak under the round-4 default arithmetic (bf16x3 GEMMs and recurrences) + the 1024-per-direction reading of configs[4]
cd "$GRAFT_REPO_ROOT"
for w in cfg1 es_en_20h; do timeout -k 10 300 python3 scratch/soak.py $w 300 2>&1 | tail -n 4; done > gpurun_out/r4_soak.log 2>&1
cat gpurun_out/r4_soak.log
timeout -k 10 500 python3 bench.py --model cfg5 --hidden 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --profile-steps 0 > gpurun_out/r4_cfg5_wide.log 2>&1
python3 - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r4_cfg5_wide.log') if l.startswith('{')][-1])
    print('cfg5 --hidden 2048:', d['ms_per_step'], 'ms', d['value'], 'frames/s', d['paths'])
except Exception as e:
    print('cfg5 wide failed', e); print(open('gpurun_out/r4_cfg5_wide.log').read()[-1500:])
PY
