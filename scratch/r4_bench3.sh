#!/bin/bash
# three headline bench lines (kernel figures only)
cd "$GRAFT_REPO_ROOT"
for m in ${MODELS:-cfg1 cfg1 cfg1}; do
python3 bench.py --model $m --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$m', d['precision'], d['ms_per_step'], d['kernels'])"
done
