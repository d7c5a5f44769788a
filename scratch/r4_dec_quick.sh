#!/bin/bash
# quick check of a decoder-kernel change: decoder op tests + goldens, then three bench lines (cfg1) and two (es_en_20h)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "(decoder and bf16x3) or (fullsize_golden and bf16x3) or (train_step_parity and bf16x3) or timeout_raises or bounded_spins" > gpurun_out/r4_dec_tests.log 2>&1 || { tail -n 30 gpurun_out/r4_dec_tests.log; exit 1; }
tail -n 2 gpurun_out/r4_dec_tests.log
for m in cfg1 cfg1 cfg1 es_en_20h es_en_20h; do
python3 bench.py --model $m --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$m', d['precision'], d['ms_per_step'], d['kernels'])"
done
