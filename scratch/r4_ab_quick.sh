#!/bin/bash
# quick check of a library change: lstm op tests + three bench lines
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "lstm_stack" > gpurun_out/r4_quick_tests.log 2>&1 || { tail -n 30 gpurun_out/r4_quick_tests.log; exit 1; }
tail -n 2 gpurun_out/r4_quick_tests.log
for i in 1 2 3; do
python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(d['precision'], d['ms_per_step'], 'gemm', r['ms_per_step'], d['kernels'])"
done
