#!/bin/bash
# split tiles under the ticket protocol: GEMM tests, then the bench line with and without it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/r5_ticket_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5_ticket_tests.log
[ $rc -ne 0 ] && exit $rc
for t in 1 0; do
  ASTK_GEMM_TICKET=$t timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt-precisions > gpurun_out/r5_ticket_bench_$t.json 2> gpurun_out/r5_ticket_bench_$t.err || exit 1
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r5_ticket_bench_$t.json').read().strip().splitlines()[-1])
print('ticket=$t', d['ms_per_step'], 'gemm_ms', d.get('roofline',{}).get('ms_per_step'), 'also', [ (a.get('ms_per_step')) for a in d.get('also',[])] if isinstance(d.get('also'),list) else d.get('also',{}).get('ms_per_step'))"
done
