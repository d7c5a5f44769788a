#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for t in 1 0 1 0; do
  ASTK_GEMM_TICKET=$t timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions > gpurun_out/r5_small2_bench_$t.json 2> gpurun_out/r5_small2_bench_$t.err || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/r5_small2_bench_$t.json').read().strip().splitlines()[-1])
print('ticket=$t', d['ms_per_step'], 'gemm_ms', d.get('roofline',{}).get('ms_per_step'), 'also', [a.get('ms_per_step') for a in d.get('also',[])] if isinstance(d.get('also'),list) else d.get('also'))"
done
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_cfg1.txt
grep -v "gemm_f32\|persist" gpurun_out/r5_small_trace_cfg1.txt | cut -c1-100
