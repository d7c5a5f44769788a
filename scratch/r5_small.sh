#!/bin/bash
# small-launch merges: full GPU suite, then the ordered trace of a step and a bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_small_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r5_small_tests.log
[ $rc -ne 0 ] && exit $rc
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_cfg1.txt
PROF_ARGS="--model es_en_20h --no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_es.txt
timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt-precisions > gpurun_out/r5_small_bench.json 2> gpurun_out/r5_small_bench.err || exit 1
python3 -c "
import json
d=json.loads(open('gpurun_out/r5_small_bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], 'also', [a.get('ms_per_step') for a in d.get('also',[])] if isinstance(d.get('also'),list) else d.get('also'))"
tail -3 gpurun_out/r5_small_trace_cfg1.txt
