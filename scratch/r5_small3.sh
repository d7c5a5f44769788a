#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_golden.py -x -q -m gpu -k "cnn or golden or train_step or optimizer or reproducible or trajectory" > gpurun_out/r5_small3_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r5_small3_tests.log
[ $rc -ne 0 ] && exit $rc
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt gpurun_out/r5_small_trace_cfg1.txt
grep -v "gemm_f32\|persist" gpurun_out/r5_small_trace_cfg1.txt | cut -c1-100
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions > gpurun_out/r5_small3_bench.json 2> gpurun_out/r5_small3_bench.err || exit 1
python3 -c "
import json
d=json.loads(open('gpurun_out/r5_small3_bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], 'gemm_ms', d.get('roofline',{}).get('ms_per_step'), 'also', [a.get('ms_per_step') for a in d.get('also',[])])"
