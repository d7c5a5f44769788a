#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or cnn" > gpurun_out/r5_chunk_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_chunk_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do for c in 0 1; do
ASTK_GEMM_CHUNK=$c python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('chunk=$c', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'])"; done; done
for c in 0 1; do ASTK_GEMM_CHUNK=$c bash scratch/gemm_step_table.sh > gpurun_out/r5_chunk_table_$c.txt 2>&1; done
paste -d'|' <(cut -c1-22 gpurun_out/r5_chunk_table_0.txt) <(cut -c1-110 gpurun_out/r5_chunk_table_1.txt) | grep "TN\|total"
grep "cs=[1-9]" gpurun_out/gemm_step/run.log | sort | uniq -c | head
