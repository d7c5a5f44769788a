#!/bin/bash
# round 4, first call: bf16x3 kernel with and without the in-loop split arithmetic (upper bound of operands that arrive as planes)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for lib in ast_amd/libastk.so scratch/libastk_nosplit.so; do
  for prec in bf16x3 fp16x2; do
    echo "== $lib $prec"
    ASTK_GEMM_PREC=$prec ASTK_LIB_PATH=$PWD/$lib python3 scratch/gemm_nosplit_bench.py
  done
done > gpurun_out/r4_nosplit.log 2>&1
ASTK_GEMM_PREC=bf16x3 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > gpurun_out/r4_bench_bf16x3_base.log 2>&1
ASTK_GEMM_PREC=bf16x3 python3 bench.py --model es_en_20h --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > gpurun_out/r4_bench_bf16x3_es.log 2>&1
tail -n 30 gpurun_out/r4_nosplit.log
