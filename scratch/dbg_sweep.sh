#!/bin/bash
# GEMM timing experiments: each ASTK_GEMM_DBG build at 4096^3, both split precisions
for d in 8 16 4 1; do
  for p in bf16x3 f16x2; do
    echo "== dbg $d prec $p"
    ASTK_LIB_PATH=$PWD/scratch/libastk_dbg$d.so ASTK_GEMM_PREC=$p timeout -k 10 120 python scratch/gemm_bench_one.py 2>&1 | grep layout
  done
done
