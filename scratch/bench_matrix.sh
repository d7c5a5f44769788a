#!/bin/bash
# the round's results table: one bench line per workload -> gpurun_out/matrix_*.log
run() { name=$1; shift; python bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/matrix_$name.log 2>&1; }
run cfg1 &&
run es_en_20h --model es_en_20h &&
run cfg1_t1200 --frames 1200 &&
run cfg1_t1680 --frames 1680 &&
run es_en_20h_t1200 --model es_en_20h --frames 1200 &&
run cfg1_fp16 --gemm-operands fp16 &&
run cfg5 --model cfg5 --steps 10 --warmup 3 &&
run cfg5_fp16 --model cfg5 --gemm-operands fp16 --steps 10 --warmup 3 &&
run cfg1_b64 --batch 64 --steps 20 --warmup 5
