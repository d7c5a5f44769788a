#!/bin/bash
# LDS-DMA staging trial (review item 2 ii): scratch/libastk_ldsdma.so (= product objects, gemm.hip with -DASTK_GEMM_LDSDMA=1: the 12-wave NT
# kernel's raw f32 tiles go global -> LDS by buffer_load_dwordx4 ... lds) against the product library, same box, alternating.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_ldsdma.txt; : > $O
V=$PWD/scratch/libastk_ldsdma.so; P=$PWD/ast_amd/libastk.so
echo "== parity (variant library): gemm tests + golden" >> $O
ASTK_LIB_PATH=$V timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" 2>&1 | tail -3 >> $O || exit 1
ASTK_LIB_PATH=$V timeout -k 10 600 python -m pytest tests/test_golden.py -q -x -m gpu -k "cfg1 and bf16x3" 2>&1 | tail -3 >> $O || exit 1
echo "== single launches (NT, 12-wave kernel)" >> $O
for rep in 1 2; do
for sh in "0 38400 512 1152 0 30" "0 6400 1024 3072 0 30" "0 4096 4096 4096 0 30" "0 38400 128 2560 0 30"; do
  for v in product ldsdma; do
    L=$P; [ $v = ldsdma ] && L=$V
    echo -n "$v: " >> $O; ASTK_LIB_PATH=$L python3 scratch/gemm_one.py $sh 2>&1 | tail -1 >> $O
  done
done
done
echo "== train step, side stream off (family ms by the bench's events)" >> $O
for rep in 1 2; do
  for v in product ldsdma; do
    L=$P; [ $v = ldsdma ] && L=$V
    ASTK_LIB_PATH=$L ASTK_SIDE_STREAM=0 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > gpurun_out/r6_ldsdma_bench.log 2>&1
    python3 - >> $O <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6_ldsdma_bench.log") if l.startswith("{")][-1])
print("$v rep $rep step", d["ms_per_step"], "gemm family", d["roofline"]["ms_per_step"], "TFLOP/s", d["roofline"]["achieved"])
PY
  done
done
cat $O
