#!/bin/bash
# 12-wave 256x128 kernel (ASTK_GEMM_TILE=256) against the default 128x128 one: GEMM tests under the forced tile, per-launch tables, bench A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_t256; rm -rf $O; mkdir -p $O
ASTK_GEMM_TILE=256 timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -k "gemm and bf16x3" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for t in 128 256; do
  ASTK_GEMM_TILE=$t bash scratch/gemm_step_table.sh > $O/step_table_t$t.txt 2>&1
  echo "== per-launch table tile=$t"; cat $O/step_table_t$t.txt
done
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt-precisions --no-also"
for rep in 1 2; do
  for t in 128 256; do
    echo -n "tile=$t rep $rep: "
    ASTK_GEMM_TILE=$t $B 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], 'loss', d.get('loss'))"
  done
done
for sh in "0 38400 512 1152" "0 6400 1024 3072" "1 6400 3072 1024" "2 1024 3072 6400 2" "0 4096 4096 4096"; do
  for t in 128 256; do echo -n "tile=$t "; ASTK_GEMM_TILE=$t python3 scratch/gemm_one.py $sh; done
done
