#!/bin/bash
# Round 5, item 1: hybrid GEMM schedule (XCD-local data-parallel waves + stream-K remainder) against the round-4 schedule
# (ASTK_GEMM_HYBRID=0), inside ONE gpurun call: GEMM tests, bench A/B, per-launch table, HBM-side traffic of both.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_gemm; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt-precisions --no-also"
for rep in 1 2; do
  for h in 0 1; do
    echo -n "hybrid=$h rep $rep: "
    ASTK_GEMM_HYBRID=$h $B 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], d['kernels'])"
  done
done
for h in 0 1; do
  ASTK_GEMM_HYBRID=$h bash scratch/gemm_step_table.sh > $O/step_table_h$h.txt 2>&1
  echo "== per-launch table hybrid=$h"; cat $O/step_table_h$h.txt
done
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0"
for h in 0 1; do
  ASTK_GEMM_HYBRID=$h rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_h$h -- $P > $O/fetch_h$h.log 2>&1
  ASTK_GEMM_HYBRID=$h rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_h$h -- $P > $O/write_h$h.log 2>&1
  ASTK_GEMM_HYBRID=$h rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc_h$h -- $P > $O/tcc_h$h.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/r5_gemm"
for h in (0, 1):
    tot = {}
    for c, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("TCC_HIT_sum", "tcc"), ("TCC_MISS_sum", "tcc")):
        v = 0.0; n = 0
        for f in glob.glob(f"{O}/{d}_h{h}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_f32_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v += float(r["Counter_Value"]); n += 1
        tot[c] = (v, n)
    steps = 4
    f, w = tot["FETCH_SIZE"][0], tot["WRITE_SIZE"][0]
    print(f"hybrid={h}: gemm dispatches/step {tot['FETCH_SIZE'][1] / steps:.1f}  FETCH {f / steps / 1e6:.3f} GB(KB units)  WRITE {w / steps / 1e6:.3f}  "
          f"hbm-side bytes/step (2F+W) {(2 * f + w) * 1024 / steps / 1e9:.2f} GB   L2 hit rate {tot['TCC_HIT_sum'][0] / max(1.0, tot['TCC_HIT_sum'][0] + tot['TCC_MISS_sum'][0]):.3f}")
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
