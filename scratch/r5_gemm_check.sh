#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_check; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_golden.py -x -q -m gpu -k "gemm or cnn or lstm_stack or golden" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
bash scratch/gemm_step_table.sh > $O/step_table.txt 2>&1; cat $O/step_table.txt
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt-precisions --no-also"
for rep in 1 2; do
  for v in "ASTK_GEMM_TILE=128 ASTK_GEMM_HYBRID=0" "ASTK_GEMM_HYBRID=1"; do
    echo -n "$v rep $rep: "
    env $v $B 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gemm', d['roofline']['ms_per_step'], 'loss', d.get('loss'))"
  done
done
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $P > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $P > $O/write.log 2>&1
python3 - <<'PY'
import csv, glob
O = "gpurun_out/r5_check"
tot = {}
for c, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    v = 0.0
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_f32_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: v += float(r["Counter_Value"])
    tot[c] = v
print(f"hbm-side bytes/step (2F+W) {(2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / 4 / 1e9:.2f} GB")
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
