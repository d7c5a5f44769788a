#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_det; rm -rf $O; mkdir -p $O
timeout -k 10 500 python3 scratch/soak.py cfg1 3000 2>&1 | tail -n 3
timeout -k 10 500 python3 scratch/soak.py es_en_20h 1500 2>&1 | tail -n 3
for i in 1 2; do python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt-precisions --no-also 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['ms_per_step'])"; done
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $P > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $P > $O/write.log 2>&1
python3 - <<'PY'
import csv, glob
O = "gpurun_out/r5_det"
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
