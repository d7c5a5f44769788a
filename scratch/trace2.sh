#!/bin/bash
# kernel trace of the last train step with START OFFSETS and queue ids (overlapping launches on two streams): gpurun_out/$1.txt
cd /tmp && export TMPDIR=/tmp
OUT=${1:-r6_trace}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_tmp -- python3 $GRAFT_REPO_ROOT/scratch/bench_knobs.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-alt-precisions --no-also --histogram none $PROF_ARGS > $GRAFT_REPO_ROOT/gpurun_out/$OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(root + "/gpurun_out/trace_tmp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_amsgrad" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
out = open(root + "/gpurun_out/" + sys.argv[1] + ".txt", "w")
qs = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    name = r["Kernel_Name"].replace("astk::(anonymous namespace)::", "").replace("void ", "")
    out.write(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  q{q}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>7}  {name[:80]}\n")
out.write(f"step span {(int(rows[b-1]['End_Timestamp']) - t0) / 1e6:.3f} ms, {b - a} launches\n")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_tmp
