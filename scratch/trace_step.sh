#!/bin/bash
# ordered kernel trace of the last train step of a short bench run -> gpurun_out/trace_step.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_step
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 $PROF_ARGS > $GRAFT_REPO_ROOT/gpurun_out/trace_step.log 2>&1
python3 - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(root + "/gpurun_out/trace_step/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last k_fill_normal / first kernel after the second-to-last k_amsgrad to the last k_amsgrad
idx = [i for i, r in enumerate(rows) if "k_amsgrad" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
out = open(root + "/gpurun_out/trace_step.txt", "w")
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].replace("astk::(anonymous namespace)::", "").replace("void ", "")
    out.write(f"{(e - s) / 1e3:9.1f} us  gap {gap:6.1f}  {name[:90]}\n")
    prev_end = e
out.write(f"step span {(int(rows[b-1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6:.3f} ms, {b - a} launches\n")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_step
