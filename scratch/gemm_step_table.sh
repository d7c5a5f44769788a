#!/bin/bash
# per-launch table of the train step's batched products: logged shapes (tuning knob gemm.log) matched in order with the kernel trace (side stream off)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gemm_step
rm -rf $OUT && mkdir -p $OUT
export ASTK_BENCH_KNOBS="gemm.log=1,SIDE=0${EXTRA_KNOBS:+,$EXTRA_KNOBS}"
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 scratch/bench_knobs.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-alt-precisions --no-also --histogram none ${BENCH_ARGS} > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = sorted(glob.glob('gpurun_out/gemm_step/**/*kernel_trace.csv', recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if 'gemm_f32_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
launches = []   # one entry per dispatch: list of (layout, M, N, K, batch, mode, G, tile)
for line in open('gpurun_out/gemm_step/run.log'):
    m = re.match(r'astk_gemm layout=(\d+) M=(\d+) N=(\d+) K=(\d+) batch=(\d+) mode=(\d+) twolvl=(\d+) group=(\d+)/(\d+) G=(\d+) kt=(\d+) tile=(\d+)', line)
    if not m: continue
    v = list(map(int, m.groups()))
    if v[7] == 0: launches.append([])
    launches[-1].append(v)
print(len(rows), 'dispatches,', len(launches), 'logged launches')
n = min(len(rows), len(launches))
rows, launches = rows[-n:], launches[-n:]
per = n // 4 if n % 4 == 0 else None      # 1 warmup + 3 steps
start = n - per if per else 0
tot_t = tot_f = 0
for r, L in list(zip(rows, launches))[start:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    fl = sum(2.0 * v[1] * v[2] * v[3] * v[4] for v in L)
    tot_t += d; tot_f += fl
    name = r['Kernel_Name']; tmpl = name[name.index('<'):name.index('>') + 1]
    desc = ' + '.join(f"{'NT NN TN'.split()[v[0]]} {v[1]}x{v[2]}x{v[3]}" + (f" b{v[4]}" if v[4] > 1 else '') + f" m{v[5]}" for v in L)
    print(f"{d:8.1f} us {fl / d / 1e6:7.1f} TF  G={L[0][9]:4d} {tmpl:38s} {desc}")
print(f"total {tot_t:.1f} us, {tot_f / 1e9:.1f} GFLOP, {tot_f / tot_t / 1e6:.1f} TFLOP/s (kernels only)")
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
