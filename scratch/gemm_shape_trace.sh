#!/bin/bash
# kernel-trace of the GEMM microbench: kernel time (without the absolute-maximum passes) per shape
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gemm_trace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 scratch/gemm_nosplit_bench.py > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/gemm_trace/**/*kernel_trace.csv', recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if any(k in r['Kernel_Name'] for k in ('gemm_f32', 'absmax', 'zero_split'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ng, per = 0, {}
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    g = 'gemm_f32' in r['Kernel_Name']
    e = per.setdefault(ng // 23, {'gemm': 0, 'other': 0, 'name': ''})
    e['gemm' if g else 'other'] += d
    if g:
        e['name'] = r['Kernel_Name'][r['Kernel_Name'].index('<'):r['Kernel_Name'].index('>') + 1] + ' wg ' + str(int(r['Grid_Size_X']) // 512)
        ng += 1
for i, e in per.items(): print(i, e['name'], f"gemm {e['gemm'] / 23e3:.1f} us  max/zero passes {e['other'] / 23e3:.1f} us")
PY
cat $OUT/run.log | grep layout
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
