"""per-step kernel table from a rocprofv3 --stats directory: kstat2.py <dir> <steps incl. warm-up> [top N]"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('steps', steps, 'kernel time per step ms', round(tot / steps / 1e6, 3))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:top]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])/steps:7.1f} calls {float(r['TotalDurationNs'])/steps/1e3:9.1f} us/step {float(r['AverageNs'])/1e3:8.1f} us avg")
