import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=[int(r['Calls']) for r in rows if 'lstm_persist_fwd_g' in r['Name']][0]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('steps',steps,'total per step ms', round(tot/steps/1e6,3))
pat=sys.argv[2:] 
for r in rows:
    if not pat or any(p in r['Name'] for p in pat):
        print(f"{r['Name'][:70]:70s} {int(r['Calls'])/steps:6.1f} {float(r['TotalDurationNs'])/steps/1e3:8.1f} us")
