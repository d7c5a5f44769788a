import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True))[-1]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_im2col0' in r['Kernel_Name']]
sp=[(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3 for a,b in zip(idx,idx[1:])]
print('step spans us', [round(x) for x in sp])
a,b=idx[-2],idx[-1]
# time from decoder_persist_bwd end to lstm_persist_bwd end, and to first gemm after
d=[r for r in rows[a:b] if 'decoder_persist_bwd' in r['Kernel_Name']][0]
l=[r for r in rows[a:b] if 'lstm_persist_bwd' in r['Kernel_Name']][0]
print('dec bwd end -> lstm bwd start', (int(l['Start_Timestamp'])-int(d['End_Timestamp']))/1e3, 'lstm bwd dur', (int(l['End_Timestamp'])-int(l['Start_Timestamp']))/1e3)
nxt=[r for r in rows[a:b] if int(r['Start_Timestamp'])>=int(l['End_Timestamp'])][0]
print('lstm bwd end -> next kernel start', (int(nxt['Start_Timestamp'])-int(l['End_Timestamp']))/1e3, nxt['Kernel_Name'][:50])
