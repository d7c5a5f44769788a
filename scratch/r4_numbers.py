"""Prints the figures of a scratch/profile_r4.sh run (gpurun_out/prof_r4) that the docs quote."""
import json, glob, csv, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r4"
os.chdir(root)
for f in ["bench_default.log","bench_cfg5.log","bench_cfg5_fp16.log","bench_b64.log","bench_t1200.log","bench_t1680.log","bench_stats_bf16x3.log","bench_stats_f32.log","bench_stats_fp16x2.log","bench_stats_es.log"]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, "ERR", e); continue
    r=d.get('roofline') or {}
    print(f"{f:26s} {d['precision']:7s} ms {d['ms_per_step']:7.3f} frames/s {d['value']/1e6:5.2f}M gemm {r.get('ms_per_step')} ach {r.get('achieved')} frac {r.get('frac')} exec {r.get('executed_frac_of_pipe_peak')} k {d.get('kernels')}")
    for a in d.get('alt_precisions',[]): print("     alt", a['scheme'], a['ms_per_step'], a['value'])
    for a in d.get('also',[]): print("     also", a['ms_per_step'], a['value'], a.get('kernels'), (a.get('cpu_baseline') or {}).get('value'), (a.get('cpu_baseline') or {}).get('sample','')[:90])
    if 'cpu_baseline' in d: print("     cpu", d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sample'][:100])
    if d.get('roofline_scan'): print("     scan", d['roofline_scan']['frac'], d['roofline_scan']['us_per_decoder_step'], (d['roofline_scan'].get('phase') or {}).get('us'))
for tag in ("bf16x3","f32","fp16x2","es"):
    f=max(glob.glob(f'stats_{tag}/*/*kernel_stats.csv'), key=os.path.getmtime)
    rows=list(csv.DictReader(open(f)))
    steps=28
    tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/steps
    fam={}; n_other=0; t_other=0
    for r in rows:
        nm=r['Name']; t=float(r['TotalDurationNs'])/1e6/steps; c=int(r['Calls'])/steps
        key=None
        for k in ("gemm_f32_kernel","decoder_persist_fwd","decoder_persist_bwd","lstm_persist_fwd_g","lstm_persist_bwd_rs","k_zero_split_tiles","k_absmax"):
            if k in nm: key=k
        if key: fam[key]=fam.get(key,[0,0]); fam[key][0]+=t; fam[key][1]+=c
        else: n_other+=c; t_other+=t
    print(tag, "total ms/step %.3f"%tot, {k:(round(v[0],3),round(v[1],1)) for k,v in fam.items()}, "other %.3f ms in %.1f launches"%(t_other,n_other))
d=json.load(open('pmc_summary.json'))
for k,v in d['kernels'].items():
    print(k, {a:b for a,b in v.items() if a in ('mean_us','hbm_bytes_per_step','mfma_pipe_busy_frac','SQ_LDS_BANK_CONFLICT','SQ_WAIT_ANY','SQ_WAVE_CYCLES')})
print(open('gemm_traffic.json').read()[:400])
