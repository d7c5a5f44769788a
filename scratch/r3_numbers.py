"""prints the figures the round-3 docs quote, from gpurun_out/prof_r3 (scratch/profile_r3.sh)"""
import csv, glob, json, os
P = "gpurun_out/prof_r3"
def line(f):
    return json.loads([l for l in open(os.path.join(P, f)) if l.startswith("{")][-1])
for f in ("bench_default", "bench_es_en_20h", "bench_cfg5", "bench_cfg5_fp16", "bench_cfg1_fp16", "bench_b64", "bench_t1200", "bench_t1680", "bench_es_t1200"):
    d = line(f + ".log"); r = d["roofline"]
    print(f"{f:18s} ms {d['ms_per_step']:7.3f}  frames/s {d['value']/1e6:5.2f} M  frac {r['frac']:.3f} useful {r['useful_tflops']:6.1f} gemm ms {r['ms_per_step']:.3f}",
          [(a['scheme'], a['ms_per_step'], round(a['value']/1e6, 2)) for a in d.get('alt_precisions', [])], (d.get('cpu_baseline') or {}).get('value'),
          (d.get('cpu_baseline') or {}).get('sample', '')[:70])
for tag in ("stats", "stats_es"):
    f = sorted(glob.glob(f"{P}/{tag}/runc/*kernel_stats.csv"), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    steps = [int(r['Calls']) for r in rows if 'lstm_persist_fwd_g' in r['Name']][0]
    tot = sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e3
    fam = lambda keys: sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in keys)) / steps / 1e3
    print(tag, f, "steps", steps, "kernel us/step %.1f" % tot, "gemm_f32 %.1f" % fam(['gemm_f32_kernel']), "absmax %.1f" % fam(['k_absmax']), "zero %.1f" % fam(['k_zero_split']),
          "dec fwd %.1f bwd %.1f" % (fam(['decoder_persist_fwd']), fam(['decoder_persist_bwd'])), "enc fwd %.1f bwd %.1f" % (fam(['lstm_persist_fwd']), fam(['lstm_persist_bwd'])))
d = json.load(open(f"{P}/pmc_summary.json"))
for k, v in d['kernels'].items():
    print(k, v.get('mean_us'), v.get('hbm_bytes_per_dispatch'), v.get('mfma_pipe_busy_frac'))
print(open(f"{P}/gemm_traffic.json").read()[:200])
