"""Copies the summaries of a scratch/profile_r5.sh run (gpurun_out/prof_r5) into profiles/ under their round-5 names."""
import glob, json, os, shutil
P = "gpurun_out/prof_r5"
for tag, name in (("bf16x3", "cfg1_bf16x3"), ("f32", "cfg1_f32"), ("fp16x2", "cfg1_fp16x2"), ("es", "es_en_20h_bf16x3")):
    shutil.copy(max(glob.glob(f"{P}/stats_{tag}/*/*kernel_stats.csv"), key=os.path.getmtime), f"profiles/r5_kernel_stats_{name}.csv")
for a, b in (("pmc_summary.json", "r5_pmc_summary.json"), ("gemm_traffic.json", "r5_gemm_traffic.json"), ("attn_traffic.json", "r5_attn_traffic.json"),
             ("trace_step_cfg1.txt", "r5_trace_step_cfg1.txt"), ("trace_step_es_en_20h.txt", "r5_trace_step_es_en_20h.txt"),
             ("gemm_step_table.txt", "r5_gemm_step_table.txt"), ("gemm_step_table_r4_schedule.txt", "r5_gemm_step_table_r4_schedule.txt")):
    shutil.copy(f"{P}/{a}", f"profiles/{b}")
fresh = []
for f in ("phase_stamps.log", "phase_stamps_es.log"):
    fresh.append(f"==== {f} (libastk_test.so, ASTK_PERSIST_DBG=8)")
    fresh += [l.rstrip("\n") for l in open(f"{P}/{f}") if not l.startswith("{") and ("persist" in l or "bwd_rs" in l or "dec" in l)]
open("profiles/r5_phase_stamps.txt", "w").write("\n".join(fresh) + "\n")
out = {}
for n in ["bench_default", "bench_cfg5", "bench_cfg5_fp16", "bench_cfg5_wide", "bench_b64", "bench_t1200", "bench_t1680"]:
    out[n] = json.loads([l for l in open(f"{P}/{n}.log") if l.startswith("{")][-1])
json.dump(out, open("profiles/r5_bench_lines.json", "w"), indent=1)
print({k: v["ms_per_step"] for k, v in out.items()})
