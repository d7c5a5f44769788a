"""Copies the summaries of a scratch/profile_r6.sh run (gpurun_out/prof_r6) into profiles/ under their round-6 names."""
import glob, json, os, shutil
P = "gpurun_out/prof_r6"
for tag, name in (("bf16x3", "cfg1_bf16x3"), ("f32", "cfg1_f32"), ("fp16x2", "cfg1_fp16x2"), ("es", "es_en_20h_bf16x3"), ("side", "cfg1_bf16x3_sidestream")):
    shutil.copy(max(glob.glob(f"{P}/stats_{tag}/*/*kernel_stats.csv"), key=os.path.getmtime), f"profiles/r6_kernel_stats_{name}.csv")
for a, b in (("pmc_summary.json", "r6_pmc_summary.json"), ("gemm_traffic.json", "r6_gemm_traffic.json"), ("attn_traffic.json", "r6_attn_traffic.json"),
             ("trace_step_cfg1.txt", "r6_trace_step_cfg1.txt"), ("trace_step_es_en_20h.txt", "r6_trace_step_es_en_20h.txt"),
             ("trace_step_cfg1_inline.txt", "r6_trace_step_cfg1_inline.txt"), ("gemm_step_table.txt", "r6_gemm_step_table.txt"), ("ab_side.txt", "r6_ab_side.txt")):
    if os.path.exists(f"{P}/{a}"):
        shutil.copy(f"{P}/{a}", f"profiles/{b}")
fresh = []
for f in ("phase_stamps.log", "phase_stamps_es.log"):
    fresh.append(f"==== {f} (libastk_test.so, ASTK_PERSIST_DBG=8, side stream off)")
    fresh += [l.rstrip("\n") for l in open(f"{P}/{f}") if not l.startswith("{") and ("persist" in l or "bwd_rs" in l or "dec" in l)]
open("profiles/r6_phase_stamps.txt", "w").write("\n".join(fresh) + "\n")
out = {}
for n in ["bench_default", "bench_inline", "bench_deterministic", "bench_bucket_batch", "bench_cfg5", "bench_cfg5_fp16", "bench_cfg5_wide", "bench_b64", "bench_b64_rows16", "bench_b64_mt2", "bench_duo_inline", "bench_mt2_inline", "bench_duo_side", "bench_t1200", "bench_t1680"]:
    lines = [l for l in open(f"{P}/{n}.log") if l.startswith("{")]
    if lines:
        out[n] = json.loads(lines[-1])
json.dump(out, open("profiles/r6_bench_lines.json", "w"), indent=1)
print({k: v["ms_per_step"] for k, v in out.items()})
for k, v in out.items():
    if v.get("epoch"):
        print(k, "epoch", v["epoch"]["frames_per_s_real"], v["epoch"]["frames_per_s_padded"], v["epoch"]["epoch_s"])
