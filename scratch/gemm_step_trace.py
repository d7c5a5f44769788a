"""Per-launch GEMM table of one train step.  Two phases:
  python scratch/gemm_step_trace.py run      (under: ASTK_GEMM_LOG=1 rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scratch/gemm_step_trace.py run 2> DIR/gemm.log)
  python scratch/gemm_step_trace.py join DIR (joins the kernel trace with the logged shapes of the LAST step)
"""
import csv, glob, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

def run():
    import copy, random, torch, bench
    from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
    from ast_amd import optimizers as O
    from oracle.ast_ref import synth_batch
    cfg = copy.deepcopy(bench.MODEL_CFG)
    B, T, D, L, V = 32, 800, 80, 40, cfg["rnn_config"]["dec_vocab_size"]
    m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
    opt = O.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(m)
    opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
    X, y = synth_batch(B, T, D, L, V, 20)
    X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    random.seed("seed-ast-20h")
    for i in range(3):
        torch.cuda.synchronize()
        print("STEP", i, file=sys.stderr, flush=True)
        with using_config("train", True):
            l = m.forward_loss(X=X, y=y, teach_ratio=0.8, random_out=0, add_noise=0.25)
            m.cleargrads(); l.backward(); opt.update()
    torch.cuda.synchronize()

def join(d):
    log = open(os.path.join(d, "gemm.log")).read().split("STEP 2")[1]
    groups = []
    for ln in log.splitlines():
        m = re.match(r"astk_gemm layout=(\d) M=(\d+) N=(\d+) K=(\d+) batch=(\d+) mode=(\d) twolvl=(\d) group=(\d+)/(\d+) G=(\d+) kt=(\d+)", ln)
        if not m: continue
        lay, M, N, K, bt, mode, tl, gi, gn, G, kt = map(int, m.groups())
        if gi == 0: groups.append(dict(layout=lay, tl=tl, G=G, shapes=[], flops=0.0))
        groups[-1]["shapes"].append(f"{M}x{N}x{K}" + (f"*{bt}" if bt > 1 else ""))
        groups[-1]["flops"] += 2.0 * M * N * K * bt
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm_f32_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(groups)
    rows = rows[-n:]
    tot = 0.0
    for g, r in zip(groups, rows):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += us
        sh = " ".join(g["shapes"][:3]) + (f" (+{len(g['shapes']) - 3})" if len(g["shapes"]) > 3 else "")
        print(f"layout {g['layout']} tl{g['tl']} G={g['G']:4d} {us:8.1f} us {g['flops'] / us / 1e6:7.1f} TF/s  {sh}")
    print(f"total {tot:.1f} us over {n} launches")

if __name__ == "__main__":
    run() if sys.argv[1] == "run" else join(sys.argv[2])
