"""bench.py -- speech frames/s of one full train step (BASELINE.json metric).

step = speech noise -> forward_loss -> cleargrads -> backward -> [RCCL all-reduce of the flat gradient] -> WeightDecay +
GradientClipping + AMSGrad, data already resident in HBM.  Workload (N=1 and per GPU for N>1, weak scaling): BASELINE.json
configs[1]: synthetic fbank (T=800, 80-d) batch 32, 2x[Conv+BN+ReLU] -> 3-layer 2x256 LSTM encoder -> attention ->
1-layer LSTM-512 decoder, V=1098, L=40, shipped training knobs (dropout .3, speech noise .25, teacher forcing .8).

    python bench.py --gpus 1 --steps 50 --warmup 10            # BASELINE configs[1] (1-layer decoder): the metric's workload
    python bench.py --model es_en_20h                          # the shipped model (3 decoder layers, experiments/es_en_20h)
    python bench.py --gpus N                                   # N > 1 without a launcher: starts the N ranks itself (child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import copy
import ctypes as C
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL_CFG = {
    "dropout": {"embed": 0.3, "rnn": 0.3, "out": 0},
    "rnn_config": {"bi_rnn": True, "enc_layers": 3, "dec_layers": 1, "hidden_units": 512, "embedding_units": 128,
                   "attn_units": 512, "n_attn": 1, "feed_attn": True, "ln": False, "dec_vocab_size": 1098},
    "cnn_config": {"bn": True, "cnn_layers": [
        {"in_channels": None, "out_channels": 128, "ksize": [9, 13], "stride": [2, 13], "pad": [4, 0]},
        {"in_channels": None, "out_channels": 512, "ksize": [9, 1], "stride": [2, 1], "pad": [4, 0]}]},
}
TRAIN = {"teach_ratio": float(os.environ.get("ASTK_BENCH_TEACH", "0.8")),      # (shipped: 0.8; the override is for kernel experiments only)
         "speech_noise": 0.25, "lr": 1e-3, "l2": 1e-4, "grad_clip": 2}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA dense peak
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # fp16 / bf16 dense MFMA peak (MI355X_MICROARCH.md; no sparsity)
PRECISIONS = {"fp16x2": "f32 storage / f32 accumulate, products as fp16x2 splits (two fp16 terms per operand behind a power-of-two scale: 22 significant "
                        "bits per operand value, limited exponent range -- NARROWER than the reference's float32; opt-in) in the batched GEMMs and the "
                        "encoder recurrences; decoder loop, attention, softmax-CE, optimizer in IEEE f32",
              "bf16x3": "f32 storage / f32 accumulate; batched GEMMs as bf16x3 splits on the bf16 MFMA pipe (every f32 operand value represented EXACTLY by "
                        "three bf16 terms, f32 exponent range, six term products, each product exact to 2^-26: >= float32 on any data); the encoder "
                        "recurrences multiply the same way (resident weight fragments and each step's activations as exact bf16x3 terms, "
                        "v_mfma_f32_16x16x32_bf16); decoder loop, attention, softmax-CE, optimizer in IEEE f32 (f32-input MFMAs / f32 FMAs)",
              "f32": "IEEE f32 products everywhere (f32-input MFMA v_mfma_f32_32x32x2_f32 / 16x16x4_f32), f32 accumulate"}
SCHEME_OF_MODE = {0: "fp16x2", 1: "bf16x3", 2: "f32"}       # astk_get_gemm_precision()


def synth_batch(B, T, D, L, V, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, T, D)).astype(np.float32)
    y = np.zeros((B, L), dtype=np.int32)
    for b in range(B):
        n = L if b == 0 else int(rng.integers(max(L // 2, 3), L + 1))
        y[b, 0] = 1
        y[b, 1:n - 1] = rng.integers(4, V, size=n - 2)
        y[b, n - 1] = 2
    return X, y


def cpu_baseline(model_cfg, B, T, D, L, V, budget_s=30.0):
    """Times the CPU oracle (the faithful float32 NumPy restatement of the Chainer path, oracle/ast_ref.py) on a bounded
    sample of the same workload: same model, same T/D/L and -- when one step fits the budget -- the same batch.  A batch-2
    step is timed first; the full batch runs only if its predicted cost (measured ratio batch 32 : batch 2 = 6-7x) fits
    `budget_s`, otherwise the largest power-of-two batch that does.  Reported, never the target."""
    import numpy as np
    from oracle import ast_ref as R
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    P = R.init_params(model_cfg, D, V, seed=0, dtype=np.float32)
    opt_cfg = {"type": 0, "lr": TRAIN["lr"], "l2": TRAIN["l2"], "grad_clip": TRAIN["grad_clip"], "grad_noise_eta": 0, "freeze": []}

    def one(Bs):
        m = R.RefModel(model_cfg, {k: v.copy() for k, v in P.items()}, V)
        m.masks = R.RecordingMasks(1)
        opt = R.RefOptimizer(m, opt_cfg)
        X, y = synth_batch(Bs, T, D, L, V, 20)
        noise = np.random.default_rng(1).normal(1.0, TRAIN["speech_noise"], X.shape).astype(np.float32)
        t0 = time.time()
        R.train_step(m, opt, X, y, TRAIN["teach_ratio"], add_noise=TRAIN["speech_noise"], noise=noise, pyrandom=random.Random("seed-ast-20h"))
        return time.time() - t0
    b2 = min(2, B)
    one(b2)                    # BLAS / allocator warm-up
    t2 = one(b2)
    Bs, best = b2, t2
    cand = B
    while cand > 2 and t2 * (0.6 + 0.2 * cand) > budget_s:      # predicted cost of a step at batch `cand`
        cand //= 2
    if cand > 2:
        Bs, best = cand, one(cand)
    return {"value": round(Bs * T / best, 1), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"one full train step of the same model at batch {Bs} (GPU batch {B}; T={T}, D={D}, L={L}) in {best:.2f} s, after a "
                      f"batch-{b2} step of {t2:.2f} s = {round(b2 * T / t2, 1)} frames/s; float32 NumPy restatement of the Chainer "
                      "path, per-step Python loop and one BLAS sgemm per Linear like Chainer-on-NumPy (Chainer is not installable offline)"}


def bucket_plan(path, B, tokens_per_word, bucket_batch=None, num_b=20, width_b=80, max_sp=1680, max_pred=175, seed="seed-ast-20h"):
    """The Fisher-20h training set cut the way the reference cuts it: bucket = min(frames // width_b, num_b - 1) (prep_buckets.py:52), every
    bucket shuffled and sliced into batches (dataloader.py:127-140; the OLD path's per-bucket sizes nmt_run.py:421-426 with `bucket_batch`),
    every batch padded to its longest utterance (<= max_sp frames, dataloader.py:95-108) and to its longest target
    [GO] + ids[:max_pred - 2] + [EOS] (dataloader.py:139-147), target length = round(tokens_per_word x words).  Returns one row per bucket:
    utterances, batch size, batches, the PADDED extent a step of that bucket is timed at (T = the bucket's upper edge, L = the mean batch
    maximum), real and padded frames."""
    import math
    d = json.load(open(path))
    frames, words = d["frames"]["fisher_train"], d["en_w"]["fisher_train"]
    rnd = random.Random(seed)
    buckets = [[] for _ in range(num_b)]
    for f, wd in zip(frames, words):
        buckets[min(f // width_b, num_b - 1)].append((min(f, max_sp), min(max_pred, 2 + int(math.ceil(tokens_per_word * wd)))))
    rows = []
    for b, bk in enumerate(buckets):
        if not bk:
            continue
        size = B
        if bucket_batch:
            size = bucket_batch[0] if b < num_b // 3 else bucket_batch[1] if b < (num_b * 2) // 3 else bucket_batch[2]
        rnd.shuffle(bk)
        batches = [bk[i:i + size] for i in range(0, len(bk), size)]
        Lmean = sum(max(u[1] for u in bt) for bt in batches) / len(batches)
        Tb = min((b + 1) * width_b, max_sp) if b < num_b - 1 else max_sp
        rows.append({"bucket": b, "utts": len(bk), "batch": size, "batches": len(batches), "T": Tb, "L": max(3, int(round(Lmean))),
                     "real_frames": sum(u[0] for u in bk), "rows": sum(len(bt) for bt in batches)})
    return rows


def histogram_leg(w, args, B, T, L, barrier):
    """One timed train step per bucket (same model, optimizer, scheme and step definition as the headline), weighted by the batches per
    bucket: what an EPOCH of the reference's es_en_20h training set costs, not one bucket of it."""
    bb = [int(x) for x in args.bucket_batch.split(",")] if args.bucket_batch else None
    rows = bucket_plan(args.histogram, B, args.tokens_per_word, bb)
    total_ms = real = padded = 0.0
    table = []
    for r in rows:
        w.set_batch(r["batch"], r["T"], r["L"], seed=100 + r["bucket"])
        dt, _ = w.timed(3, args.hist_steps)
        ms = dt / args.hist_steps * 1e3
        # (a bucket's last batch is short: priced at its share of a full batch's time -- the latency-bound kernels do not care, so this is
        #  slightly optimistic, by less than one batch per bucket)
        full = r["rows"] / r["batch"]
        total_ms += ms * full
        real += r["real_frames"]
        padded += r["rows"] * r["T"]
        table.append([r["bucket"], r["utts"], r["batch"], r["T"], r["L"], round(ms, 3)])
    return {"what": "one epoch of the reference's Fisher-20h training set (17 306 utterances; tests/golden/fisher_20h_frames.json) on the model of "
                    + w.name + ": one timed step per bucket x batches per bucket",
            "tokens_per_word": args.tokens_per_word, "bucket_batch": bb or B,
            "columns": ["bucket", "utts", "batch", "T", "L", "ms_per_step"], "buckets": table,
            "epoch_s": round(total_ms * 1e-3, 3), "frames_per_s_real": round(real / (total_ms * 1e-3), 1),
            "frames_per_s_padded": round(padded / (total_ms * 1e-3), 1), "real_frames": int(real)}


def self_launch_command(gpus, argv, port=None):
    """The command `python bench.py --gpus N` (N > 1, no launcher in the environment) runs as a CHILD process: one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), the same arguments."""
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(args, argv):
    """N > 1 ranks asked for, but this process was not started by a launcher (WORLD_SIZE unset): start them, relay rank 0's JSON line,
    exit with the launcher's code.  A child process, never an exec; nothing here has touched HIP / torch.cuda yet."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = self_launch_command(args.gpus, argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this stack (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    raise SystemExit(rc if rc != 0 or line is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # >= 50 timed steps after >= 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="cfg1", choices=["cfg1", "es_en_20h", "cfg5"],
                    help="cfg1 = BASELINE configs[1] (1-layer LSTM-512 decoder, the metric's workload); es_en_20h = the shipped "
                         "experiments/es_en_20h model (3 decoder layers) on the same synthetic batch; cfg5 = the shape of BASELINE "
                         "configs[4] (6-layer encoder, hidden 1024 = 512 per direction, V = 8004): use with --gemm-operands fp16")
    ap.add_argument("--gemm-operands", default="f32", choices=["f32", "fp16"],
                    help="fp16: operands of the CNN / encoder-input GEMMs rounded to fp16, f32 accumulation (configs[4]); f32 (default): "
                         "f32-accurate products")
    ap.add_argument("--hidden", type=int, default=0, help="cfg5 only: hidden_units (= attn_units) override, e.g. 2048 = 1024 per direction, the "
                    "other reading of BASELINE configs[4] (encoder and decoder then run their per-launch fall-back kernels: see `paths`)")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--feat", type=int, default=80)
    ap.add_argument("--tgt-len", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="N>1: BatchNorm statistics over the global batch (4 extra 5-10 KB all-reduces)")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra steps with per-kernel HIP-event timing")
    ap.add_argument("--no-alt-precisions", action="store_true", help="skip the re-timings of the same step under the other two arithmetic schemes")
    ap.add_argument("--precision", default=None, choices=["bf16x3", "f32", "fp16x2"],
                    help="arithmetic of the batched products for the headline (default: the library's default, bf16x3 = exact f32 operands "
                         "as three bf16 terms; f32 = f32-input MFMAs; fp16x2 = two scaled fp16 terms, narrower than float32)")
    ap.add_argument("--no-also", action="store_true", help="skip the second workload (the shipped es_en_20h model) of the default cfg1 run")
    ap.add_argument("--histogram", default=os.path.join(ROOT, "tests", "golden", "fisher_20h_frames.json"),
                    help="corpus description (per-utterance frame and word counts of the reference's data/fisher/fisher_20h.info, extracted by "
                         "tests/golden/make_fisher_frames.py): one timed step per bucket of the reference's bucketing rule (prep_buckets.py:52, "
                         "20 buckets of 80 frames), weighted by the batches per bucket => epoch frames/s on the REAL shape distribution "
                         "(`epoch` in the JSON line); 'none' skips the leg")
    ap.add_argument("--hist-steps", type=int, default=8, help="timed steps per bucket of the --histogram leg (after 3 warm-ups)")
    ap.add_argument("--tokens-per-word", type=float, default=1.3, help="--histogram: BPE-1k tokens per English word (the token counts live in the "
                    "LDC-licensed text the reference does not ship; the .info file holds word counts)")
    ap.add_argument("--bucket-batch", default=None, help="--histogram: per-bucket batch sizes 'max,med,min' of the reference's OLD path "
                    "(nmt_run.py:421-426: first third of the buckets / second third / rest), e.g. 64,48,32; default: --batch everywhere")
    args = ap.parse_args()
    maybe_self_launch(args, sys.argv[1:])

    import numpy as np
    import torch
    from ast_amd import _lib, dist as adist
    from ast_amd import optimizers as O
    from ast_amd.seq2seq import SpeechEncoderDecoder, using_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    torch.cuda.set_device(local)
    # ASTK_BENCH_FORCE_DP=1: a process group of ONE rank takes the data-parallel branch (bucketed asynchronous all-reduces, 1/world on the
    # fly, rank-dependent seeds) against the real RCCL backend -- the only multi-rank code a one-GPU box can execute (tests/)
    dp = world > 1 or os.environ.get("ASTK_BENCH_FORCE_DP", "0") == "1"
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29613")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # persistent grids: encoder (h/16) x ceil(B/16) x cells; the decoder loop takes every CU but never overlaps a collective
        rc = MODEL_CFG["rnn_config"]
        need = (rc["hidden_units"] // 2 // 16) * ((args.batch + 15) // 16) * 2 * rc["enc_layers"]
        chan_cap = adist.reserve_cus_for_recurrence(min(need, 224))
        import torch.distributed as td
        if world > 1:
            adist.init("nccl")
        else:
            td.init_process_group("nccl", world_size=1, rank=0)
            adist.is_distributed = lambda: True
    lib = _lib.load()
    B, T, D, L = args.batch, args.frames, args.feat, args.tgt_len
    # the headline's arithmetic: the library default (bf16x3) unless --precision names another scheme; it travels in the op descriptors
    base_scheme = args.precision or SCHEME_OF_MODE[lib.astk_get_gemm_precision()]
    compute = torch.cuda.Stream()
    # The step runs on a stream of its own, not on the legacy default stream (measured 0.05 ms per step faster; also what the
    # side stream of ast_amd/seq2seq.py needs: the legacy default stream synchronises implicitly with every other stream).

    def model_cfg(name):
        cfg = copy.deepcopy(MODEL_CFG)
        if name == "es_en_20h":
            cfg["rnn_config"]["dec_layers"] = 3          # /root/reference/experiments/es_en_20h/model_cfg.json:12
        if name == "cfg5":                               # BASELINE configs[4]: "BiLSTM-1024" read as the concatenated width (512 per direction)
            hu = args.hidden if args.hidden > 0 else 1024
            cfg["rnn_config"].update(enc_layers=6, hidden_units=hu, attn_units=hu, dec_vocab_size=8004)
        return cfg

    def barrier():
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    class Workload:
        """One model + optimizer + resident synthetic batch; step() = the whole train step of SURVEY.md 8(d)."""

        def __init__(self, name):
            self.name, self.cfg = name, model_cfg(name)
            self.V = self.cfg["rnn_config"]["dec_vocab_size"]
            m = self.model = SpeechEncoderDecoder(local, self.cfg).materialize(D, seed=0)       # identical replicas
            m.gemm_precision = base_scheme
            m.gemm_operands = args.gemm_operands
            o = self.opt = O.Adam(alpha=TRAIN["lr"], beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(m)
            o.add_hook(O.WeightDecay(TRAIN["l2"]))
            o.add_hook(O.GradientClipping(TRAIN["grad_clip"]))
            if dp:
                # overlapped exchange: decoder / encoder / CNN gradient ranges are all-reduced as soon as their backward is enqueued
                m.grad_buckets = adist.make_grad_buckets(m)
                o.grad_sync = m.grad_buckets.finish
                m.rng_seed = (m.rng_seed + 0x9E3779B97F4A7C15 * rank) & 0xFFFFFFFFFFFFFFFF   # per-replica dropout / noise streams
                if args.sync_bn:
                    m.stat_exchange = adist.StatExchange()
            random.seed("seed-ast-20h")                                       # same teacher-forcing stream on every rank
            Xh, yh = synth_batch(B, T, D, L, self.V, 20 + rank)               # each rank owns its shard of the global batch
            self.X, self.y = torch.from_numpy(Xh).cuda(), torch.from_numpy(yh).cuda()
            torch.cuda.synchronize()

        def set_batch(self, Bs, Ts, Ls, seed=20):
            """Another resident synthetic batch (the --histogram leg: one shape per bucket); the model's buffers are grow-only pools."""
            Xh, yh = synth_batch(Bs, Ts, D, Ls, self.V, seed + rank)
            self.X, self.y = torch.from_numpy(Xh).cuda(), torch.from_numpy(yh).cuda()
            torch.cuda.synchronize()

        def step(self):
            with torch.cuda.stream(compute), using_config("train", True):
                loss = self.model.forward_loss(X=self.X, y=self.y, teach_ratio=TRAIN["teach_ratio"], random_out=0, add_noise=TRAIN["speech_noise"])
                self.model.cleargrads()
                loss.backward()
                self.opt.update()
            return loss

        def timed(self, warmup, steps):
            """W untimed warm-ups, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
            for _ in range(warmup):
                loss = self.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = self.step()
            barrier()
            dt = time.perf_counter() - t0
            if dp:
                tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                dt = float(tt.item())
            lv = float(loss)                    # raises AstkError if a persistent kernel timed out (status word next to the loss)
            assert np.isfinite(lv), "loss is not finite"
            return dt, lv

        def profile(self, nsteps):
            """`nsteps` extra steps with per-kernel HIP-event timing on the launch stream (astk_prof_*): res[] of include/astk.h."""
            m = self.model
            # per-kernel figures are taken with every kernel alone on the device: with the side-stream work (decoder parameter gradients
            # and the time-chunked layer-0 products on the CUs the recurrences leave free) those GEMMs would be priced as slow ones
            overlap, m.side_stream_on = m.side_stream_on, False
            lib.astk_prof_begin()
            for _ in range(nsteps):
                self.step()
            torch.cuda.synchronize()
            m.side_stream_on = overlap
            res = (C.c_double * 24)()
            lib.astk_prof_end(res)
            return res

        def describe(self):
            rc = self.cfg["rnn_config"]
            return ({"cfg1": "BASELINE configs[1]: ", "es_en_20h": "shipped es_en_20h model (configs[0]'s model at configs[1]'s batch): ",
                     "cfg5": f"shape of BASELINE configs[4] (6-layer encoder, 2x{rc['hidden_units'] // 2}, V=8004): "}[self.name] +
                    f"synthetic fbank T={T} D={D} batch {B}/GPU, 2xConv+BN -> {rc['enc_layers']}-layer 2x{rc['hidden_units'] // 2} LSTM enc -> "
                    f"attention -> {rc['dec_layers']}-layer LSTM-{rc['hidden_units']} dec, V={self.V}, L={L}, dropout .3, noise .25, teach .8, Adam(amsgrad)+L2+clip")

    w = Workload(args.model)
    model, cfg, V = w.model, w.cfg, w.V
    dt, loss_val = w.timed(args.warmup, args.steps)
    ms = dt / args.steps * 1e3
    value = world * B * T / (dt / args.steps)

    # ---- which kernels ran (a silent fall-back to the per-launch paths must show in the driver's line)
    def paths_of(model):
        st = model._cur
        epath = lib.astk_lstm_stack_path(C.byref(st["ld"]))
        paths = {"encoder": {1: "persistent wavefront kernels (one launch for all steps of all cells)",
                             2: "persistent kernels, hoisted form (h = 1024: one launch per layer for all steps, input projections and down gradients as batched products)"}.get(
                                 epath, "per-step fused-cell launches (fallback)")}
        dpath = lib.astk_decoder_path(C.byref(st["dd"]))
        paths["decoder"] = (f"persistent loop, {dpath >> 8} layer(s) fused" + (", attention phase specialised (H=512, chunk<=32)" if dpath & 2 else ", generic attention phase")
                            + (", two launches over halves of the batch rows" if dpath & 4 else "")) \
            if dpath & 1 else ("persistent loops of the wide decoder (decoder_wide.hip): one forward and one backward launch for all steps"
                               if dpath & 16 else "per-launch loop (fallback)")
        paths["encoder_persistent"], paths["decoder_persistent"] = bool(lib.astk_lstm_stack_path(C.byref(st["ld"]))), bool(dpath & 17)
        paths["cu_count"] = int(lib.astk_device_cu_count())
        paths["side_stream"] = bool(model._side is not None)
        paths["free_cus_beside_recurrences"] = int(lib.astk_lstm_stack_free_cus(C.byref(st["ld"])))
        return paths
    paths = paths_of(model)

    def profile_json(name):
        f = os.path.join(ROOT, "profiles", name)
        return json.load(open(f)) if os.path.exists(f) else {}

    def kernel_extras(res, nsteps, T2):
        extra = {}
        if res[5] > 0:
            extra["gemm_family_ms_per_step"] = round(res[4] / nsteps, 3)
        if res[8] > 0:
            extra["encoder_lstm_persistent_ms_per_step"] = round(res[7] / nsteps, 3)
            extra["encoder_us_per_time_step"] = round(res[7] / nsteps * 1e3 / 2 / T2, 2)     # fwd + bwd launches
        if res[17] > 0:
            extra["decoder_persistent_ms_per_step"] = round((res[16] + res[18]) / nsteps, 3)
            S_ = L - 1
            extra["decoder_us_per_decoder_step"] = {"fwd": round(res[16] / res[17] * 1e3 / S_, 2), "bwd": round(res[18] / res[19] * 1e3 / S_, 2)}
        return extra

    # ---- per-kernel timing with HIP events on the launch stream (in situ), plus in-kernel phase stamps of the persistent decoder
    roof, scan, extra = None, None, {}
    if args.profile_steps > 0:
        res = w.profile(args.profile_steps)
        T2 = model._cur["T2"]
        H = cfg["rnn_config"]["hidden_units"]
        S = L - 1
        bytes_scan = B * T2 * H * 4                                    # one streaming read of enc_states
        if res[5] > 0:
            # THE dominant kernel family: every batched dense product of the step.  `achieved` / `frac` price what the matrix pipe EXECUTES
            # against the pipe that executes it: under bf16x3 every f32 operand is split in the kernel into three bf16 terms and a useful
            # product is six v_mfma_f32_32x32x16_bf16 -- 6x the useful flops on the 2.5 PFLOP/s bf16 pipe (fp16x2: 3x on the fp16 pipe;
            # f32: 1x on the 157.3 TFLOP/s f32-input MFMA pipe).  `useful_*` carries the algorithmic 2*M*N*K figure and its fraction of
            # the executing pipe's peak -- the number to compare across schemes.
            tfl = res[6] / (res[4] * 1e-3) / 1e12
            ms_gemm, n_gemm = res[4] / args.profile_steps, int(res[5] / args.profile_steps)
            nprod, peak, pipe = {"fp16x2": (3, MFMA_16BIT_PEAK_TFLOPS, "fp16 MFMA (v_mfma_f32_32x32x16_f16), 3 per useful 16-k product block (fp16x2 split)"),
                                 "bf16x3": (6, MFMA_16BIT_PEAK_TFLOPS, "bf16 MFMA (v_mfma_f32_32x32x16_bf16), 6 per useful 16-k product block (bf16x3 split)"),
                                 "f32": (1, MFMA_F32_PEAK_TFLOPS, "f32-input MFMA (v_mfma_f32_32x32x2_f32)")}[base_scheme]
            tj = (profile_json("r6_gemm_traffic.json") or profile_json("r5_gemm_traffic.json") or profile_json("r4_gemm_traffic.json") or {}).get(base_scheme) or {}
            # `achieved` / `frac` = ALGORITHMIC flops (2*M*N*K) over the launches' time against the dense peak of the pipe that executes
            # them -- the task's definition; `executed_*` = the same rate times the MFMAs a useful product costs under the scheme (what
            # the matrix pipe really does: 6x under bf16x3), `algorithmic_over_f32_mfma_peak` = the same algorithmic rate against what
            # the native f32-input MFMA pipe could deliver at best (157.3 TFLOP/s) -- the ceiling an f32 computation has WITHOUT the split.
            roof = {"bound": "mfma", "kernel": "gemm_f32_kernel<NT|NN|TN> (+ k_zero_split_tiles; fp16x2 also k_absmax): all batched dense products of the step with their preparation launches",
                    "scheme": base_scheme, "pipe": pipe,
                    "achieved": round(tfl, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tfl / peak, 4),
                    "executed_tflops": round(tfl * nprod, 2), "executed_frac_of_pipe_peak": round(tfl * nprod / peak, 4),
                    "algorithmic_over_f32_mfma_peak": round(tfl / MFMA_F32_PEAK_TFLOPS, 4),
                    "traffic": tj.get("hbm_bytes_per_step"),
                    "traffic_source": ("STATIC FILE, not measured in this run: " + tj.get("source", "profiles/r6_gemm_traffic.json")) if tj else None,
                    "mfma_pipe_busy_frac_pmc": tj.get("mfma_pipe_busy_frac"),
                    # logical operand / result bytes (window operands of the conv GEMMs counted at their im2col extent, not at the
                    # smaller extent of the arrays they address); `traffic` is what crosses the L2's memory side, Infinity-Cache hits
                    # included: the stream-K workgroups of an XCD sit at unrelated k positions, so a panel is re-fetched by every tile
                    "algorithmic_bytes_per_step": round(res[20] / args.profile_steps),
                    "flops_per_step": round(res[6] / args.profile_steps), "ms_per_step": round(ms_gemm, 3),
                    "avg_launch_us": round(ms_gemm * 1e3 / max(1, n_gemm), 1), "launches_per_step": n_gemm,
                    "method": "algorithmic flops = 2*M*N*K per launch (summed by the launcher), executed = algorithmic x MFMAs per product block; time = HIP "
                              "events around every launch (preparation launches included) on the launch stream, live in this run"}
            if args.gemm_operands == "fp16":
                roof["pipe"] += "; single-term fp16 (1 MFMA) for the launches marked eligible (--gemm-operands fp16): executed flops are over-counted for those"
        if res[17] > 0 and res[19] > 0:
            # Attention scan (north_star's "HBM roofline on the attention scan").  The scan is a PHASE of the two persistent decoder
            # launches, not a launch of its own, and its enc / encA slices are LDS-resident after step 0: the launches are LATENCY-bound
            # (4-5 dependent hand-offs per decoder step), not HBM-bound.  Kernel-level: algorithmic bytes of all S scans of a launch
            # over the launch's duration.  Phase-level (in-kernel stamps, hand-off satisfied -> partial published) is given for
            # reference only: it is an HBM-EQUIVALENT rate of an on-chip pass, not HBM traffic.
            k_fwd_us, k_bwd_us = res[16] / res[17] * 1e3, res[18] / res[19] * 1e3
            attn_traffic = profile_json("r6_attn_traffic.json") or profile_json("r5_attn_traffic.json") or profile_json("r4_attn_traffic.json") or profile_json("r3_attn_traffic.json") or profile_json("attn_traffic.json")
            ach = S * bytes_scan / ((k_fwd_us + k_bwd_us) / 2 * 1e-6) / 1e9
            scan = {"bound": "latency (priced against the HBM roofline north_star names; slices are LDS-resident, measured HBM traffic < algorithmic bytes)",
                    "kernel": "decoder_persist_fwd / decoder_persist_bwd (all S decoder steps per launch)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": attn_traffic.get("hbm_bytes_per_launch"), "traffic_per_scan": attn_traffic.get("hbm_bytes_per_scan"),
                    "traffic_source": "STATIC FILE, not measured in this run: " + attn_traffic.get("source", "profiles/attn_traffic.json") if attn_traffic else None,
                    "bytes_per_launch": S * bytes_scan, "avg_launch_us": round((k_fwd_us + k_bwd_us) / 2, 1), "launches_per_step": 2,
                    "us_per_decoder_step": {"fwd": round(k_fwd_us / S, 2), "bwd": round(k_bwd_us / S, 2)},
                    "method": "algorithmic bytes (S scans x B*T''*H*4) / HIP-event duration of the launch",
                    "phase": None}
            if res[11] > 0 and res[14] > 0:
                us = 0.5 * (res[10] + res[13])
                scan["phase"] = {"what": "scan phase only, in-kernel s_memrealtime stamps (slowest workgroup's mean), HBM-equivalent rate of an LDS-resident pass",
                                 "bytes": bytes_scan, "us": round(us, 3), "equiv_GBps": round(bytes_scan / (us * 1e-6) / 1e9, 1),
                                 "equiv_frac_of_hbm_peak": round(bytes_scan / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                 "fwd_mean_us": round(res[9], 3), "fwd_slowest_wg_us": round(res[10], 3),
                                 "bwd_mean_us": round(res[12], 3), "bwd_slowest_wg_us": round(res[13], 3)}
        else:
            n_attn, ms_attn = res[1] + res[3], res[0] + res[2]
            if n_attn > 0:
                avg_us = ms_attn / n_attn * 1e3
                ach = bytes_scan / (avg_us * 1e-6) / 1e9
                scan = {"bound": "hbm", "kernel": "attn_fwd_partial+attn_bwd_partial (per-launch decoder loop)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": bytes_scan,
                        "avg_launch_us": round(avg_us, 3), "launches_per_step": int(n_attn / args.profile_steps)}
        extra = kernel_extras(res, args.profile_steps, T2)

    # ---- the SAME step, same process, under the other two arithmetic schemes (the model's gemm_precision -> the descriptors' precision
    # field): driver-timed figures for the literal f32 MFMA chain and for the narrower opt-in fp16x2 split next to the headline's
    alt = []
    if not args.no_alt_precisions:
        for scheme in ("bf16x3", "f32", "fp16x2"):
            if scheme == base_scheme:
                continue
            model.gemm_precision = scheme
            d1, l2_ = w.timed(args.warmup, args.steps)       # (round 5: the headline's warm-ups, not 3 -- a scheme's first steps load its kernels' code objects and run at unsettled clocks)
            alt.append({"scheme": scheme, "dtype": PRECISIONS[scheme], "steps": args.steps, "ms_per_step": round(d1 / args.steps * 1e3, 3),
                        "value": round(world * B * T / (d1 / args.steps), 1), "loss": round(l2_, 4)})
        model.gemm_precision = base_scheme

    # ---- second workload of the default run: the SHIPPED es_en_20h model (3 decoder layers: the model north_star's >= 50x target is
    # quoted on), same batch, same scheme, same step definition, `--steps` timed steps after `--warmup` warm-ups
    also = []
    w2 = None
    if args.model == "cfg1" and not args.no_also:
        w2 = Workload("es_en_20h")
        d2, l2v = w2.timed(max(1, args.warmup), args.steps)
        e2 = {"workload": w2.describe(), "precision": base_scheme, "steps": args.steps, "ms_per_step": round(d2 / args.steps * 1e3, 3),
              "value": round(world * B * T / (d2 / args.steps), 1), "unit": "frames/s", "loss": round(l2v, 4), "paths": paths_of(w2.model)}
        if args.profile_steps > 0:
            e2["kernels"] = kernel_extras(w2.profile(args.profile_steps), args.profile_steps, w2.model._cur["T2"])
        also.append(e2)

    # ---- the reference's REAL shape distribution (round 6): one timed step per bucket of the Fisher-20h training set
    epoch = None
    if args.histogram and args.histogram != "none" and os.path.exists(args.histogram) and world == 1:
        epoch = histogram_leg(w2 if w2 is not None else w, args, B, T, L, barrier)
        (w2 if w2 is not None else w).set_batch(B, T, L)

    out = {"metric": "speech frames/s (train step)", "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": PRECISIONS[base_scheme] + ("" if args.gemm_operands == "f32" else "; single-term fp16 operands in the CNN / encoder-input / decoder-output GEMMs (--gemm-operands fp16)"),
           "precision": base_scheme,
           "data": "synthetic",
           "config": {"workload": w.describe(),
                      "global_batch": world * B, "frames": T, "feat_dim": D, "tgt_len": L, "parallelism": f"dp{world}", "batchnorm": "global-batch statistics" if (world > 1 and args.sync_bn) else "per-replica statistics",
                      "launches": "plain stream launches (the step is GPU-bound: the host enqueues it in 0.6 ms)"},
           "loss": round(loss_val, 4), "alt_precisions": alt, "also": also, "paths": paths, "roofline": roof, "roofline_scan": scan}
    if dp:
        out["dp"] = {"rccl_ranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                     "buckets": list(model.grad_buckets.ranges), "bucket_launch": "dec+enc behind the encoder recurrence, cnn behind the CNN backward",
                     "NCCL_MAX_NCHANNELS": chan_cap, "recurrence_grid_cus": need}
    out.update({"kernels": extra} if extra else {})
    if epoch is not None:
        out["epoch"] = epoch
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            # bounded CPU samples (the NumPy oracle on this box's host cores): ~20 s each for the headline's model and the shipped one,
            # both at the GPU's batch
            out["cpu_baseline"] = cpu_baseline(cfg, B, T, D, L, V)
            if also:
                also[0]["cpu_baseline"] = cpu_baseline(w2.cfg, B, T, D, L, w2.V, budget_s=25.0)
        if also:
            # LAST key of the line (a record that keeps only the line's tail still shows the model north_star's >= 50x target names):
            # the shipped es_en_20h model, compact
            e2 = also[0]
            cb = e2.get("cpu_baseline")
            out["es_en_20h"] = {"model": "shipped es_en_20h (3 decoder layers), same batch / scheme / step as the headline", "precision": base_scheme,
                                "ms_per_step": e2["ms_per_step"], "value": e2["value"], "unit": "frames/s",
                                "decoder_us_per_step": (e2.get("kernels") or {}).get("decoder_us_per_decoder_step"),
                                "decoder_ms_per_step": (e2.get("kernels") or {}).get("decoder_persistent_ms_per_step"),
                                "paths": {k: e2["paths"][k] for k in ("encoder_persistent", "decoder_persistent", "side_stream") if k in e2["paths"]},
                                "cpu_baseline": ({"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                                  "sample": cb["sample"].split(" in ")[0]} if cb else None),
                                "gpu_over_cpu": round(e2["value"] / cb["value"], 1) if cb else None,
                                "epoch_frames_per_s_real_shapes": (epoch or {}).get("frames_per_s_real")}
        print(json.dumps(out), flush=True)
    if dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
