"""Soak / repeatability run: N different batches through forward + backward at FIXED weights, the whole sequence twice; per-batch loss and clip
norm of the two passes must agree (the only run-to-run freedom is the order of float atomic adds), no persistent kernel may time out, nothing
may be non-finite.  A race like the softmax-CE one shows up here as an outlier.
    python3 scratch/soak.py <cfg1|es_en_20h|cfg5> <batches> [det=1] [stream=1] [batch=64] [tuning knobs like dec.persist=0]
det=1: the model's deterministic backward (astk.h `deterministic`): then EVERY gradient must be bit-identical between the two passes (a
64-bit checksum of the whole gradient arena per batch), not just the norm to 2e-5 -- a backward hand-off race shows as a bit difference.
stream=1: the steps run on a stream of their own (what bench.py and NN.train_epoch do), so that the side-stream work beside the
recurrences is part of the soak."""
import copy, os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from ast_amd import _lib, optimizers as O
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from oracle.ast_ref import synth_batch
which, N = sys.argv[1], int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:])
DET, OWN_STREAM, BATCH = opts.pop("det", "0") == "1", opts.pop("stream", "0") == "1", int(opts.pop("batch", "32"))
for k, v in opts.items():
    _lib.set_tuning(k, float(v))
cfg = copy.deepcopy(bench.MODEL_CFG)
V = cfg["rnn_config"]["dec_vocab_size"]
if which == "es_en_20h":
    cfg["rnn_config"]["dec_layers"] = 3
if which == "cfg5":
    cfg["rnn_config"].update(enc_layers=6, hidden_units=1024, attn_units=1024, dec_vocab_size=8004); V = 8004
B, T, D, L = BATCH, 800, 80, 40
m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
m.deterministic = DET
compute = torch.cuda.Stream() if OWN_STREAM else torch.cuda.current_stream()
opt = O.Adam(alpha=1e-3, amsgrad=True).setup(m)
opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
batches = []
for i in range(8):                                   # 8 distinct batches, cycled (host RAM); the teacher-forcing stream makes every step different
    X, y = synth_batch(B, T, D, L, V, 100 + i)
    batches.append((torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()))
lib = _lib.load()
import ctypes as C
def run():
    random.seed("soak")
    m.rng_seed = 12345; m._rng_offset = 0
    out = []
    for i in range(N):
        X, y = batches[i % 8]
        with torch.cuda.stream(compute), using_config("train", True):
            l = m.forward_loss(X=X, y=y, teach_ratio=float(os.environ.get('SOAK_TEACH', 0.8)), random_out=0, add_noise=0.25)
            m.cleargrads(); l.backward()
            gr = m.arena.grad
            g = float(torch.sqrt((gr.double() ** 2).sum()))
            h = int(gr.view(torch.int32).to(torch.int64).mul(torch.arange(1, gr.numel() + 1, device=gr.device) % 1000003).sum())      # bit checksum of every gradient
        out.append((float(l.data), g, float(h % (1 << 52))))
    return np.asarray(out)
a = run(); b = run()
mask = C.c_uint(0); lib.astk_persist_status(C.byref(mask), 1)
rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-9)
print("bit-identical gradient arenas:", int((a[:, 2] == b[:, 2]).sum()), "of", N, "(det=%d stream=%d side=%s)" % (DET, OWN_STREAM, m._side is not None))
print(which, sys.argv[3:], "batches", N, "finite", bool(np.isfinite(a).all() and np.isfinite(b).all()), "status", mask.value,
      "max rel diff loss %.2e grad-norm %.2e" % (rel[:, 0].max(), rel[:, 1].max()), "loss sum %.6f grad-norm sum %.6f" % (a[:, 0].sum(), a[:, 1].sum()), "loss range", a[:, 0].min(), a[:, 0].max(), "paths", m.paths()["decoder"][:40])
bad = np.flatnonzero((rel[:, :2] > 2e-5).any(axis=1))
print("outliers", bad[:10], a[bad[:5]], b[bad[:5]])
