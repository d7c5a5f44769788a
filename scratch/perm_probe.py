"""Which tensor breaks the batch-permutation property (tests/test_gpu_model.py) under a given ASTK_GEMM_X3_BELOW?"""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from oracle.ast_ref import synth_batch
T, dec_layers, D = 800, 3, 13
cfg = copy.deepcopy(bench.MODEL_CFG)
cfg["dropout"] = {"embed": 0.0, "rnn": 0.0, "out": 0}
cfg["rnn_config"]["dec_layers"] = dec_layers
B, L, V = 32, 40, cfg["rnn_config"]["dec_vocab_size"]
X, y = synth_batch(B, T, D, L, V, 20, dtype=np.float32)
X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
m.inject = {"use_truth": [1] * (L - 1)}
def run(Xb, yb):
    with using_config("train", True):
        l = m.forward_loss(X=Xb, y=yb, teach_ratio=1.0, random_out=0, add_noise=0)
        m.cleargrads()
        l.backward()
    torch.cuda.synchronize()
    return float(l.data), m.arena.grad.clone()
perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).cuda()
Xp, yp = X[perm].contiguous(), y[perm].contiguous()
if os.environ.get("PERTURB"):
    gen = torch.Generator(device="cuda").manual_seed(11)
    Xp = Xp * (1.0 + float(os.environ["PERTURB"]) * torch.randn(Xp.shape, device="cuda", generator=gen))
a = m.arena
off = a.offsets["CNN_1/W"]; n = int(np.prod(a.shapes["CNN_1/W"]))
tag = os.environ.get("ASTK_GEMM_X3_BELOW", "def")
order = os.environ.get("ORDER", "opop")
res = []
for k, ch in enumerate(order):
    l, g = run(X, y) if ch == "o" else run(Xp, yp)
    res.append(g[off:off+n].cpu().numpy())
    np.save(f"gpurun_out/perm_{tag}_{order}_{k}.npy", res[-1])
print("done", tag, order)
