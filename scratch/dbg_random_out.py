import sys, random, copy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from test_gpu_options import _cfg, _oracle_step, _gpu, _rel
from oracle import ast_ref as R
from ast_amd.seq2seq import using_config
cfg = _cfg(); V = cfg["rnn_config"]["dec_vocab_size"]
B, T, D, L = 5, 64, 80, 7
P = R.init_params(cfg, D, V, seed=2, dtype=np.float32)
X, y = R.synth_batch(B, T, D, L, V, seed=3, dtype=np.float32)
for ro in (0.0, 0.5):
    ids = list(np.random.default_rng(1).integers(4, V + 1, size=400)); ids[0] = V
    it1, it2 = iter(ids), iter(ids)
    ref = _oracle_step(cfg, P, X, y, V, 0.7, False, random_out=ro, randint=lambda lo, hi: next(it1), seed=11)
    g = _gpu(cfg, P, D, V)
    g.inject["randint"] = lambda lo, hi: next(it2)
    random.seed(11)
    with using_config("train", True):
        loss = g.forward_loss(X=torch.from_numpy(X), y=torch.from_numpy(y), teach_ratio=0.7, random_out=ro)
        g.cleargrads(); loss.backward()
    grads = g.arena.to_numpy(grads=True)
    print("random_out", ro, "loss", float(loss.data), ref["loss"], "flags", g.use_truth, ref["flags"])
    for k, want in ref["grads"].items():
        print(f"  {k:28s} err {np.abs(grads[k]-want).max():.3e} max {np.abs(want).max():.3e}")
