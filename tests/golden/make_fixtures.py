"""Generates the golden vectors under tests/golden/ from the float64 CPU oracle (oracle/ast_ref.py).

    python tests/golden/make_fixtures.py

The reference ships no test vectors and Chainer is not installable offline, so these fixtures are produced by the
restatement itself (pinned by finite differences and the independent torch restatement, tests/test_oracle.py);
they freeze its results so that later edits to the oracle or the kernels cannot drift silently.  Each .npz holds:
inputs (X, y, teacher-forcing flags), Chainer-layout weights, the loss, the clip norm sqrt(sum (g + l2 p)^2),
every gradient, and the parameters after 3 AMSGrad steps on the same batch."""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ast_ref as R  # noqa: E402

OPT = {"type": 0, "lr": 1e-3, "l2": 1e-4, "grad_clip": 2, "grad_noise_eta": 0, "freeze": []}


def small_cfg(V):
    return {"dropout": {"embed": 0.0, "rnn": 0.0, "out": 0},
            "rnn_config": {"bi_rnn": True, "enc_layers": 3, "dec_layers": 3, "hidden_units": 16, "embedding_units": 8,
                           "attn_units": 16, "n_attn": 1, "feed_attn": True, "ln": False, "dec_vocab_size": V},
            "cnn_config": {"bn": True, "cnn_layers": [
                {"in_channels": None, "out_channels": 8, "ksize": [9, 13], "stride": [2, 13], "pad": [4, 0]},
                {"in_channels": None, "out_channels": 12, "ksize": [9, 1], "stride": [2, 1], "pad": [4, 0]}]}}


def make(name, B, T, D, L, V, seed):
    cfg = small_cfg(V)
    P = R.init_params(cfg, D, V, seed=seed, dtype=np.float64)
    X, y = R.synth_batch(B, T, D, L, V, seed=seed + 1, dtype=np.float64)
    m = R.RefModel(cfg, {k: v.copy() for k, v in P.items()}, V)
    opt = R.RefOptimizer(m, OPT)
    rnd = random.Random("seed-ast-20h")
    # raw gradients of step 1 (before the hooks touch them)
    loss = m.forward_loss(X, y, 0.8, pyrandom=rnd)
    flags = list(m.use_truth)
    m.cleargrads()
    loss.backward()
    grads = {k: p.grad.copy() for k, p in m.params()}
    out = {"cfg": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), "X": X, "y": y, "flags": np.asarray(flags, np.int32),
           "loss": np.asarray(float(loss.data))}
    opt.update()
    out["grad_norm"] = np.asarray(opt.last_grad_norm)
    losses = [float(loss.data)]

    class Fixed:
        def __init__(self):
            self.it = iter([])

        def random(self):
            return 0.0 if next(self.it) else 1.0
    fx = Fixed()
    for _ in range(2):
        fx.it = iter(flags[1:-1])
        l2, _ = R.train_step(m, opt, X, y, 0.5, pyrandom=fx)
        losses.append(l2)
    out["losses3"] = np.asarray(losses)
    for k, v in P.items():
        out["param/" + k] = v
    for k, v in grads.items():
        out["grad/" + k] = v
    for k, p in m.params():
        out["after3/" + k] = p.data
    for k in ("CNN_0_bn/avg_mean", "CNN_0_bn/avg_var", "CNN_1_bn/avg_mean", "CNN_1_bn/avg_var"):
        out["after3/" + k] = m.p[k]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(path, "loss", float(loss.data), "grad_norm", opt.last_grad_norm, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    make("step_B2_T37_D13_L6", 2, 37, 13, 6, 23, seed=3)
    make("step_B2_T37_D80_L6", 2, 37, 80, 6, 23, seed=5)
