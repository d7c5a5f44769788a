"""Margins of tests/test_gpu_model.py::test_full_size_batch_permutation_and_gradient_accumulation: the same quantities, over several
permutations and repeats (atomics order differs run to run), printed against the test's bounds."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config  # noqa: E402
from oracle.ast_ref import synth_batch  # noqa: E402

worst = {}
for T, dec_layers, D in [(800, 1, 80), (1200, 1, 80), (800, 3, 80), (1200, 3, 80), (800, 3, 13)]:
    cfg = copy.deepcopy(bench.MODEL_CFG)
    cfg["dropout"] = {"embed": 0.0, "rnn": 0.0, "out": 0}
    cfg["rnn_config"]["dec_layers"] = dec_layers
    B, L, V = 32, 40, cfg["rnn_config"]["dec_vocab_size"]
    X, y = synth_batch(B, T, D, L, V, 20, dtype=np.float32)
    X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
    m.inject = {"use_truth": [1] * (L - 1)}

    def run(Xb, yb, clear=True):
        with using_config("train", True):
            l = m.forward_loss(X=Xb, y=yb, teach_ratio=1.0, random_out=0, add_noise=0)
            if clear:
                m.cleargrads()
            l.backward()
        torch.cuda.synchronize()
        return float(l.data), m.arena.grad.clone(), m.enc_states.clone()

    l0, g0, e0 = run(X, y)
    gscale, gnorm = float(g0.abs().max()), float(g0.norm())
    for seed in (5, 6, 7, 8, 9, 10):
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(seed)).cuda()
        l1, g1, e1 = run(X[perm].contiguous(), y[perm].contiguous())
        _, g2, _ = run(X[perm].contiguous(), y[perm].contiguous(), clear=False)
        r = dict(loss=abs(l1 - l0) / abs(l0) / (2e-5 if T <= 800 else 1e-4),
                 enc=float((e1 - e0[perm]).abs().max()) / float(e0.abs().max()) / 2e-4,
                 gnorm=float((g1 - g0).norm()) / gnorm / 5e-4,
                 gmax=float((g1 - g0).abs().max()) / gscale / 2e-3,
                 acc_norm=float((g2 - 2 * g1).norm()) / gnorm / 5e-4,
                 acc_max=float((g2 - 2 * g1).abs().max()) / gscale / 2e-3)
        print(f"T={T} dec={dec_layers} D={D} seed={seed}: fraction of the bound used: " + "  ".join(f"{k} {v:.3f}" for k, v in r.items()), flush=True)
        for k, v in r.items():
            worst[k] = max(worst.get(k, 0), v)
        if True:
            rows = []
            for name in m.arena.shapes:
                o, n = m.arena.range_of(name)
                a, b = g0[o:o + n], g1[o:o + n]
                rows.append((float((a - b).norm()) / max(float(a.norm()), 1e-30), float((a - b).norm()) / gnorm, name))
            rows.sort(reverse=True)
            for xx, yy, nm in rows:
                k = "own_cnn" if nm.startswith("CNN_") else "own_rest"
                worst[k] = max(worst.get(k, 0), xx)
            rows2 = []
            for name in m.arena.shapes:
                o, n = m.arena.range_of(name)
                a, b = g1[o:o + n], g2[o:o + n]
                rows2.append(float((b - 2 * a).norm()) / max(float(a.norm()), 1e-30))
            worst["own_acc"] = max(worst.get("own_acc", 0), max(rows2))
            rows = [t for t in rows if not t[2].startswith("CNN_")][:4] + [t for t in rows if t[2].startswith("CNN_")][:3]
            print("    by tensor (own-norm relative, share of the global bound):  " + "  ".join(f"{nm} {xx:.1e}/{yy / 5e-4:.3f}" for xx, yy, nm in rows[:7]), flush=True)
    del m
    torch.cuda.empty_cache()
print("worst fraction of each bound:", {k: round(v, 3) for k, v in worst.items()})
