"""GPU parity tests, whole hot path: ast_amd.seq2seq.SpeechEncoderDecoder (HIP kernels through the C ABI) against
the CPU oracle (oracle/ast_ref.py) on identical inputs and weights -- the north_star gate: loss and global
gradient norm within 1e-4 relative (fp32), plus every individual gradient and the parameters after 3 updates."""
import random

import numpy as np
import pytest
import torch

from conftest import tiny_cfg

pytestmark = pytest.mark.gpu

OPT = {"type": 0, "lr": 1e-3, "l2": 1e-4, "grad_clip": 2, "grad_noise_eta": 0, "freeze": []}


def _mid_cfg(drop):
    cfg = tiny_cfg(enc_layers=3, dec_layers=2, H=64, E=16, A=64, c0=16, c1=32, V=57, drop=drop)
    return cfg


def _make(cfg, B, T, D, L, V, seed=0):
    from oracle import ast_ref as R
    P = R.init_params(cfg, D, V, seed=seed, dtype=np.float32)
    X, y = R.synth_batch(B, T, D, L, V, seed=seed + 1, dtype=np.float32)
    return P, X, y


def _gpu_model(cfg, P, D, V):
    from ast_amd.seq2seq import SpeechEncoderDecoder
    import copy
    c = copy.deepcopy(cfg)
    c["rnn_config"]["dec_vocab_size"] = V
    m = SpeechEncoderDecoder(0, c)
    m.materialize(D, values=P)
    return m


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


@pytest.mark.parametrize("name,cfgf,B,T,D,L,V,drop,teach", [
    ("tiny", lambda d: tiny_cfg(c1=8, drop=d), 3, 21, 26, 6, 11, 0.0, 1.0),
    ("tiny-drop", lambda d: tiny_cfg(c1=8, drop=d), 3, 21, 26, 6, 11, 0.3, 0.5),
    ("mid-80d", _mid_cfg, 4, 120, 80, 9, 57, 0.0, 0.8),
    ("mid-80d-drop", _mid_cfg, 4, 120, 80, 9, 57, 0.3, 0.8),
    ("mid-13d", _mid_cfg, 5, 90, 13, 7, 57, 0.0, 0.8),
    # hidden_units 128 -> 64 per direction: the persistent wavefront encoder kernels
    ("persist-h64", lambda d: tiny_cfg(enc_layers=3, dec_layers=3, H=128, E=16, A=64, c0=8, c1=16, V=57, drop=d), 18, 70, 80, 8, 57, 0.0, 0.8),
    ("persist-h64-drop", lambda d: tiny_cfg(enc_layers=2, dec_layers=1, H=128, E=16, A=64, c0=8, c1=16, V=57, drop=d), 4, 70, 80, 8, 57, 0.3, 0.8),
])
def test_train_step_parity(name, cfgf, B, T, D, L, V, drop, teach, gemm_scheme):
    from oracle import ast_ref as R
    from oracle.ast_ref_torch import masks_from_recording
    from ast_amd import optimizers as O
    from ast_amd.seq2seq import using_config
    cfg = cfgf(drop)
    P, X, y = _make(cfg, B, T, D, L, V)
    # ---- oracle, float64 (reference truth) and float32 (what Chainer-on-NumPy would compute)
    res = {}
    for dt in (np.float64, np.float32):
        m = R.RefModel(cfg, {k: v.astype(dt) for k, v in P.items()}, V)
        rec = R.RecordingMasks(3) if drop > 0 else None
        if rec:
            m.masks = rec
        noise = np.random.default_rng(9).normal(1.0, 0.25, X.shape).astype(np.float32) if drop > 0 else None
        opt = R.RefOptimizer(m, OPT)
        rnd = random.Random("seed-ast-20h")
        loss, _ = R.train_step(m, opt, X.astype(dt), y, teach, add_noise=0.25 if drop > 0 else 0, noise=noise, pyrandom=rnd)
        res[dt] = dict(loss=loss, gnorm=opt.last_grad_norm, model=m, opt=opt, flags=list(m.use_truth), rec=rec, noise=noise,
                       enc=m.enc_states.data.copy())
    ref = res[np.float64]
    # ---- HIP path
    g = _gpu_model(cfg, P, D, V)
    g.gemm_precision = gemm_scheme
    T2 = ref["enc"].shape[1]
    if drop > 0:
        packed = masks_from_recording(cfg, ref["rec"].masks, T2, L - 1, B)
        g.inject = {k: torch.from_numpy(v) for k, v in packed.items()}
        g.inject["noise"] = torch.from_numpy(ref["noise"])
    g.inject["use_truth"] = ref["flags"]
    opt = O.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True)
    opt.setup(g)
    opt.add_hook(O.WeightDecay(1e-4))
    opt.add_hook(O.GradientClipping(2))
    with using_config("train", True):
        loss = g.forward_loss(X=torch.from_numpy(X), y=torch.from_numpy(y), teach_ratio=teach, random_out=0,
                              add_noise=0.25 if drop > 0 else 0)
        g.cleargrads()
        loss.backward()
        grads = g.arena.to_numpy(grads=True)
        opt.update()
    torch.cuda.synchronize()
    lv = float(loss.data)
    np.testing.assert_allclose(g.enc_states.cpu().numpy(), ref["enc"], rtol=0, atol=2e-4 * np.abs(ref["enc"]).max(), err_msg="enc_states")
    assert _rel(lv, ref["loss"]) < 1e-4, (name, lv, ref["loss"])
    assert _rel(opt.last_grad_norm, ref["gnorm"]) < 1e-4, (name, opt.last_grad_norm, ref["gnorm"])
    # the f32 oracle itself sits this far from the f64 one (context for the tolerance)
    assert _rel(res[np.float32]["loss"], ref["loss"]) < 1e-4
    # note: after update() the oracle's grads include decay and clip; recompute raw grads for the per-tensor check
    m2 = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    if drop > 0:
        m2.masks = lambda shape, ratio, tag: ref["rec"].masks[tag]
    class _Fixed:
        def __init__(s, flags): s.it = iter(flags[1:-1])
        def random(s): return 0.0 if next(s.it) else 1.0
    l2 = m2.forward_loss(X.astype(np.float64), y, 0.5, add_noise=0.25 if drop > 0 else 0, noise=ref["noise"], pyrandom=_Fixed(ref["flags"]))
    m2.cleargrads()
    l2.backward()
    gmax = max(np.abs(p.grad).max() for _, p in m2.params())
    for k, p in m2.params():
        err = np.abs(grads[k] - p.grad).max()
        tol = 3e-4 * max(np.abs(p.grad).max(), 1e-3 * gmax)
        assert err <= tol, f"{name}: grad {k}: err {err:.3e} tol {tol:.3e}"
    # ---- two more steps (no dropout, all teacher-forced), then compare losses and the parameter deltas
    import copy
    mo = ref["model"]
    nodrop = copy.deepcopy(mo.cfg)
    nodrop["dropout"] = {"embed": 0.0, "rnn": 0.0, "out": 0}
    mo.cfg = nodrop
    for step in range(2):
        X2, y2 = R.synth_batch(B, T, D, L, V, seed=100 + step, dtype=np.float32)
        g.inject = {"use_truth": [1] * (L - 1), "enc_masks": None, "emb_mask": None, "rnn_masks": None}
        with using_config("train", True):
            ls = g.forward_loss(X=torch.from_numpy(X2), y=torch.from_numpy(y2), teach_ratio=1.0)
            g.cleargrads()
            ls.backward()
            opt.update()
        lref, _ = R.train_step(mo, ref["opt"], X2.astype(np.float64), y2, 1.0, pyrandom=random.Random(1))
        assert _rel(float(ls.data), lref) < 2e-3, (name, step, float(ls.data), lref)
    after = g.arena.to_numpy()
    num = den = 0.0
    for k, p in mo.params():
        num += float(((after[k].astype(np.float64) - p.data) ** 2).sum())
        den += float(((p.data - P[k]) ** 2).sum())
    # AMSGrad's first steps move every weight by ~lr*sign(g): elements whose gradient is below f32 noise may flip
    assert np.sqrt(num / den) < 5e-2, f"{name}: parameter delta after 3 updates off by {np.sqrt(num / den):.3e} (relative L2)"
    # BN running statistics follow Chainer-sem A4
    for i in range(2):
        for s in ("avg_mean", "avg_var"):
            np.testing.assert_allclose(g.persist[f"CNN_{i}_bn/{s}"].cpu().numpy(), ref["model"].p[f"CNN_{i}_bn/{s}"], rtol=2e-3, atol=1e-5)


def test_predict_greedy_matches_oracle():
    from oracle import ast_ref as R
    cfg = _mid_cfg(0.3)
    B, T, D, L, V = 4, 100, 80, 8, 57
    P, X, y = _make(cfg, B, T, D, L, V, seed=4)
    m = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    ref = m.predict(X.astype(np.float64), R.GO_ID, R.EOS_ID, 12)
    g = _gpu_model(cfg, P, D, V)
    got = g.predict(torch.from_numpy(X), R.GO_ID, R.EOS_ID, 12)
    assert got.shape == ref.shape and (got == ref).all(), (got, ref)


def test_beam_search_matches_oracle():
    """nn.py:235-322 (decode_beam / decode_beam_step / init_hyp) through the model's state API against the oracle's restatement:
    same N-best hypotheses, scores within 1e-4."""
    from oracle import ast_ref as R
    from ast_amd import nn as gnn
    cfg = tiny_cfg(enc_layers=3, dec_layers=3, H=64, E=16, A=64, c0=8, c1=16, V=57, drop=0.3)
    B, T, D, L, V = 1, 90, 80, 8, 57
    P, X, y = _make(cfg, B, T, D, L, V, seed=11)
    m = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    ref = R.decode_beam(m, X.astype(np.float64), stop_limit=7, N=3, K=4)
    g = _gpu_model(cfg, P, D, V)
    got = gnn.decode_beam(g, torch.from_numpy(X), stop_limit=7, N=3, K=4)
    assert len(got) == len(ref) == 3
    for a, b in zip(got, ref):
        assert a["hyp"] == b["hyp"], (a["hyp"], b["hyp"])
        assert abs(a["score"] - b["score"]) <= 1e-4 * max(1.0, abs(b["score"])), (a["score"], b["score"])
        assert len(a["attn_history"]) == len(b["attn_history"]) == len(a["hyp"]) - 1
        np.testing.assert_allclose(a["attn_history"][-1], b["attn_history"][-1], rtol=0, atol=1e-5)


def test_checkpoint_roundtrip(tmp_path):
    from ast_amd import serializers
    from ast_amd.seq2seq import SpeechEncoderDecoder
    import copy
    cfg = tiny_cfg(c1=8)
    P, X, y = _make(cfg, 2, 21, 26, 5, 11)
    g = _gpu_model(cfg, P, 26, 11)
    path = str(tmp_path / "seq2seq_3.model")
    serializers.save_npz(path, g)
    z = np.load(path)
    assert {"CNN_0/W", "CNN_0_bn/avg_var", "CNN_0_bn/N", "L0_rev_enc/lateral/W", "attn_Wa/b", "embed_dec/W", "out/b"} <= set(z.files)
    c2 = copy.deepcopy(cfg)
    g2 = SpeechEncoderDecoder(0, c2)
    serializers.load_npz(path, g2)
    assert g2.in_dim == 26 and g2.V == 11
    for k, v in g.arena.views.items():
        assert torch.equal(v, g2.arena.views[k]), k


def test_checkpoints_cross_load_between_oracle_and_hip_model(tmp_path):
    """train.py:73-75 / nn.py:141-152 at the file level: a Chainer-layout .npz WRITTEN BY THE ORACLE (oracle.ast_ref.save_npz: the keys,
    layouts and BatchNorm persistents chainer.serializers.save_npz produces, after one oracle train step so that the running
    statistics and N are not the initial ones) loads into a HIP model that then computes the oracle's eval-mode greedy decode and
    train-mode loss; and a checkpoint written by the HIP model after a train step loads into a fresh oracle model with the same result."""
    from oracle import ast_ref as R
    from ast_amd import serializers
    from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
    import copy
    cfg = tiny_cfg(enc_layers=2, dec_layers=2, H=32, E=8, A=32, c0=8, c1=16, V=37, drop=0.0)
    B, T, D, L, V = 3, 60, 26, 6, 37
    P, X, y = _make(cfg, B, T, D, L, V, seed=7)
    ref = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    opt = R.RefOptimizer(ref, OPT)
    R.train_step(ref, opt, X.astype(np.float64), y, 1.0, pyrandom=random.Random(0))
    path = str(tmp_path / "seq2seq_1.model")
    R.save_npz(path, ref)
    z = np.load(path)
    assert int(z["CNN_0_bn/N"]) == 1 and int(z["CNN_1_bn/N"]) == 1 and z["L0_dec/upward/W"].shape == (4 * 32, 8 + 32)
    # oracle file -> HIP model
    g = SpeechEncoderDecoder(0, copy.deepcopy(cfg))
    serializers.load_npz(path, g)
    assert g.bn_N == 1 and g.in_dim == D
    want = ref.predict(X.astype(np.float64), R.GO_ID, R.EOS_ID, 8)                    # eval mode: running statistics from the file
    got = g.predict(torch.from_numpy(X), R.GO_ID, R.EOS_ID, 8)
    assert got.shape == want.shape and (got == want).all()
    flags = [1] * (L - 1)
    lref = ref.forward_loss(X.astype(np.float64), y, 1.0, pyrandom=random.Random(0))
    g.inject["use_truth"] = flags
    with using_config("train", True):
        lg = g.forward_loss(torch.from_numpy(X), torch.from_numpy(y), 1.0)
        g.cleargrads()
        lg.backward()
    assert _rel(float(lg.data), float(lref.data)) < 1e-4
    assert g.bn_N == 2
    # HIP file -> oracle model
    path2 = str(tmp_path / "seq2seq_2.model")
    serializers.save_npz(path2, g)
    ref2 = R.RefModel(cfg, {k: np.zeros_like(v, dtype=np.float64) for k, v in P.items()}, V)
    R.load_npz(path2, ref2)
    assert ref2.bn["CNN_0_bn"].N == 2
    np.testing.assert_allclose(ref2.p["CNN_1_bn/avg_var"], ref.p["CNN_1_bn/avg_var"], rtol=2e-3, atol=1e-6)   # both saw the same second forward
    want2 = ref2.predict(X.astype(np.float64), R.GO_ID, R.EOS_ID, 8)
    got2 = g.predict(torch.from_numpy(X), R.GO_ID, R.EOS_ID, 8)
    assert (got2 == want2).all()


def test_missing_library_fails_loudly(monkeypatch):
    from ast_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libastk.so")
    with pytest.raises(_lib.AstkError):
        _lib.load()


@pytest.mark.parametrize("dec_layers,D", [(1, 80), (3, 80), (3, 13)])   # configs[1]; shipped es_en_20h (3 decoder layers); asr_gpfr shape (13-d)
def test_full_size_directional_derivative_and_repeatability(dec_layers, D):
    """BASELINE configs[1] at full size (B=32, T=800, D=80, h=256, H=512, V=1098, L=40): the oracle is too slow here, so
    the check is a size-independent property.  With dropout / noise off and every step teacher-forced the loss is a smooth
    function of the parameters: (L(p + e d) - L(p - e d)) / 2e must equal <grad L, d> -- the persistent encoder / decoder
    kernels, the stream-K GEMMs and the CNN backward at exactly the benchmark's shapes, forward against backward.
    Also: the same step twice gives the same loss up to the atomics' summation order."""
    import copy
    import bench
    from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
    from oracle.ast_ref import synth_batch
    cfg = copy.deepcopy(bench.MODEL_CFG)
    cfg["dropout"] = {"embed": 0.0, "rnn": 0.0, "out": 0}
    cfg["rnn_config"]["dec_layers"] = dec_layers
    B, T, L, V = 32, 800, 40, cfg["rnn_config"]["dec_vocab_size"]
    X, y = synth_batch(B, T, D, L, V, 20, dtype=np.float32)
    X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
    m.inject = {"use_truth": [1] * (L - 1)}

    def loss_only():
        with using_config("train", True):
            l = m.forward_loss(X=X, y=y, teach_ratio=1.0, random_out=0, add_noise=0)
        return l

    l0 = loss_only()
    m.cleargrads()
    l0.backward()
    torch.cuda.synchronize()
    base = float(l0.data)
    g = m.arena.grad.clone()
    assert np.isfinite(base) and bool(torch.isfinite(g).all())
    l1 = float(loss_only().data)
    assert abs(l1 - base) <= 1e-5 * abs(base), (base, l1)
    p0 = m.arena.data.clone()
    gen = torch.Generator(device="cuda").manual_seed(3)
    for trial in range(2):
        # direction scaled like the parameters themselves; pad elements of the arena carry zero gradient and stay untouched
        d = torch.randn(p0.shape, device="cuda", generator=gen) * p0.abs().clamp_min(1e-3)
        d[g == 0] = 0
        gd = float((g.double() * d.double()).sum())
        eps = 0.05 / max(abs(gd), 1e-6)               # aim at |L+ - L-| ~ 0.1 on a loss of ~140
        eps = min(eps, 1e-2)
        m.arena.data.copy_(p0 + eps * d)
        lp = float(loss_only().data)
        m.arena.data.copy_(p0 - eps * d)
        lm = float(loss_only().data)
        m.arena.data.copy_(p0)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - gd) <= 3e-2 * abs(gd) + 1e-3, (trial, fd, gd, eps, lp, lm)


@pytest.mark.parametrize("c0", [8, 16])
def test_sync_batchnorm_two_replicas_match_one_process_on_the_concatenated_batch(c0):
    """(c0 = 16: layer 0 as the direct convolution, whose per-tile sums feed the statistics kernel -- the exchange sits between that kernel
    and a finalize launch of its own; c0 = 8: the im2col path.)
    Global-batch BatchNorm under data parallelism (include/astk.h astk_conv_bn_relu_*_sync, ast_amd.dist.StatExchange).
    Two replicas with the two halves of a batch are emulated in ONE process: the exchange callback is replaced by one that
    plays back the sum of both replicas' statistics, which are collected exchange point by exchange point (4 per step: two
    layers forward, two backward) over repeated passes.  Result: the replicas' mean loss, mean gradient, encoder states and
    BatchNorm running statistics must be those of the oracle run once on the whole batch."""
    from oracle import ast_ref as R
    from ast_amd.dist import StatExchange
    from ast_amd.seq2seq import using_config
    cfg = tiny_cfg(enc_layers=2, dec_layers=1, H=16, E=8, A=16, c0=c0, c1=8, V=23, drop=0.0)
    B, T, D, L, V, world = 6, 37, 26, 6, 23, 2
    P, X, y = _make(cfg, B, T, D, L, V)
    ref = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    flags = [1] * (L - 1)

    class _Truth:
        def random(self): return 0.0
    lref = ref.forward_loss(X.astype(np.float64), y, 1.0, pyrandom=_Truth())
    ref.cleargrads()
    lref.backward()
    gref = {k: p.grad.copy() for k, p in ref.params()}

    shards = [slice(r * B // world, (r + 1) * B // world) for r in range(world)]
    models = [_gpu_model(cfg, P, D, V) for _ in range(world)]
    known = []            # global statistics of exchange points 0..len-1

    def run(m, rows, record):
        state = {"k": 0}

        def reduce(view):
            k = state["k"]
            state["k"] += 1
            if k < len(known):
                view.copy_(known[k])
            elif k == len(known):
                record.append(view.clone())
        m.stat_exchange = StatExchange(world=world, reduce=reduce)
        m.inject = {"use_truth": flags}
        for name in ("avg_mean", "avg_var"):          # every pass starts from the initial running statistics
            for i in range(2):
                m.persist[f"CNN_{i}_bn/{name}"].copy_(torch.from_numpy(P[f"CNN_{i}_bn/{name}"]))
        with using_config("train", True):
            loss = m.forward_loss(X=torch.from_numpy(X[rows]), y=torch.from_numpy(y[rows]), teach_ratio=1.0)
            m.cleargrads()
            loss.backward()
        torch.cuda.synchronize()
        assert state["k"] == 4, state
        return float(loss.data)

    for _ in range(4):
        rec = []
        for m, rows in zip(models, shards):
            run(m, rows, rec)
        assert len(rec) == world
        known.append(sum(rec))
    losses = [run(m, rows, []) for m, rows in zip(models, shards)]
    assert _rel(sum(losses) / world, float(lref.data)) < 1e-4, (losses, float(lref.data))
    enc = np.concatenate([m.enc_states.cpu().numpy() for m in models])
    np.testing.assert_allclose(enc, ref.enc_states.data, rtol=0, atol=2e-4 * np.abs(ref.enc_states.data).max())
    grads = [m.arena.to_numpy(grads=True) for m in models]
    gmax = max(np.abs(g).max() for g in gref.values())
    for k, g in gref.items():
        got = sum(gr[k] for gr in grads) / world
        err = np.abs(got - g).max()
        tol = 3e-4 * max(np.abs(g).max(), 1e-3 * gmax)
        assert err <= tol, f"grad {k}: err {err:.3e} tol {tol:.3e}"
    for i in range(2):
        for s in ("avg_mean", "avg_var"):
            for m in models:
                np.testing.assert_allclose(m.persist[f"CNN_{i}_bn/{s}"].cpu().numpy(), ref.p[f"CNN_{i}_bn/{s}"], rtol=2e-3, atol=1e-5)
    # and without the exchange the replicas differ from the one-process result (the test is sensitive to the statistics)
    models[0].stat_exchange = None
    models[0].inject = {"use_truth": flags}
    with using_config("train", True):
        models[0].forward_loss(X=torch.from_numpy(X[shards[0]]), y=torch.from_numpy(y[shards[0]]), teach_ratio=1.0)
    torch.cuda.synchronize()
    assert np.abs(models[0].enc_states.cpu().numpy() - ref.enc_states.data[shards[0]]).max() > 1e-3


def test_nn_trains_predicts_and_resumes_through_train_py(tmp_path):
    """The drop-in surface end to end on the GPU (train.py:34-75, nn.py:141-233): `python train.py -m <dir> -e N` on an experiment
    directory in the reference's schema (synthetic loader), in a child process like a user would run it; the loss falls over
    the epochs, checkpoints appear, a second invocation resumes from the newest one, and NN.predict decodes the dev set."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # no dropout / noise / frame zeroing and full teacher forcing: the epoch averages of 5 batches are then smooth enough to assert on
    mcfg = tiny_cfg(enc_layers=2, dec_layers=1, H=32, E=16, A=32, c0=8, c1=16, V=31, drop=0.0)
    del mcfg["rnn_config"]["dec_vocab_size"]                      # injected by Config from the data section
    tcfg = {"seed": "seed-ast-20h", "gpuid": 0, "batch_size": 8, "train_set": "syn_train", "dev_set": "syn_dev", "iters_save": 2,
            "save_optimizer": True,      # extension: Adam moments travel with the checkpoint, so the resumed epoch continues the curve
            "optimizer": {"type": 0, "lr": 2e-3, "l2": 1e-4, "grad_clip": 2, "grad_noise_eta": 0, "freeze": []},
            "extras": {"teach_ratio": 1.0, "random_out": 0, "speech_noise": 0},
            "data": {"dataloader": "synthetic", "vocab_size": 31, "feat_dim": 13, "n_utts": {"syn_train": 40, "syn_dev": 6},
                     "frames": [60, 300], "targets": [2, 9], "buckets_num": 4, "buckets_width": 80, "max_pred": 12,
                     "zero_input": 0.0, "train_scale": 1, "dec_key": "bpe_w"}}
    json.dump(mcfg, open(tmp_path / "model_cfg.json", "w"))
    json.dump(tcfg, open(tmp_path / "train_cfg.json", "w"))

    def run(epochs):
        r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "-m", str(tmp_path), "-e", str(epochs)], cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout
    run(6)
    log = [l.split(",") for l in open(tmp_path / "train.log").read().split("\n") if l.strip()]
    epochs, losses = [int(a) for a, _ in log], [float(b) for _, b in log]
    assert epochs == [1, 2, 3, 4, 5, 6]
    assert sum(losses[-2:]) < 0.9 * sum(losses[:2]) and all(np.isfinite(losses)), losses
    saved = sorted(f for f in os.listdir(tmp_path) if f.endswith(".model"))
    assert saved == ["seq2seq_2.model", "seq2seq_4.model", "seq2seq_6.model"], saved
    out = run(2)                                                    # resumes at epoch 7 from seq2seq_6.model
    assert "seq2seq_6.model" in out and "optimizer state restored" in out
    log = [l.split(",") for l in open(tmp_path / "train.log").read().split("\n") if l.strip()]
    assert [int(a) for a, _ in log[-2:]] == [7, 8]
    assert sum(float(b) for _, b in log[-2:]) < 1.05 * sum(losses[-2:]), log      # the curve continues, it does not restart
    # in-process: the resumed model decodes the dev set (greedy) into token lists that start with GO and stay inside the vocabulary
    from ast_amd.nn import NN
    nn = NN(str(tmp_path))
    assert nn.max_epoch == 8
    preds = nn.predict("syn_dev")
    assert len(preds) == 6 and all(1 <= len(p) <= 12 and all(0 <= t < 31 for t in p) for _, p in preds)
    # beam.py (beam.py:46-146): references in the reference's file layout, n-best pickle, BLEU line, hypothesis file, --resume
    refs = tmp_path / "refs" / "syn_dev"
    os.makedirs(refs)
    utts = sorted(nn.data_loader.info["syn_dev"])
    truth = nn.data_loader.get_hyps([(u, list(nn.data_loader.ids["syn_dev"][u])) for u in utts])
    (refs / "eval.ids").write_text("".join(u + "\n" for u in utts))
    (refs / "ref.en0").write_text("".join(" ".join(truth[u]) + "\n" for u in utts))
    tcfg["data"].update(refs_path=str(tmp_path / "refs"), n_evals=1)
    json.dump(tcfg, open(tmp_path / "train_cfg.json", "w"))
    del nn
    torch.cuda.empty_cache()
    for extra in ([], ["--resume"]):
        r = subprocess.run([sys.executable, os.path.join(root, "beam.py"), "-m", str(tmp_path), "-n", "3", "-k", "4", "-s", "syn_dev", "-w", "0.6"] + extra,
                           cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "BLEU = " in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(tmp_path / "syn_dev_beam_N-3_K-4.p")
    lines = open(tmp_path / "syn_dev_beam_N-3_K-4_W-0.60.en").read().split("\n")
    assert len(lines) == 7 and lines[-1] == ""


def test_descriptor_precision_overrides_the_process_default_for_every_op():
    """include/astk.h: the arithmetic of an op's products travels in its descriptor (`precision`), the process-wide setter only supplies
    what ASTK_PREC_DEFAULT resolves to.  A model that ASKS for f32 while the process default is bf16x3 must produce exactly the bits of
    a model without a wish while the process default is f32 (CNN, encoder stack incl. the persistent recurrences, decoder, all batched
    products) -- and other bits than the default arithmetic."""
    from oracle import ast_ref as R
    from ast_amd import _lib
    from ast_amd.seq2seq import using_config
    lib = _lib.load()
    cfg = tiny_cfg(enc_layers=2, dec_layers=1, H=128, E=16, A=64, c0=8, c1=16, V=57, drop=0.0)
    B, T, D, L, V = 18, 70, 80, 8, 57
    P = R.init_params(cfg, D, V, seed=4, dtype=np.float32)
    X, y = R.synth_batch(B, T, D, L, V, seed=41, dtype=np.float32)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()

    def run(model_prec):
        g = _gpu_model(cfg, P, D, V)
        g.gemm_precision = model_prec
        g.inject = {"use_truth": [1] * (L - 1)}
        with using_config("train", True):
            loss = g.forward_loss(X=Xd, y=yd, teach_ratio=1.0)
            g.cleargrads()
            loss.backward()
        torch.cuda.synchronize()
        return float(loss.data), g.enc_states.clone(), g.arena.grad.clone()
    assert lib.astk_get_gemm_precision() == 1                      # the library default: bf16x3
    l_def, e_def, _ = run(None)
    l_ask, e_ask, g_ask = run("f32")                               # descriptor asks for f32, process default bf16x3
    prev = lib.astk_set_gemm_precision(2)
    try:
        l_f32, e_f32, g_f32 = run(None)                            # process default f32, descriptor silent
        l_x3, e_x3, _ = run("bf16x3")                              # ... and a descriptor that asks for bf16x3 under it
    finally:
        lib.astk_set_gemm_precision(prev)
    assert l_ask == l_f32 and torch.equal(e_ask, e_f32)
    # (gradients: float atomics in the split tiles make the last bits run-dependent; everything else is bit-identical)
    assert float((g_ask - g_f32).abs().max()) <= 1e-6 * float(g_f32.abs().max())
    assert torch.equal(e_x3, e_def) and l_x3 == l_def
    assert not torch.equal(e_def, e_f32), "the two arithmetics produced identical bits: the descriptor field is not in force"


@pytest.mark.parametrize("scheme", ["fp16x2", "bf16x3", "f32"])
def test_thirty_update_trajectory_against_the_float64_oracle(scheme):
    """The arithmetic schemes of the batched GEMMs and the encoder recurrences over a TRAJECTORY, not only at initialisation: 30 updates
    (Adam + L2 + clip, a cycle of three batches, teacher-forced so that no argmax tie can fork the runs) of a model on the persistent
    kernels, under each scheme, against the float64 oracle: the loss of EVERY update within 2e-3, the parameter displacement after the
    last one within 2 % of its length.  (fp16x2 forced on every GEMM size; by default small launches take bf16x3.)"""
    import ctypes as C
    from oracle import ast_ref as R
    from ast_amd import _lib, optimizers as O
    from ast_amd.seq2seq import using_config
    lib = _lib.load()
    cfg = tiny_cfg(enc_layers=2, dec_layers=1, H=128, E=16, A=64, c0=8, c1=16, V=57, drop=0.0)
    B, T, D, L, V = 18, 70, 80, 8, 57
    P = R.init_params(cfg, D, V, seed=4, dtype=np.float32)
    batches = [R.synth_batch(B, T, D, L, V, seed=40 + i, dtype=np.float32) for i in range(3)]
    ref = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    ropt = R.RefOptimizer(ref, OPT)
    want = []
    for it in range(30):
        X, y = batches[it % 3]
        want.append(R.train_step(ref, ropt, X.astype(np.float64), y, 1.0, pyrandom=random.Random(0))[0])
    assert want[-1] < 0.9 * want[0]                                           # it does train
    below = lib.astk_set_gemm_bf16_split_below(C.c_double(0.0))
    try:
        g = _gpu_model(cfg, P, D, V)
        g.gemm_precision = scheme            # -> the descriptors' `precision` field (per call)
        assert lib.astk_lstm_stack_path(C.byref(LstmStackDescFor(g, B, T, D))) == 1
        g.inject = {"use_truth": [1] * (L - 1)}
        opt = O.Adam(alpha=1e-3, amsgrad=True).setup(g)
        opt.add_hook(O.WeightDecay(1e-4))
        opt.add_hook(O.GradientClipping(2))
        dev = [(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()) for X, y in batches]
        got = []
        for it in range(30):
            X, y = dev[it % 3]
            with using_config("train", True):
                loss = g.forward_loss(X=X, y=y, teach_ratio=1.0)
                g.cleargrads()
                loss.backward()
                opt.update()
            got.append(float(loss.data))
    finally:
        lib.astk_set_gemm_bf16_split_below(C.c_double(below))
    worst = max(_rel(a, b) for a, b in zip(got, want))
    assert worst < 2e-3, (scheme, worst, got[-3:], want[-3:])
    after = g.arena.to_numpy()
    num = sum(float(((after[k].astype(np.float64) - p.data) ** 2).sum()) for k, p in ref.params())
    den = sum(float(((p.data - P[k]) ** 2).sum()) for k, p in ref.params())
    assert num <= 0.02 ** 2 * den, (scheme, num, den)


def LstmStackDescFor(model, B, T, D):
    """The encoder stack descriptor the model would build for a (B, T, D) batch (to ask the library which path it takes)."""
    from ast_amd._lib import LstmStackDesc
    T2 = ((T + 8 - 9) // 2 + 1 + 8 - 9) // 2 + 1
    feat = model.arena.shapes["L0_enc/upward/W"][1]
    return LstmStackDesc(T2, B, feat, model.h, len(model.rnn_enc), model.n_dirs)


def _full_cfg(model):
    """bench.py's models without dropout: cfg1 (configs[1]), es_en_20h (3 decoder layers), cfg5 (configs[4]'s shape)."""
    import copy
    import bench
    cfg = copy.deepcopy(bench.MODEL_CFG)
    cfg["dropout"] = {"embed": 0.0, "rnn": 0.0, "out": 0}
    if model == "es_en_20h":
        cfg["rnn_config"]["dec_layers"] = 3
    if model == "cfg5":
        cfg["rnn_config"].update(enc_layers=6, hidden_units=1024, attn_units=1024, dec_vocab_size=8004)
    return cfg


# 1200 frames: T'' = 300 > 256 (long utterances); es_en_20h = the shipped 3-layer decoder; D = 13: asr_gpfr features; batch 64: grouped
# encoder launches and the decoder's 64-row path; cfg5: the shape of BASELINE configs[4] (6-layer 2 x 512 encoder, H = 1024, V = 8004),
# in both GEMM operand modes (fp16: the reduced-precision mode configs[4] names)
@pytest.mark.parametrize("model,B,T,D,operands", [
    ("cfg1", 32, 800, 80, "f32"), ("cfg1", 32, 1200, 80, "f32"), ("es_en_20h", 32, 800, 80, "f32"), ("es_en_20h", 32, 1200, 80, "f32"),
    ("es_en_20h", 32, 800, 13, "f32"), ("cfg1", 64, 800, 80, "f32"), ("cfg1", 64, 800, 80, "fp16"), ("cfg5", 32, 800, 80, "f32"),
    ("cfg5", 32, 800, 80, "fp16")])
def test_full_size_batch_permutation_and_gradient_accumulation(model, B, T, D, operands):
    """Two size-independent properties at BASELINE's full sizes (no dropout / noise, teacher-forced):
    * nothing in the model couples batch rows except BatchNorm's statistics and the mean of the loss, and both are symmetric:
      permuting the rows of (X, y) permutes enc_states the same way and leaves the loss and every gradient unchanged -- although
      every row then runs in a different batch tile, workgroup and attention slice;
    * gradients are ACCUMULATED (cleargrads is the caller's business, nn.py:177): backward twice without clearing doubles them.
    Every gradient tensor must agree to 5e-5 of its norm -- the Conv+BN tensors too, once the ReLU units whose pre-activation changed
    SIGN between the two evaluations are named and their upstream gradient is dropped in both (see below)."""
    import ctypes as C
    from ast_amd import _lib
    from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
    from oracle.ast_ref import synth_batch
    cfg = _full_cfg(model)
    L, V = 40, cfg["rnn_config"]["dec_vocab_size"]
    X, y = synth_batch(B, T, D, L, V, 20, dtype=np.float32)
    X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    # the pre-activation read-back and the unit kill list are test hooks: they exist in libastk_test.so only (the same sources built with
    # -DASTK_TEST_HOOKS), and inside this block the model runs on that library
    hooks = _lib.load_test_hooks()
    lib = hooks.__enter__()
    try:
        m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
        m.gemm_operands = operands           # -> the descriptors' `gemm_operands` field
        m.inject = {"use_truth": [1] * (L - 1)}
        stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def preact_signs():
            """sign bits of the post-BatchNorm pre-activations of both conv layers, [(b, f*t)][c] per layer, after a forward pass"""
            st = m._cur
            ws = m._workspace("cnn", st["ws_cnn"])
            out = []
            for i, l in enumerate(cfg["cnn_config"]["cnn_layers"]):
                n = lib.astk_conv_bn_relu_workspace_bytes(C.byref(st["cd"]))
                assert n <= ws.numel()
                t2 = C.c_int()
                # rows of layer i = B * F' * T_i: take them from the buffer size the library reports per layer via a probe descriptor
                Ti = (T + 2 * 4 - 9) // 2 + 1
                for _ in range(i):
                    Ti = (Ti + 2 * 4 - 9) // 2 + 1
                F = (D - 13) // 13 + 1
                z = torch.empty(B, F * Ti, l["out_channels"], device="cuda")
                _lib.check(lib.astk_conv_debug_preact(C.byref(st["cd"]), C.c_void_p(ws.data_ptr()), ws.numel(), i, C.c_void_p(z.data_ptr()), stream()))
                out.append(z)
            return out

        def run(Xb, yb, clear=True, want_z=False):
            with using_config("train", True):
                l = m.forward_loss(X=Xb, y=yb, teach_ratio=1.0, random_out=0, add_noise=0)
                z = preact_signs() if want_z else None
                if clear:
                    m.cleargrads()
                l.backward()
            torch.cuda.synchronize()
            return float(l.data), m.arena.grad.clone(), m.enc_states.clone(), z

        l0, g0, e0, z0 = run(X, y, want_z=True)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).cuda()
        assert not bool((perm == torch.arange(B, device="cuda")).all())
        Xp, yp = X[perm].contiguous(), y[perm].contiguous()
        l1, g1, e1, z1 = run(Xp, yp, want_z=True)
        # a row's partial sums are split differently over the stream-K workgroups when it moves (fp32 atomics): 1e-5-level noise that a
        # 200- to 300-step recurrence carries into the loss
        assert abs(l1 - l0) <= (2e-5 if T <= 800 else 1e-4) * abs(l0), (l0, l1)
        assert float((e1 - e0[perm]).abs().max()) <= 2e-4 * float(e0.abs().max())
        gnorm = float(g0.norm())

        def worst(ga, gb):
            w = {"cnn": 0.0, "rest": 0.0}
            for name in m.arena.shapes:
                o, n = m.arena.range_of(name)
                d = float((ga[o:o + n] - gb[o:o + n]).norm()) / max(float(gb[o:o + n].norm()), 1e-6 * gnorm)
                k = "cnn" if name.startswith("CNN_") else "rest"
                w[k] = max(w[k], d)
            return w

        # ---- the ReLU units on which the two evaluations disagree: named, few, and all within rounding of the kink
        flips0, flips1 = [], []          # (layer, row, channel) in the coordinates of run 0 / of the permuted run
        for i, (za, zb) in enumerate(zip(z0, z1)):
            diff = (za[perm] > 0) != (zb > 0)                      # row j of the permuted run is row perm[j] of run 0
            idx = diff.nonzero()
            for j, r, c in idx.tolist():
                va, vb = float(za[perm[j], r, c]), float(zb[j, r, c])
                # (pre-activations are O(1): BatchNorm output; with fp16 operands the conv output itself carries 11-bit products and
                #  the two evaluations' statistics differ a little more)
                # (fp16 bound: 2e-4, the mode's own self-agreement -- the two evaluations of one pre-activation differ by up to ~1.3e-4
                #  there, so a unit at +1.1e-4 in one of them can sit below zero in the other)
                assert max(abs(va), abs(vb)) < (1e-5 if operands == "f32" else 2e-4), (i, j, r, c, va, vb)
                rows_per_b = za.shape[1]
                flips1.append((i, j * rows_per_b + r, c))
                flips0.append((i, int(perm[j]) * rows_per_b + r, c))
        assert len(flips0) <= (8 if operands == "f32" else 24), flips0
        # (fp16 operands: an element of an activation or gradient matrix that sits at a rounding boundary of its 11-bit operand form turns a
        #  1e-7 difference between the two evaluations into a 5e-4 one: the reduced-precision mode agrees with itself to 2e-4 / 1e-3, which
        #  is the drift SURVEY 8d asks to be reported for it, not the f32 gate)
        tol_rest, tol_cnn = (5e-5, 5e-5) if operands == "f32" else (2e-4, 1e-3)
        w = worst(g1, g0)
        assert w["rest"] <= tol_rest, w
        if not flips0:
            assert w["cnn"] <= tol_cnn, w
        else:
            # one unit among N incoherent contributions is ~1/sqrt(N) of a weight gradient's norm (1e-4 .. 3e-3 seen): drop the upstream
            # gradient of exactly those units in BOTH evaluations -- everything else must then agree like the rest of the model
            assert w["cnn"] <= 2e-2, w
            res = []
            for Xb, yb, fl in ((X, y, flips0), (Xp, yp, flips1)):
                units = torch.tensor(fl, dtype=torch.int32, device="cuda").contiguous()
                _lib.check(lib.astk_conv_debug_kill_units(C.c_void_p(units.data_ptr()), len(fl)))
                try:
                    res.append(run(Xb, yb)[1])
                finally:
                    torch.cuda.synchronize()
                    _lib.check(lib.astk_conv_debug_kill_units(None, 0))
            wk = worst(res[1], res[0])
            assert wk["rest"] <= tol_rest and wk["cnn"] <= tol_cnn, (wk, w, flips0)
        # accumulate: same batch again without cleargrads.  (A third evaluation may again sit on the other side of a kink -- the split-tile
        # atomics of the conv GEMMs are not ordered -- and this property is about accumulate-versus-overwrite, an error of 50 % or
        # more: the Conv+BN tensors get the one-unit allowance here.)
        _, g2, _, _ = run(Xp, yp, clear=False)
        w2 = worst(g2, 2 * g1)
        assert w2["rest"] <= tol_rest and w2["cnn"] <= 2e-2, w2
    finally:
        torch.cuda.synchronize()
        hooks.__exit__(None, None, None)


def test_forward_pass_is_bit_reproducible_from_run_to_run():
    """The forward pass at BASELINE configs[1]'s shape, 25 times over the same batch and weights: loss and encoder states bit-identical.  The
    batched products split some output tiles between workgroups and add the pieces with float atomics; two pieces into a zeroed tile
    commute, three do not -- forward entry points therefore run the GEMM schedule that keeps every split tile at two contributors
    (common.h GemmForwardScope; round 5: with three, the soak saw 1-ulp differences and one flipped fed-back argmax in 6000 batch passes).
    The recurrences, the attention scan and the CE role are deterministic by construction."""
    import bench
    from ast_amd.seq2seq import using_config
    import copy
    cfg = copy.deepcopy(bench.MODEL_CFG)
    cfg["dropout"] = {"embed": 0.0, "rnn": 0.0, "out": 0}          # (fresh masks per call would be a difference of their own)
    V = cfg["rnn_config"]["dec_vocab_size"]
    B, T, D, L = 32, 800, 80, 40
    P, X, y = _make(cfg, B, T, D, L, V)
    m = _gpu_model(cfg, P, D, V)
    m.inject["use_truth"] = [1 if (i % 5) else 0 for i in range(L - 1)]
    m.inject["use_truth"][0] = 1
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    ref = None
    with using_config("train", True):
        for it in range(25):
            loss = m.forward_loss(Xd, yd, 0.8)
            got = (float(loss.data), m.enc_states.clone())
            if ref is None:
                ref = got
            else:
                assert got[0] == ref[0], (it, got[0], ref[0])
                assert torch.equal(got[1], ref[1]), it


@pytest.mark.parametrize("model_name,B", [("cfg1", 32), ("es_en_20h", 32), ("cfg1", 64)])
def test_backward_pass_is_bit_reproducible_in_deterministic_mode(model_name, B):
    """model.deterministic (-> astk.h `deterministic` of the three descriptors): the whole train step -- forward AND backward -- at BASELINE
    configs[1]'s shape (and the shipped 3-decoder-layer model, and batch 64), 12 times over two alternating batches at fixed weights: every
    evaluation of a batch must leave the SAME BITS in every gradient.  Without the mode the backward's weight-gradient products sum their
    split tiles with float atomics in arrival order (differences of 2-6e-10 of the clip norm from run to run: harmless, but they hide
    exactly the kind of error a hand-off race in the backward makes -- round 4's was 1e-4 of two tensors once in ~4000 launches); with
    it the split tiles go through the fix-up workspace (gemm.hip), column / embedding / bias sums are ordered, nothing runs on a side
    stream.  Also: the deterministic gradients agree with the default mode's to float-atomics accuracy (same sums, other order)."""
    import bench
    import copy
    from ast_amd.seq2seq import using_config
    cfg = copy.deepcopy(bench.MODEL_CFG)
    if model_name == "es_en_20h":
        cfg["rnn_config"]["dec_layers"] = 3
    V = cfg["rnn_config"]["dec_vocab_size"]
    T, D, L = 800, 80, 40
    P, X, y = _make(cfg, B, T, D, L, V)
    X2 = np.roll(X, 1, axis=0) * 0.9
    m = _gpu_model(cfg, P, D, V)
    m.inject["use_truth"] = [1 if (i % 5) else 0 for i in range(L - 1)]
    m.inject["use_truth"][0] = 1
    sets = [(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()), (torch.from_numpy(np.ascontiguousarray(X2, np.float32)).cuda(), torch.from_numpy(y).cuda())]
    compute = torch.cuda.Stream()

    def evaluate(which):
        m.rng_seed, m._rng_offset = 777, 0                  # the same dropout masks and speech noise in every evaluation
        with torch.cuda.stream(compute), using_config("train", True):
            loss = m.forward_loss(sets[which][0], sets[which][1], 0.8, add_noise=0.25)
            m.cleargrads()
            loss.backward()
            out = (float(loss.data), m.arena.grad.clone())
        torch.cuda.synchronize()
        return out
    m.deterministic = True
    ref = [evaluate(0), evaluate(1)]
    assert m._side is None, "deterministic steps must not use the side stream"
    for it in range(5):
        for which in (1, 0):
            loss, grad = evaluate(which)
            assert loss == ref[which][0], (it, which, loss, ref[which][0])
            assert torch.equal(grad, ref[which][1]), (it, which, float((grad - ref[which][1]).abs().max()))
    m.deterministic = False
    loss, grad = evaluate(0)
    assert loss == ref[0][0]
    scale = float(ref[0][1].abs().max())
    assert float((grad - ref[0][1]).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("dec_layers", [1, 2])
def test_deferred_cleargrads_cannot_be_observed(dec_layers):
    """cleargrads() behind a forward_loss defers its zero fill to the backward pass (the decoder backward's first fill launch takes the
    gradient arena along: include/astk.h zero_ptr, a launch less per train step).  Whoever reads the gradients in between sees zeros; the
    gradients of the step are those of an eagerly cleared arena."""
    from ast_amd.seq2seq import using_config
    cfg = tiny_cfg(enc_layers=1, dec_layers=dec_layers, H=64, E=16, A=64, c0=16, c1=8, V=23, drop=0.0)
    B, T, D, L, V = 3, 48, 80, 6, 23
    P, X, y = _make(cfg, B, T, D, L, V)
    m = _gpu_model(cfg, P, D, V)
    m.inject["use_truth"] = [1] * (L - 1)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    with using_config("train", True):
        # reference: eager zero (cleargrads in FRONT of forward_loss is never deferred)
        m.arena.grad.fill_(7.0)
        m.cleargrads()
        assert not m.arena._zero_pending
        loss = m.forward_loss(Xd, yd, 1.0)
        loss.backward()
        want = m.arena.grad.clone()
        assert float(want.abs().max()) > 0
        # deferred: garbage in the arena, cleargrads behind forward_loss, backward
        m.arena.grad.fill_(7.0)
        loss = m.forward_loss(Xd, yd, 1.0)
        m.cleargrads()
        assert m.arena._zero_pending, "the fill was expected to be deferred here"
        loss.backward()
        assert not m.arena._zero_pending
        assert torch.allclose(m.arena.grad, want, rtol=0, atol=1e-6 * float(want.abs().max())), float((m.arena.grad - want).abs().max())
        # a reader in between sees zeros, and the backward that follows still starts from zero
        m.arena.grad.fill_(7.0)
        loss = m.forward_loss(Xd, yd, 1.0)
        m.cleargrads()
        assert m.arena._zero_pending
        assert float(m.arena.to_numpy(grads=True)["out/W"].max()) == 0.0 and not m.arena._zero_pending
        assert float(m.arena.grad.abs().max()) == 0.0
        loss.backward()
        assert torch.allclose(m.arena.grad, want, rtol=0, atol=1e-6 * float(want.abs().max()))


def test_cnn_backward_refuses_a_workspace_whose_forward_took_the_other_layer0_path():
    """the layer-0 path (direct convolution + window-matrix weight gradient under the default arithmetic, im2col + GEMM
    otherwise) is re-derived at every call from the arithmetic in force then.  With the descriptor at ASTK_PREC_DEFAULT a change of the
    PROCESS default between forward and backward would make the backward read a window matrix that was never written: it must refuse."""
    from ast_amd import _lib
    from ast_amd.seq2seq import using_config
    lib = _lib.load()
    cfg = tiny_cfg(enc_layers=1, dec_layers=1, H=16, E=8, A=16, c0=16, c1=8, V=23, drop=0.0)
    B, T, D, L, V = 2, 48, 80, 5, 23
    P, X, y = _make(cfg, B, T, D, L, V)
    m = _gpu_model(cfg, P, D, V)
    assert m.gemm_precision is None and lib.astk_get_gemm_precision() == 1          # descriptors say DEFAULT, the default is bf16x3
    with using_config("train", True):
        loss = m.forward_loss(torch.from_numpy(X), torch.from_numpy(y), 1.0)
        m.cleargrads()
        prev = lib.astk_set_gemm_precision(2)                                       # the process default moves to the f32 chain ...
        try:
            with pytest.raises(_lib.AstkError, match="layer-0 path"):               # ... and the backward call says so instead of reading stale workspace
                loss.backward()
        finally:
            lib.astk_set_gemm_precision(prev)
            torch.cuda.synchronize()
        loss = m.forward_loss(torch.from_numpy(X), torch.from_numpy(y), 1.0)        # same default on both sides: fine
        m.cleargrads()
        loss.backward()
    assert np.isfinite(float(loss.data))


@pytest.mark.parametrize("dec_layers", [1, 2])     # persistent decoder loop / per-launch decoder loop
def test_overlapped_backward_equals_inline_backward(dec_layers):
    """On a stream of its own the model runs the decoder's parameter gradients (astk_decoder_bwd_phase, grids capped at the CUs the recurrence
    leaves free: astk_decoder_desc.side_wgs) and the library's time-chunked layer-0 products (astk_lstm_stack_desc.side_stream) on an ordinary
    second stream beside the encoder's recurrences.  Same batch, same weights: the loss must equal that of the in-line schedule on the default
    stream bit for bit, the gradients up to the order of float atomics."""
    from ast_amd.seq2seq import using_config
    cfg = tiny_cfg(enc_layers=2, dec_layers=dec_layers, H=128, E=16, A=64, c0=8, c1=16, V=57, drop=0.0)
    B, T, D, L, V = 4, 70, 80, 8, 57
    P, X, y = _make(cfg, B, T, D, L, V)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    res = []
    for own_stream in (False, True):
        m = _gpu_model(cfg, P, D, V)
        m.inject = {"use_truth": [1] * (L - 1)}
        torch.cuda.synchronize()
        s = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
        with torch.cuda.stream(s), using_config("train", True):
            loss = m.forward_loss(X=Xd, y=yd, teach_ratio=1.0)
            m.cleargrads()
            loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss.data), m.arena.grad.clone(), m._side))
    assert res[0][2] is None and res[1][2] is not None, "the side stream was not used"
    assert res[0][0] == res[1][0]
    scale = float(res[0][1].abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * scale


def test_data_parallel_path_against_rccl_with_one_rank():
    """tests/dp_single_rank_rccl.py: a process group of ONE rank on the `nccl` (= RCCL) backend drives the real data-parallel code path
    on this GPU -- bucketed asynchronous all-reduces launched from the backward pass, the BatchNorm statistics exchange through the
    C callback, masked side streams, non-default compute stream -- and must reproduce the plain step (the collectives are
    identities).  Multi-rank semantics are covered by tests/test_dist_gloo.py and the two-replica emulation above."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dp_single_rank_rccl.py")], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1500:] + r.stderr[-1500:]


def test_prefetching_loader_yields_the_same_batches_as_the_plain_one(tmp_path):
    """On a GPU the loader stages batch k+1 on a helper thread into a ring of three reused pinned buffers while batch k is in use
    (ast_amd/dataloader.py:_PinnedRing); the batches must be those of the plain host path -- also when the consumer is slow or
    fast, and across more batches than the ring has slots."""
    import time
    from ast_amd.dataloader import SyntheticDataLoader
    data = {"dataloader": "synthetic", "vocab_size": 31, "feat_dim": 13, "n_utts": {"syn_train": 61, "syn_dev": 5},
            "frames": [20, 400], "targets": [1, 30], "buckets_num": 4, "buckets_width": 80, "max_pred": 12,
            "zero_input": 0.0, "train_scale": 1, "dec_key": "bpe_w"}
    for delay in (0.0, 0.01):
        # fresh loaders per pass: batch_plan shuffles the buckets in place, like the reference
        ref = SyntheticDataLoader(data, str(tmp_path), -1)
        gpu = SyntheticDataLoader(data, str(tmp_path), 0)
        random.seed("seed-ast-20h")
        want = [(b["utts"], b["X"].clone(), b["y"].clone()) for b in ref.get_batch(4, "syn_train", train=True, labels=True)]
        assert len(want) > 9
        random.seed("seed-ast-20h")
        got = []
        for b in gpu.get_batch(4, "syn_train", train=True, labels=True):
            assert b["X"].is_cuda and b["y"].is_cuda
            time.sleep(delay)
            got.append((b["utts"], b["X"].cpu(), b["y"].cpu()))
        assert len(got) == len(want)
        for (u0, x0, y0), (u1, x1, y1) in zip(want, got):
            assert u0 == u1 and torch.equal(x0, x1) and torch.equal(y0, y1)


def test_device_loader_reproduces_the_loader_oracle_with_its_own_draws_injected(tmp_path):
    """The GPU loader (pinned staging ring + frame zeroing ON the uploaded batch, astk_zero_frames) against oracle/loader_ref.py -- the
    restated dataloader.py:83-164.  The reference draws the zeroed frames from NumPy's unseeded global RNG (quirk Q7); the device draws
    them from a counter-based stream.  astk_zero_frames_draws writes that stream out, the oracle's `choice` hook replays it, and every
    batch of two epochs must then agree bit for bit: order, utterances, zeroed frames, padding, targets.  The number of draws per
    utterance must be Python's int(zero_input * frames)."""
    import ctypes as C
    from ast_amd import _lib
    from ast_amd.dataloader import SyntheticDataLoader
    from oracle import loader_ref as LR
    lib = _lib.load()
    rate = 0.1
    data = {"dataloader": "synthetic", "vocab_size": 31, "feat_dim": 13, "n_utts": {"syn_train": 45, "syn_dev": 5},
            "frames": [20, 400], "targets": [1, 30], "buckets_num": 4, "buckets_width": 80, "max_pred": 12,
            "zero_input": rate, "train_scale": 1, "dec_key": "bpe_w"}
    gpu = SyntheticDataLoader(data, str(tmp_path), 0)
    gpu.zero_seed = 0xABCDEF
    host = SyntheticDataLoader(data, str(tmp_path), -1)       # same synthetic corpus (same seeds): source of speech / ids for the oracle
    info = host.info
    vocab = {"bpe_w": {"w2i": {i: i for i in range(31)}}}                                  # ids are their own words
    mp = {k: {u: {"bpe_w": list(map(int, host.ids[k][u]))} for u in host.ids[k]} for k in host.ids}
    buckets = {k: LR.create_buckets(info[k], 4, 80, "sp", 1, "haha") for k in info}

    def oracle(choice):
        bk = {k: {"buckets": [list(b) for b in v["buckets"]], "num_b": v["num_b"], "width_b": v["width_b"]} for k, v in buckets.items()}
        return LR.RefLoader(data, bk, vocab, mp, lambda u, k: host.speech[k][u], choice=choice)
    # pass 1 (no zeroing needed for the plan): batch composition of two epochs
    random.seed("seed-ast-20h")
    ld = oracle(lambda n, k: np.zeros(k, dtype=np.int64))
    plan = [b for _ in range(2) for b in ld.get_batch(4, "syn_train", True, labels=True)]
    # the device's draws for every batch, in the loader's counter order
    draws, off = [], 0
    for b in plan:
        lens = np.array([min(info["syn_train"][u]["sp"], 400) for u in b["utts"]], dtype=np.int32)
        Bk, Tk = b["X"].shape[0], b["X"].shape[1]
        ld_ = torch.from_numpy(lens).cuda()
        idx = torch.full((Bk, 64), -1, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(Bk, dtype=torch.int32, device="cuda")
        _lib.check(lib.astk_zero_frames_draws(Bk, Tk, C.c_void_p(ld_.data_ptr()), rate, 0xABCDEF, off, C.c_void_p(idx.data_ptr()), 64,
                                              C.c_void_p(cnt.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        off += Bk * Tk
        idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
        for i, n_u in enumerate(lens):
            assert int(cnt[i]) == int(rate * int(n_u))                                     # Python's count, in double precision
            draws.append(idx[i, :cnt[i]].astype(np.int64))
    it = iter(draws)

    def replay(n_frames, n_drop):
        d = next(it)
        while len(d) == 0:                  # (the oracle does not ask for utterances that lose no frame)
            d = next(it)
        assert len(d) == n_drop and (d < n_frames).all()
        return d
    random.seed("seed-ast-20h")
    ld = oracle(replay)                 # ONE loader for both epochs: the bucket lists are shuffled in place, epoch after epoch
    want = [b for _ in range(2) for b in ld.get_batch(4, "syn_train", True, labels=True)]
    random.seed("seed-ast-20h")
    got = [{"utts": b["utts"], "X": b["X"].cpu().numpy(), "y": b["y"].cpu().numpy()} for _ in range(2)
           for b in gpu.get_batch(4, "syn_train", train=True, labels=True)]
    assert len(got) == len(want) > 10
    lost = 0
    for w, g in zip(want, got):
        assert w["utts"] == g["utts"]
        assert np.array_equal(w["y"], g["y"]) and np.array_equal(w["X"], g["X"])
        lost += int((w["X"] == 0).all(2).sum())
    assert lost > 0


def test_persistent_kernel_timeout_raises_instead_of_training_on(tune):
    """Every spin of the persistent encoder / decoder kernels is bounded; a time-out (grid not fully resident) drains the grid and
    sets a sticky status word that rides next to the loss scalar (include/astk.h astk_persist_status_snapshot).  Forced here with
    the tuning knob persist.spin_limit = 1 (a wait that is not satisfied by its second poll gives up): reading the loss must raise, through
    float(loss) and through NN.train_epoch's one-step-late read-back helper alike, the word must clear, and the next clean step
    must give the oracle's loss again."""
    from oracle import ast_ref as R
    from ast_amd import _lib
    from ast_amd.seq2seq import using_config, raise_if_aborted
    cfg = tiny_cfg(enc_layers=2, dec_layers=1, H=128, E=16, A=64, c0=8, c1=16, V=57, drop=0.0)     # persistent encoder and decoder
    B, T, D, L, V = 18, 70, 80, 8, 57
    P, X, y = _make(cfg, B, T, D, L, V)
    ref = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    rl = ref.forward_loss(X.astype(np.float64), y, 1.0, pyrandom=random.Random(0))
    g = _gpu_model(cfg, P, D, V)
    g.inject["use_truth"] = [1] * (L - 1)
    lib = _lib.load()
    import ctypes as C
    mask = C.c_uint(0)
    assert lib.astk_persist_status(C.byref(mask), 1) == 0

    def step():
        with using_config("train", True):
            loss = g.forward_loss(X=torch.from_numpy(X), y=torch.from_numpy(y), teach_ratio=1.0)
            g.cleargrads()
            loss.backward()
        return loss
    tune("persist.spin_limit", 1)
    loss = step()
    with pytest.raises(_lib.AstkError, match="timed out"):
        float(loss)
    assert lib.astk_persist_status(C.byref(mask), 0) == 0 and mask.value == 0          # cleared by the raise
    loss = step()
    with pytest.raises(_lib.AstkError, match="timed out"):                             # the accessor nn.py:189 uses: float(loss.data)
        float(loss.data)
    loss = step()
    with pytest.raises(_lib.AstkError, match="timed out"):
        loss.data.item()
    # the update of a timed-out step must not move the parameters or the moments (the host learns of the time-out one step late)
    from ast_amd import optimizers as O
    opt = O.Adam(alpha=1e-3, amsgrad=True).setup(g)
    opt.add_hook(O.WeightDecay(1e-4))
    opt.add_hook(O.GradientClipping(2))
    loss = step()
    before = g.arena.data.clone()
    opt.update()
    torch.cuda.synchronize()
    assert torch.equal(g.arena.data, before) and float(opt.m.abs().max()) == 0.0
    pair = loss.pair.clone()                                                            # what NN.train_epoch keeps for its late read
    with pytest.raises(_lib.AstkError, match="NN.train_epoch"):
        raise_if_aborted(pair.tolist()[1], "NN.train_epoch")
    tune("persist.spin_limit", 0)
    loss = step()
    assert _rel(float(loss.data), float(rl.data)) < 1e-4
    opt.update()
    torch.cuda.synchronize()
    assert not torch.equal(g.arena.data, before)                                       # a clean step updates again


def test_device_loader_zeroes_frames_like_the_host_loader(tmp_path):
    """zero_input on the device path of the loader: every training utterance of a batch has between 1 and int(0.1 T_u) all-zero frames
    inside its own length (the host path's rule, dataloader.py:83-93), targets are untouched, evaluation batches are not zeroed."""
    from ast_amd.dataloader import SyntheticDataLoader
    data = {"dataloader": "synthetic", "vocab_size": 31, "feat_dim": 13, "n_utts": {"syn_train": 40, "syn_dev": 8},
            "frames": [60, 300], "targets": [2, 9], "buckets_num": 4, "buckets_width": 80, "max_pred": 12,
            "zero_input": 0.1, "train_scale": 1, "dec_key": "bpe_w"}
    gpu = SyntheticDataLoader(data, str(tmp_path), 0)
    random.seed("seed-ast-20h")
    n_checked = 0
    for b in gpu.get_batch(8, "syn_train", train=True, labels=True):
        X = b["X"].cpu().numpy()
        for row, u in zip(X, b["utts"]):
            t_u = min(int(gpu.info["syn_train"][u]["sp"]), 400)
            zero = np.where((row[:t_u] == 0).all(axis=1))[0]
            assert 1 <= len(zero) <= int(0.1 * t_u), (u, len(zero), t_u)
            assert (row[t_u:] == 0).all()                       # padding
            n_checked += 1
    assert n_checked >= 36
    for b in gpu.get_batch(8, "syn_dev", train=False, labels=False):
        X = b["X"].cpu().numpy()
        for row, u in zip(X, b["utts"]):
            t_u = int(gpu.info["syn_dev"][u]["sp"])
            assert not (row[:t_u] == 0).all(axis=1).any()


@pytest.mark.parametrize("model,B,T,L", [("cfg1", 16, 400, 20), ("cfg5", 32, 800, 40)])
def test_fp16_operand_gemms_loss_drift_against_fp32(model, B, T, L):
    """BASELINE configs[4] asks for fp16 MFMA GEMMs; SURVEY.md 8(d): the fp32 parity gate does not apply there, report the loss drift
    against fp32 instead.  Same model, batch and weights with gemm_operands "f32" / "fp16" (astk_*_desc.gemm_operands) -- fp16 operands in the batched products
    of K6, K9, K18 and K24 and their backward, everything else f32-accurate: the fp16-operand step's loss and clip norm stay within 2e-3
    of the f32-accurate step's (11 significant bits per operand, f32 accumulation), 3 updates keep the losses within 5e-3, and the mode
    really changes the arithmetic (the results are not bitwise those of the f32 path).  cfg5 = configs[4]'s own shape (6-layer 2 x 512
    encoder, H = A = 1024, V = 8004) at its full batch."""
    import copy
    from ast_amd import _lib, optimizers as O
    from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
    from oracle.ast_ref import synth_batch
    lib = _lib.load()
    cfg = _full_cfg(model)
    D, V = 80, cfg["rnn_config"]["dec_vocab_size"]
    X, y = synth_batch(B, T, D, L, V, 20, dtype=np.float32)
    X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    res = {}
    assert lib.astk_get_low_precision_gemms() == 0
    if True:
        for mode in (0, 1):
            m = SpeechEncoderDecoder(0, copy.deepcopy(cfg)).materialize(D, seed=0)
            m.gemm_operands = ("f32", "fp16")[mode]          # -> the descriptors' `gemm_operands` field (per call, no process-wide state)
            m.inject = {"use_truth": [1] * (L - 1)}
            opt = O.Adam(alpha=1e-3, amsgrad=True).setup(m)
            opt.add_hook(O.WeightDecay(1e-4))
            opt.add_hook(O.GradientClipping(2))
            losses, norms = [], []
            for step in range(3):
                with using_config("train", True):
                    l = m.forward_loss(X=X, y=y, teach_ratio=1.0)
                    m.cleargrads()
                    l.backward()
                    opt.update()
                losses.append(float(l))
                norms.append(opt.last_grad_norm)
            res[mode] = (losses, norms)
    (l32, n32), (l16, n16) = res[0], res[1]
    drift = [abs(a - b) / abs(a) for a, b in zip(l32, l16)]
    print("fp16-operand loss drift per step:", drift, "clip-norm drift:", [abs(a - b) / a for a, b in zip(n32, n16)])
    assert l16[0] != l32[0], "the low-precision mode did not change the arithmetic"
    assert drift[0] < 2e-3 and abs(n16[0] - n32[0]) < 2e-3 * n32[0], (l32, l16, n32, n16)
    assert max(drift) < 5e-3, (l32, l16)
