"""GPU parity tests of the reference model's OPTIONAL features (seq2seq.py:43-57, 81-121, 244-291, 369-394,
456-465): rnn_config.ln / linear_proj / n_attn > 1 / feed_attn = false, cnn_config.bn = false, dropout.out, forward_loss's random_out;
the old path's cnn_pool (enc_dec.py:444-456).
Every case is one full train step of the HIP path (through the C ABI) against the float64 CPU oracle on identical inputs, weights,
dropout masks and random draws -- loss and clip norm within 1e-4, every gradient tensor, the encoder states, and the parameters after
the update -- plus greedy decoding in eval mode, and op-level checks of the new kernels (csrc/norm.hip)."""
import copy
import ctypes as C
import random

import numpy as np
import pytest
import torch

from conftest import tiny_cfg

pytestmark = pytest.mark.gpu

OPT = {"type": 0, "lr": 1e-3, "l2": 1e-4, "grad_clip": 2, "grad_noise_eta": 0, "freeze": []}


def _cfg(enc_layers=2, dec_layers=2, H=64, E=16, A=32, c0=8, c1=16, V=41, drop=0.0, out=0.0, **rc):
    cfg = tiny_cfg(enc_layers=enc_layers, dec_layers=dec_layers, H=H, E=E, A=A, c0=c0, c1=c1, V=V, drop=drop)
    cfg["dropout"]["out"] = out
    bn = rc.pop("bn", True)
    cfg["cnn_config"]["bn"] = bn
    if "cnn_pool" in rc:
        cfg["cnn_config"]["cnn_pool"] = rc.pop("cnn_pool")
    cfg["rnn_config"].update(rc)
    return cfg


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


CASES = {
    # name: (cfg kwargs, B, T, D, L)
    "ln": (dict(ln=True), 5, 64, 80, 7),
    "ln-drop-3x1": (dict(ln=True, drop=0.3, enc_layers=3, dec_layers=1), 4, 64, 80, 7),
    "ln-persist-h64": (dict(ln=True, H=128, A=64, enc_layers=2, dec_layers=2), 18, 70, 80, 8),   # one-layer stacks on the persistent encoder kernels
    "n_attn3": (dict(n_attn=3), 5, 64, 80, 7),
    "n_attn2-drop": (dict(n_attn=2, drop=0.3, dec_layers=1), 4, 64, 13, 6),
    "no-feed": (dict(feed_attn=False), 5, 64, 80, 7),
    "no-bn": (dict(bn=False), 5, 64, 80, 7),
    "no-bn-13d": (dict(bn=False, drop=0.2), 3, 90, 13, 6),
    "out-drop": (dict(out=0.4, drop=0.2), 5, 64, 80, 7),
    "proj-3": (dict(linear_proj=True, enc_layers=3, dec_layers=2), 5, 64, 80, 7),
    "proj-2-drop": (dict(linear_proj=True, enc_layers=2, dec_layers=2, drop=0.3), 4, 64, 80, 6),
    "proj-persist-h64": (dict(linear_proj=True, H=128, A=64, enc_layers=2, dec_layers=1), 17, 70, 80, 6),
    # OLD-path extra (enc_dec.py:444-456): max-pool between convolution and BatchNorm, (time, frequency) windows per layer, -1 = whole extent,
    # cover_all (ragged last windows: 19 frames by 3; 3 bins by 2)
    "pool-t2": (dict(cnn_pool=[[2, 1], [1, 1]]), 5, 64, 80, 7),
    "pool-f2-t3-ragged": (dict(cnn_pool=[[1, 2], [3, 2]], drop=0.2), 4, 76, 80, 6),
    "pool-whole-freq": (dict(cnn_pool=[[1, -1], [2, 1]]), 5, 64, 80, 7),
    "pool-nobn": (dict(bn=False, cnn_pool=[[2, 2], [1, 1]], drop=0.2), 3, 90, 80, 6),
    "all": (dict(ln=True, n_attn=2, feed_attn=False, bn=False, out=0.3, drop=0.2, enc_layers=2, dec_layers=3), 4, 64, 80, 7),
}


def _oracle_step(cfg, P, X, y, V, teach, drop, dt=np.float64, random_out=0, randint=None, seed="seed-ast-20h"):
    from oracle import ast_ref as R
    m = R.RefModel(cfg, {k: v.astype(dt) for k, v in P.items()}, V)
    rec = R.RecordingMasks(3) if drop else None
    if rec:
        m.masks = rec
    noise = np.random.default_rng(9).normal(1.0, 0.25, X.shape).astype(np.float32) if drop else None
    m.train = True
    loss = m.forward_loss(X.astype(dt), y, teach, random_out, 0.25 if drop else 0, noise, random.Random(seed), randint)
    m.cleargrads()
    loss.backward()
    grads = {k: (p.grad.copy() if p.grad is not None else np.zeros_like(p.data)) for k, p in m.params()}
    opt = R.RefOptimizer(m, OPT)
    opt.update()
    return dict(loss=float(loss.data), gnorm=opt.last_grad_norm, grads=grads, model=m, rec=rec, noise=noise, flags=list(m.use_truth),
                enc=m.enc_states.data.copy(), after={k: p.data.copy() for k, p in m.params()})


def _gpu(cfg, P, D, V):
    from ast_amd.seq2seq import SpeechEncoderDecoder
    c = copy.deepcopy(cfg)
    c["rnn_config"]["dec_vocab_size"] = V
    return SpeechEncoderDecoder(0, c).materialize(D, values=P)


@pytest.mark.parametrize("name", sorted(CASES))
def test_optional_features_train_step_parity(name, gemm_scheme):
    from oracle import ast_ref as R
    from oracle.ast_ref_torch import masks_from_recording
    from ast_amd import optimizers as O
    from ast_amd.seq2seq import using_config
    kw, B, T, D, L = CASES[name]
    cfg = _cfg(**kw)
    V = cfg["rnn_config"]["dec_vocab_size"]
    drop = max(cfg["dropout"].values()) > 0
    P = R.init_params(cfg, D, V, seed=2, dtype=np.float32)
    rng = np.random.default_rng(12)
    for k in P:          # LayerNorm / BatchNorm scales and shifts and the plain biases away from their 1 / 0 initial values
        if k.endswith(("gamma", "beta", "/b")) and "upward" not in k:
            P[k] = (P[k] + 0.2 * rng.standard_normal(P[k].shape)).astype(np.float32)
    X, y = R.synth_batch(B, T, D, L, V, seed=3, dtype=np.float32)
    ref = _oracle_step(cfg, P, X, y, V, 0.6, drop)
    g = _gpu(cfg, P, D, V)
    g.gemm_precision = gemm_scheme
    assert g.paths()["options"], name
    if drop:
        packed = masks_from_recording(cfg, ref["rec"].masks, ref["enc"].shape[1], L - 1, B)
        g.inject = {k: torch.from_numpy(v) for k, v in packed.items()}
        g.inject["noise"] = torch.from_numpy(ref["noise"])
    g.inject["use_truth"] = ref["flags"]
    opt = O.Adam(alpha=1e-3, amsgrad=True).setup(g)
    opt.add_hook(O.WeightDecay(1e-4))
    opt.add_hook(O.GradientClipping(2))
    with using_config("train", True):
        loss = g.forward_loss(X=torch.from_numpy(X), y=torch.from_numpy(y), teach_ratio=0.6, add_noise=0.25 if drop else 0)
        g.cleargrads()
        loss.backward()
        grads = g.arena.to_numpy(grads=True)
        opt.update()
    torch.cuda.synchronize()
    np.testing.assert_allclose(g.enc_states.cpu().numpy(), ref["enc"], rtol=0, atol=2e-4 * np.abs(ref["enc"]).max(), err_msg="enc_states")
    assert _rel(float(loss.data), ref["loss"]) < 1e-4, (name, float(loss.data), ref["loss"])
    assert _rel(opt.last_grad_norm, ref["gnorm"]) < 1e-4, (name, opt.last_grad_norm, ref["gnorm"])
    assert set(grads) == set(ref["grads"])
    gmax = max(np.abs(v).max() for v in ref["grads"].values())
    for k, want in ref["grads"].items():
        err = np.abs(grads[k] - want).max()
        tol = 3e-4 * max(np.abs(want).max(), 1e-3 * gmax)
        if k.startswith("enc_proj") and k.endswith("/b"):
            # a bias in front of a BatchNorm has NO gradient (the normalisation removes it): the exact value is 0 (1e-17 in the float64
            # oracle) and what float32 leaves is the rounding of a sum of dz entries that cancel -- priced against the weight's gradient
            tol = 3e-4 * np.abs(ref["grads"][k[:-1] + "W"]).max()
        assert err <= tol, f"{name}: grad {k}: err {err:.3e} tol {tol:.3e}"
    after = g.arena.to_numpy()
    num = sum(float(((after[k].astype(np.float64) - ref["after"][k]) ** 2).sum()) for k in after)
    den = sum(float(((ref["after"][k] - P[k]) ** 2).sum()) for k in after)
    assert num <= (2e-3) ** 2 * den, (name, num, den)
    if cfg["rnn_config"].get("linear_proj"):
        # the projection's BatchNorm: running statistics after T'' sequential per-step updates, and its call counter
        T2 = ref["enc"].shape[1]
        assert g.proj_bn_N[0] == T2 == ref["model"].bn["enc_proj0_bn"].N
        for s_ in ("avg_mean", "avg_var"):
            np.testing.assert_allclose(g.persist[f"enc_proj0_bn/{s_}"].cpu().numpy(), ref["model"].p[f"enc_proj0_bn/{s_}"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["ln", "n_attn3", "no-feed", "no-bn", "proj-3", "all"])
def test_optional_features_greedy_predict_matches_oracle(name):
    """Eval mode (chainer.config.train = False): BatchNorm on running statistics, dropout off, LayerNorm / extra heads / no input
    feeding as in training -- the argmax token sequences of predict() must be the oracle's, token for token."""
    from oracle import ast_ref as R
    kw, B, T, D, L = CASES[name]
    cfg = _cfg(**kw)
    V = cfg["rnn_config"]["dec_vocab_size"]
    P = R.init_params(cfg, D, V, seed=5, dtype=np.float32)
    rng = np.random.default_rng(13)
    for k in P:
        if k.endswith(("gamma", "beta", "/b", "avg_mean")) and "upward" not in k:
            P[k] = (P[k] + 0.2 * rng.standard_normal(P[k].shape)).astype(np.float32)
    P["out/W"] = (P["out/W"] * 4).astype(np.float32)          # wider logit margins: argmax ties would make the comparison meaningless
    X, _ = R.synth_batch(B, T, D, L, V, seed=6, dtype=np.float32)
    ref = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    want = ref.predict(X.astype(np.float64), stop_limit=6)
    g = _gpu(cfg, P, D, V)
    got = g.predict(torch.from_numpy(X), 1, 2, 6)
    np.testing.assert_array_equal(got, want)


def test_random_out_replaces_scored_targets_like_the_reference():
    """forward_loss(random_out > 0), seq2seq.py:456-465: the draws come from the same Python `random` stream as the teacher-forcing
    coins, in the reference's order; the replacement ids from an injected randint (the reference's is the unseeded global RNG); ids of
    dec_vocab_size -- which the reference can draw, quirk Q8 -- are clamped to dec_vocab_size - 1.  The fed tokens stay the true ones."""
    from oracle import ast_ref as R
    from ast_amd.seq2seq import using_config
    for persist in (False, True):
        # (persist: the persistent decoder loop -- its CE role and its post kernel read the scored-targets matrix; else: the per-launch
        #  loop, forced by a second attention head)
        cfg = _cfg(H=128, A=64, dec_layers=1, V=57) if persist else _cfg(n_attn=2)
        V = cfg["rnn_config"]["dec_vocab_size"]
        B, T, D, L = (18, 70, 80, 8) if persist else (5, 64, 80, 7)
        P = R.init_params(cfg, D, V, seed=2, dtype=np.float32)
        X, y = R.synth_batch(B, T, D, L, V, seed=3, dtype=np.float32)
        ids = list(np.random.default_rng(1).integers(4, V + 1, size=400))
        ids[0] = V                                              # the out-of-range id
        it1, it2 = iter(ids), iter(ids)
        ref = _oracle_step(cfg, P, X, y, V, 0.7, False, random_out=0.5, randint=lambda lo, hi: next(it1), seed=11)
        assert any((t != y[:, i + 1]).any() for i, t in enumerate(ref["model"].targets))
        g = _gpu(cfg, P, D, V)
        g.inject["randint"] = lambda lo, hi: next(it2)
        random.seed(11)
        with using_config("train", True):
            loss = g.forward_loss(X=torch.from_numpy(X), y=torch.from_numpy(y), teach_ratio=0.7, random_out=0.5)
            g.cleargrads()
            loss.backward()
        assert g.use_truth == [int(f) for f in ref["flags"]]
        assert _rel(float(loss.data), ref["loss"]) < 1e-4, (persist, float(loss.data), ref["loss"])
        grads = g.arena.to_numpy(grads=True)
        gmax = max(np.abs(v).max() for v in ref["grads"].values())
        for k, want in ref["grads"].items():
            assert np.abs(grads[k] - want).max() <= 3e-4 * max(np.abs(want).max(), 1e-3 * gmax), (persist, k)
        import ast_amd._lib as L_
        assert bool(L_.load().astk_decoder_path(C.byref(g._cur["dd"])) & 1) == persist


def _vp(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("rows,n,ld", [(7, 32, 32), (130, 256, 512), (33, 100, 104), (2000, 512, 512)])
def test_layernorm_kernels(rows, n, ld):
    """astk_layernorm_fwd / _bwd against a float64 torch restatement of F.layer_normalization (biased variance, eps inside the root),
    strided rows (halves of a wider buffer), accumulating parameter gradients."""
    from ast_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(rows + n)
    x = torch.randn(rows, ld, device="cuda", generator=gen) * 2 + 0.5
    gamma = torch.randn(n, device="cuda", generator=gen)
    beta = torch.randn(n, device="cuda", generator=gen)
    dy = torch.randn(rows, ld, device="cuda", generator=gen)
    y = torch.zeros(rows, ld, device="cuda")
    _lib.check(lib.astk_layernorm_fwd(rows, n, _vp(x), ld, _vp(gamma), _vp(beta), 1e-6, _vp(y), ld, _stream()))
    xd = x[:, :n].double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mu = xd.mean(1, keepdim=True)
    var = ((xd - mu) ** 2).mean(1, keepdim=True)
    yd = (xd - mu) / torch.sqrt(var + 1e-6) * gd + bd
    assert float((y[:, :n].double() - yd).abs().max()) < 1e-5 * float(yd.abs().max())
    assert float(y[:, n:].abs().max()) == 0.0 if ld > n else True
    yd.backward(dy[:, :n].double())
    dx = torch.zeros(rows, ld, device="cuda")
    dgam, dbet = torch.ones(n, device="cuda"), torch.full((n,), 2.0, device="cuda")            # accumulated into
    _lib.check(lib.astk_layernorm_bwd(rows, n, _vp(x), ld, _vp(gamma), 1e-6, _vp(dy), ld, _vp(dx), ld, _vp(dgam), _vp(dbet), _stream()))
    assert float((dx[:, :n].double() - xd.grad).abs().max()) < 2e-5 * float(xd.grad.abs().max())
    assert float((dgam.double() - 1 - gd.grad).abs().max()) < 2e-5 * float(gd.grad.abs().max()) + 1e-5
    assert float((dbet.double() - 2 - bd.grad).abs().max()) < 2e-5 * float(bd.grad.abs().max()) + 1e-5


@pytest.mark.parametrize("T,B,Cc", [(5, 3, 8), (40, 32, 64), (17, 2, 100)])
def test_step_batchnorm_relu_kernels(T, B, Cc):
    """astk_step_bn_relu_fwd / _bwd: per-time-step BatchNorm over the B rows of the step + ReLU, running statistics advanced T times in step
    order with the unbiased-variance factor B / (B - 1) (Chainer-sem A4 on a 2-D input), eval mode on the running statistics."""
    from ast_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(T * B + Cc)
    z = torch.randn(T, B, Cc, device="cuda", generator=gen) * 1.5 + 0.3
    gamma, beta = torch.randn(Cc, device="cuda", generator=gen), torch.randn(Cc, device="cuda", generator=gen)
    am0, av0 = torch.randn(Cc, device="cuda", generator=gen), torch.rand(Cc, device="cuda", generator=gen) + 0.5
    am, av = am0.clone(), av0.clone()
    out, stats = torch.empty_like(z), torch.empty(T, 2, Cc, device="cuda")
    _lib.check(lib.astk_step_bn_relu_fwd(T, B, Cc, _vp(z), _vp(gamma), _vp(beta), _vp(am), _vp(av), 2e-5, 0.9, 1, _vp(out), _vp(stats), _stream()))
    zd = z.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mu = zd.mean(1, keepdim=True)
    var = ((zd - mu) ** 2).mean(1, keepdim=True)
    od = torch.relu((zd - mu) / torch.sqrt(var + 2e-5) * gd + bd)
    assert float((out.double() - od).abs().max()) < 2e-5 * float(od.abs().max())
    wm, wv = am0.double(), av0.double()
    for t in range(T):
        wm = 0.9 * wm + 0.1 * mu[t, 0].detach()
        wv = 0.9 * wv + 0.1 * (B / max(B - 1.0, 1.0)) * var[t, 0].detach()
    assert float((am.double() - wm).abs().max()) < 1e-5 and float((av.double() - wv).abs().max()) < 1e-5 * float(wv.abs().max()) + 1e-5
    d_out = torch.randn(T, B, Cc, device="cuda", generator=gen)
    od.backward(d_out.double())
    dz = torch.empty_like(z)
    dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    _lib.check(lib.astk_step_bn_relu_bwd(T, B, Cc, _vp(z), _vp(stats), _vp(gamma), 2e-5, _vp(out), _vp(d_out), _vp(dz), _vp(dg), _vp(db), _stream()))
    assert float((dz.double() - zd.grad).abs().max()) < 5e-5 * float(zd.grad.abs().max())
    assert float((dg.double() - gd.grad).abs().max()) < 5e-5 * float(gd.grad.abs().max()) + 1e-5
    assert float((db.double() - bd.grad).abs().max()) < 5e-5 * float(bd.grad.abs().max()) + 1e-5
    # eval mode: the running statistics, no update
    am2, av2 = am.clone(), av.clone()
    _lib.check(lib.astk_step_bn_relu_fwd(T, B, Cc, _vp(z), _vp(gamma), _vp(beta), _vp(am2), _vp(av2), 2e-5, 0.9, 0, _vp(out), None, _stream()))
    oe = torch.relu((z.double() - am.double()) / torch.sqrt(av.double() + 2e-5) * gamma.double() + beta.double())
    assert float((out.double() - oe).abs().max()) < 2e-5 * float(oe.abs().max()) and torch.equal(am2, am) and torch.equal(av2, av)


def test_weight_noise_of_the_old_path():
    """enc_dec.py:587-624 / nmt_run.py:850-853: N(mu, sigma) on every LSTM's upward W, upward b, lateral W and on the decoder embedding,
    nothing else.  With the oracle's draws replayed the parameters must be the oracle's bit for bit; drawn on the device they must have
    the asked mean and spread, and differ from call to call."""
    from oracle import ast_ref as R
    cfg = _cfg(enc_layers=2, dec_layers=2)
    V, D = cfg["rnn_config"]["dec_vocab_size"], 80
    P = R.init_params(cfg, D, V, seed=2, dtype=np.float32)
    ref = R.RefModel(cfg, {k: v.copy() for k, v in P.items()}, V)
    rng = np.random.default_rng(3)
    draws = ref.add_weight_noise(0.01, 0.05, lambda mu, sigma, shape: rng.normal(mu, sigma, shape).astype(np.float32))
    g = _gpu(cfg, P, D, V)
    g.inject["weight_noise"] = {k: torch.from_numpy(v) for k, v in draws.items()}
    g.add_weight_noise(0.01, 0.05)
    after = g.arena.to_numpy()
    touched = set(draws)
    assert touched == {k for k in P if ("/upward/" in k or "/lateral/" in k) or k == "embed_dec/W"}
    for k in after:
        want = ref.p[k].data if k in touched else P[k]
        assert np.array_equal(after[k], want), k
    g2 = _gpu(cfg, P, D, V)
    g2.add_weight_noise(0.01, 0.05)
    a1 = g2.arena.to_numpy()
    g2.add_weight_noise(0.0, 0.05)
    a2 = g2.arena.to_numpy()
    d1 = np.concatenate([(a1[k] - P[k]).ravel() for k in sorted(touched)])
    d2 = np.concatenate([(a2[k] - a1[k]).ravel() for k in sorted(touched)])
    assert abs(d1.mean() - 0.01) < 5 * 0.05 / np.sqrt(d1.size) + 1e-4 and abs(d1.std() - 0.05) < 0.01 * 0.05
    assert abs(d2.mean()) < 5 * 0.05 / np.sqrt(d2.size) + 1e-4 and abs(np.corrcoef(d1, d2)[0, 1]) < 0.01
    assert all(np.array_equal(a2[k], P[k]) for k in a2 if k not in touched)


@pytest.mark.parametrize("name", ["all", "proj-3"])
def test_beam_search_and_checkpoints_with_optional_features(name, tmp_path):
    """The callers either side of the train step on a model WITH the optional features: (i) a Chainer-layout checkpoint written by the
    oracle after one train step (LayerNorm gamma / beta under '<link>_ln/', extra heads 'attn_Wa1/', conv biases 'CNN_i/b', the projection's
    'enc_proj{i}/' + '_bn/' links with avg_mean / avg_var / N = one count per TIME STEP) loads into a HIP model; (ii) beam search through
    the state API (nn.py:235-322) finds the oracle's N-best hypotheses and scores; (iii) the HIP model's own checkpoint loads back into
    the oracle with the same counters."""
    from oracle import ast_ref as R
    from ast_amd import nn as gnn, serializers
    from ast_amd.seq2seq import SpeechEncoderDecoder
    kw, B, T, D, L = CASES[name]
    kw = {k: v for k, v in kw.items() if k not in ("drop", "out")}
    if name == "all":
        # (beam search seeds the decoder with the encoder's final states, layer by layer: set_decoder_states -- the reference's too,
        #  seq2seq.py:562-568 -- needs at least as many encoder layers as decoder layers)
        kw.update(enc_layers=3, dec_layers=3)
    cfg = _cfg(**kw)
    V = cfg["rnn_config"]["dec_vocab_size"]
    P = R.init_params(cfg, D, V, seed=8, dtype=np.float32)
    P["out/W"] = (P["out/W"] * 4).astype(np.float32)
    X, y = R.synth_batch(B, T, D, L, V, seed=9, dtype=np.float32)
    ref = R.RefModel(cfg, {k: v.astype(np.float64) for k, v in P.items()}, V)
    opt = R.RefOptimizer(ref, OPT)
    R.train_step(ref, opt, X.astype(np.float64), y, 1.0, pyrandom=random.Random(0))
    path = str(tmp_path / "seq2seq_1.model")
    R.save_npz(path, ref)
    z = np.load(path)
    T2 = ref.enc_states.shape[1]
    if name == "proj-3":
        assert int(z["enc_proj0_bn/N"]) == T2 and "enc_proj1/W" in z.files and "enc_proj2/W" not in z.files
    else:
        assert {"L1_rev_enc_ln/gamma", "L2_dec_ln/beta", "attn_Wa1/W", "CNN_1/b"} <= set(z.files) and "CNN_0_bn/N" not in z.files
    c = copy.deepcopy(cfg)
    g = SpeechEncoderDecoder(0, c)
    serializers.load_npz(path, g)
    assert g.V == V             # (in_dim is inferred from the first LSTM's fan-in: 78 of the 80 feature dims reach the conv)
    if name == "proj-3":
        assert g.proj_bn_N == [T2, T2]
    X1 = X[:1]
    want = R.decode_beam(ref, X1.astype(np.float64), stop_limit=6, N=3, K=4)
    got = gnn.decode_beam(g, torch.from_numpy(X1), stop_limit=6, N=3, K=4)
    assert len(got) == len(want) == 3
    for a, b in zip(got, want):
        assert a["hyp"] == b["hyp"], (a["hyp"], b["hyp"])
        assert abs(a["score"] - b["score"]) <= 1e-4 * max(1.0, abs(b["score"]))
        np.testing.assert_allclose(a["attn_history"][-1], b["attn_history"][-1], rtol=0, atol=1e-5)
    path2 = str(tmp_path / "seq2seq_2.model")
    serializers.save_npz(path2, g)
    ref2 = R.RefModel(cfg, {k: np.zeros_like(v, dtype=np.float64) for k, v in P.items()}, V)
    R.load_npz(path2, ref2)
    for k, p in ref.params():
        np.testing.assert_allclose(ref2.p[k].data, p.data, rtol=0, atol=1e-6 * max(1.0, float(np.abs(p.data).max())), err_msg=k)
    if name == "proj-3":
        assert ref2.bn["enc_proj1_bn"].N == T2
