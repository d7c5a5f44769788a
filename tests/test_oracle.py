"""CPU tests that pin the oracle (it has no reference vectors to lean on -- 'parity unpinned'):
float64 finite differences, an independent torch-autograd restatement, and hand-derivable
known-answer checks for the Chainer semantics listed in SURVEY.md Appendix A / B."""
import random

import numpy as np
import pytest
import torch

from oracle import ast_ref as R
from oracle import minichainer as F
from oracle.ast_ref_torch import forward_loss_torch
from conftest import tiny_cfg


def _setup(dtype=np.float64, drop=0.0, enc_layers=2, dec_layers=2, B=3, T=21, D=26, L=6, V=11, seed=0):
    cfg = tiny_cfg(enc_layers=enc_layers, dec_layers=dec_layers, drop=drop, V=V)
    P = R.init_params(cfg, D, V, seed=seed, dtype=dtype)
    X, y = R.synth_batch(B, T, D, L, V, seed=seed + 1, dtype=dtype)
    return cfg, P, X, y, V


def _loss(cfg, P, X, y, V, teach=1.0, masks=None, noise=None):
    m = R.RefModel(cfg, {k: v.copy() for k, v in P.items()}, V)
    if masks is not None:
        m.masks = masks
    loss = m.forward_loss(X, y, teach, add_noise=0.25 if noise is not None else 0, noise=noise,
                          pyrandom=random.Random("seed-ast-20h"))
    return m, loss


def test_finite_differences_f64():
    cfg, P, X, y, V = _setup()
    m, loss = _loss(cfg, P, X, y, V)
    m.cleargrads()
    loss.backward()
    rng = np.random.default_rng(5)
    for k, p in m.params():
        g = p.grad
        assert g is not None, k
        for _ in range(2):
            idx = tuple(int(rng.integers(0, s)) for s in g.shape)
            eps = 1e-6
            Pp = {n: v.copy() for n, v in P.items()}
            Pm = {n: v.copy() for n, v in P.items()}
            Pp[k][idx] += eps
            Pm[k][idx] -= eps
            num = (float(_loss(cfg, Pp, X, y, V)[1].data) - float(_loss(cfg, Pm, X, y, V)[1].data)) / (2 * eps)
            assert abs(num - g[idx]) <= 1e-6 * max(1.0, abs(num)) + 2e-8, (k, idx, num, g[idx])


@pytest.mark.parametrize("drop,enc_layers,dec_layers,teach", [(0.0, 2, 2, 1.0), (0.3, 3, 1, 0.5), (0.3, 1, 2, 0.8)])
def test_against_independent_torch_restatement(drop, enc_layers, dec_layers, teach):
    cfg, P, X, y, V = _setup(drop=drop, enc_layers=enc_layers, dec_layers=dec_layers, L=7)
    masks = R.RecordingMasks(3) if drop > 0 else None
    noise = np.random.default_rng(9).normal(1.0, 0.25, X.shape) if drop > 0 else None
    m, loss = _loss(cfg, P, X, y, V, teach, masks, noise)
    m.cleargrads()
    loss.backward()
    Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=R.is_trainable(k)) for k, v in P.items()}
    lt, enc_t = forward_loss_torch(cfg, Pt, torch.tensor(X), y, m.use_truth, V,
                                   masks=masks.masks if masks else None,
                                   noise=torch.tensor(noise) if noise is not None else None)
    lt.backward()
    assert abs(float(lt.detach()) - float(loss.data)) < 1e-10 * max(1, abs(float(lt.detach())))
    np.testing.assert_allclose(m.enc_states.data, enc_t.detach().numpy(), rtol=1e-10, atol=1e-12)
    for k, p in m.params():
        np.testing.assert_allclose(p.grad, Pt[k].grad.numpy(), rtol=1e-8, atol=1e-11, err_msg=k)


def test_lstm_gate_interleave_A1():
    # unit j gate k lives in column 4j+k; a=tanh, i,f,o = sigmoid
    z = np.zeros((1, 8))
    z[0, 0:4] = [0.5, 100, -100, 100]      # unit 0: a=.5, i->1, f->0, o->1
    z[0, 4:8] = [100, -100, 100, 100]      # unit 1: i->0, f->1: c = c_prev
    c_prev = np.array([[3.0, 0.25]])
    c, h = F.lstm(F.Variable(c_prev), F.Variable(z))
    np.testing.assert_allclose(c.data, [[np.tanh(0.5), 0.25]], atol=1e-12)
    np.testing.assert_allclose(h.data, [[np.tanh(np.tanh(0.5)), np.tanh(0.25)]], atol=1e-12)


def test_softmax_ce_weight_and_denominator_A6():
    x = np.log(np.array([[0.5, 0.25, 0.25], [0.1, 0.2, 0.7], [0.3, 0.3, 0.4]]))
    t = np.array([1, 0, 2])                # middle row is PAD: weight 0 but counted in the denominator
    w = np.array([0.0, 1.0, 1.0])
    xv = F.Variable(x)
    loss = F.softmax_cross_entropy(xv, t, class_weight=w)
    assert abs(float(loss.data) - (-(np.log(0.25) + np.log(0.4)) / 3)) < 1e-12
    loss.backward()
    np.testing.assert_allclose(xv.grad[1], 0, atol=1e-15)
    np.testing.assert_allclose(xv.grad[0], (np.array([0.5, 0.25, 0.25]) - [0, 1, 0]) / 3, atol=1e-12)


def test_reverse_stack_order_Q1():
    """The reverse stack consumes frames 0, T''-1, ..., 1; row p of its half of enc_states is the state
    after frame p+1 (p <= T''-2) and the last row is the state after frame 0 only."""
    cfg, P, X, y, V = _setup(enc_layers=1, dec_layers=1)
    m = R.RefModel(cfg, P, V)
    feats = m.forward_cnn(F.Variable(X))
    m.forward_rnn_encode(feats)
    T2 = feats.shape[0]
    link = F.LSTMLink(*(F.Variable(P[f"L0_rev_enc/{n}"]) for n in ("upward/W", "upward/b", "lateral/W")))
    first = link(F.Variable(feats.data[0])).data
    Hh = cfg["rnn_config"]["hidden_units"] // 2
    np.testing.assert_allclose(m.enc_states.data[:, T2 - 1, Hh:], first, atol=1e-12)
    second = link(F.Variable(feats.data[T2 - 1])).data
    np.testing.assert_allclose(m.enc_states.data[:, T2 - 2, Hh:], second, atol=1e-12)


def test_teacher_forcing_flags_Q4():
    rnd = random.Random("seed-ast-20h")
    flags = R.teacher_flags(8, 0.8, rnd)
    assert len(flags) == 7 and flags[0] and flags[-1]
    rnd2 = random.Random("seed-ast-20h")
    draws = [rnd2.random() < 0.8 for _ in range(5)]
    assert flags[1:6] == draws
    # str seeding is sha512-based and stable across CPython 3.x: pin the first draws
    rnd3 = random.Random("seed-ast-20h")
    assert [rnd3.random() for _ in range(4)] == pytest.approx(
        [0.12586440007881605, 0.06907301332172022, 0.6615765874083718, 0.36126063757535964], abs=1e-15)


def test_hooks_and_amsgrad_A7_A8():
    class M:
        def __init__(self):
            self.p = {"a/W": F.Parameter(np.array([3.0, -4.0])), "b/W": F.Parameter(np.array([12.0]))}

        def params(self):
            return list(self.p.items())
    m = M()
    cfg = {"type": 0, "lr": 0.1, "l2": 0.5, "grad_clip": 2, "grad_noise_eta": 0, "freeze": []}
    opt = R.RefOptimizer(m, cfg)
    m.p["a/W"].grad = np.array([1.5, 2.0])
    m.p["b/W"].grad = np.array([0.0])
    opt.update()
    # decay first: g = [3, 0], [6]; norm = sqrt(45); rate = 2/sqrt(45) < 1 -> clipped
    g = np.array([3.0, 0.0, 6.0]) * (2 / np.sqrt(45))
    assert abs(opt.last_grad_norm - np.sqrt(45)) < 1e-12
    m1, v1 = 0.1 * g, 0.001 * g * g
    lr1 = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = np.array([3.0, -4.0, 12.0]) - lr1 * m1 / (np.sqrt(v1) + 1e-8)
    np.testing.assert_allclose(np.concatenate([m.p["a/W"].data, m.p["b/W"].data]), exp, rtol=1e-12)
    # second step with a smaller gradient: vhat keeps the max
    m.p["a/W"].grad = np.array([0.0, 0.0])
    m.p["b/W"].grad = np.array([0.0])
    for p in m.p.values():
        p.data[...] = 0
    opt.update()
    st = opt.state["b/W"]
    assert st["vhat"][0] == pytest.approx(v1[2]) and st["v"][0] == pytest.approx(0.999 * v1[2])


def test_batchnorm_train_and_running_stats_A4():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 4, 5)) * 2 + 1
    bn = F.BatchNormState(F.Parameter(np.ones(3)), F.Parameter(np.zeros(3)), np.zeros(3), np.ones(3))
    y = bn(F.Variable(x), train=True)
    np.testing.assert_allclose(y.data.mean(axis=(0, 2, 3)), 0, atol=1e-12)
    m = 2 * 4 * 5
    np.testing.assert_allclose(bn.avg_var, 0.9 + 0.1 * x.var(axis=(0, 2, 3)) * m / (m - 1), rtol=1e-12)
    np.testing.assert_allclose(bn.avg_mean, 0.1 * x.mean(axis=(0, 2, 3)), rtol=1e-12)
    xt = torch.tensor(x)
    yt = torch.nn.functional.batch_norm(xt, None, None, torch.ones(3, dtype=torch.float64),
                                        torch.zeros(3, dtype=torch.float64), training=True, eps=2e-5)
    np.testing.assert_allclose(y.data, yt.numpy(), atol=1e-12)


def test_train_step_reports_loss_over_batch_Q5():
    cfg, P, X, y, V = _setup(dtype=np.float32)
    m = R.RefModel(cfg, {k: v.copy() for k, v in P.items()}, V)
    opt = R.RefOptimizer(m, {"type": 0, "lr": 1e-3, "l2": 1e-4, "grad_clip": 2, "grad_noise_eta": 0, "freeze": []})
    before = m.p["out/W"].data.copy()
    loss, rep = R.train_step(m, opt, X, y, 0.8, pyrandom=random.Random("seed-ast-20h"))
    assert rep == pytest.approx(loss / len(y))
    assert opt.last_grad_norm > 0 and not np.allclose(before, m.p["out/W"].data)
    assert m.p["out/W"].data.dtype == np.float32
