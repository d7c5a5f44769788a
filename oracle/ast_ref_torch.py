"""oracle/ast_ref_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Second, independent CPU restatement of the same hot path (seq2seq.py:158-473) written against
torch's CPU autograd and library ops (conv2d, batch_norm, cross_entropy) instead of the hand-derived
backward formulas of oracle/minichainer.py.  Its only job is to cross-check oracle/ast_ref.py
(loss, every gradient, grad norm) so that a mistake in either restatement shows up as a disagreement.
Structure differs on purpose: the input projection is batched over time, gates are de-interleaved
into torch's block order, the reverse direction is expressed through an index list.
"""
import numpy as np
import torch
import torch.nn.functional as TF


def _lstm_run(x_seq, Wu, b, Wl, h0=None, c0=None, masks=None):
    """x_seq (n,B,in) in consumption order.  Chainer interleaved gates (A1): column 4j+k, k=a,i,f,o.
    Returns raw h per step, final (c,h) and the dropped outputs."""
    n, B, _ = x_seq.shape
    hdim = Wl.shape[1]
    zx = x_seq @ Wu.t() + b                                   # batched upward
    h, c = h0, c0
    if c is None:
        c = torch.zeros(B, hdim, dtype=x_seq.dtype)
    outs = []
    for t in range(n):
        z = zx[t] if h is None else zx[t] + h @ Wl.t()
        z = z.view(B, hdim, 4)
        a, i, f, o = torch.tanh(z[..., 0]), torch.sigmoid(z[..., 1]), torch.sigmoid(z[..., 2]), torch.sigmoid(z[..., 3])
        c = a * i + f * c
        h = o * torch.tanh(c)
        outs.append(h if masks is None else h * masks[t])
    return torch.stack(outs, 0), c, h


def forward_loss_torch(cfg, P, X, y, use_truth, V, masks=None, noise=None, bn_eps=2e-5):
    """P: dict name -> torch tensor (requires_grad on trainables).  use_truth: the Q4 flag list.
    masks: dict tag -> numpy scaled mask as recorded by oracle.ast_ref.RecordingMasks (or None)."""
    rc = cfg["rnn_config"]
    dt = X.dtype
    B = X.shape[0]
    mk = (lambda tag: torch.from_numpy(masks[tag]).to(dt)) if masks else None
    if noise is not None:
        X = X * noise
    h = X.unsqueeze(1)                                        # (B,1,T,D)
    for i, l in enumerate(cfg["cnn_config"]["cnn_layers"]):
        h = TF.conv2d(h, P[f"CNN_{i}/W"], stride=tuple(l["stride"]), padding=tuple(l["pad"]))
        if cfg["cnn_config"]["bn"]:
            h = TF.batch_norm(h, None, None, P[f"CNN_{i}_bn/gamma"], P[f"CNN_{i}_bn/beta"], training=True, eps=bn_eps)
        h = torch.relu(h)
    Bc, C, T2, F2 = h.shape
    feats = h.permute(2, 0, 1, 3).reshape(T2, B, C * F2)      # (T'',B,C*F'), index c*F'+f
    nl = rc["enc_layers"]
    order_rev = [(-i) % T2 for i in range(T2)]                # Q1: 0, T''-1, ..., 1
    finals = {}
    xs_f, xs_r = feats, feats[order_rev]
    for k in range(nl):
        mf = [mk(("enc", k, t)) for t in range(T2)] if masks and cfg["dropout"]["rnn"] > 0 else None
        mr = [mk(("rev", k, t)) for t in range(T2)] if masks and cfg["dropout"]["rnn"] > 0 else None
        xs_f, cf, hf = _lstm_run(xs_f, P[f"L{k}_enc/upward/W"], P[f"L{k}_enc/upward/b"], P[f"L{k}_enc/lateral/W"], masks=mf)
        xs_r, cr, hr = _lstm_run(xs_r, P[f"L{k}_rev_enc/upward/W"], P[f"L{k}_rev_enc/upward/b"], P[f"L{k}_rev_enc/lateral/W"], masks=mr)
        finals[k] = (torch.cat([cf, cr], 1), torch.cat([hf, hr], 1))
    enc = torch.cat([xs_f, torch.flip(xs_r, [0])], 2).transpose(0, 1)   # (B,T'',H)
    nd = rc["dec_layers"]
    st = [finals.get(k, (None, None)) for k in range(nd)]
    cs, hs = [s[0] for s in st], [s[1] for s in st]
    A = rc["attn_units"]
    ht = torch.zeros(B, A, dtype=dt)
    yT = torch.as_tensor(np.asarray(y)).long().t()
    L = yT.shape[0]
    w = torch.ones(V, dtype=dt)
    w[0] = 0
    loss = torch.zeros((), dtype=dt)
    dec_in = None
    dr = cfg["dropout"]
    for s in range(L - 1):
        if use_truth[s]:
            dec_in = yT[s]
        e = P["embed_dec/W"][dec_in]
        if masks and dr["embed"] > 0:
            e = e * mk(("emb", 0, s))
        x = torch.cat([e, ht], 1)
        for k in range(nd):
            Wu, b, Wl = P[f"L{k}_dec/upward/W"], P[f"L{k}_dec/upward/b"], P[f"L{k}_dec/lateral/W"]
            z = x @ Wu.t() + b
            if hs[k] is not None:
                z = z + hs[k] @ Wl.t()
            c_prev = cs[k] if cs[k] is not None else torch.zeros(B, Wl.shape[1], dtype=dt)
            z = z.view(B, -1, 4)
            a, i, f, o = torch.tanh(z[..., 0]), torch.sigmoid(z[..., 1]), torch.sigmoid(z[..., 2]), torch.sigmoid(z[..., 3])
            cs[k] = a * i + f * c_prev
            hs[k] = o * torch.tanh(cs[k])
            x = hs[k]
            if masks and dr["rnn"] > 0:
                x = x * mk(("dec", k, s))
        q = x @ P["attn_Wa/W"].t() + P["attn_Wa/b"]
        alpha = torch.softmax(torch.einsum("bth,bh->bt", enc, q), dim=1)
        cv = torch.einsum("bth,bt->bh", enc, alpha)
        ht = torch.tanh(torch.cat([cv, x], 1) @ P["context/W"].t() + P["context/b"])
        logits = ht @ P["out/W"].t() + P["out/b"]
        dec_in = logits.argmax(1)
        loss = loss + TF.cross_entropy(logits, yT[s + 1], weight=w, reduction="sum") / B   # Q6: /B, PAD rows weigh 0
    return loss, enc
