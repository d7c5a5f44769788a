"""oracle/ast_ref_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Second, independent CPU restatement of the same hot path (seq2seq.py:158-473) written against
torch's CPU autograd and library ops (conv2d, batch_norm, cross_entropy) instead of the hand-derived
backward formulas of oracle/minichainer.py.  Its jobs: (1) cross-check oracle/ast_ref.py (loss, every
gradient, grad norm) so that a mistake in either restatement shows up as a disagreement; (2) give the GPU
tests per-operator references (cnn / encoder / decoder) with gradients w.r.t. the operator inputs.
Structure differs on purpose: the input projection is batched over time, the reverse direction is an index
list, the blocks are plain functions.
"""
import numpy as np
import torch
import torch.nn.functional as TF


def cnn_torch(cfg, P, X, noise=None, bn_eps=2e-5):
    """seq2seq.py:158-180 -> (T'',B,C*F') with feature index c*F'+f."""
    if noise is not None:
        X = X * noise
    h = X.unsqueeze(1)                                        # (B,1,T,D)
    for i, l in enumerate(cfg["cnn_config"]["cnn_layers"]):
        h = TF.conv2d(h, P[f"CNN_{i}/W"], stride=tuple(l["stride"]), padding=tuple(l["pad"]))
        if "cnn_pool" in cfg["cnn_config"]:                     # old path, enc_dec.py:444-456: max_pooling_nd, stride = window, cover_all
            kt, kf = cfg["cnn_config"]["cnn_pool"][i]
            k = (h.shape[2] if kt == -1 else max(kt, 1), h.shape[3] if kf == -1 else max(kf, 1))
            h = TF.max_pool2d(h, k, stride=k, ceil_mode=True)
        if cfg["cnn_config"]["bn"]:
            h = TF.batch_norm(h, None, None, P[f"CNN_{i}_bn/gamma"], P[f"CNN_{i}_bn/beta"], training=True, eps=bn_eps)
        h = torch.relu(h)
    B, C, T2, F2 = h.shape
    return h.permute(2, 0, 1, 3).reshape(T2, B, C * F2)


def _lstm_run(x_seq, Wu, b, Wl, masks=None):
    """x_seq (n,B,in) in consumption order.  Chainer interleaved gates (A1): column 4j+k, k=a,i,f,o.
    Returns dropped outputs per step and the final un-dropped (c,h)."""
    n, B, _ = x_seq.shape
    hdim = Wl.shape[1]
    zx = x_seq @ Wu.t() + b                                   # batched upward
    h = None
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


def encoder_torch(cfg, P, feats, masks=None):
    """seq2seq.py:205-242.  feats (T'',B,in).  masks: None or array (2, n_layers, T'', B, h) indexed by loop step.
    Returns enc_states (B,T'',H) and the per-layer finals cT,hT as (2,n_layers,B,h)."""
    rc = cfg["rnn_config"]
    T2 = feats.shape[0]
    nl = rc["enc_layers"]
    order_rev = [(-i) % T2 for i in range(T2)]                # Q1: 0, T''-1, ..., 1
    xs_f, xs_r = feats, feats[order_rev]
    cT, hT = [[], []], [[], []]
    for k in range(nl):
        mf = masks[0][k] if masks is not None else None
        mr = masks[1][k] if masks is not None else None
        xs_f, cf, hf = _lstm_run(xs_f, P[f"L{k}_enc/upward/W"], P[f"L{k}_enc/upward/b"], P[f"L{k}_enc/lateral/W"], mf)
        xs_r, cr, hr = _lstm_run(xs_r, P[f"L{k}_rev_enc/upward/W"], P[f"L{k}_rev_enc/upward/b"], P[f"L{k}_rev_enc/lateral/W"], mr)
        cT[0].append(cf); cT[1].append(cr); hT[0].append(hf); hT[1].append(hr)
    enc = torch.cat([xs_f, torch.flip(xs_r, [0])], 2).transpose(0, 1)   # (B,T'',H)
    cT = torch.stack([torch.stack(cT[0]), torch.stack(cT[1])])
    hT = torch.stack([torch.stack(hT[0]), torch.stack(hT[1])])
    return enc, cT, hT


def decoder_torch(cfg, P, enc, c0, h0, y, use_truth, V, emb_mask=None, rnn_masks=None):
    """seq2seq.py:361-473.  c0,h0 (n_layers,B,H); emb_mask (S,B,E); rnn_masks (n_layers,S,B,H).
    Returns loss, preds (S,B)."""
    rc = cfg["rnn_config"]
    dt = enc.dtype
    B = enc.shape[0]
    nd = rc["dec_layers"]
    cs, hs = [c0[k] for k in range(nd)], [h0[k] for k in range(nd)]
    ht = torch.zeros(B, rc["attn_units"], dtype=dt)
    yT = torch.as_tensor(np.asarray(y)).long().t()
    L = yT.shape[0]
    w = torch.ones(V, dtype=dt)
    w[0] = 0
    loss = torch.zeros((), dtype=dt)
    dec_in = None
    preds = []
    for s in range(L - 1):
        if use_truth[s]:
            dec_in = yT[s]
        e = P["embed_dec/W"][dec_in]
        if emb_mask is not None:
            e = e * emb_mask[s]
        x = torch.cat([e, ht], 1)
        for k in range(nd):
            Wu, b, Wl = P[f"L{k}_dec/upward/W"], P[f"L{k}_dec/upward/b"], P[f"L{k}_dec/lateral/W"]
            z = (x @ Wu.t() + b + hs[k] @ Wl.t()).view(B, -1, 4)
            a, i, f, o = torch.tanh(z[..., 0]), torch.sigmoid(z[..., 1]), torch.sigmoid(z[..., 2]), torch.sigmoid(z[..., 3])
            cs[k] = a * i + f * cs[k]
            hs[k] = o * torch.tanh(cs[k])
            x = hs[k] if rnn_masks is None else hs[k] * rnn_masks[k][s]
        q = x @ P["attn_Wa/W"].t() + P["attn_Wa/b"]
        alpha = torch.softmax(torch.einsum("bth,bh->bt", enc, q), dim=1)
        cv = torch.einsum("bth,bt->bh", enc, alpha)
        ht = torch.tanh(torch.cat([cv, x], 1) @ P["context/W"].t() + P["context/b"])
        logits = ht @ P["out/W"].t() + P["out/b"]
        dec_in = logits.argmax(1)
        preds.append(dec_in)
        loss = loss + TF.cross_entropy(logits, yT[s + 1], weight=w, reduction="sum") / B   # Q6: /B, PAD rows weigh 0
    return loss, torch.stack(preds)


def masks_from_recording(cfg, rec, T2, S, B):
    """Packs oracle.ast_ref.RecordingMasks' tag->mask dict into the dense layouts the C ABI takes."""
    rc = cfg["rnn_config"]
    h = rc["hidden_units"] // 2
    H, E = rc["hidden_units"], rc["embedding_units"]
    nl, nd = rc["enc_layers"], rc["dec_layers"]
    out = {}
    if cfg["dropout"]["rnn"] > 0:
        enc = np.ones((2, nl, T2, B, h), np.float32)
        for d_, name in enumerate(("enc", "rev")):
            for k in range(nl):
                for t in range(T2):
                    enc[d_, k, t] = rec[(name, k, t)]
        out["enc_masks"] = enc
        dec = np.ones((nd, S, B, H), np.float32)
        for k in range(nd):
            for s in range(S):
                dec[k, s] = rec[("dec", k, s)]
        out["rnn_masks"] = dec
    if cfg["dropout"]["embed"] > 0:
        emb = np.ones((S, B, E), np.float32)
        for s in range(S):
            emb[s] = rec[("emb", 0, s)]
        out["emb_mask"] = emb
    if cfg["dropout"].get("out", 0) > 0:                      # dropout on the logits (seq2seq.py:394)
        V = rec[("out", 0, 0)].shape[1]
        om = np.ones((S, B, V), np.float32)
        for s in range(S):
            om[s] = rec[("out", 0, s)]
        out["out_mask"] = om
    return out


def forward_loss_torch(cfg, P, X, y, use_truth, V, masks=None, noise=None):
    """Whole forward (seq2seq.py:399-473).  masks: dict tag -> numpy scaled mask as recorded by
    oracle.ast_ref.RecordingMasks (or None)."""
    rc = cfg["rnn_config"]
    dt = X.dtype
    B = X.shape[0]
    feats = cnn_torch(cfg, P, X, noise)
    T2 = feats.shape[0]
    S = np.asarray(y).shape[1] - 1
    packed = masks_from_recording(cfg, masks, T2, S, B) if masks else {}
    tt = lambda a: None if a is None else torch.from_numpy(a).to(dt)
    enc, cT, hT = encoder_torch(cfg, P, feats, tt(packed.get("enc_masks")))
    nd, H = rc["dec_layers"], rc["hidden_units"]
    c0 = [torch.zeros(B, H, dtype=dt) for _ in range(nd)]
    h0 = [torch.zeros(B, H, dtype=dt) for _ in range(nd)]
    for k in range(min(nd, rc["enc_layers"])):                # seq2seq.py:318-333
        c0[k] = torch.cat([cT[0, k], cT[1, k]], 1)
        h0[k] = torch.cat([hT[0, k], hT[1, k]], 1)
    loss, _ = decoder_torch(cfg, P, enc, c0, h0, y, use_truth, V, tt(packed.get("emb_mask")), tt(packed.get("rnn_masks")))
    return loss, enc
