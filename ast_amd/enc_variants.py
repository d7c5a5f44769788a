"""The reference's OPTIONAL encoder variants, composed layer by layer from the C-ABI ops:

  rnn_config.ln           L.LayerNormalization behind every encoder LSTM: hs = LN(dropout(LSTM(hs)))      (seq2seq.py:81-87, 192-203)
  rnn_config.linear_proj  forward_rnn_encode_proj: layer-by-layer stacks with Linear + BatchNorm + ReLU between the layers,
                          as written -- see LinearProjEncoder                                              (seq2seq.py:89-100, 244-291)

The shipped configs use neither; the default encoder (one persistent wavefront launch over all layers, ast_amd/seq2seq.py) is untouched
by this module.  Here every LSTM layer and direction is one `astk_lstm_stack_*` call of a ONE-layer stack (which still takes the
persistent recurrence kernels when the shape allows), and the normalisation / projection between two layers is a kernel of
csrc/norm.hip or a GEMM.  torch only moves data (transposes, flips, broadcasts of one frame): no arithmetic outside libastk.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import LstmGrads, LstmParams, LstmStackDesc, check

LN_EPS = 1e-6            # L.LayerNormalization's default eps
BN_EPS, BN_DECAY = 2e-5, 0.9


def _vp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _Layerwise:
    """Shared plumbing: one-layer stack calls with their own workspaces (the forward call leaves its saved activations there for the
    backward call), per-cell parameter structs, per-cell views of the model's dropout masks and final-state buffers."""

    def __init__(self, model):
        self.m = model

    def _cell(self, d, l):
        m = self.m
        n = (m.rnn_enc, m.rnn_rev_enc)[d][l]
        a = m.arena
        p, g = (LstmParams * 1)(), (LstmGrads * 1)()
        p[0].Wu, p[0].b, p[0].Wl = a.p(n + "/upward/W"), a.p(n + "/upward/b"), a.p(n + "/lateral/W")
        g[0].dWu, g[0].db, g[0].dWl = a.g(n + "/upward/W"), a.g(n + "/upward/b"), a.g(n + "/lateral/W")
        return n, p, g

    def _desc(self, T, B, in_dim, h, nl, nd):
        """A stack descriptor that carries the model's per-call arithmetic (include/astk.h: precision / gemm_operands)."""
        d = LstmStackDesc(T, B, in_dim, h, nl, nd)
        d.precision, d.gemm_operands = _lib.PREC_BY_NAME[self.m.gemm_precision], _lib.OPERANDS_BY_NAME[self.m.gemm_operands]
        return d

    def _ws(self, key, desc):
        lib = _lib.load()
        nbytes = int(lib.astk_lstm_stack_workspace_bytes(C.byref(desc)))
        assert nbytes, lib.astk_last_error().decode()
        return self.m._workspace(("encv",) + key, nbytes)

    def _fwd(self, key, desc, params, x, masks, out, cT, hT):
        lib = _lib.load()
        ws = self._ws(key, desc)
        check(lib.astk_lstm_stack_fwd(C.byref(desc), params, _vp(x), _vp(masks), _vp(out), _vp(cT), _vp(hT), _vp(ws), ws.numel(), self.m._stream()))

    def _bwd(self, key, desc, params, grads, x, masks, d_out, d_cT, d_hT, dx):
        lib = _lib.load()
        ws = self._ws(key, desc)
        check(lib.astk_lstm_stack_bwd(C.byref(desc), params, grads, _vp(x), _vp(masks), _vp(d_out), _vp(d_cT), _vp(d_hT), _vp(dx), _vp(ws),
                                      ws.numel(), self.m._stream()))

    def _ln(self, name, x, ldx, rows, n, y, ldy):
        a = self.m.arena
        check(_lib.load().astk_layernorm_fwd(rows, n, _vp(x), ldx, a.p(name + "_ln/gamma"), a.p(name + "_ln/beta"), LN_EPS, _vp(y), ldy, self.m._stream()))

    def _ln_bwd(self, name, x, ldx, rows, n, dy, lddy, dx, lddx):
        a = self.m.arena
        check(_lib.load().astk_layernorm_bwd(rows, n, _vp(x), ldx, a.p(name + "_ln/gamma"), LN_EPS, _vp(dy), lddy, _vp(dx), lddx,
                                             a.g(name + "_ln/gamma"), a.g(name + "_ln/beta"), self.m._stream()))


class _Ptr:
    """A device address inside a tensor (column offset of a strided view) for _vp()."""

    def __init__(self, t, offset_floats):
        self.t, self.off = t, offset_floats

    def data_ptr(self):
        return self.t.data_ptr() + 4 * self.off


class LayerNormEncoder(_Layerwise):
    """rnn_config.ln: every layer's dropped output goes through its own LayerNormalization before the next layer / enc_states see it;
    the LSTM's recurrent state and the final (c, h) that seed the decoder stay raw.  The reference interleaves the layers per time step
    (feed_rnn); nothing crosses between time steps except inside a layer, so running layer after layer over the whole sequence is the same
    computation.  Layer 0 is the usual two-direction stack on the CNN output (direction 1 through the Q1 frame order 0, T-1, .., 1); from
    layer 1 on each direction is fed its own layer below IN LOOP-STEP ORDER, as one-direction stacks."""

    def forward(self, st):
        m = self.m
        T2, B, h, nd, nl = st["T2"], st["B"], m.h, m.n_dirs, len(m.rnn_enc)
        masks = st["enc_masks"]                                   # (nd, nl, T2, B, h) or None
        sv = st["encv"] = {}
        # ---- layer 0: both directions in one call, output (B, T2, nd*h) with direction 1 flipped to frame positions
        d0 = self._desc(T2, B, st["feat"], h, 1, nd)
        p0, g0 = (LstmParams * nd)(), (LstmGrads * nd)()
        for d in range(nd):
            _, p, g = self._cell(d, 0)
            p0[d], g0[d] = p[0], g[0]
        mk0 = masks[:, 0:1].contiguous() if masks is not None else None
        U0 = m._pool("encv_U0", (B, T2, nd * h))
        cT0, hT0 = m._pool("encv_cT0", (nd, 1, B, h)), m._pool("encv_hT0", (nd, 1, B, h))
        self._fwd((0,), d0, p0, st["xlstm"], mk0, U0, cT0, hT0)
        st["cT"][:, 0].copy_(cT0[:, 0])
        st["hT"][:, 0].copy_(hT0[:, 0])
        V0 = m._pool("encv_V0", (B, T2, nd * h))
        for d in range(nd):
            self._ln((m.rnn_enc, m.rnn_rev_enc)[d][0], _Ptr(U0, d * h), nd * h, B * T2, h, _Ptr(V0, d * h), nd * h)
        sv.update(d0=d0, p0=p0, g0=g0, mk0=mk0, U0=U0)
        # per direction, (B, T2, h) in LOOP-STEP order (direction 1: position p of the flipped half is loop step T2-1-p)
        seq = [V0[:, :, :h], V0[:, :, h:].flip(1)] if nd == 2 else [V0]
        d1 = self._desc(T2, B, h, h, 1, 1)
        for l in range(1, nl):
            for d in range(nd):
                name, p, g = self._cell(d, l)
                x = m._pool(f"encv_x{l}_{d}", (T2, B, h))
                x.copy_(seq[d].transpose(0, 1))
                mk = masks[d, l].contiguous() if masks is not None else None
                U = m._pool(f"encv_U{l}_{d}", (B, T2, h))
                cT, hT = m._pool(f"encv_cT{l}_{d}", (1, 1, B, h)), m._pool(f"encv_hT{l}_{d}", (1, 1, B, h))
                self._fwd((l, d), d1, p, x, mk, U, cT, hT)
                st["cT"][d, l].copy_(cT[0, 0])
                st["hT"][d, l].copy_(hT[0, 0])
                V = m._pool(f"encv_V{l}_{d}", (B, T2, h))
                self._ln(name, U, h, B * T2, h, V, h)
                sv[(l, d)] = dict(p=p, g=g, x=x, mk=mk, U=U, name=name)
                seq[d] = V
        enc = st["enc_states"]
        enc[:, :, :h].copy_(seq[0])
        if nd == 2:
            enc[:, :, h:].copy_(seq[1].flip(1))
        sv["d1"] = d1

    def backward(self, st, d_enc, d_cT, d_hT, d_xlstm):
        m = self.m
        T2, B, h, nd, nl = st["T2"], st["B"], m.h, m.n_dirs, len(m.rnn_enc)
        sv = st["encv"]
        dseq = [d_enc[:, :, :h].contiguous(), d_enc[:, :, h:].flip(1).contiguous()] if nd == 2 else [d_enc]
        for l in range(nl - 1, 0, -1):
            for d in range(nd):
                c = sv[(l, d)]
                dU = m._pool(f"encv_dU_{d}", (B, T2, h))
                self._ln_bwd(c["name"], c["U"], h, B * T2, h, dseq[d], h, dU, h)
                dx = m._pool(f"encv_dx_{d}", (T2, B, h))
                self._bwd((l, d), sv["d1"], c["p"], c["g"], c["x"], c["mk"], dU, d_cT[d, l].contiguous().view(1, 1, B, h),
                          d_hT[d, l].contiguous().view(1, 1, B, h), dx)
                nxt = m._pool(f"encv_dseq{l}_{d}", (B, T2, h))
                nxt.copy_(dx.transpose(0, 1))
                dseq[d] = nxt
        dV0 = m._pool("encv_dV0", (B, T2, nd * h))
        dV0[:, :, :h].copy_(dseq[0])
        if nd == 2:
            dV0[:, :, h:].copy_(dseq[1].flip(1))
        dU0 = m._pool("encv_dU0", (B, T2, nd * h))
        for d in range(nd):
            self._ln_bwd((m.rnn_enc, m.rnn_rev_enc)[d][0], _Ptr(sv["U0"], d * h), nd * h, B * T2, h, _Ptr(dV0, d * h), nd * h, _Ptr(dU0, d * h), nd * h)
        self._bwd((0,), sv["d0"], sv["p0"], sv["g0"], st["xlstm"], sv["mk0"], dU0, d_cT[:, 0:1].contiguous(), d_hT[:, 0:1].contiguous(), d_xlstm)


class LinearProjEncoder(_Layerwise):
    """rnn_config.linear_proj (seq2seq.py:244-291), as written:
      * a layer runs over the WHOLE sequence before the next one starts; no LayerNorm on this path (feed_rnn is not used);
      * the reverse stack of a layer is fed the LAST frame of that layer's input at every step (`enc_states[-1]`, :256 -- quirk Q8) and its
        outputs are flipped;
      * between layers: currH_t = relu(BN(Linear([fwd_t ; rev_t]))) with the BatchNormalization link called once per time step, i.e.
        statistics over the B rows of that step, running averages / N advanced T'' times;
      * `enc_states` is only reassigned inside the projection branch: the attention memory is the LAST PROJECTION's output (the CNN
        output itself for a one-layer encoder, which then needs C*F' == hidden_units); the top LSTM layer reaches the loss only through
        its final (c, h), which seed the decoder."""

    def forward(self, st, train):
        m = self.m
        lib = _lib.load()
        T2, B, h, nd, nl, H = st["T2"], st["B"], m.h, m.n_dirs, len(m.rnn_enc), m.H
        masks = st["enc_masks"]
        sv = st["encv"] = {}
        cur, width = st["xlstm"], st["feat"]
        s = m._stream()
        for l in range(nl):
            desc = self._desc(T2, B, width, h, 1, 1)
            U = []
            for d in range(nd):
                _, p, g = self._cell(d, l)
                if d == 0:
                    x = cur
                else:                                             # the last frame of the layer's input, at every step
                    x = m._pool(f"encv_xr{l}", (T2, B, width))
                    x.copy_(cur[T2 - 1:T2].expand(T2, B, width))
                mk = masks[d, l].contiguous() if masks is not None else None
                Ud = m._pool(f"encv_U{l}_{d}", (B, T2, h))
                cT, hT = m._pool(f"encv_cT{l}_{d}", (1, 1, B, h)), m._pool(f"encv_hT{l}_{d}", (1, 1, B, h))
                self._fwd((l, d), desc, p, x, mk, Ud, cT, hT)
                st["cT"][d, l].copy_(cT[0, 0])
                st["hT"][d, l].copy_(hT[0, 0])
                sv[(l, d)] = dict(p=p, g=g, x=x, mk=mk, desc=desc)
                U.append(Ud)
            if l < nl - 1:
                S = m._pool(f"encv_S{l}", (T2, B, H))              # rnn_states: [fwd_t ; flipud(rev)_t]
                S[:, :, :h].copy_(U[0].transpose(0, 1))
                if nd == 2:
                    S[:, :, h:].copy_(U[1].flip(1).transpose(0, 1))
                a = m.arena
                n = f"enc_proj{l}"
                Z = m._pool(f"encv_Z{l}", (T2, B, H))
                check(lib.astk_gemm_f32(0, T2 * B, H, H, _vp(S), H, a.p(n + "/W"), H, _vp(Z), H, a.p(n + "/b"), 0, 1, 1, 0, 0, 0, s))
                O = m._pool(f"encv_O{l}", (T2, B, H))
                stats = m._pool(f"encv_stats{l}", (T2, 2, H))
                check(lib.astk_step_bn_relu_fwd(T2, B, H, _vp(Z), a.p(n + "_bn/gamma"), a.p(n + "_bn/beta"), _vp(m.persist[n + "_bn/avg_mean"]),
                                                _vp(m.persist[n + "_bn/avg_var"]), BN_EPS, BN_DECAY, 1 if train else 0, _vp(O), _vp(stats), s))
                if train:
                    m.proj_bn_N[l] += T2                          # the link's persistent N: one call per time step
                sv[("proj", l)] = dict(S=S, Z=Z, O=O, stats=stats, name=n)
                cur, width = O, H
        if width != H:
            raise ValueError(f"linear_proj with a one-layer encoder leaves the CNN output ({width} features) as the attention memory, "
                             f"which must equal hidden_units ({H}) -- the reference's batch_matmul fails on this shape too")
        st["enc_states"].copy_(cur.transpose(0, 1))

    def backward(self, st, d_enc, d_cT, d_hT, d_xlstm):
        m = self.m
        lib = _lib.load()
        T2, B, h, nd, nl, H = st["T2"], st["B"], m.h, m.n_dirs, len(m.rnn_enc), m.H
        sv = st["encv"]
        s = m._stream()
        a = m.arena
        # gradient wrt what the attention read: the last projection's output (or the CNN output)
        dcur = m._pool("encv_dcur", (T2, B, H))
        dcur.copy_(d_enc.transpose(0, 1))
        for l in range(nl - 1, -1, -1):
            width = sv[(l, 0)]["desc"].in_dim
            dU = []
            if l == nl - 1:
                for d in range(nd):                               # the top layer's outputs are not read by anything
                    z = m._pool(f"encv_dU_{d}", (B, T2, h))
                    z.zero_()
                    dU.append(z)
            else:
                pr = sv[("proj", l)]
                n = pr["name"]
                dZ = m._pool("encv_dZ", (T2, B, H))
                check(lib.astk_step_bn_relu_bwd(T2, B, H, _vp(pr["Z"]), _vp(pr["stats"]), a.p(n + "_bn/gamma"), BN_EPS, _vp(pr["O"]), _vp(dcur), _vp(dZ),
                                                a.g(n + "_bn/gamma"), a.g(n + "_bn/beta"), s))
                # dW (H,H) += dZ^T S ; db += column sums ; dS = dZ W
                check(lib.astk_gemm_f32(2, H, H, T2 * B, _vp(dZ), H, _vp(pr["S"]), H, a.g(n + "/W"), H, None, 2, 1, 1, 0, 0, 0, s))
                check(lib.astk_colsum_add_f32(a.g(n + "/b"), _vp(dZ), H, T2 * B, H, s))
                dS = m._pool("encv_dS", (T2, B, H))
                check(lib.astk_gemm_f32(1, T2 * B, H, H, _vp(dZ), H, a.p(n + "/W"), H, _vp(dS), H, None, 0, 1, 1, 0, 0, 0, s))
                for d in range(nd):
                    z = m._pool(f"encv_dU_{d}", (B, T2, h))
                    z.copy_(dS[:, :, :h].transpose(0, 1) if d == 0 else dS[:, :, h:].transpose(0, 1).flip(1))
                    dU.append(z)
            dprev = d_xlstm if l == 0 else m._pool(f"encv_dprev{l % 2}", (T2, B, width))
            for d in range(nd):
                c = sv[(l, d)]
                dx = dprev if d == 0 else m._pool("encv_dxr", (T2, B, width))
                self._bwd((l, d), c["desc"], c["p"], c["g"], c["x"], c["mk"], dU[d], d_cT[d, l].contiguous().view(1, 1, B, h),
                          d_hT[d, l].contiguous().view(1, 1, B, h), dx)
                if d == 1:                                        # every step of the reverse stack read the same frame: sum over the steps
                    check(lib.astk_colsum_add_f32(_vp(dprev[T2 - 1]), _vp(dx), B * width, T2, B * width, s))
            if l == nl - 1 and l > 0:
                # the input of the top layer is also the attention memory: both gradients meet here
                check(lib.astk_add_f32(_vp(dprev), _vp(dcur), dprev.numel(), s))
            elif l == 0 and nl == 1:
                check(lib.astk_add_f32(_vp(d_xlstm), _vp(dcur), d_xlstm.numel(), s))
            dcur = dprev
