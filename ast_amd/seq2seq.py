"""Drop-in for the reference's model class (seq2seq.py:22-568): SpeechEncoderDecoder(gpuid, cfg) with the same
method names, argument meaning and state protocol, running on libastk.so (hand-written HIP for gfx950).

PyTorch is used only as the owner of device memory and streams; every arithmetic step of the hot path is a
C-ABI call (include/astk.h).  There is no CPU / eager-PyTorch fallback: without the HIP library and a GPU
the compute methods raise.
"""
import ctypes as C
import os
import random

import numpy as np
import torch

from . import _lib
from ._lib import (CnnDesc, CnnLayerGrads, CnnLayerParams, DecoderDesc, DecoderGrads, DecoderParams, LstmGrads,
                   LstmParams, LstmStackDesc, check)
from .params import ParamArena, init_values, param_shapes


class _Config:
    """Stand-in for chainer.config / chainer.using_config('train', flag) (nn.py:174, 216, 248)."""
    train = True


config = _Config()


class using_config:
    def __init__(self, name, value):
        assert name == "train"
        self.value = value

    def __enter__(self):
        self.old = config.train
        config.train = self.value

    def __exit__(self, *a):
        config.train = self.old


class Link:
    """What `model[name]` returns (nn.py:113-116 calls .disable_update(); copy_params.py reads child params)."""

    def __init__(self, model, name):
        self._model, self.name = model, name

    def disable_update(self):
        self._model._frozen.add(self.name)

    def enable_update(self):
        self._model._frozen.discard(self.name)

    def namedparams(self):
        a = self._model.arena
        return [(k, v) for k, v in a.views.items() if k.split("/")[0] == self.name]


def raise_if_aborted(status, where="train step"):
    """`status` = the persistent kernels' sticky status word as read back next to the loss (include/astk.h
    astk_persist_status_snapshot).  Non-zero: a bounded spin timed out (the grid was not fully resident: another tenant on the GPU,
    RCCL kernels holding CUs, a partitioned device), the kernels drained, and the step's loss and gradients are garbage."""
    mask = int(status)
    if mask:
        names = [n for b, n in ((1, "encoder forward"), (2, "encoder backward"), (4, "decoder forward"), (8, "decoder backward"),
                                (16, "reported by a peer rank of the data-parallel job")) if mask & b]
        _lib.load().astk_persist_status(None, 1)          # clear the sticky word: the caller may retry (e.g. with the *.persist knobs at 0)
        raise _lib.AstkError(f"{where}: persistent kernel(s) timed out waiting for a peer workgroup ({', '.join(names)}); the results "
                             "of this step are invalid.  The whole grid must be resident (one workgroup per CU); set "
                             "the tuning knobs lstm.persist / dec.persist to 0 (ast_amd._lib.set_tuning) to use the per-launch kernels on a shared device")


def draw_flags_and_targets(yh, teach_ratio, random_out, V, randint, rank=None, world=None, gather=None):
    """forward_loss's draws from the seeded Python `random` stream in the reference's order (seq2seq.py:431-436, 456-465): per decoder
    step one teacher-forcing coin (0 < i < L-2), then one draw per target >= 4 of THE BATCH -- above `random_out` the scored target is
    replaced by randint(4, V + 1), clamped to V - 1 (quirk Q8).  Returns (flags, scored targets of the local rows).
    Data parallelism: the stream also feeds next epoch's shuffles and must stay IDENTICAL on every rank, but the number of draws depends
    on the targets -- and the ranks hold different rows.  So every rank makes the draws of the WHOLE global batch, in the row order of
    the unsharded batch (the loader gives rank r rows r::world: global row b * world + r), and keeps the replacements of its own rows:
    the `y` of all ranks is gathered first (`gather(yh) -> [yh of rank 0, ..]`; default torch.distributed.all_gather_object)."""
    from . import dist as adist
    world = adist.world_size() if world is None else world
    rank = adist.rank() if rank is None else rank
    B, L = yh.shape
    S = L - 1
    tg = yh.copy()
    if world > 1:
        if gather is None:
            def gather(a):
                out = [None] * world
                td_ = __import__("torch.distributed", fromlist=["x"])
                td_.all_gather_object(out, np.ascontiguousarray(a))
                return out
        parts = gather(yh)
        assert len(parts) == world and all(p.shape == yh.shape for p in parts), "shards of one bucketed batch share (B, L) (dataloader.get_batch)"
        rows = [(parts[g % world], g // world, g % world == rank) for g in range(B * world)]
    else:
        rows = [(yh, b, True) for b in range(B)]
    flags = []
    for i in range(S):
        flags.append(int(random.random() < teach_ratio) if 0 < i < L - 2 else 1)
        for src, b, mine in rows:
            if int(src[b, i + 1]) >= 4 and random.random() > random_out:
                if mine:
                    tg[b, i + 1] = min(int(randint(4, V + 1)), V - 1)
    return flags, tg


class LossData:
    """`loss.data` (nn.py:189: `float(loss.data)`): the 0-d device scalar, read through the persistent kernels' status word.  float(),
    .item(), .tolist() and NumPy conversion all read the [loss, status] pair in ONE copy and raise AstkError when a kernel of the step
    timed out -- a drop-in caller that only ever touches `.data` cannot train on with garbage.  `.tensor` is the raw 0-d tensor (no
    check, no synchronisation) for callers that keep the value on the device."""

    def __init__(self, pair):
        self._pair = pair

    @property
    def tensor(self):
        return self._pair[0]

    def _checked(self):
        v = self._pair.tolist()
        raise_if_aborted(v[1])
        return v[0]

    def __float__(self):
        return self._checked()

    def item(self):
        return self._checked()

    def tolist(self):
        return self._checked()

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._checked(), dtype=dtype or np.float32)

    def __getattr__(self, name):              # shape / dtype / device / clone() ... of the 0-d tensor
        if name.startswith("_"):              # (copy / pickle probe dunder names before _pair exists: no recursion)
            raise AttributeError(name)
        return getattr(self._pair[0], name)


class Loss:
    """What forward_loss returns: `.data` (the 0-d loss, status-checked when it is read: LossData), float(), `.backward()`
    (nn.py:175-189).  `.pair` = [loss, status] on the device: the status word of the persistent kernels rides next to the scalar, so one
    read-back serves both."""

    def __init__(self, model, pair):
        self._model = model
        self.pair = pair
        self.data = LossData(pair)

    def backward(self):
        self._model._backward()

    def __float__(self):
        return float(self.data)


def _vp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_LAZY_CLEARGRADS = os.environ.get("ASTK_LAZY_CLEARGRADS", "1") != "0"      # (0: cleargrads() always fills at once -- A/B runs)


class SpeechEncoderDecoder:
    def __init__(self, gpuid, cfg):
        self.gpuid = gpuid
        self.cfg = cfg
        rc, cc = cfg["rnn_config"], cfg["cnn_config"]
        # Optional features of the reference's model (seq2seq.py:43-57, 81-121, 244-291, 369-394; none is set by a shipped config).
        # They run on per-launch kernels / layer-by-layer stacks instead of the persistent kernels of the default model: `paths()` says so.
        self.rnn_ln = bool(rc.get("ln", False))
        self.rnn_linear_proj = bool(rc.get("linear_proj", False))
        self.feed_attn = bool(rc.get("feed_attn", True))
        n_attn = int(rc.get("n_attn", 1))
        if not 1 <= n_attn <= _lib.MAX_ATTN:
            raise ValueError(f"rnn_config.n_attn = {n_attn}: 1..{_lib.MAX_ATTN} attention heads are supported")
        if not rc["bi_rnn"]:
            self.n_dirs = 1
        else:
            self.n_dirs = 2
        self.cnns = [f"CNN_{i}" for i in range(len(cc["cnn_layers"]))]
        self.cnn_bn = cc["bn"]
        self.rnn_enc = [f"L{i}_enc" for i in range(rc["enc_layers"])]
        self.rnn_rev_enc = [f"L{i}_rev_enc" for i in range(rc["enc_layers"])] if rc["bi_rnn"] else []
        self.rnn_dec = [f"L{i}_dec" for i in range(rc["dec_layers"])]
        self.bi_rnn = rc["bi_rnn"]
        self.n_attn = n_attn
        self.enc_variant = None         # ast_amd.enc_variants.LayerNormEncoder / LinearProjEncoder (set in materialize)
        self.proj_bn_N = [0] * max(0, rc["enc_layers"] - 1)     # BatchNormalization N of the enc_proj{i}_bn links (one call per time step)
        self.h = rc["hidden_units"] // 2 if rc["bi_rnn"] else rc["hidden_units"]
        self.H, self.E, self.A = rc["hidden_units"], rc["embedding_units"], rc["attn_units"]
        self.V = rc.get("dec_vocab_size")
        self.device = torch.device(f"cuda:{gpuid}") if gpuid is not None and gpuid >= 0 else torch.device("cpu")
        self.arena = None
        self.persist = {}
        self.in_dim = None
        self._frozen = set()
        self._ws = {}
        self._shape_cache = {}
        self._pools = {}                # grow-only device buffers shared by all batch shapes (_pool)
        self._bstate = {}               # small zero-initialised state tensors, per batch size
        self.inject = {}          # test hooks: enc_masks / emb_mask / rnn_masks / noise / use_truth
        self.rng_seed = 0x5EED
        self._rng_offset = 0
        self._predrawn = {}             # random tensors of the current train step drawn ahead in one launch (_predraw)
        self._pending_flags = None      # the step's teacher-forcing flags on their way to the device with that launch
        self.bn_N = 0                   # training-mode forward passes so far = Chainer's BatchNormalization persistent N (A10)
        self.enc_states = None
        self.loss = 0
        self._cur = None
        self._dec_c = self._dec_h = None
        self.grad_buckets = None        # ast_amd.dist.GradBuckets under data parallelism: ranges are all-reduced as they become final
        self.stat_exchange = None       # ast_amd.dist.StatExchange: BatchNorm statistics over the global batch (train mode only)
        # Work BESIDE the latency-bound encoder recurrences (round 6): an ORDINARY second stream carries (i) the decoder's parameter
        # gradients beside the encoder's backward recurrence and (ii) -- inside the library, astk_lstm_stack_desc.side_stream -- the
        # time-chunked layer-0 products beside both recurrences; every launch on it is capped at the CUs the recurrence grid leaves
        # free, so neither order of dispatch can keep that grid from becoming resident.  (Rounds 1-5 needed CU-masked streams for
        # this and kept it opt-in: uncapped stream-K grids that got the CUs first blocked the recurrence.)  ASTK_SIDE_STREAM=0: in line.
        self.side_stream_on = os.environ.get("ASTK_SIDE_STREAM", "1") != "0"
        self._side = None
        self._side_by_main = {}
        self.mask_pad_id = None
        # Arithmetic of the batched products, per model (-> the descriptors' `precision` / `gemm_operands` fields): None = the library's
        # process-wide default (bf16x3: exact f32 operands on the 16-bit matrix pipe); "bf16x3" | "f32" | "fp16x2" (narrower, opt-in);
        # gemm_operands "fp16" = single-term fp16 operands for the eligible products (BASELINE configs[4]; reduced precision).
        self.gemm_precision = None
        self.gemm_operands = None
        # Deterministic backward (-> the descriptors' `deterministic` field, astk.h): every accumulated sum of the backward pass in a fixed
        # order -- split tiles of the batched products through the fix-up workspace instead of float atomics, ordered column / embedding /
        # bias sums, no side-stream work.  A few per cent slower; two runs over the same batches are then bit-identical in EVERY gradient,
        # which is what lets a soak see a hand-off race in the backward half of the step (train_cfg.json: extras.deterministic).
        self.deterministic = False

    # ------------------------------------------------------------------ parameters
    def materialize(self, in_dim, values=None, seed=0):
        """Allocates the (lazily shaped, seq2seq.py:52,83) parameters once the feature dim is known."""
        if self.V is None:
            raise ValueError("cfg['rnn_config']['dec_vocab_size'] must be set (config.py:24 injects it)")
        train, persist = param_shapes(self.cfg, in_dim, self.V)
        self.in_dim = in_dim
        self.arena = ParamArena(train, self.device)
        vals = values if values is not None else init_values(self.cfg, in_dim, self.V, seed)
        self.arena.load(vals)
        self.persist = {k: torch.as_tensor(np.asarray(vals[k], dtype=np.float32)).to(self.device).contiguous() for k in persist}
        w = torch.ones(self.V, dtype=torch.float32)
        w[0] = 0                                             # seq2seq.py:152-156
        self.mask_pad_id = w.to(self.device)
        self._shape_cache.clear()
        if self.rnn_linear_proj:                             # (seq2seq.py:311-314: linear_proj takes precedence; that path has no LayerNorm)
            from .enc_variants import LinearProjEncoder
            self.enc_variant = LinearProjEncoder(self)
        elif self.rnn_ln:
            from .enc_variants import LayerNormEncoder
            self.enc_variant = LayerNormEncoder(self)
        return self

    def add_weight_noise(self, mu, sigma):
        """The OLD path's Gaussian weight noise (enc_dec.py:587-624, driven per epoch by nmt_run.py:850-853): N(mu, sigma) added, in place, to
        every LSTM's upward W and b and lateral W (encoder, reverse encoder, decoder) and to the decoder embedding.  Drawn on the
        device (the reference's draw is the unseeded global RNG, quirk Q7); inject['weight_noise'] = {name: tensor} replays given draws."""
        lib = self._require_gpu()
        names = []
        for n in self.rnn_enc + self.rnn_rev_enc + self.rnn_dec:
            names += [n + "/upward/W", n + "/upward/b", n + "/lateral/W"]
        names.append("embed_dec/W")
        for name in names:
            w = self.arena.views[name]
            if "weight_noise" in self.inject:
                z = self.inject["weight_noise"][name].to(self.device, torch.float32).contiguous()
            else:
                z = self._pool("weight_noise", (w.numel() + (w.numel() & 1),))[:w.numel()]
                check(lib.astk_fill_normal(_vp(z), w.numel(), float(mu), float(sigma), self.rng_seed ^ 0x5EED0015E, self._rng(w.numel() + 1), self._stream()))
            check(lib.astk_add_f32(_vp(w), _vp(z), w.numel(), self._stream()))

    def paths(self):
        """Which kernels this model's train step takes (bench.py / logs): the optional features leave the persistent kernels."""
        opts = [n for n, on in (("ln", self.rnn_ln), ("linear_proj", self.rnn_linear_proj), (f"n_attn={self.n_attn}", self.n_attn > 1),
                                ("feed_attn=false", not self.feed_attn), ("cnn bn=false", not self.cnn_bn),
                                ("dropout.out", bool(self.cfg["dropout"].get("out", 0))),
                                ("cnn_pool (max-pool, old path)", "cnn_pool" in self.cfg["cnn_config"])) if on]
        return {"options": opts,
                "encoder": "layer-by-layer one-layer stacks + " + ("Linear/BatchNorm/ReLU projections" if self.rnn_linear_proj else "LayerNorm kernels")
                if self.enc_variant is not None else "one stack call (persistent wavefront kernels when the shape fits)",
                "decoder": "per-launch loop (optional features)" if (self.rnn_ln or self.n_attn > 1 or not self.feed_attn or self.cfg["dropout"].get("out", 0))
                else "persistent loop when the shape fits"}

    def to_gpu(self, gpuid=None):
        if gpuid is not None and gpuid != self.gpuid:
            assert self.arena is None, "move before materialize()"
            self.gpuid = gpuid
            self.device = torch.device(f"cuda:{gpuid}")
        return self

    def __getitem__(self, name):
        return Link(self, name)

    def namedparams(self):
        return list(self.arena.views.items())

    def params(self):
        return list(self.arena.views.values())

    def cleargrads(self):
        """Zero gradients (nn.py:180 calls it between forward_loss and backward).  Behind a forward_loss on the device the fill is DEFERRED to
        the backward pass, whose first fill launch takes the arena along (a launch less per step); anything that reads the gradients before
        that performs it first (ParamArena.flush_zero), so the deferral cannot be observed."""
        if self.arena is None:
            return
        st = self._cur
        if st and st.get("train_mode") and "dd" in st and st.get("pending_backward") and _LAZY_CLEARGRADS:
            self.arena.defer_zero()
        else:
            self.arena.grad.zero_()

    def enabled_ranges(self):
        """Contiguous [offset, n) slices of the arena whose links have updates enabled (nn.py:113-116)."""
        out = []
        for name in self.arena.shapes:
            if name.split("/")[0] in self._frozen:
                continue
            o, n = self.arena.range_of(name)
            if out and out[-1][0] + out[-1][1] == o:
                out[-1] = (out[-1][0], out[-1][1] + n)
            else:
                out.append((o, n))
        return out

    # ------------------------------------------------------------------ plumbing
    def _require_gpu(self):
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("ast_amd compute path needs an MI355X GPU and libastk.so (no CPU fallback)")
        return _lib.load()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, key, nbytes):
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes * 1.0) + 256, dtype=torch.uint8, device=self.device)
            self._ws[key] = t
        return t

    def _pool(self, name, shape, dtype=torch.float32):
        """A view of a persistent, grow-only device buffer: real batches come in hundreds of (T, L) combinations (bucket width 80
        frames, 2..max_pred targets), and a fresh set of 80 MB activation tensors per combination costs 40 ms per step in
        allocations and never gives the memory back.  Only data that is fully rewritten by every step lives here."""
        n = 1
        for d_ in shape:
            n *= int(d_)
        buf = self._pools.get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty(int(n * 1.25) + 64, dtype=dtype, device=self.device)
            self._pools[name] = buf
            self._shape_cache.clear()          # cached views would keep the outgrown storage alive
        return buf[:n].view(shape)

    def _shape_state(self, B, T, D, L):
        """The descriptors, parameter tables and buffers of one batch shape (cached); the arithmetic the model currently asks for
        (`gemm_precision`, `gemm_operands`) goes into the descriptors' per-call fields (include/astk.h: ASTK_PREC_* / ASTK_OPERANDS_*)."""
        st = self._shape_state_cached(B, T, D, L)
        prec, ops = _lib.PREC_BY_NAME[self.gemm_precision], _lib.OPERANDS_BY_NAME[self.gemm_operands]
        for k in ("cd", "ld", "dd"):
            st[k].precision, st[k].gemm_operands = prec, ops
            st[k].deterministic = 1 if self.deterministic else 0
        return st

    def _shape_state_cached(self, B, T, D, L):
        key = (B, T, D, L)
        st = self._shape_cache.get(key)
        if st is not None:
            return st
        lib = self._require_gpu()
        if self.arena is None:
            self.materialize(D)
        dev = self.device
        cc = self.cfg["cnn_config"]["cnn_layers"]
        cd = CnnDesc()
        cd.B, cd.T, cd.D, cd.n_layers = B, T, D, len(cc)
        for i, l in enumerate(cc):
            cd.C[i] = l["out_channels"]
            cd.kt[i], cd.kf[i] = l["ksize"]
            cd.st[i], cd.sf[i] = l["stride"]
            cd.pt[i] = l["pad"][0]
            if l["pad"][1] != 0 or l.get("dilate", 1) != 1:
                raise NotImplementedError("frequency padding / dilation are not used by the shipped configs")
        cd.bn_eps, cd.bn_decay = 2e-5, 0.9
        cd.no_bn = 0 if self.cnn_bn else 1
        for i, (kt, kf) in enumerate(self.cfg["cnn_config"].get("cnn_pool", [])):     # OLD-path extra (enc_dec.py:444-456), -1 = whole extent
            cd.pool_t[i], cd.pool_f[i] = int(kt), int(kf)
        t2, f2, feat = C.c_int(), C.c_int(), C.c_int()
        check(lib.astk_conv_bn_relu_out_dims(C.byref(cd), C.byref(t2), C.byref(f2), C.byref(feat)))
        T2, feat = t2.value, feat.value
        rnn_in = self.arena.shapes["L0_enc/upward/W"][1]
        if feat != rnn_in:
            raise ValueError(f"feature dim {D} gives {feat} LSTM inputs but the model was built for {rnn_in} "
                             f"(in_dim {self.in_dim}); the reference's lazily-shaped links fix this at the first batch")
        a = self.arena
        cp = (CnnLayerParams * len(cc))()
        cg = (CnnLayerGrads * len(cc))()
        for i in range(len(cc)):
            n = f"CNN_{i}"
            cp[i].W, cg[i].dW = a.p(n + "/W"), a.g(n + "/W")
            if self.cnn_bn:
                cp[i].gamma, cp[i].beta = a.p(n + "_bn/gamma"), a.p(n + "_bn/beta")
                cp[i].avg_mean = self.persist[n + "_bn/avg_mean"].data_ptr()
                cp[i].avg_var = self.persist[n + "_bn/avg_var"].data_ptr()
                cg[i].dgamma, cg[i].dbeta = a.g(n + "_bn/gamma"), a.g(n + "_bn/beta")
            else:                                            # Convolution2D with bias, no BatchNormalization (seq2seq.py:52-57)
                cp[i].bias, cg[i].dbias = a.p(n + "/b"), a.g(n + "/b")
        nl, nd, h, H = len(self.rnn_enc), self.n_dirs, self.h, self.H
        ld = LstmStackDesc(T2, B, feat, h, nl, nd)
        # hints that spare the GEMMs absolute-maximum passes (include/astk.h): the layer outputs are bounded by the dropout scale; the
        # frames' maximum is taken by the CNN kernel that writes them (set per call in encode(): it lives in the CNN workspace)
        ratio = float(self.cfg["dropout"]["rnn"])
        ld.out_bound = 1.0 / (1.0 - ratio) if 0 <= ratio < 1 else 0.0
        lp = (LstmParams * (nl * nd))()
        lg = (LstmGrads * (nl * nd))()
        for d_, names in enumerate([self.rnn_enc, self.rnn_rev_enc][:nd]):
            for l, n in enumerate(names):
                k = d_ * nl + l
                lp[k].Wu, lp[k].b, lp[k].Wl = a.p(n + "/upward/W"), a.p(n + "/upward/b"), a.p(n + "/lateral/W")
                lg[k].dWu, lg[k].db, lg[k].dWl = a.g(n + "/upward/W"), a.g(n + "/upward/b"), a.g(n + "/lateral/W")
        nld = len(self.rnn_dec)
        dd = DecoderDesc(B, max(L, 2), T2, H, self.E, self.A, self.V, nld, self.n_attn, 0 if self.feed_attn else 1, 1 if self.rnn_ln else 0)
        dp, dg = DecoderParams(), DecoderGrads()
        dp.embed, dg.d_embed = a.p("embed_dec/W"), a.g("embed_dec/W")
        for l, n in enumerate(self.rnn_dec):
            dp.lstm[l].Wu, dp.lstm[l].b, dp.lstm[l].Wl = a.p(n + "/upward/W"), a.p(n + "/upward/b"), a.p(n + "/lateral/W")
            dg.lstm[l].dWu, dg.lstm[l].db, dg.lstm[l].dWl = a.g(n + "/upward/W"), a.g(n + "/upward/b"), a.g(n + "/lateral/W")
            if self.rnn_ln:
                dp.ln_gamma[l], dp.ln_beta[l] = a.p(n + "_ln/gamma"), a.p(n + "_ln/beta")
                dg.d_ln_gamma[l], dg.d_ln_beta[l] = a.g(n + "_ln/gamma"), a.g(n + "_ln/beta")
        for k in range(1, self.n_attn):
            dp.Wa_x[k - 1], dp.ba_x[k - 1] = a.p(f"attn_Wa{k}/W"), a.p(f"attn_Wa{k}/b")
            dg.dWa_x[k - 1], dg.dba_x[k - 1] = a.g(f"attn_Wa{k}/W"), a.g(f"attn_Wa{k}/b")
        dp.Wa, dp.ba, dp.Wc, dp.bc = a.p("attn_Wa/W"), a.p("attn_Wa/b"), a.p("context/W"), a.p("context/b")
        dp.Wo, dp.bo, dp.class_weight = a.p("out/W"), a.p("out/b"), self.mask_pad_id.data_ptr()
        dg.dWa, dg.dba, dg.dWc, dg.dbc = a.g("attn_Wa/W"), a.g("attn_Wa/b"), a.g("context/W"), a.g("context/b")
        dg.dWo, dg.dbo = a.g("out/W"), a.g("out/b")
        S = max(L, 2) - 1
        f32 = dict(dtype=torch.float32, device=dev)
        # pooled first (growing a pool clears the shape cache), then the per-batch-size state, then the views of this shape
        big = dict(xlstm=self._pool("xlstm", (T2, B, feat)), d_xlstm=self._pool("d_xlstm", (T2, B, feat)),
                   enc_states=self._pool("enc_states", (B, T2, H)), d_enc=self._pool("d_enc", (B, T2, H)),
                   pred=self._pool("pred", (S, B), torch.int32), flags=self._pool("flags", (S,), torch.int32))
        bs = self._bstate.get(B)
        if bs is None:
            # zero-initialised and only partly written (encoder layers without a decoder counterpart keep their zero gradient)
            bs = dict(cT=torch.zeros(nd, nl, B, h, **f32), hT=torch.zeros(nd, nl, B, h, **f32),
                      d_cT=torch.zeros(nd, nl, B, h, **f32), d_hT=torch.zeros(nd, nl, B, h, **f32),
                      c0=torch.zeros(nld, B, H, **f32), h0=torch.zeros(nld, B, H, **f32),
                      d_c0=torch.zeros(nld, B, H, **f32), d_h0=torch.zeros(nld, B, H, **f32), loss=torch.zeros(2, **f32))
            self._bstate[B] = bs
        st = dict(key=key, B=B, T=T, D=D, L=L, T2=T2, feat=feat, S=S, cd=cd, cp=cp, cg=cg, ld=ld, lp=lp, lg=lg, dd=dd, dp=dp, dg=dg,
                  ws_cnn=int(lib.astk_conv_bn_relu_workspace_bytes(C.byref(cd))),
                  ws_lstm=int(lib.astk_lstm_stack_workspace_bytes(C.byref(ld))),
                  ws_dec=int(lib.astk_decoder_workspace_bytes(C.byref(dd))))
        st.update(big)
        st.update(bs)
        # the persistent kernels' status word rides in loss[1]: written by the decoder forward (the kernel that writes the loss) and once
        # more by the CNN backward's last kernel -- include/astk.h status_dst, a launch less each than astk_persist_status_snapshot
        dd.status_dst = cd.status_dst = bs["loss"].data_ptr() + 4
        assert st["ws_cnn"] and st["ws_lstm"] and st["ws_dec"], lib.astk_last_error().decode()
        self._shape_cache[key] = st
        return st

    def _rng(self, n):
        off = self._rng_offset
        self._rng_offset += n
        return off

    def _masks(self, name, shape, ratio):
        """Scaled keep-masks: injected (parity tests) or drawn on device; None when dropout is inactive."""
        if name in self.inject:
            m = self.inject[name]
            return None if m is None else m.to(self.device, torch.float32).contiguous()
        if not config.train or ratio <= 0:
            return None
        pre = self._predrawn.pop(name, None)
        if pre is not None and tuple(pre.shape) == tuple(shape):
            return pre
        t = self._pool("mask_" + name, shape)
        lib = _lib.load()
        check(lib.astk_fill_dropout_mask(_vp(t), t.numel(), float(ratio), self.rng_seed, self._rng(t.numel()), self._stream()))
        return t

    def _predraw(self, X_shape, add_noise, st, L, with_decoder):
        """All random tensors of a train step -- the speech noise and the four kinds of dropout masks -- in ONE launch (astk_fill_random)
        instead of up to five; the (seed, offset) counters are consumed in the order the separate draws used, so every value is the one
        they would have produced.  Injected tensors (parity tests) are left alone, as before."""
        self._predrawn = {}
        if not config.train and not (with_decoder and self._pending_flags is not None):
            return
        dr = self.cfg["dropout"]
        B, S = X_shape[0], L - 1
        want = []
        if config.train and add_noise > 0 and "noise" not in self.inject:
            want.append(("noise", tuple(X_shape), _lib.RAND_NORMAL, 1.0, float(add_noise), self.rng_seed ^ 0xABCDEF))
        masks = [("enc_masks", (self.n_dirs, len(self.rnn_enc), st["T2"], B, self.h), dr["rnn"])]
        if with_decoder:
            masks += [("emb_mask", (S, B, self.E), dr["embed"]), ("rnn_masks", (len(self.rnn_dec), S, B, self.H), dr["rnn"]),
                      ("out_mask", (S, B, self.V), dr.get("out", 0))]
        for name, shape, ratio in masks:
            if config.train and name not in self.inject and ratio and ratio > 0:
                want.append((name, shape, _lib.RAND_DROPOUT, float(ratio), 0.0, self.rng_seed))
        flags = self._pending_flags if with_decoder else None
        if len(want) > _lib.RAND_SEG_MAX or (len(want) < 2 and flags is None):
            return                                    # (a single draw keeps its own launch)
        segs = (_lib.RandSeg * max(1, len(want)))()
        for i, (name, shape, kind, a, b, seed) in enumerate(want):
            t = self._pool("noise" if name == "noise" else "mask_" + name, shape)
            segs[i].out, segs[i].n, segs[i].kind, segs[i].a, segs[i].b = t.data_ptr(), t.numel(), kind, a, b
            segs[i].seed, segs[i].offset = seed & 0xFFFFFFFFFFFFFFFF, self._rng(t.numel())
            self._predrawn[name] = t
        if flags is None:
            check(_lib.load().astk_fill_random(segs, len(want), self._stream()))
        else:
            words = (C.c_int32 * len(flags))(*flags)
            check(_lib.load().astk_fill_random_ex(segs, len(want), words, len(flags), _vp(st["flags"]), self._stream()))
            self._pending_flags = None

    # ------------------------------------------------------------------ encoder (seq2seq.py:293-314)
    def _as_input(self, X):
        if isinstance(X, np.ndarray):
            X = torch.from_numpy(X)
        if hasattr(X, "data") and not isinstance(X, torch.Tensor):
            X = X.data
        return X.to(self.device, torch.float32).contiguous()

    def encode(self, X, add_noise=0):
        lib = self._require_gpu()
        X = self._as_input(X)
        B, T, D = X.shape
        in_forward_loss = bool(self._cur and self._cur.get("pending_L"))
        L = self._cur["L"] if in_forward_loss else 2
        st = self._shape_state(B, T, D, L)
        self._cur = st
        st["X"] = X
        noise = None
        self._predraw(tuple(X.shape), add_noise, st, L, in_forward_loss)
        if add_noise > 0 and config.train:
            if "noise" in self.inject:
                noise = self.inject["noise"].to(self.device, torch.float32).contiguous()
            elif "noise" in self._predrawn:
                noise = self._predrawn.pop("noise")
            else:
                noise = self._pool("noise", tuple(X.shape))
                check(lib.astk_fill_normal(_vp(noise), noise.numel(), 1.0, float(add_noise), self.rng_seed ^ 0xABCDEF,
                                           self._rng(noise.numel()), self._stream()))
        st["noise"] = noise
        nl, nd = len(self.rnn_enc), self.n_dirs
        st["enc_masks"] = self._masks("enc_masks", (nd, nl, st["T2"], B, self.h), self.cfg["dropout"]["rnn"])
        st["train_mode"] = bool(config.train)
        s = self._stream()
        wc = self._workspace("cnn", st["ws_cnn"])
        sx = self.stat_exchange if config.train else None
        if sx is None:
            check(lib.astk_conv_bn_relu_fwd(C.byref(st["cd"]), st["cp"], _vp(X), _vp(noise), _vp(st["xlstm"]), _vp(wc), wc.numel(),
                                            1 if config.train else 0, s))
        else:
            rc = lib.astk_conv_bn_relu_fwd_sync(C.byref(st["cd"]), st["cp"], _vp(X), _vp(noise), _vp(st["xlstm"]), _vp(wc), wc.numel(), 1,
                                                C.cast(sx.bind(wc).callback, C.c_void_p), None, sx.world, s)
            if sx.error is not None:
                raise sx.error
            check(rc)
        st["bn_world"] = 1 if sx is None else sx.world
        if config.train:
            self.bn_N += 1
        st["ld"].x_amax = lib.astk_conv_out_amax(C.byref(st["cd"]), _vp(wc), wc.numel())      # (None for bad arguments: the GEMM measures x itself)
        if self.enc_variant is not None:                     # rnn_config.ln / linear_proj: layer-by-layer (ast_amd/enc_variants.py)
            if self.rnn_linear_proj:
                self.enc_variant.forward(st, bool(config.train))
            else:
                self.enc_variant.forward(st)
        else:
            wl = self._workspace("lstm", st["ws_lstm"])
            # work beside the recurrences: the library cuts the layer-0 products into time chunks on this stream (astk.h side_stream), and
            # sizes its recurrence workgroups (16 / 32 batch rows) by whether it has one -- the backward call sees the same descriptor
            side = self._side_stream()
            st["ld"].side_stream = side.cuda_stream if side is not None else None
            check(lib.astk_lstm_stack_fwd(C.byref(st["ld"]), st["lp"], _vp(st["xlstm"]), _vp(st["enc_masks"]), _vp(st["enc_states"]),
                                          _vp(st["cT"]), _vp(st["hT"]), _vp(wl), wl.numel(), s))
        self.enc_states = st["enc_states"]
        self.loss = 0

    def _bridge_states(self, st, dec_c, dec_h, enc_c, enc_h, to_decoder):
        """[fwd_k ; rev_k] of the encoder <-> decoder layer k, for both tensors and every bridged layer in ONE launch (the twelve strided
        torch copies of a 3-layer model were 58 us of a 7.2 ms step)."""
        n = min(len(self.rnn_enc), len(self.rnn_dec))
        if self.h % 4:                      # (the kernel moves 16-byte pieces)
            for k in range(n):
                for d_, e_ in ((dec_c, enc_c), (dec_h, enc_h)):
                    if to_decoder:
                        d_[k].view(-1, self.n_dirs, self.h).copy_(e_[:, k].permute(1, 0, 2))
                    else:
                        e_[:, k].copy_(d_[k].view(-1, self.n_dirs, self.h).permute(1, 0, 2))
            return
        check(_lib.load().astk_bridge_states(_vp(dec_c), _vp(dec_h), _vp(enc_c), _vp(enc_h), self.n_dirs, len(self.rnn_enc), n, st["B"], self.h,
                                             1 if to_decoder else 0, self._stream()))

    # ------------------------------------------------------------------ seq2seq.py:318-333
    def init_decoder_state(self):
        st = self._cur
        h, nd = self.h, self.n_dirs
        # decoder layer k starts from [fwd_k ; rev_k] of the encoder; layers without an encoder counterpart keep the zeros they
        # were allocated with (c0/h0 are only ever written here).  One strided copy per tensor and layer.
        self._bridge_states(st, st["c0"], st["h0"], st["cT"], st["hT"], True)
        if config.train:
            self._dec_c, self._dec_h = st["c0"], st["h0"]      # the step API (decode_step) clones before it advances them
        else:
            self._dec_c, self._dec_h = st["c0"].clone(), st["h0"].clone()

    # ------------------------------------------------------------------ seq2seq.py:399-473
    def _side_stream(self):
        """The model's second stream (ordinary, from torch's pool) for the stream the step is running on, or None: side-stream work off, the
        step on the legacy default stream (which synchronises implicitly with every other stream), or no stream found that really runs
        CONCURRENTLY with it.  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams that
        share one execute in order whatever their events say -- the flag-gated hand-offs between the recurrence kernels and the chunked
        products beside them need true concurrency (in one queue the consumer would sit in front of its producer until its bounded spin
        times out), so every candidate is probed once (a 3 ms spin on one stream, a trivial kernel on the other)."""
        if not self.side_stream_on or self.enc_variant is not None or self.deterministic:
            return None
        main = torch.cuda.current_stream(self.device)
        if main.cuda_stream == 0:
            return None
        if main.cuda_stream not in self._side_by_main:
            chosen = None
            for _ in range(6):
                cand = torch.cuda.Stream(device=self.device)
                if self._concurrent(main, cand) and self._concurrent(cand, main):
                    chosen = cand
                    break
            self._side_by_main[main.cuda_stream] = chosen
        self._side = self._side_by_main[main.cuda_stream]
        return self._side

    def _concurrent(self, a, b):
        """True if a kernel on stream b runs while stream a is busy (one 3 ms spin on a, a trivial kernel on b)."""
        lib = _lib.load()
        probe = self._ws.get("probe")
        if probe is None:
            probe = self._ws["probe"] = torch.zeros(4, device=self.device)
            check(lib.astk_spin(10, None, C.c_void_p(a.cuda_stream)))              # (first launches load the code objects: not inside the probe)
            check(lib.astk_scale_f32(_vp(probe), 4, 1.0, C.c_void_p(a.cuda_stream)))
        torch.cuda.synchronize(self.device)
        check(lib.astk_spin(3000, None, C.c_void_p(a.cuda_stream)))
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(a)
        check(lib.astk_scale_f32(_vp(probe), 4, 1.0, C.c_void_p(b.cuda_stream)))
        eb.record(b)
        eb.synchronize()
        overlapped = not ea.query()
        ea.synchronize()
        return overlapped

    def close(self):
        """Kept for callers of rounds 2-5 (the CU-masked streams it used to destroy are gone; torch owns the side streams)."""
        self._side = None
        self._side_by_main = {}

    def _upload_flags(self, dst, flags):
        """Host -> device copy of the teacher-forcing flags WITHOUT a host sync: a copy from pageable memory would make the host wait
        for the encoder of this step, and the GPU then idles while the host catches up at the start of the next one.  The flags
        go through a small ring of pinned buffers; a slot is reused only after the copy that read it has executed."""
        ring = self._ws.get("flag_ring")
        if ring is None or ring["bufs"][0][0].numel() < len(flags):
            ring = {"i": 0, "bufs": [(torch.empty(max(64, len(flags)), dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(8)]}
            self._ws["flag_ring"] = ring
        buf, ev = ring["bufs"][ring["i"]]
        ring["i"] = (ring["i"] + 1) % len(ring["bufs"])
        ev.synchronize()                       # returns at once unless the host is 8 steps ahead
        buf[:len(flags)] = torch.tensor(flags, dtype=torch.int32)
        # (round 5: a side stream for this copy, beside the encoder, was measured -- it trades the 4.5 us copy for a 6 us cross-queue wait on
        #  the compute stream: nothing, and one more stream per model next to RCCL's; the copy stays in stream order)
        dst.copy_(buf[:len(flags)], non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))

    def forward_loss(self, X, y, teach_ratio, random_out=0, add_noise=0, y_global=None):
        """seq2seq.py:399-473.  y_global (data parallelism with random_out > 0 only): the targets of the WHOLE unsharded batch as a host
        array, global row b * world + r = row b of rank r -- the loader has them before it shards (`batch["y_global"]`); without it the
        ranks' targets are all-gathered, a host-blocking collective in front of every step."""
        lib = self._require_gpu()
        X = self._as_input(X)
        if isinstance(y, np.ndarray):
            y = torch.from_numpy(y)
        if hasattr(y, "data") and not isinstance(y, torch.Tensor):
            y = y.data
        y_host = y if (random_out and y.device.type == "cpu") else None
        y = y.to(self.device, torch.int32).contiguous()
        B, L = y.shape
        assert L >= 2, "targets need at least GO and EOS"
        S = L - 1
        # quirk Q4: one Python-`random` coin per step for 0 < i < L-2, truth otherwise (seq2seq.py:431-436).  With random_out > 0
        # (seq2seq.py:456-465) the SAME stream also decides, behind each step's coin, which targets >= 4 are replaced (draw ABOVE
        # random_out) by a class id from xp.random.randint(4, dec_vocab_size + 1).  That range is inclusive of dec_vocab_size, one past the
        # last class (quirk Q8): Chainer's softmax_cross_entropy raises on it with NumPy and reads out of bounds with CuPy.  DEVIATION: the
        # drawn id is clamped to dec_vocab_size - 1.  Nothing in a decoder step consumes either stream, so all draws are made here, in
        # the reference's order, and the scored targets go to the device as a second (B, L) matrix.
        targets = None
        if "use_truth" in self.inject:
            flags = [int(bool(v)) for v in self.inject["use_truth"]]
            assert not random_out, "inject['use_truth'] bypasses the random stream random_out shares"
        elif random_out:
            yh = (y_host if y_host is not None else y.cpu()).numpy()
            randint = self.inject.get("randint", np.random.randint)      # the reference's draw is the unseeded global xp RNG (quirk Q7)
            gather = None
            if y_global is not None:
                from . import dist as adist
                yg, w_ = np.asarray(y_global), adist.world_size()
                if w_ > 1 and yg.shape == (yh.shape[0] * w_, yh.shape[1]):
                    gather = lambda a: [yg[r::w_] for r in range(w_)]          # noqa: E731  (no collective: every rank holds the same global rows)
            flags, tg = draw_flags_and_targets(yh, teach_ratio, random_out, self.V, randint, gather=gather)
            targets = torch.from_numpy(np.ascontiguousarray(tg, dtype=np.int32)).to(self.device)
        else:
            flags = [int(random.random() < teach_ratio) if 0 < i < L - 2 else 1 for i in range(S)]
        self.use_truth = flags
        # (the flags go to the device with the step's random tensors, in the kernel arguments of that one launch: _predraw)
        self._pending_flags = flags if len(flags) <= _lib.RAND_WORDS_MAX else None
        self._cur = {"L": L, "pending_L": True}
        self.encode(X, add_noise=add_noise)
        st = self._cur
        self.init_decoder_state()
        if self._pending_flags is not None or len(flags) > _lib.RAND_WORDS_MAX:      # (not taken by _predraw: a copy of their own)
            self._pending_flags = None
            self._upload_flags(st["flags"], flags)
        st["flags_host"] = (C.c_int32 * S)(*flags)          # the per-launch loop scores the teacher-forced steps behind the loop (astk.h)
        st["dd"].use_truth_host = C.cast(st["flags_host"], C.POINTER(C.c_int32))
        st["y"] = y
        st["targets"] = targets
        dr = self.cfg["dropout"]
        st["emb_mask"] = self._masks("emb_mask", (S, B, self.E), dr["embed"])
        st["rnn_masks"] = self._masks("rnn_masks", (len(self.rnn_dec), S, B, self.H), dr["rnn"])
        st["out_mask"] = self._masks("out_mask", (S, B, self.V), dr.get("out", 0))       # dropout on the logits (seq2seq.py:394)
        wd = self._workspace("dec", st["ws_dec"])
        check(lib.astk_decoder_fwd_ex(C.byref(st["dd"]), C.byref(st["dp"]), _vp(st["enc_states"]), _vp(st["c0"]), _vp(st["h0"]),
                                      _vp(y), _vp(st["flags"]), _vp(st["emb_mask"]), _vp(st["rnn_masks"]), _vp(st["out_mask"]), _vp(targets),
                                      _vp(st["loss"]), _vp(st["pred"]), _vp(wd), wd.numel(), self._stream()))
        # (loss[1] = the persistent kernels' status word: written by the op itself, astk_decoder_desc.status_dst)
        st["pending_backward"] = True
        self.loss = Loss(self, st["loss"])
        return self.loss

    def _backward(self):
        lib = _lib.load()
        st = self._cur
        s = self._stream()
        wd = self._workspace("dec", st["ws_dec"])
        st["pending_backward"] = False
        # a cleargrads() deferred to this pass: the decoder backward zeroes the arena in front of everything it accumulates (astk.h zero_ptr)
        if self.arena.take_zero():
            st["dd"].zero_ptr, st["dd"].zero_bytes = self.arena._grad.data_ptr(), 4 * self.arena.size
        else:
            st["dd"].zero_ptr, st["dd"].zero_bytes = None, 0

        def dec_bwd(phase, stream):
            check(lib.astk_decoder_bwd_phase_ex(C.byref(st["dd"]), C.byref(st["dp"]), C.byref(st["dg"]), _vp(st["enc_states"]), _vp(st["c0"]),
                                                _vp(st["h0"]), _vp(st["y"]), _vp(st["emb_mask"]), _vp(st["rnn_masks"]), _vp(st["out_mask"]),
                                                _vp(st["d_enc"]), _vp(st["d_c0"]), _vp(st["d_h0"]), _vp(wd), wd.numel(), phase, stream))
        joined = None
        side = self._side_stream()
        free = lib.astk_lstm_stack_free_cus(C.byref(st["ld"])) if side is not None else 0
        if side is not None and free >= 16:
            # The encoder's backward recurrence (0.7 ms; one workgroup on each of 96-192 CUs, one wave per SIMD) leaves the rest of the
            # device idle and needs nothing from the decoder's parameter gradients (0.17 ms of GEMMs at full speed).  They run beside
            # it on the side stream, every grid capped at the free CUs, ordered after the chain phase by an event and joined before
            # anything reads the gradient arena or reuses the workspace.
            main = torch.cuda.current_stream(self.device)
            dec_bwd(1, s)                                    # ASTK_DEC_BWD_CHAIN
            fork = torch.cuda.Event()
            fork.record(main)
            st["dd"].side_wgs = free
            with torch.cuda.stream(side):
                side.wait_event(fork)
                dec_bwd(2, C.c_void_p(side.cuda_stream))     # ASTK_DEC_BWD_PARAMS
                joined = torch.cuda.Event()
                joined.record(side)
            st["dd"].side_wgs = 0
        else:
            dec_bwd(0, s)
        # The decoder's gradient range is final here, but its all-reduce is NOT launched yet: an RCCL kernel holds its CUs until every
        # peer has arrived, and the encoder's backward recurrence (next) needs up to 192 CUs to itself to become resident -- a peer
        # that is late would turn into a time-out of the recurrence's bounded spins.  Both ranges go out behind the recurrence and
        # hide behind the encoder's batched weight-gradient products and the CNN backward instead.
        h, nd = self.h, self.n_dirs
        # encoder layers without a decoder counterpart keep the zero gradient they were allocated with
        self._bridge_states(st, st["d_c0"], st["d_h0"], st["d_cT"], st["d_hT"], False)
        if self.enc_variant is not None:
            self.enc_variant.backward(st, st["d_enc"], st["d_cT"], st["d_hT"], st["d_xlstm"])
        else:
            wl = self._workspace("lstm", st["ws_lstm"])
            check(lib.astk_lstm_stack_bwd(C.byref(st["ld"]), st["lp"], st["lg"], _vp(st["xlstm"]), _vp(st["enc_masks"]), _vp(st["d_enc"]),
                                          _vp(st["d_cT"]), _vp(st["d_hT"]), _vp(st["d_xlstm"]), _vp(wl), wl.numel(), s))
        if joined is not None:
            # from here on everything is on one stream again, and the gradient exchange is launched from it
            torch.cuda.current_stream(self.device).wait_event(joined)
        if self.grad_buckets is not None:
            self.grad_buckets.launch("dec")
            self.grad_buckets.launch("enc")
        wc = self._workspace("cnn", st["ws_cnn"])
        sx = self.stat_exchange if st.get("bn_world", 1) > 1 else None   # backward of the statistics the forward pass used
        if sx is None:
            check(lib.astk_conv_bn_relu_bwd(C.byref(st["cd"]), st["cp"], st["cg"], _vp(st["d_xlstm"]), _vp(wc), wc.numel(), s))
        else:
            rc = lib.astk_conv_bn_relu_bwd_sync(C.byref(st["cd"]), st["cp"], st["cg"], _vp(st["d_xlstm"]), _vp(wc), wc.numel(),
                                                C.cast(sx.bind(wc).callback, C.c_void_p), None, sx.world, s)
            if sx.error is not None:
                raise sx.error
            check(rc)
        if self.grad_buckets is not None:
            self.grad_buckets.launch("cnn")
        # (the status word once more, behind the backward recurrences: astk_cnn_desc.status_dst, written by the CNN backward's last kernel)
        if self.grad_buckets is not None:
            # ... and once more behind the gradient exchange, when the peers' words have been merged (GradBuckets.finish): every rank's
            # pair of THIS step then carries the merged word
            self.grad_buckets.status_dest = st["loss"][1:2]

    # ------------------------------------------------------------------ inference (seq2seq.py:361-396, 475-568)
    def decode_step(self, word, ht):
        """Eval-mode step for predict()/beam: returns (logits (B,V), ht (B,A), alphas (B,T'',1))."""
        lib = self._require_gpu()
        if config.train:
            raise NotImplementedError("decode_step outside forward_loss is provided for eval mode (predict / beam)")
        st = self._cur
        if isinstance(word, np.ndarray):
            word = torch.from_numpy(word)
        word = word.to(self.device, torch.int32).contiguous()
        B = word.shape[0]
        ht = ht.to(self.device, torch.float32).contiguous().clone()
        logits = torch.empty(B, self.V, dtype=torch.float32, device=self.device)
        alpha = torch.empty(B, st["T2"], dtype=torch.float32, device=self.device)
        wd = self._workspace("dec", st["ws_dec"])
        check(lib.astk_decoder_step_infer(C.byref(st["dd"]), C.byref(st["dp"]), _vp(st["enc_states"]), _vp(self._dec_c),
                                          _vp(self._dec_h), _vp(ht), _vp(word), _vp(logits), _vp(alpha), None, _vp(wd), wd.numel(),
                                          self._stream()))
        return logits, ht, alpha.unsqueeze(2)

    def predict(self, X, start_token, end_token, stop_limit):
        with using_config("train", False):
            X = self._as_input(X)
            B = X.shape[0]
            self._cur = None
            self.encode(X)
            self.init_decoder_state()
            ht = torch.zeros(B, self.A, dtype=torch.float32, device=self.device)
            word = torch.full((B,), start_token, dtype=torch.int32, device=self.device)
            done = torch.zeros(B, dtype=torch.bool, device=self.device)
            rows, npred = [], 0
            while npred < stop_limit:
                logits, ht, _ = self.decode_step(word, ht)
                word = logits.argmax(dim=1).to(torch.int32)
                rows.append(word)
                done |= word == end_token
                if bool(done.all()):
                    break
                npred += 1
            return torch.stack(rows, 0).T.cpu().numpy()

    def get_encoder_states(self):
        st = self._cur
        h = self.h
        out = {"c": [], "h": []}
        for k in range(len(self.rnn_enc)):
            out["c"].append(torch.cat([st["cT"][d_, k] for d_ in range(self.n_dirs)], 1))
            out["h"].append(torch.cat([st["hT"][d_, k] for d_ in range(self.n_dirs)], 1))
        return out

    def get_decoder_states(self):
        return {"c": [self._dec_c[l].clone() for l in range(len(self.rnn_dec))],
                "h": [self._dec_h[l].clone() for l in range(len(self.rnn_dec))]}

    def set_decoder_states(self, rnn_states):
        n = len(self.rnn_dec)
        B = rnn_states["c"][0].shape[0]
        c = torch.zeros(n, B, self.H, dtype=torch.float32, device=self.device)
        hh = torch.zeros(n, B, self.H, dtype=torch.float32, device=self.device)
        for l in range(min(n, len(rnn_states["c"]))):
            c[l] = rnn_states["c"][l]
            hh[l] = rnn_states["h"][l]
        self._dec_c, self._dec_h = c, hh
