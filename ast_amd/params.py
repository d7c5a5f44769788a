"""Flat parameter / gradient arenas in HBM.

All trainable parameters live in ONE contiguous float32 buffer (and their gradients in another), each
parameter 16-byte aligned, in Chainer's layouts and under Chainer's save_npz names.  One
flat buffer makes `cleargrads` a single memset, the WeightDecay -> GradientClipping -> Adam hooks of
nn.py:81-119 two kernel launches, and the data-parallel gradient exchange a single RCCL all-reduce.
"""
from collections.abc import Mapping
import numpy as np
import torch


def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def param_shapes(cfg, in_dim, vocab_size=None):
    """Ordered {name: shape} for trainables and persistents, following seq2seq.py:35-156.
    `in_dim` resolves the reference's lazily-shaped links (in_channels=None / L.LSTM(None, ...))."""
    rc, cc = cfg["rnn_config"], cfg["cnn_config"]
    V = vocab_size if vocab_size is not None else rc["dec_vocab_size"]
    train, persist = {}, {}
    cin, fdim = 1, in_dim
    for i, l in enumerate(cc["cnn_layers"]):
        kh, kw = l["ksize"]
        co = l["out_channels"]
        train[f"CNN_{i}/W"] = (co, cin, kh, kw)
        if cc["bn"]:
            train[f"CNN_{i}_bn/gamma"] = (co,)
            train[f"CNN_{i}_bn/beta"] = (co,)
            persist[f"CNN_{i}_bn/avg_mean"] = (co,)
            persist[f"CNN_{i}_bn/avg_var"] = (co,)
        else:
            train[f"CNN_{i}/b"] = (co,)                     # nobias=self.cnn_bn (seq2seq.py:52-54)
        fdim = conv_out(fdim, kw, l["stride"][1], l["pad"][1])
        if "cnn_pool" in cc:                                # OLD-path extra (enc_dec.py:444-456): max-pool, cover_all; -1 = whole extent
            kf = cc["cnn_pool"][i][1]
            fdim = 1 if kf == -1 else -(-fdim // max(int(kf), 1))
        cin = co
    rnn_in = cin * fdim
    Hh = rc["hidden_units"] // 2 if rc["bi_rnn"] else rc["hidden_units"]
    ln, proj = bool(rc.get("ln", False)), bool(rc.get("linear_proj", False))

    def lstm(n, n_in, n_out):
        train[f"{n}/upward/W"] = (4 * n_out, n_in)
        train[f"{n}/upward/b"] = (4 * n_out,)
        train[f"{n}/lateral/W"] = (4 * n_out, n_out)
        if ln:                                              # L.LayerNormalization(units) behind the LSTM (seq2seq.py:85-87, 141-143)
            train[f"{n}_ln/gamma"] = (n_out,)
            train[f"{n}_ln/beta"] = (n_out,)
    for pat in ["L{}_enc"] + (["L{}_rev_enc"] if rc["bi_rnn"] else []):
        n_in = rnn_in
        for i in range(rc["enc_layers"]):
            lstm(pat.format(i), n_in, Hh)
            # the lazily-shaped L.LSTM(None, units) takes the width of what it is first fed: the layer below (units per direction), or
            # -- with linear_proj -- the projection of the concatenated states (hidden_units), seq2seq.py:250-286
            n_in = rc["hidden_units"] if proj else Hh
    H, E, A = rc["hidden_units"], rc["embedding_units"], rc["attn_units"]
    if proj:                                                # seq2seq.py:89-99
        for i in range(rc["enc_layers"] - 1):
            train[f"enc_proj{i}/W"] = (H, H)
            train[f"enc_proj{i}/b"] = (H,)
            train[f"enc_proj{i}_bn/gamma"] = (H,)
            train[f"enc_proj{i}_bn/beta"] = (H,)
            persist[f"enc_proj{i}_bn/avg_mean"] = (H,)
            persist[f"enc_proj{i}_bn/avg_var"] = (H,)
    n_attn = rc.get("n_attn", 1)
    train["attn_Wa/W"] = (H, H)
    train["attn_Wa/b"] = (H,)
    for i in range(1, n_attn):                              # seq2seq.py:114-116
        train[f"attn_Wa{i}/W"] = (H, H)
        train[f"attn_Wa{i}/b"] = (H,)
    train["context/W"] = (A, (n_attn + 1) * H)
    train["context/b"] = (A,)
    train["embed_dec/W"] = (V, E)
    n_in = E + A if rc.get("feed_attn", True) else E
    for i in range(rc["dec_layers"]):
        lstm(f"L{i}_dec", n_in, H)
        n_in = H
    train["out/W"] = (V, A)
    train["out/b"] = (V,)
    return train, persist


def init_values(cfg, in_dim, vocab_size=None, seed=0):
    """Reference initialisers: conv HeNormal, Linear/LSTM LeCunNormal, forget-gate bias 1,
    EmbedID N(0,1), BN gamma 1 / beta 0, running mean 0 / var 1."""
    train, persist = param_shapes(cfg, in_dim, vocab_size)
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in train.items():
        leaf = name.split("/")[-1]
        if name.startswith("CNN_") and leaf == "W":
            fan_in = shp[1] * shp[2] * shp[3]
            v = rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)
        elif leaf == "gamma":
            v = np.ones(shp)
        elif leaf in ("beta",):
            v = np.zeros(shp)
        elif name == "embed_dec/W":
            v = rng.standard_normal(shp)
        elif leaf == "W":
            v = rng.standard_normal(shp) * np.sqrt(1.0 / shp[1])
        elif leaf == "b":
            v = np.zeros(shp)
            if "/upward/" in name:
                v[2::4] = 1.0
        else:
            raise KeyError(name)
        out[name] = v.astype(np.float32)
    for name, shp in persist.items():
        out[name] = (np.ones(shp) if name.endswith("avg_var") else np.zeros(shp)).astype(np.float32)
    return out


class _FlushingViews(Mapping):
    """name -> gradient view; ANY read performs the arena's deferred zero fill first (ParamArena.flush_zero).  A Mapping, not a dict
    subclass (round-5 advice: dict(gviews), {**gviews}, .copy() and CPython's fast iteration paths bypass overridden dict methods and
    would hand out gradients cleargrads() has cleared): every way in -- [], get, items, values, iteration, dict(...) -- goes through
    __getitem__ / __iter__ here.  (A view a caller obtained EARLIER and kept is an alias of the arena like any tensor slice; it shows the
    fill once something has flushed it.)"""

    def __init__(self, arena):
        self._arena = arena
        self._views = {}

    def _put(self, k, v):                 # (construction only)
        self._views[k] = v

    def raw(self, k):                     # an address, not a read: no flush
        return self._views[k]

    def __getitem__(self, k):
        self._arena.flush_zero()
        return self._views[k]

    def __iter__(self):
        self._arena.flush_zero()
        return iter(self._views)

    def __len__(self):
        return len(self._views)


class ParamArena:
    TAIL = 4

    def __init__(self, shapes, device):
        self.shapes = dict(shapes)
        self.offsets = {}
        off = 0
        for name, shp in self.shapes.items():
            self.offsets[name] = off
            n = int(np.prod(shp))
            off += (n + 3) // 4 * 4                 # 16-byte aligned slices; pad elements stay zero forever
        self.size = off
        self.device = device
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        # the gradient buffer carries TAIL extra floats behind the last parameter: [0] = the abort status word that rides in the last
        # range's all-reduce under data parallelism (ast_amd.dist.GradBuckets); not a parameter, never seen by cleargrads / the optimizer
        self.grad_full = torch.zeros(off + self.TAIL, dtype=torch.float32, device=device)
        # cleargrads() may DEFER its zero fill to the backward pass that follows it (defer_zero / take_zero: the decoder backward's first
        # fill launch takes the arena along, include/astk.h astk_decoder_desc.zero_ptr); every other reader of the gradients -- `grad`,
        # `gviews[...]`, to_numpy(grads=True) -- goes through flush_zero() first, so nobody ever sees gradients cleargrads() has cleared
        self._grad = self.grad_full[:off]
        self._zero_pending = False
        self.status_tail = self.grad_full[off:off + 1]
        self.views, self.gviews = {}, _FlushingViews(self)
        for name, shp in self.shapes.items():
            o, n = self.offsets[name], int(np.prod(shp))
            self.views[name] = self.data[o:o + n].view(shp)
            self.gviews._put(name, self._grad[o:o + n].view(shp))

    @property
    def grad(self):
        self.flush_zero()
        return self._grad

    def defer_zero(self):
        self._zero_pending = True

    def take_zero(self):
        """True once per deferred zero fill: the caller has taken it over."""
        p, self._zero_pending = self._zero_pending, False
        return p

    def flush_zero(self):
        if self._zero_pending:
            self._zero_pending = False
            self._grad.zero_()

    def range_of(self, name):
        o = self.offsets[name]
        n = int(np.prod(self.shapes[name]))
        return o, (n + 3) // 4 * 4

    def p(self, name):
        return self.views[name].data_ptr()

    def g(self, name):
        return self.gviews.raw(name).data_ptr()       # (an address, not a read: no flush)

    def load(self, values):
        for name in self.shapes:
            self.views[name].copy_(torch.as_tensor(values[name], dtype=torch.float32).reshape(self.shapes[name]))

    def to_numpy(self, grads=False):
        src = self.gviews if grads else self.views
        return {k: v.detach().cpu().numpy().copy() for k, v in src.items()}
