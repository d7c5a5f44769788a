"""oracle/minichainer.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A tiny define-by-run reverse-mode autograd on NumPy that restates exactly the
Chainer primitives the reference's encoder-decoder hot path calls (SURVEY.md
section 2.3 / Appendix A).  Chainer itself is a third-party dependency of the
reference, is not vendored under /root/reference and is not installable
offline, so its *published* semantics (Chainer v5/v6 docs) are restated here;
every function names the Appendix-A item and the reference call site it
serves.

PARITY UNPINNED: the reference ships no tests / golden vectors for this path
and Chainer cannot be executed here.  This restatement is pinned instead by
(i) float64 central-difference gradient checks and (ii) an independent
torch-autograd restatement (oracle/ast_ref_torch.py); see tests/test_oracle_*.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
import heapq
import numpy as np


# --------------------------------------------------------------------------- core
class Variable:
    __slots__ = ("data", "grad", "creator", "rank", "name")

    def __init__(self, data, name=None):
        self.data = np.asarray(data)
        self.grad = None
        self.creator = None
        self.rank = 0
        self.name = name

    # array-ish protocol used by the reference code (`len(y)`, iteration, X[i], X[-i])
    @property
    def shape(self):
        return self.data.shape

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def ndim(self):
        return self.data.ndim

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        return get_item(self, idx)

    def __iter__(self):
        for i in range(len(self)):
            yield get_item(self, i)

    def __add__(self, other):
        return add(self, other)

    def __radd__(self, other):          # `loss = 0; loss += curr_loss` (seq2seq.py:415,470)
        if isinstance(other, (int, float)) and other == 0:
            return self
        return add(self, other)

    def __mul__(self, other):
        return mul(self, other)

    def cleargrad(self):
        self.grad = None

    def backward(self):
        """Reverse pass in decreasing creation rank (Chainer's Variable.backward)."""
        if self.grad is None:
            self.grad = np.ones_like(self.data)
        heap, seen, tick = [], set(), 0

        def push(f):
            nonlocal tick
            if f is not None and id(f) not in seen:
                seen.add(id(f))
                heapq.heappush(heap, (-f.rank, tick, f))
                tick += 1

        push(self.creator)
        while heap:
            _, _, f = heapq.heappop(heap)
            gys = [o.grad for o in f.outputs]
            gxs = f.backward(gys)
            if not isinstance(gxs, (tuple, list)):
                gxs = (gxs,)
            for x, gx in zip(f.inputs, gxs):
                if gx is None:
                    continue
                x.grad = gx if x.grad is None else x.grad + gx
                push(x.creator)
            for o in f.outputs:          # free intermediate grads like Chainer (retain_grad=False)
                if o is not self and o.creator is not None:
                    o.grad = None


class Parameter(Variable):
    __slots__ = ()


def as_variable(x):
    return x if isinstance(x, Variable) else Variable(x)


class Function:
    def __call__(self, *inputs):
        inputs = [as_variable(x) for x in inputs]
        ys = self.forward([v.data for v in inputs])
        if not isinstance(ys, tuple):
            ys = (ys,)
        self.inputs = inputs
        self.rank = max((v.rank for v in inputs), default=0)
        outs = []
        for y in ys:
            o = Variable(y)
            o.creator = self
            o.rank = self.rank + 1
            outs.append(o)
        self.outputs = outs
        return outs[0] if len(outs) == 1 else tuple(outs)


# --------------------------------------------------------------------------- shape ops
class _GetItem(Function):
    def __init__(self, idx):
        self.idx = idx

    def forward(self, xs):
        self.in_shape, self.in_dtype = xs[0].shape, xs[0].dtype
        return xs[0][self.idx]

    def backward(self, gys):
        gx = np.zeros(self.in_shape, dtype=self.in_dtype)
        gx[self.idx] = gys[0]
        return gx


def get_item(x, idx):
    return _GetItem(idx)(x)


class _Reshape(Function):
    def __init__(self, shape):
        self.shape = shape

    def forward(self, xs):
        self.in_shape = xs[0].shape
        return xs[0].reshape(self.shape)

    def backward(self, gys):
        return gys[0].reshape(self.in_shape)


def reshape(x, shape):
    return _Reshape(shape)(x)


def expand_dims(x, axis):
    x = as_variable(x)
    shp = list(x.shape)
    if axis < 0:
        axis = len(shp) + 1 + axis
    shp.insert(axis, 1)
    return reshape(x, tuple(shp))


def squeeze(x, axis):
    x = as_variable(x)
    shp = list(x.shape)
    assert shp[axis] == 1
    del shp[axis]
    return reshape(x, tuple(shp))


class _Transpose(Function):
    def __init__(self, axes):
        self.axes = tuple(axes)

    def forward(self, xs):
        return xs[0].transpose(self.axes)

    def backward(self, gys):
        return gys[0].transpose(np.argsort(self.axes))


def swapaxes(x, a, b):
    x = as_variable(x)
    axes = list(range(x.ndim))
    axes[a], axes[b] = axes[b], axes[a]
    return _Transpose(axes)(x)


def rollaxis(x, axis, start=0):
    """F.rollaxis(x, 1): move `axis` to position `start` (A5)."""
    x = as_variable(x)
    axes = list(range(x.ndim))
    axes.remove(axis)
    axes.insert(start, axis)
    return _Transpose(axes)(x)


class _FlipUD(Function):
    def forward(self, xs):
        return xs[0][::-1]

    def backward(self, gys):
        return gys[0][::-1]


def flipud(x):
    return _FlipUD()(x)


class _Concat(Function):
    def __init__(self, axis):
        self.axis = axis

    def forward(self, xs):
        self.sizes = [x.shape[self.axis] for x in xs]
        return np.concatenate(xs, axis=self.axis)

    def backward(self, gys):
        cuts = np.cumsum(self.sizes)[:-1]
        return tuple(np.split(gys[0], cuts, axis=self.axis))


def concat(xs, axis=1):
    """F.concat, default axis=1 (A5)."""
    return _Concat(axis)(*xs)


# --------------------------------------------------------------------------- arithmetic
class _Add(Function):
    def forward(self, xs):
        return xs[0] + xs[1]

    def backward(self, gys):
        return gys[0], gys[0]


def add(a, b):
    return _Add()(a, b)


class _Mul(Function):
    def forward(self, xs):
        self.xs = xs
        return xs[0] * xs[1]

    def backward(self, gys):
        return gys[0] * self.xs[1], gys[0] * self.xs[0]


def mul(a, b):
    return _Mul()(a, b)


class _Tanh(Function):
    def forward(self, xs):
        self.y = np.tanh(xs[0])
        return self.y

    def backward(self, gys):
        return gys[0] * (1 - self.y * self.y)


def tanh(x):
    return _Tanh()(x)


class _ReLU(Function):
    def forward(self, xs):
        self.y = np.maximum(xs[0], 0)
        return self.y

    def backward(self, gys):
        return gys[0] * (self.y > 0)


def relu(x):
    return _ReLU()(x)


# --------------------------------------------------------------------------- links' math
class _Linear(Function):
    """A2: y = x W^T + b, W (out,in).  One GEMM per call (Chainer-on-NumPy)."""

    def forward(self, xs):
        self.x, self.W = xs[0], xs[1]
        y = self.x.dot(self.W.T)
        if len(xs) == 3:
            y += xs[2]
        return y

    def backward(self, gys):
        gy = gys[0]
        gx = gy.dot(self.W)
        gW = gy.T.dot(self.x)
        if len(self.inputs) == 3:
            return gx, gW, gy.sum(axis=0)
        return gx, gW


def linear(x, W, b=None):
    return _Linear()(x, W) if b is None else _Linear()(x, W, b)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


class _LSTM(Function):
    """A1: F.lstm(c_prev, z).  Gate k of unit j sits at column 4j+k, k = a,i,f,o."""

    def forward(self, xs):
        c_prev, z = xs
        B = z.shape[0]
        r = z.reshape(B, z.shape[1] // 4, 4)
        self.a = np.tanh(r[:, :, 0])
        self.i = _sigmoid(r[:, :, 1])
        self.f = _sigmoid(r[:, :, 2])
        self.o = _sigmoid(r[:, :, 3])
        self.c_prev = c_prev
        c = self.a * self.i + self.f * c_prev
        self.tc = np.tanh(c)
        return c, self.o * self.tc

    def backward(self, gys):
        gc, gh = gys
        a, i, f, o, tc = self.a, self.i, self.f, self.o, self.tc
        if gh is None:
            gh = np.zeros_like(tc)
        if gc is None:
            gc = np.zeros_like(tc)
        dc = gh * o * (1 - tc * tc) + gc
        gz = np.empty(a.shape + (4,), dtype=a.dtype)
        gz[:, :, 0] = dc * i * (1 - a * a)
        gz[:, :, 1] = dc * a * i * (1 - i)
        gz[:, :, 2] = dc * self.c_prev * f * (1 - f)
        gz[:, :, 3] = gh * tc * o * (1 - o)
        return dc * f, gz.reshape(a.shape[0], -1)


def lstm(c_prev, z):
    return _LSTM()(c_prev, z)


class _Dropout(Function):
    """A5: train-mode dropout, y = x * mask / (1 - ratio); mask is injected (scaled)."""

    def __init__(self, scaled_mask):
        self.m = scaled_mask

    def forward(self, xs):
        return xs[0] * self.m

    def backward(self, gys):
        return gys[0] * self.m


def dropout(x, ratio, mask_source=None, tag=None, train=True):
    """mask_source(shape, ratio, tag) -> scaled mask (0 or 1/(1-ratio)).  ratio==0 or eval: identity."""
    if not train or ratio == 0:
        return as_variable(x)
    x = as_variable(x)
    m = mask_source(x.shape, ratio, tag).astype(x.dtype, copy=False)
    return _Dropout(m)(x)


class _EmbedID(Function):
    """A9: row gather; backward = scatter-add into a dense (V,E) gradient."""

    def __init__(self, ids):
        self.ids = np.asarray(ids)

    def forward(self, xs):
        self.W_shape, self.W_dtype = xs[0].shape, xs[0].dtype
        return xs[0][self.ids]

    def backward(self, gys):
        gW = np.zeros(self.W_shape, dtype=self.W_dtype)
        np.add.at(gW, self.ids, gys[0])
        return gW


def embed_id(ids, W):
    ids = ids.data if isinstance(ids, Variable) else ids
    return _EmbedID(ids)(W)


class _BatchMatMul(Function):
    """A5: F.batch_matmul(a, b); a 2-D operand (B,n) is treated as (B,n,1)."""

    def forward(self, xs):
        a, b = xs
        self.a_shape, self.b_shape = a.shape, b.shape
        if a.ndim == 2:
            a = a[:, :, None]
        if b.ndim == 2:
            b = b[:, :, None]
        self.a, self.b = a, b
        return np.matmul(a, b)

    def backward(self, gys):
        gy = gys[0]
        ga = np.matmul(gy, self.b.transpose(0, 2, 1)).reshape(self.a_shape)
        gb = np.matmul(self.a.transpose(0, 2, 1), gy).reshape(self.b_shape)
        return ga, gb


def batch_matmul(a, b):
    return _BatchMatMul()(a, b)


class _Softmax(Function):
    def __init__(self, axis):
        self.axis = axis

    def forward(self, xs):
        x = xs[0]
        e = np.exp(x - x.max(axis=self.axis, keepdims=True))
        self.y = e / e.sum(axis=self.axis, keepdims=True)
        return self.y

    def backward(self, gys):
        gx = self.y * gys[0]
        return gx - self.y * gx.sum(axis=self.axis, keepdims=True)


def softmax(x, axis=1):
    """F.softmax default axis=1 (A5) -- over *time* for the (B,T'',1) score tensor (seq2seq.py:351)."""
    return _Softmax(axis)(x)


def log_softmax_np(x):
    m = x.max(axis=1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=1, keepdims=True))


class _SoftmaxCrossEntropy(Function):
    """A6: class-weighted, normalised by the count of t != ignore_label(-1)  (== B here, Q6)."""

    def __init__(self, t, class_weight):
        self.t = np.asarray(t)
        self.w = class_weight

    def forward(self, xs):
        x = xs[0]
        logp = log_softmax_np(x)
        self.y = np.exp(logp)
        rows = np.arange(len(self.t))
        wt = self.w[self.t].astype(x.dtype) if self.w is not None else np.ones(len(self.t), x.dtype)
        self.wt = wt
        self.count = max(int((self.t != -1).sum()), 1)
        return np.asarray(-(logp[rows, self.t] * wt).sum() / self.count, dtype=x.dtype)

    def backward(self, gys):
        rows = np.arange(len(self.t))
        gx = self.y * self.wt[:, None]
        gx[rows, self.t] -= self.wt
        return gx * (gys[0] / self.count)


def softmax_cross_entropy(x, t, class_weight=None):
    t = t.data if isinstance(t, Variable) else t
    return _SoftmaxCrossEntropy(t, class_weight)(x)


def argmax(x, axis=1):
    x = x.data if isinstance(x, Variable) else x
    return x.argmax(axis=axis).astype(np.int32)


# --------------------------------------------------------------------------- conv / BN
def _im2col(x, kh, kw, sy, sx, ph, pw):
    B, C, H, W = x.shape
    oh = (H + 2 * ph - kh) // sy + 1
    ow = (W + 2 * pw - kw) // sx + 1
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    s = xp.strides
    col = np.lib.stride_tricks.as_strided(
        xp, shape=(B, C, kh, kw, oh, ow),
        strides=(s[0], s[1], s[2], s[3], s[2] * sy, s[3] * sx), writeable=False)
    return col, oh, ow


def _col2im(gcol, x_shape, kh, kw, sy, sx, ph, pw):
    B, C, H, W = x_shape
    oh, ow = gcol.shape[4], gcol.shape[5]
    gxp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), dtype=gcol.dtype)
    for i in range(kh):
        for j in range(kw):
            gxp[:, :, i:i + sy * oh:sy, j:j + sx * ow:sx] += gcol[:, :, i, j]
    return gxp[:, :, ph:ph + H, pw:pw + W]


class _Conv2D(Function):
    """A3: NCHW cross-correlation, W (out,in,kh,kw), cover_all=False; bias only when cnn_config.bn is false (seq2seq.py:52-54:
    nobias=self.cnn_bn)."""

    def __init__(self, stride, pad):
        self.sy, self.sx = stride
        self.ph, self.pw = pad

    def forward(self, xs):
        x, W = xs[0], xs[1]
        self.x_shape, self.W = x.shape, W
        kh, kw = W.shape[2], W.shape[3]
        self.col, oh, ow = _im2col(x, kh, kw, self.sy, self.sx, self.ph, self.pw)
        y = np.tensordot(self.col, W, ((1, 2, 3), (1, 2, 3)))      # (B,oh,ow,O): im2col + one sgemm
        if len(xs) == 3:
            y = y + xs[2]
        return np.ascontiguousarray(np.rollaxis(y, 3, 1))

    def backward(self, gys):
        gy = gys[0]
        kh, kw = self.W.shape[2], self.W.shape[3]
        gW = np.tensordot(gy, self.col, ((0, 2, 3), (0, 4, 5)))
        gcol = np.tensordot(self.W, gy, (0, 1))                     # (C,kh,kw,B,oh,ow)
        gcol = np.rollaxis(gcol, 3)
        gx = _col2im(gcol, self.x_shape, kh, kw, self.sy, self.sx, self.ph, self.pw)
        if len(self.inputs) == 3:
            return gx, gW, gy.sum(axis=(0, 2, 3))
        return gx, gW


def convolution_2d(x, W, stride, pad, b=None):
    return _Conv2D(stride, pad)(x, W) if b is None else _Conv2D(stride, pad)(x, W, b)


class _MaxPoolingND(Function):
    """F.max_pooling_nd(x, ksize) on (B,C,H,W) as the OLD path calls it (enc_dec.py:456): stride = ksize, pad = 0, cover_all = True
    (Chainer's default for max pooling: out = ceil(in / k), the last window may be partial and is padded with -inf).  The gradient goes
    to the first maximum of the window in (kh, kw) order, like Chainer's argmax over the flattened window."""

    def __init__(self, ksize):
        self.kh, self.kw = ksize

    def forward(self, xs):
        x = xs[0]
        B, C, H, W = x.shape
        kh, kw = self.kh, self.kw
        oh, ow = -(-H // kh), -(-W // kw)
        xp = np.full((B, C, oh * kh, ow * kw), -np.inf, dtype=x.dtype)
        xp[:, :, :H, :W] = x
        win = xp.reshape(B, C, oh, kh, ow, kw).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, oh, ow, kh * kw)
        self.arg = win.argmax(axis=4)                      # first maximum in (kh, kw) order
        self.x_shape = x.shape
        return np.ascontiguousarray(win.max(axis=4))

    def backward(self, gys):
        gy = gys[0]
        B, C, H, W = self.x_shape
        kh, kw = self.kh, self.kw
        oh, ow = gy.shape[2], gy.shape[3]
        gwin = np.zeros((B, C, oh, ow, kh * kw), dtype=gy.dtype)
        np.put_along_axis(gwin, self.arg[..., None], gy[..., None], axis=4)
        gxp = gwin.reshape(B, C, oh, ow, kh, kw).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, oh * kh, ow * kw)
        return np.ascontiguousarray(gxp[:, :, :H, :W])


def max_pooling_nd(x, ksize):
    return _MaxPoolingND(ksize)(x)


class _BatchNormTrain(Function):
    """A4: statistics over every axis but the channel axis 1 -- (0,2,3) for the CNN's 4-D input, (0,) for the 2-D (B, units) input of
    the linear_proj encoder (seq2seq.py:281) -- biased variance, eps=2e-5."""

    def __init__(self, eps):
        self.eps = eps

    def forward(self, xs):
        x, gamma, beta = xs
        ax = self.ax = (0,) + tuple(range(2, x.ndim))
        self.ex = (None, slice(None)) + (None,) * (x.ndim - 2)
        self.mean = x.mean(axis=ax)
        self.var = x.var(axis=ax)
        self.inv_std = (self.var + x.dtype.type(self.eps)) ** x.dtype.type(-0.5)
        ex = self.ex
        self.x_hat = (x - self.mean[ex]) * self.inv_std[ex]
        self.gamma = gamma
        return gamma[ex] * self.x_hat + beta[ex]

    def backward(self, gys):
        gy = gys[0]
        ax, ex = self.ax, self.ex
        m = gy.size // self.gamma.size
        gbeta = gy.sum(axis=ax)
        ggamma = (gy * self.x_hat).sum(axis=ax)
        gx = (self.gamma * self.inv_std)[ex] * (
            gy - (self.x_hat * ggamma[ex] + gbeta[ex]) / m)
        return gx, ggamma, gbeta


class BatchNormState:
    """Link state of L.BatchNormalization (A4, A10): gamma, beta + running stats, decay 0.9."""

    def __init__(self, gamma, beta, avg_mean, avg_var, eps=2e-5, decay=0.9):
        self.gamma, self.beta = gamma, beta
        self.avg_mean, self.avg_var = avg_mean, avg_var
        self.N = 0
        self.eps, self.decay = eps, decay

    def __call__(self, x, train=True):
        if train:
            f = _BatchNormTrain(self.eps)
            y = f(x, self.gamma, self.beta)
            m = x.data.size // self.gamma.data.size
            adjust = m / max(m - 1.0, 1.0)
            dt = self.avg_mean.dtype.type
            self.avg_mean *= dt(self.decay)
            self.avg_mean += dt(1 - self.decay) * f.mean
            self.avg_var *= dt(self.decay)
            self.avg_var += dt((1 - self.decay) * adjust) * f.var
            self.N += 1
            return y
        ex = (None, slice(None)) + (None,) * (x.ndim - 2)
        inv = (self.avg_var + self.avg_var.dtype.type(self.eps)) ** -0.5
        scale = as_variable((self.gamma.data * inv)[ex])
        shift = as_variable((self.beta.data - self.gamma.data * inv * self.avg_mean)[ex])
        return add(mul(x, scale), shift)


# --------------------------------------------------------------------------- L.LayerNormalization
class _LayerNorm(Function):
    """F.layer_normalization(x, gamma, beta, eps) on (B, units): per-row mean and BIASED variance over the units axis,
    x_hat = (x - mu) / sqrt(var + eps), y = x_hat * gamma + beta.  L.LayerNormalization(size) passes its own eps = 1e-6 (the function's
    default would be 1e-5); gamma initialised to 1, beta to 0.  Used behind every LSTM when rnn_config.ln is set (seq2seq.py:85-87,
    141-143, 200-202)."""

    def __init__(self, eps):
        self.eps = eps

    def forward(self, xs):
        x, gamma, beta = xs
        mu = x.mean(axis=1, keepdims=True)
        xm = x - mu
        var = (xm * xm).mean(axis=1, keepdims=True)
        self.inv_std = 1.0 / np.sqrt(var + x.dtype.type(self.eps))
        self.x_hat = xm * self.inv_std
        self.gamma = gamma
        return self.x_hat * gamma[None, :] + beta[None, :]

    def backward(self, gys):
        gy = gys[0]
        gbeta = gy.sum(axis=0)
        ggamma = (gy * self.x_hat).sum(axis=0)
        g = gy * self.gamma[None, :]
        n = gy.shape[1]
        gx = self.inv_std * (g - g.mean(axis=1, keepdims=True) - self.x_hat * (g * self.x_hat).sum(axis=1, keepdims=True) / n)
        return gx, ggamma, gbeta


def layer_normalization(x, gamma, beta, eps=1e-6):
    return _LayerNorm(eps)(x, gamma, beta)


# --------------------------------------------------------------------------- L.LSTM link state
class LSTMLink:
    """A1: L.LSTM(in,out) call protocol: upward(x) [+ lateral(h) when h is not None]; c=None -> zeros."""

    def __init__(self, Wu, bu, Wl):
        self.Wu, self.bu, self.Wl = Wu, bu, Wl
        self.h = None
        self.c = None

    def reset_state(self):
        self.h = None
        self.c = None

    def set_state(self, c, h):
        self.c, self.h = c, h

    def __call__(self, x):
        z = linear(x, self.Wu, self.bu)
        if self.h is not None:
            z = add(z, linear(self.h, self.Wl))
        if self.c is None:
            self.c = Variable(np.zeros((len(x), self.Wl.shape[1]), dtype=x.dtype))
        self.c, y = lstm(self.c, z)
        self.h = y
        return y
