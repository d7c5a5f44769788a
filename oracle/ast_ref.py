"""oracle/ast_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (NumPy, float64 or float32) of the reference's encoder-decoder
training hot path, executed the way Chainer-on-NumPy executes it: one Python
iteration per time step and per layer, one GEMM per Linear, the per-step
concat growth kept, conv via im2col + GEMM, a define-by-run reverse pass, and a
per-parameter decay -> clip -> AMSGrad update.

Follows (file:line under /root/reference):
  seq2seq.py:35-156   parameter set / initialisers        -> init_params
  seq2seq.py:158-180  forward_cnn                         -> RefModel.forward_cnn
  seq2seq.py:182-242  reset / feed_rnn / forward_rnn_encode (quirk Q1 kept) -> feed_rnn, forward_rnn_encode
  seq2seq.py:293-314  encode (+ multiplicative speech noise)
  seq2seq.py:318-333  init_decoder_state
  seq2seq.py:336-357  compute_context_vector (unmasked softmax over time, Q2)
  seq2seq.py:361-396  decode_step (input feeding)
  seq2seq.py:399-473  forward_loss (teacher forcing Q4, CE denominator Q6)
  seq2seq.py:475-527  predict (greedy)
  nn.py:81-119        optimizer + hook order              -> RefOptimizer
  nn.py:168-194       train step and reported loss (Q5)   -> train_step

PARITY UNPINNED (see oracle/minichainer.py header and DESIGN.md): no reference
test vectors exist and Chainer is not runnable here.
"""
import random as _pyrandom
import numpy as np

from . import minichainer as F
from .minichainer import Variable, Parameter

PAD_ID, GO_ID, EOS_ID, UNK_ID = 0, 1, 2, 3      # dataloader.py:26-36


# --------------------------------------------------------------------------- shapes
def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1                # A3, cover_all=False


def cnn_out_dims(cnn_layers, T, D):
    """(T'', F', C_last) after the conv stack."""
    c = 1
    for l in cnn_layers:
        kh, kw = l["ksize"]
        sy, sx = l["stride"]
        ph, pw = l["pad"]
        T, D, c = conv_out(T, kh, sy, ph), conv_out(D, kw, sx, pw), l["out_channels"]
    return T, D, c


def pool_window(cnn_config, i, extent, axis):
    """enc_dec.py:444-451 (OLD path, `cnn_pool`): per-layer (time_pool, freq_pool), -1 = the whole extent; absent: no pooling."""
    if "cnn_pool" not in cnn_config:
        return 1
    k = cnn_config["cnn_pool"][i][axis]
    return extent if k == -1 else max(int(k), 1)


# --------------------------------------------------------------------------- parameters
def init_params(cfg, in_dim, vocab_size=None, seed=0, dtype=np.float32):
    """Reference initialisers (A9): conv HeNormal, Linear/LSTM LeCunNormal, forget bias 1,
    EmbedID N(0,1), BN gamma=1 beta=0.  Names follow Chainer's save_npz keys (A10)."""
    rng = np.random.default_rng(seed)
    rc, cc = cfg["rnn_config"], cfg["cnn_config"]
    V = vocab_size if vocab_size is not None else rc["dec_vocab_size"]
    P = {}

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(dtype)

    cin, fdim = 1, in_dim
    for i, l in enumerate(cc["cnn_layers"]):
        kh, kw = l["ksize"]
        co = l["out_channels"]
        P[f"CNN_{i}/W"] = normal((co, cin, kh, kw), np.sqrt(2.0 / (cin * kh * kw)))
        if cc["bn"]:
            P[f"CNN_{i}_bn/gamma"] = np.ones(co, dtype)
            P[f"CNN_{i}_bn/beta"] = np.zeros(co, dtype)
            P[f"CNN_{i}_bn/avg_mean"] = np.zeros(co, dtype)
            P[f"CNN_{i}_bn/avg_var"] = np.ones(co, dtype)
        else:
            P[f"CNN_{i}/b"] = np.zeros(co, dtype)
        fdim = conv_out(fdim, kw, l["stride"][1], l["pad"][1])
        fdim = -(-fdim // pool_window(cc, i, fdim, 1))      # max-pool over frequency, cover_all (enc_dec.py:449-456)
        cin = co
    rnn_in = cin * fdim

    ln = rc.get("ln", False)

    def lstm_params(name, n_in, n_out):
        P[f"{name}/upward/W"] = normal((4 * n_out, n_in), np.sqrt(1.0 / n_in))
        b = np.zeros(4 * n_out, dtype)
        b[2::4] = 1                                 # forget gate entries b[4j+2] (A9)
        P[f"{name}/upward/b"] = b
        P[f"{name}/lateral/W"] = normal((4 * n_out, n_out), np.sqrt(1.0 / n_out))
        if ln:                                      # seq2seq.py:85-87, 141-143: L.LayerNormalization(units), gamma 1 / beta 0
            P[f"{name}_ln/gamma"] = np.ones(n_out, dtype)
            P[f"{name}_ln/beta"] = np.zeros(n_out, dtype)

    Hh = rc["hidden_units"] // 2 if rc["bi_rnn"] else rc["hidden_units"]
    stacks = ["L{}_enc"] + (["L{}_rev_enc"] if rc["bi_rnn"] else [])
    for pat in stacks:
        n_in = rnn_in
        for i in range(rc["enc_layers"]):
            lstm_params(pat.format(i), n_in, Hh)
            # L.LSTM(None, units) takes its input width from the first batch: the layer below's output (units per direction), or --
            # with linear_proj -- the projection of the CONCATENATED states (hidden_units), seq2seq.py:250-286
            n_in = rc["hidden_units"] if rc.get("linear_proj", False) else Hh
    H, E, A = rc["hidden_units"], rc["embedding_units"], rc["attn_units"]
    if rc.get("linear_proj", False):                # seq2seq.py:89-99: Linear(H, H) + BatchNormalization(H) between encoder layers
        for i in range(rc["enc_layers"] - 1):
            P[f"enc_proj{i}/W"] = normal((H, H), np.sqrt(1.0 / H))
            P[f"enc_proj{i}/b"] = np.zeros(H, dtype)
            P[f"enc_proj{i}_bn/gamma"] = np.ones(H, dtype)
            P[f"enc_proj{i}_bn/beta"] = np.zeros(H, dtype)
            P[f"enc_proj{i}_bn/avg_mean"] = np.zeros(H, dtype)
            P[f"enc_proj{i}_bn/avg_var"] = np.ones(H, dtype)
    n_attn = rc.get("n_attn", 1)
    P["attn_Wa/W"] = normal((H, H), np.sqrt(1.0 / H))
    P["attn_Wa/b"] = np.zeros(H, dtype)
    for i in range(1, n_attn):                      # seq2seq.py:114-116
        P[f"attn_Wa{i}/W"] = normal((H, H), np.sqrt(1.0 / H))
        P[f"attn_Wa{i}/b"] = np.zeros(H, dtype)
    P["context/W"] = normal((A, (n_attn + 1) * H), np.sqrt(1.0 / ((n_attn + 1) * H)))
    P["context/b"] = np.zeros(A, dtype)
    P["embed_dec/W"] = normal((V, E), 1.0)
    feed = rc.get("feed_attn", True)
    n_in = E + A if feed else E
    for i in range(rc["dec_layers"]):
        lstm_params(f"L{i}_dec", n_in, H)
        n_in = H
    P["out/W"] = normal((V, A), np.sqrt(1.0 / A))
    P["out/b"] = np.zeros(V, dtype)
    return P


PERSISTENT = ("avg_mean", "avg_var")               # not optimised, saved by save_npz (A10)


def is_trainable(name):
    return not name.endswith(PERSISTENT)


# --------------------------------------------------------------------------- mask sources
class NoMasks:
    def __call__(self, shape, ratio, tag):
        raise RuntimeError("dropout requested but no mask source given")


class RecordingMasks:
    """Draws dropout masks from a seeded generator and records them by tag so the identical
    masks can be injected into the HIP path (Q7: the reference's masks are unseeded)."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.masks = {}

    def __call__(self, shape, ratio, tag):
        keep = self.rng.random(shape) >= ratio      # A5: mask = rand >= ratio
        m = keep.astype(np.float32) / np.float32(1.0 - ratio)
        self.masks[tag] = m
        return m


# --------------------------------------------------------------------------- the model
class RefModel:
    def __init__(self, cfg, params, vocab_size=None):
        self.cfg = cfg
        rc = cfg["rnn_config"]
        self.V = vocab_size if vocab_size is not None else rc["dec_vocab_size"]
        self.p = {k: (Parameter(v, name=k) if is_trainable(k) else v) for k, v in params.items()}
        self.dtype = params["out/W"].dtype
        self.bi = rc["bi_rnn"]
        self.n_enc, self.n_dec = rc["enc_layers"], rc["dec_layers"]
        self.n_attn = rc.get("n_attn", 1)
        self.feed_attn = rc.get("feed_attn", True)
        self.rnn_ln = rc.get("ln", False)                        # seq2seq.py:81
        self.rnn_linear_proj = bool(rc.get("linear_proj", False))   # seq2seq.py:90-92
        self.cnn_layers = cfg["cnn_config"]["cnn_layers"]
        self.cnn_bn = cfg["cnn_config"]["bn"]
        self.bn = {}
        bn_names = [f"CNN_{i}_bn" for i in range(len(self.cnn_layers))] if self.cnn_bn else []
        if self.rnn_linear_proj:
            bn_names += [f"enc_proj{i}_bn" for i in range(self.n_enc - 1)]
        for n in bn_names:
            self.bn[n] = F.BatchNormState(self.p[n + "/gamma"], self.p[n + "/beta"], self.p[n + "/avg_mean"], self.p[n + "/avg_var"])

        def mk(n):
            link = F.LSTMLink(self.p[n + "/upward/W"], self.p[n + "/upward/b"], self.p[n + "/lateral/W"])
            link.ln = (self.p[n + "_ln/gamma"], self.p[n + "_ln/beta"]) if self.rnn_ln else None
            return link
        self.enc = [mk(f"L{i}_enc") for i in range(self.n_enc)]
        self.rev = [mk(f"L{i}_rev_enc") for i in range(self.n_enc)] if self.bi else []
        self.dec = [mk(f"L{i}_dec") for i in range(self.n_dec)]
        w = np.ones(self.V, dtype=self.dtype)       # seq2seq.py:152-156
        w[PAD_ID] = 0
        self.mask_pad_id = w
        self.train = True
        self.masks = NoMasks()
        self.enc_states = None

    # ---- enc_dec.py:587-624 (OLD path): Gaussian weight noise on every LSTM's upward W / b and lateral W and on the decoder embedding
    def add_weight_noise(self, mu, sigma, normal):
        """normal(mu, sigma, shape) stands for xp.random.normal (unseeded in the reference, quirk Q7); returns the draws by name, in the
        reference's order: per LSTM link upward.W, upward.b, lateral.W (encoder, reverse encoder, decoder links), then embed_dec.W."""
        layers = [f"L{i}_enc" for i in range(self.n_enc)] + ([f"L{i}_rev_enc" for i in range(self.n_enc)] if self.bi else [])
        layers += [f"L{i}_dec" for i in range(self.n_dec)]
        draws = {}
        for layer in layers:
            for name in (f"{layer}/upward/W", f"{layer}/upward/b", f"{layer}/lateral/W"):
                draws[name] = normal(mu, sigma, self.p[name].data.shape).astype(self.dtype)
                self.p[name].data += draws[name]
        draws["embed_dec/W"] = normal(mu, sigma, self.p["embed_dec/W"].data.shape).astype(self.dtype)
        self.p["embed_dec/W"].data += draws["embed_dec/W"]
        return draws

    # ---- chainer.Chain-like helpers
    def params(self):
        return [(k, v) for k, v in self.p.items() if isinstance(v, Parameter)]

    def cleargrads(self):
        for _, v in self.params():
            v.grad = None

    # ---- seq2seq.py:158-180
    def forward_cnn(self, X):
        h = F.swapaxes(F.expand_dims(X, 2), 1, 2)               # (B,T,D) -> (B,1,T,D)
        for i, l in enumerate(self.cnn_layers):
            h = F.convolution_2d(h, self.p[f"CNN_{i}/W"], tuple(l["stride"]), tuple(l["pad"]),
                                 None if self.cnn_bn else self.p[f"CNN_{i}/b"])      # nobias=self.cnn_bn (seq2seq.py:52-54)
            if "cnn_pool" in self.cfg["cnn_config"]:              # OLD-path extra, enc_dec.py:444-456: between convolution and BatchNorm
                kt = pool_window(self.cfg["cnn_config"], i, h.shape[-2], 0)
                kf = pool_window(self.cfg["cnn_config"], i, h.shape[-1], 1)
                h = F.max_pooling_nd(h, (kt, kf))
            if self.cnn_bn:
                h = self.bn[f"CNN_{i}_bn"](h, train=self.train)
            h = F.relu(h)
        h = F.swapaxes(h, 1, 2)                                 # (B,T'',C,F')
        h = F.reshape(h, h.shape[:2] + (-1,))                   # feature index c*F'+f  (Q9)
        return F.rollaxis(h, 1)                                 # (T'',B,C*F')

    # ---- seq2seq.py:182-203
    def reset_rnn_state(self):
        for l in self.enc + self.rev + self.dec:
            l.reset_state()

    def feed_rnn(self, x, links, stack, step):
        hs = x
        ratio = self.cfg["dropout"]["rnn"]
        for k, link in enumerate(links):
            hs = F.dropout(link(hs), ratio, self.masks, (stack, k, step), self.train)
            if self.rnn_ln:                                       # seq2seq.py:200-202: LN of the DROPPED output; the link's own h stays raw
                hs = F.layer_normalization(hs, link.ln[0], link.ln[1], 1e-6)
        return hs

    # ---- seq2seq.py:205-242  (Q1: reverse stack reads X[-i]: 0, T''-1, ..., 1)
    def forward_rnn_encode(self, X):
        self.reset_rnn_state()
        n = X.shape[0]
        h_fwd = h_rev = None
        for i in range(n):
            f = F.expand_dims(self.feed_rnn(X[i], self.enc, "enc", i), 0)
            h_fwd = f if h_fwd is None else F.concat((h_fwd, f), axis=0)    # quadratic growth kept
            if self.bi:
                r = F.expand_dims(self.feed_rnn(X[-i], self.rev, "rev", i), 0)
                h_rev = r if h_rev is None else F.concat((h_rev, r), axis=0)
        states = F.concat((h_fwd, F.flipud(h_rev)), axis=2) if self.bi else h_fwd
        self.enc_states = F.swapaxes(states, 0, 1)              # (B,T'',H)

    # ---- seq2seq.py:244-291 (rnn_config.linear_proj).  Kept as written, including: the reverse stack is fed enc_states[-1] -- the LAST
    # frame of the layer's input -- at EVERY step (:256, quirk Q8); no LayerNorm (feed_rnn is not used); the projection's
    # BatchNormalization sees one (B, H) time step per call, so its statistics are over the B rows of that step and its running
    # averages / N advance T'' times per layer; and `enc_states` is only reassigned inside the projection branch, so what the attention
    # reads is the LAST PROJECTION's output (the CNN output itself for a 1-layer encoder) -- the top LSTM layer reaches the loss only
    # through its final (c, h), which seed the decoder.
    def forward_rnn_encode_proj(self, X):
        self.reset_rnn_state()
        n = X.shape[0]
        ratio = self.cfg["dropout"]["rnn"]
        enc_states = X
        for cur in range(self.n_enc):
            h_fwd = h_rev = None
            for i in range(n):
                f = F.expand_dims(F.dropout(self.enc[cur](enc_states[i]), ratio, self.masks, ("enc", cur, i), self.train), 0)
                h_fwd = f if h_fwd is None else F.concat((h_fwd, f), axis=0)
                if self.bi:
                    r = F.expand_dims(F.dropout(self.rev[cur](enc_states[-1]), ratio, self.masks, ("rev", cur, i), self.train), 0)
                    h_rev = r if h_rev is None else F.concat((h_rev, r), axis=0)
            states = F.concat((h_fwd, F.flipud(h_rev)), axis=2) if self.bi else h_fwd
            if cur < self.n_enc - 1:
                nxt = None
                for i in range(n):
                    z = F.linear(states[i], self.p[f"enc_proj{cur}/W"], self.p[f"enc_proj{cur}/b"])
                    hcur = F.expand_dims(F.relu(self.bn[f"enc_proj{cur}_bn"](z, train=self.train)), 0)
                    nxt = hcur if nxt is None else F.concat((nxt, hcur), axis=0)
                enc_states = nxt
        self.enc_states = F.swapaxes(enc_states, 0, 1)

    # ---- seq2seq.py:293-314
    def encode(self, X, add_noise=0, noise=None):
        X = F.as_variable(X)
        if add_noise > 0 and self.train:
            assert noise is not None, "inject the N(1,sigma) tensor (Q7: the reference's draw is unseeded)"
            X = F.mul(X, Variable(noise.astype(self.dtype)))
        if self.rnn_linear_proj:
            self.forward_rnn_encode_proj(self.forward_cnn(X))
        else:
            self.forward_rnn_encode(self.forward_cnn(X))

    # ---- seq2seq.py:318-333
    def init_decoder_state(self):
        if self.bi:
            for e, r, d in zip(self.enc, self.rev, self.dec):
                d.set_state(F.concat((e.c, r.c)), F.concat((e.h, r.h)))
        else:
            for e, d in zip(self.enc, self.dec):
                d.set_state(e.c, e.h)

    # ---- seq2seq.py:336-357
    def compute_context_vector(self, dec_h, wa="attn_Wa"):
        q = F.linear(dec_h, self.p[wa + "/W"], self.p[wa + "/b"])
        scores = F.batch_matmul(self.enc_states, q)              # (B,T'',1)
        alphas = F.softmax(scores)                               # over time, no padding mask (Q2)
        cv = F.squeeze(F.batch_matmul(F.swapaxes(self.enc_states, 2, 1), alphas), 2)
        return cv, alphas

    # ---- seq2seq.py:361-396
    def decode_step(self, word, ht, step=0):
        dr = self.cfg["dropout"]
        emb = F.dropout(F.embed_id(word, self.p["embed_dec/W"]), dr["embed"], self.masks, ("emb", 0, step), self.train)
        rnn_in = F.concat((emb, ht), axis=1) if self.feed_attn else emb
        h = self.feed_rnn(rnn_in, self.dec, "dec", step)
        cv, alphas = self.compute_context_vector(h)
        for k in range(1, self.n_attn):                          # seq2seq.py:381-383: further heads on the SAME h; their alphas are dropped
            new_cv, _ = self.compute_context_vector(h, f"attn_Wa{k}")
            cv = F.concat((cv, new_cv), axis=1)
        ht = F.tanh(F.linear(F.concat((cv, h), axis=1), self.p["context/W"], self.p["context/b"]))
        logits = F.dropout(F.linear(ht, self.p["out/W"], self.p["out/b"]), dr["out"], self.masks, ("out", 0, step), self.train)
        return logits, ht, alphas

    # ---- seq2seq.py:399-473
    def forward_loss(self, X, y, teach_ratio, random_out=0, add_noise=0, noise=None, pyrandom=_pyrandom, randint=None):
        """random_out > 0 (seq2seq.py:456-465): behind each step's decode, every target >= 4 is replaced -- when
        `random.random() > random_out`, the same Python stream as the teacher-forcing coin -- by xp.random.randint(4, V + 1), whose upper
        end is one past the last class (quirk Q8): Chainer's softmax_cross_entropy raises on that id with NumPy and reads out of bounds
        with CuPy.  DEVIATION, the only one in this restatement: the drawn id is clamped to V - 1.  `randint(low, high)` is the
        replacement draw (the reference's is the unseeded global xp RNG, quirk Q7); self.targets records the targets that were scored."""
        y = np.asarray(y.data if isinstance(y, Variable) else y)
        B = X.shape[0]
        self.encode(X, add_noise, noise)
        self.init_decoder_state()
        yT = y.T                                                 # (L,B)
        L = len(yT)
        A = self.cfg["rnn_config"]["attn_units"]
        ht = Variable(np.zeros((B, A), dtype=self.dtype))
        loss = 0
        self.use_truth = []
        self.targets = []
        dec_in = None
        for i in range(L - 1):
            cur, nxt = yT[i], yT[i + 1]
            if 0 < i < L - 2:                                    # Q4: one coin per step, whole batch
                truth = pyrandom.random() < teach_ratio
            else:
                truth = True
            self.use_truth.append(bool(truth))
            if truth:
                dec_in = cur
            logits, ht, _ = self.decode_step(dec_in, ht, step=i)
            dec_in = F.argmax(logits, axis=1)
            target = nxt.copy()
            if random_out > 0:
                for b in range(len(target)):
                    if int(target[b]) >= 4 and pyrandom.random() > random_out:
                        target[b] = min(int(randint(4, self.V + 1)), self.V - 1)
            self.targets.append(target)
            loss = loss + F.softmax_cross_entropy(logits, target, class_weight=self.mask_pad_id)
        return loss

    # ---- seq2seq.py:475-527
    def predict(self, X, start_token=GO_ID, end_token=EOS_ID, stop_limit=10):
        was = self.train
        self.train = False
        try:
            B = X.shape[0]
            self.encode(X)
            self.init_decoder_state()
            A = self.cfg["rnn_config"]["attn_units"]
            ht = Variable(np.zeros((B, A), dtype=self.dtype))
            word = np.full((B,), start_token, dtype=np.int32)
            done = np.zeros(B, dtype=bool)
            rows, npred = [], 0
            while npred < stop_limit:
                logits, ht, _ = self.decode_step(word, ht, step=npred)
                word = F.argmax(logits, axis=1)
                rows.append(word)
                done[word == end_token] = True
                if done.all():
                    break
                npred += 1
            return np.stack(rows, 0).T
        finally:
            self.train = was


# --------------------------------------------------------------------------- optimizer (nn.py:81-119)
    # ---- seq2seq.py:529-568: the state API beam search uses
    def get_encoder_states(self):
        out = {"c": [], "h": []}
        if self.bi:
            for e, r in zip(self.enc, self.rev):
                out["h"].append(F.concat((e.h, r.h)))
                out["c"].append(F.concat((e.c, r.c)))
        else:
            for e, _ in zip(self.enc, self.dec):
                out["h"].append(e.h)
                out["c"].append(e.c)
        return out

    def get_decoder_states(self):
        return {"c": [d.c for d in self.dec], "h": [d.h for d in self.dec]}

    def set_decoder_states(self, rnn_states):
        for i, d in enumerate(self.dec):
            d.set_state(rnn_states["c"][i], rnn_states["h"][i])


# ---- nn.py:235-322: beam search over one utterance (batch of 1), N best kept, K expansions per live hypothesis
def decode_beam(model, X, stop_limit, N, K):
    was = model.train
    model.train = False
    try:
        model.encode(X)
        A = model.cfg["rnn_config"]["attn_units"]
        entry = {"hyp": [GO_ID], "score": 0, "dec_state": model.get_encoder_states(),
                 "attn_v": Variable(np.zeros((1, A), dtype=model.dtype)), "attn_history": []}
        n_best = [entry]
        for _ in range(stop_limit):
            if all(e["hyp"][-1] == EOS_ID for e in n_best):
                break
            cur = []
            for e in n_best:
                if e["hyp"][-1] == EOS_ID:
                    cur.append(e)
                    continue
                model.set_decoder_states(e["dec_state"])
                word = np.full((1,), e["hyp"][-1], dtype=np.int32)
                logits, ht, alphas = model.decode_step(word, e["attn_v"])
                x = np.asarray(logits.data[0], dtype=np.float64)
                logp = x - (np.log(np.exp(x - x.max()).sum()) + x.max())
                top = np.argsort(logp)[-K:]
                state = model.get_decoder_states()
                for pi in top[::-1]:
                    cur.append({"hyp": e["hyp"] + [int(pi)], "score": e["score"] + float(logp[pi]), "dec_state": state, "attn_v": ht,
                                "attn_history": e["attn_history"] + [np.squeeze(alphas.data)]})
            n_best = sorted(cur, reverse=True, key=lambda t: t["score"])[:N]
        return n_best
    finally:
        model.train = was


class RefOptimizer:
    """Adam(alpha, .9, .999, 1e-8, amsgrad=True) or SGD, with hooks in insertion order:
    WeightDecay(l2) -> GradientClipping(grad_clip)  (A7, A8)."""

    def __init__(self, model, opt_cfg):
        self.model, self.cfg = model, opt_cfg
        self.t = 0
        self.state = {}
        self.last_grad_norm = None
        self.frozen = set()
        for l in opt_cfg.get("freeze", []):
            self.frozen |= {k for k, _ in model.params() if k.split("/")[0] == l}

    def update(self):
        c = self.cfg
        plist = [(k, p) for k, p in self.model.params() if k not in self.frozen]
        for _, p in plist:                         # reallocate_cleared_grads
            if p.grad is None:
                p.grad = np.zeros_like(p.data)
        if c["l2"] > 0:
            for _, p in plist:
                p.grad = p.grad + p.dtype.type(c["l2"]) * p.data
        sq = 0.0
        for _, p in plist:
            g = p.grad.ravel()
            sq += float(g.dot(g))
        norm = np.sqrt(sq)
        self.last_grad_norm = norm                 # the "grad norm" parity observable (K29)
        rate = c["grad_clip"] / norm if norm > 0 else np.inf
        if rate < 1:
            for _, p in plist:
                p.grad = p.grad * p.dtype.type(rate)
        self.t += 1
        if c.get("grad_noise_eta", 0) > 0:
            # chainer.optimizer.GradientNoise (A7): runs behind WeightDecay and GradientClipping.  GradientMethod.update calls the
            # 'pre' hooks BEFORE `self.t += 1` and exponential_decay_noise reads the optimizer's t, so sigma^2 = eta / (1 + t)^0.55
            # with t = 0 at the first update (self.t was already incremented above for the Adam step, hence t - 1).  Chainer draws
            # from the unseeded global RNG (Q7); here the draw comes from `noise_rng` so that tests can reproduce it.
            std = np.sqrt(c["grad_noise_eta"] / (1.0 + (self.t - 1)) ** 0.55)
            rng = getattr(self, "noise_rng", None) or np.random.default_rng(0)
            self.noise_rng = rng
            for _, p in plist:
                p.grad = p.grad + (rng.standard_normal(p.grad.shape) * std).astype(p.grad.dtype)
        if c["type"] == 0:
            b1, b2, eps, alpha = 0.9, 0.999, 1e-8, c["lr"]
            lr_t = alpha * np.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
            for k, p in plist:
                st = self.state.setdefault(k, {n: np.zeros_like(p.data) for n in ("m", "v", "vhat")})
                dt = p.dtype.type
                g = p.grad
                st["m"] += dt(1 - b1) * (g - st["m"])
                st["v"] += dt(1 - b2) * (g * g - st["v"])
                np.maximum(st["vhat"], st["v"], out=st["vhat"])
                p.data -= dt(lr_t) * st["m"] / (np.sqrt(st["vhat"]) + dt(eps))
        else:
            for _, p in plist:
                p.data -= p.dtype.type(c["lr"]) * p.grad


def train_step(model, opt, X, y, teach_ratio, add_noise=0, noise=None, pyrandom=_pyrandom, random_out=0, randint=None):
    """nn.py:174-189: forward_loss -> cleargrads -> backward -> update; reports loss/B (Q5)."""
    model.train = True
    loss = model.forward_loss(X, y, teach_ratio, random_out, add_noise, noise, pyrandom, randint)
    model.cleargrads()
    loss.backward()
    opt.update()
    return float(loss.data), float(loss.data) / len(y)


# --------------------------------------------------------------------------- checkpoints (train.py:73-75, nn.py:141-152; A10)
def save_npz(path, model):
    """What chainer.serializers.save_npz(path, model) writes for this Chain: one compressed .npz with '<link>/<param>' keys --
    every parameter in Chainer's own layout, and the BatchNormalization persistents avg_mean / avg_var / N (N = number of
    training-mode calls so far).  LSTM h / c are not persistent."""
    out = {}
    for k, v in model.p.items():
        out[k] = np.asarray(v.data if isinstance(v, Parameter) else v)
    for name, bn in model.bn.items():
        out[name + "/N"] = np.asarray(bn.N, dtype=np.int64)
    with open(path, "wb") as f:
        np.savez_compressed(f, **out)


def load_npz(path, model):
    """chainer.serializers.load_npz(path, model): every parameter / persistent of the model must be in the file (strict=True)."""
    with np.load(path) as z:
        for k, v in model.p.items():
            tgt = v.data if isinstance(v, Parameter) else v
            if k not in z.files:
                raise KeyError(f"{path}: missing {k}")
            tgt[...] = np.asarray(z[k], dtype=tgt.dtype).reshape(tgt.shape)
        for name, bn in model.bn.items():
            bn.N = int(z[name + "/N"])


# --------------------------------------------------------------------------- synthetic batches
def synth_batch(B, T, D, L, V, seed=20, dtype=np.float32):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, T, D)).astype(dtype)
    y = np.zeros((B, L), dtype=np.int32)
    for b in range(B):
        # row 0 has the full padded length so that L really is the batch maximum
        n = L if b == 0 else int(rng.integers(max(L // 2, 3), L + 1))
        y[b, 0] = GO_ID
        y[b, 1:n - 1] = rng.integers(4, V, size=n - 2)
        y[b, n - 1] = EOS_ID
    return X, y


def teacher_flags(L, teach_ratio, pyrandom):
    """Q4 flag sequence for one forward_loss call with targets of padded length L."""
    flags = []
    for i in range(L - 1):
        if 0 < i < L - 2:
            flags.append(pyrandom.random() < teach_ratio)
        else:
            flags.append(True)
    return flags
