"""Data loaders with the reference's batch contract (dataloader.py:111-164, 244-297):
  * utterances are bucketed by frame count, shuffled inside each bucket, cut into batches of `batch_size`, and the
    batch list is shuffled -- all with Python's `random` (the stream nn.py:54 seeds);
  * speech is truncated to (buckets_num+1)*buckets_width frames; train sets get int(zero_input*T_u) randomly chosen
    frames (with replacement) zeroed; targets are [GO] + ids[:max_pred-2] + [EOS];
  * X (B,T_max,D) float32 and y (B,L_max) int32 are zero-padded (PAD_ID = 0) and live on the GPU.
Batches are staged through pinned host memory (one H2D copy each).  The SyntheticDataLoader produces the
SURVEY.md 8(d) batches used by bench.py and the tests (the reference's feature blobs are not distributable)."""
import os
import pickle
import random

import numpy as np
import torch

from . import prep_buckets


class SYMBOLS:
    PAD = b"_PAD"
    GO = b"_GO"
    EOS = b"_EOS"
    UNK = b"_UNK"
    START_VOCAB = [PAD, GO, EOS, UNK]
    PAD_ID = 0
    GO_ID = 1
    EOS_ID = 2
    UNK_ID = 3


def _pad_into(host, arrays):
    """Rows of `host` (a CPU tensor, possibly pinned) = the arrays, zero-padded.  Through the NumPy view: one memcpy per row
    (torch's indexed assignment spends milliseconds per row on small copies)."""
    hn = host.numpy()
    for i, a in enumerate(arrays):
        a = np.asarray(a)
        hn[i, :len(a)] = a
        hn[i, len(a):] = 0
    return host


def pad_batch(arrays, dtype, device, n_min=0):
    """F.pad_sequence(xs, padding=0) followed by to_gpu (dataloader.py:156-162); n_min: pad at least to this length."""
    n = max(max(len(a) for a in arrays), n_min)
    shape = (len(arrays), n) + tuple(arrays[0].shape[1:])
    host = torch.zeros(shape, dtype=dtype)
    if device.type == "cuda":
        host = host.pin_memory()
    return _pad_into(host, arrays).to(device, non_blocking=True)


class _PinnedRing:
    """Staging for batches that start in host memory: a few slots of REUSED pinned buffers (page-locking 8 MB per batch costs more
    than the train step) filled by a helper thread one batch ahead of the consumer; a slot is refilled only after the H2D copies
    that read it have executed."""

    def __init__(self, device, depth=3):
        self.device, self.depth = device, depth
        self.slots = [{"bufs": {}, "event": None} for _ in range(depth)]

    def host(self, slot, name, shape, dtype):
        n = int(np.prod(shape))
        buf = self.slots[slot]["bufs"].get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty(max(n, 1), dtype=dtype).pin_memory()
            self.slots[slot]["bufs"][name] = buf
        return buf[:n].view(shape)

    def wait_free(self, slot):
        ev = self.slots[slot]["event"]
        if ev is not None:
            ev.synchronize()

    def mark_in_flight(self, slot):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.slots[slot]["event"] = ev


class DataLoader:
    def __init__(self, data_cfg, model_dir, gpuid):
        self.gpuid = gpuid
        self.data_cfg = data_cfg
        self.model_dir = model_dir
        self.device = torch.device(f"cuda:{gpuid}") if gpuid is not None and gpuid >= 0 and torch.cuda.is_available() else torch.device("cpu")
        self.map, self.vocab, self.info = {}, {}, {}
        self.buckets = {}
        self.n_utts = {}
        # data-parallel sharding of every batch (ast_amd.dist): rank r takes rows r::world of each bucketed batch
        self.rank, self.world = 0, 1

    def _count(self):
        self.n_utts = {k: sum(len(b) for b in v["buckets"]) for k, v in self.buckets.items()}

    def _drop_frames(self, x, rate):
        if getattr(self, "_zero_on_device", False):      # the device path zeroes the frames of the uploaded batch (astk_zero_frames)
            return x
        n = int(rate * len(x))
        if n <= 0:
            return x
        mask = np.ones(len(x), dtype=np.float32)
        mask[np.random.choice(np.arange(len(x)), size=n)] = 0        # with replacement, unseeded (quirk Q7)
        return x * mask[:, None]

    def _speech(self, utt, set_key, max_sp):
        raise NotImplementedError

    def _targets(self, utt, set_key):
        dec_key, max_pred = self.data_cfg["dec_key"], self.data_cfg["max_pred"]
        w2i = self.vocab[dec_key]["w2i"]
        ids = [w2i.get(w, SYMBOLS.UNK_ID) for w in self.map[set_key][utt][dec_key]]
        return np.asarray([SYMBOLS.GO_ID] + ids[:max_pred - 2] + [SYMBOLS.EOS_ID], dtype=np.int32)

    def batch_plan(self, batch_size, set_key):
        """The shuffled list of (utterances, bucket width) -- consumes the seeded `random` stream exactly like the reference.
        `batch_size` may also be the OLD path's dict {'max', 'med', 'min'} (+ optional 'curriculum'), nmt_run.py:406-447 create_batches:
        the first third of the buckets (short utterances) is cut into batches of 'max', the second third of 'med', the rest of 'min';
        the BUCKET order is shuffled first (one more `random.shuffle`), or kept ascending for a curriculum, in which case the batch
        list is not shuffled either."""
        bk = self.buckets[set_key]
        plan = []
        if isinstance(batch_size, dict):
            num_b = bk["num_b"]
            order = list(range(num_b))
            curriculum = bool(batch_size.get("curriculum", False))
            if not curriculum:
                random.shuffle(order)
            for b in order:
                size = int(batch_size["max"] if b < num_b // 3 else batch_size["med"] if b < (num_b * 2) // 3 else batch_size["min"])
                bucket = bk["buckets"][b]
                random.shuffle(bucket)
                for i in range(0, len(bucket), size):
                    plan.append((bucket[i:i + size], (b + 1) * bk["width_b"]))
            if not curriculum:
                random.shuffle(plan)
            return plan
        for b, bucket in enumerate(bk["buckets"]):
            random.shuffle(bucket)
            for i in range(0, len(bucket), batch_size):
                plan.append((bucket[i:i + batch_size], (b + 1) * bk["width_b"]))
        random.shuffle(plan)
        return plan

    def get_batch(self, batch_size, set_key, train, labels=False):
        bk = self.buckets[set_key]
        max_sp = (bk["num_b"] + 1) * bk["width_b"]
        plan = []
        world = self.world if train else 1      # evaluation batches are not sharded: rank 0 decodes the whole set (train.py:55-60)
        # Under data parallelism only rank 0 evaluates, and batch_plan() consumes the seeded `random` stream that every rank's training
        # plan and teacher-forcing coins come from: an evaluation pass must leave that stream where it found it, or rank 0 would
        # build different shards and flags from the next epoch on.  (One process: the stream is consumed exactly like the reference.)
        rng_state = random.getstate() if (self.world > 1 and not train) else None
        if rng_state is not None:
            kept = [list(b) for b in bk["buckets"]]          # batch_plan shuffles the bucket lists in place
        base_plan = self.batch_plan(batch_size, set_key)
        if rng_state is not None:
            random.setstate(rng_state)
            bk["buckets"][:] = kept
        for utts, _ in base_plan:
            if world > 1:
                # equal shards on every rank, and the same number of steps: a bucket's last batch is cut to a multiple of the world
                # size (at most world-1 utterances per bucket sit out the epoch; a rank with an empty shard would skip a step
                # the others take, and their all-reduce would wait for it forever)
                utts = utts[:len(utts) // world * world]
            mine = list(utts[self.rank::world]) if world > 1 else list(utts)
            if not mine:
                continue
            pads = None
            if world > 1:
                # Every replica pads its shard to the extents of the WHOLE batch: equal (T, L) on all ranks means equal work, the
                # row count global-batch BatchNorm assumes, and -- essential -- the same number of teacher-forcing coins drawn
                # from the seeded `random` stream on every rank (one per decoder step), so that the ranks keep shuffling alike.
                t_pad = max(min(int(self.info[set_key][u]["sp"]), max_sp) for u in utts)
                l_pad = max(len(self._targets(u, set_key)) for u in utts) if labels else 0
                pads = (t_pad, l_pad)
                if labels and train:
                    # the targets of the WHOLE batch in unsharded row order (global row b * world + r = row b of rank r): forward_loss's
                    # random_out draws run over all of them on every rank, and take them from here instead of an all-gather per step
                    yg = np.zeros((len(utts), l_pad), dtype=np.int32)
                    for g_, u in enumerate(utts):
                        t_ = self._targets(u, set_key)
                        yg[g_, :len(t_)] = np.asarray(t_, dtype=np.int32)
                    pads = (t_pad, l_pad, yg)
            plan.append((mine, pads))

        def padded(arrays, n_min):
            return max(max(len(a) for a in arrays), n_min)
        if self.device.type != "cuda":
            self._zero_on_device = False
            for utts, pads in plan:
                xs = [self._speech(u, set_key, max_sp) for u in utts]
                out = {"X": pad_batch(xs, torch.float32, self.device, pads[0] if pads else 0), "utts": utts}
                if labels:
                    out["y"] = pad_batch([self._targets(u, set_key) for u in utts], torch.int32, self.device, pads[1] if pads else 0)
                if pads and len(pads) > 2:
                    out["y_global"] = pads[2]
                yield out
            return
        # device batches: loading and padding of batch k+1 run on a helper thread while batch k trains; frame zeroing
        # (dataloader.py:83-93) runs on the uploaded batch (astk_zero_frames: one launch instead of a NumPy pass per utterance)
        from concurrent.futures import ThreadPoolExecutor
        ring = self.__dict__.setdefault("_ring", _PinnedRing(self.device))
        zero_rate = float(self.data_cfg.get("zero_input", 0) or 0) if "train" in set_key else 0.0
        self._zero_on_device = zero_rate > 0 and os.environ.get("ASTK_ZERO_FRAMES_ON_HOST", "0") != "1"

        def stage(k):
            (utts, pads), slot = plan[k], k % ring.depth
            ring.wait_free(slot)
            xs = [self._speech(u, set_key, max_sp) for u in utts]
            n = padded(xs, pads[0] if pads else 0)
            host = {"X": _pad_into(ring.host(slot, "X", (len(xs), n) + tuple(xs[0].shape[1:]), torch.float32), xs)}
            if self._zero_on_device:
                lens = ring.host(slot, "len", (len(xs),), torch.int32)
                lens.copy_(torch.tensor([len(x) for x in xs], dtype=torch.int32))
                host["len"] = lens
            if labels:
                ys = [self._targets(u, set_key) for u in utts]
                host["y"] = _pad_into(ring.host(slot, "y", (len(ys), padded(ys, pads[1] if pads else 0)), torch.int32), ys)
            return host
        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(stage, 0) if plan else None
            for k, (utts, _) in enumerate(plan):
                host = nxt.result()
                nxt = pool.submit(stage, k + 1) if k + 1 < len(plan) else None
                out = {name: h.to(self.device, non_blocking=True) for name, h in host.items()}
                if "len" in out:
                    import ctypes as C
                    from . import _lib
                    X = out["X"]
                    seed = getattr(self, "zero_seed", 0x2E50F4A3E5)
                    off = self.__dict__.get("_zero_counter", 0)
                    self._zero_counter = off + X.shape[0] * X.shape[1]
                    _lib.check(_lib.load().astk_zero_frames(C.c_void_p(X.data_ptr()), X.shape[0], X.shape[1], int(X[0, 0].numel()),
                                                            C.c_void_p(out["len"].data_ptr()), zero_rate, seed, off,
                                                            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
                    del out["len"]
                ring.mark_in_flight(k % ring.depth)
                out["utts"] = utts
                if plan[k][1] and len(plan[k][1]) > 2:
                    out["y_global"] = plan[k][1][2]
                yield out

    def get_hyps(self, preds):
        dec_key = self.data_cfg["dec_key"]
        join = " " if dec_key.endswith("_w") else ""
        i2w = self.vocab[dec_key]["i2w"]
        hyps = {}
        for utt, p in preds:
            hyps[utt] = []
            if isinstance(p, list):
                s = join.join(i2w[i].decode() for i in p if i >= len(SYMBOLS.START_VOCAB))
                if "bpe_w" in dec_key:
                    s = s.replace("@@ ", "")
                hyps[utt].extend(s.strip().split())
        return hyps


class _PickledLoader(DataLoader):
    def __init__(self, data_cfg, model_dir, gpuid):
        super().__init__(data_cfg, model_dir, gpuid)
        print("Loading data dictionaries")
        for attr, key in (("map", "map_path"), ("vocab", "vocab_path"), ("info", "info_path")):
            with open(data_cfg[key], "rb") as f:
                setattr(self, attr, pickle.load(f))
        print("Organising data into buckets")
        self.buckets = prep_buckets.buckets_main(self.model_dir, data_cfg["buckets_num"], data_cfg["buckets_width"], key="sp",
                                                 scale=data_cfg["train_scale"], seed="haha", info_path=data_cfg["info_path"])
        self._count()


class FisherDataLoader(_PickledLoader):
    """Per-utterance <speech_path>/<set>/[<prefix>/]<utt>.npy float32 (T,D) files (dataloader.py:95-108)."""

    def _speech(self, utt, set_key, max_sp):
        base = os.path.join(self.data_cfg["speech_path"], set_key)
        path = os.path.join(base, "{0:s}.npy".format(utt))
        if not os.path.exists(path):
            path = os.path.join(base, utt.split("_", 1)[0], "{0:s}.npy".format(utt))
        x = np.load(path)[:max_sp]
        if "train" in set_key and self.data_cfg["zero_input"] > 0:
            x = self._drop_frames(x, self.data_cfg["zero_input"])
        return x


class GlobalPhoneDataLoader(_PickledLoader):
    """One pickled {set: {utt: ndarray}} blob (dataloader.py:185-297)."""

    def __init__(self, data_cfg, model_dir, gpuid):
        super().__init__(data_cfg, model_dir, gpuid)
        print("loading speech data from: {0:s}".format(data_cfg["speech_path"]))
        with open(data_cfg["speech_path"], "rb") as f:
            self.speech_data = pickle.load(f)

    def _speech(self, utt, set_key, max_sp):
        x = np.asarray(self.speech_data[set_key][utt][:max_sp])
        if "train" in set_key and self.data_cfg["zero_input"] > 0:
            x = self._drop_frames(x, self.data_cfg["zero_input"])
        return x


def synth_utterances(n_utts, feat_dim, vocab_size, frames_lo, frames_hi, tgt_lo, tgt_hi, seed=20):
    """Synthetic corpus with the reference's schemas: info {utt: {'sp': frames}}, speech {utt: (T,D) f32}, ids."""
    rng = np.random.default_rng(seed)
    info, speech, ids = {}, {}, {}
    for i in range(n_utts):
        u = "utt_{0:06d}".format(i)
        t = int(rng.integers(frames_lo, frames_hi + 1))
        info[u] = {"sp": t}
        speech[u] = rng.standard_normal((t, feat_dim)).astype(np.float32)
        ids[u] = rng.integers(4, vocab_size, size=int(rng.integers(tgt_lo, tgt_hi + 1))).astype(np.int32)
    return info, speech, ids


class SyntheticDataLoader(DataLoader):
    """data: {"dataloader": "synthetic", "vocab_size", "feat_dim", "n_utts": {set: n}, "frames": [lo,hi],
    "targets": [lo,hi], buckets_num, buckets_width, max_pred, zero_input, train_scale}."""

    def __init__(self, data_cfg, model_dir, gpuid):
        super().__init__(data_cfg, model_dir, gpuid)
        self.speech, self.ids, info = {}, {}, {}
        for k, (set_key, n) in enumerate(sorted(data_cfg["n_utts"].items())):
            info[set_key], self.speech[set_key], self.ids[set_key] = synth_utterances(
                n, data_cfg["feat_dim"], data_cfg["vocab_size"], data_cfg["frames"][0], data_cfg["frames"][1],
                data_cfg["targets"][0], data_cfg["targets"][1], seed=20 + k)
        self.info = info
        self.buckets = prep_buckets.buckets_from_info(info, data_cfg["buckets_num"], data_cfg["buckets_width"], "sp",
                                                      data_cfg.get("train_scale", 1), "haha")
        self._count()

    def _speech(self, utt, set_key, max_sp):
        x = self.speech[set_key][utt][:max_sp]
        if "train" in set_key and self.data_cfg.get("zero_input", 0) > 0:
            x = self._drop_frames(x, self.data_cfg["zero_input"])
        return x

    def _targets(self, utt, set_key):
        ids = list(self.ids[set_key][utt])
        return np.asarray([SYMBOLS.GO_ID] + ids[:self.data_cfg["max_pred"] - 2] + [SYMBOLS.EOS_ID], dtype=np.int32)

    def get_hyps(self, preds):
        return {utt: [str(i) for i in p if i >= 4] for utt, p in preds}
