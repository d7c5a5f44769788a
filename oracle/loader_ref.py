"""oracle/loader_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (NumPy) of the reference's batch construction -- what `FisherDataLoader` / `GlobalPhoneDataLoader` hand to the train step:

  preprocessing/prep_buckets.py:41-63   create_buckets        -> create_buckets
  dataloader.py:83-93  (:227-237)       _drop_frames          -> drop_frames
  dataloader.py:95-108 (:239-246)       _load_speech          -> RefLoader.load_speech
  dataloader.py:111-164 (:249-297)      get_batch             -> RefLoader.get_batch
  nmt_run.py:406-447 (OLD path)         create_batches        -> create_batches (per-bucket batch sizes, curriculum order)

with every random draw INJECTABLE: `pyrandom` stands for Python's `random` module (the stream nn.py:54 seeds; bucket and batch shuffles),
`choice(n_frames, n_drop)` for `np.random.choice(np.arange(n_frames), size=n_drop)` (unseeded in the reference, quirk Q7).  File I/O is
replaced by a `speech(utt, set_key)` callable; `F.pad_sequence(..., padding=0)` by NumPy zero padding.  The two loader classes of the
reference differ only in where a (T, D) array comes from, so one restatement serves both.

PARITY UNPINNED like the rest of oracle/ (the reference ships no tests); pinned by hand-derived known answers in tests/test_host.py.
"""
import random as _pyrandom

import numpy as np

PAD_ID, GO_ID, EOS_ID, UNK_ID = 0, 1, 2, 3      # dataloader.py:26-36


def create_buckets(cat_dict, num_b, width_b, key, scale, seed, pyrandom=_pyrandom):
    """prep_buckets.py:41-63: bucket = min(frames // width_b, num_b - 1); train sets down-sampled by `scale` with random.seed(seed) +
    random.sample per bucket."""
    buckets = [[] for _ in range(num_b)]
    for utt in cat_dict:
        buckets[min(cat_dict[utt][key] // width_b, num_b - 1)].append(utt)
    if scale > 1:
        pyrandom.seed(seed)
        for i in range(num_b):
            buckets[i] = pyrandom.sample(buckets[i], int(len(buckets[i]) // scale))
    return {"buckets": buckets, "num_b": num_b, "width_b": width_b}


def create_batches(b_dict, batch_size, curriculum=False, pyrandom=_pyrandom):
    """nmt_run.py:406-447 (the OLD path's batch plan): bucket order shuffled (or ascending for a curriculum); per-bucket batch size
    'max' / 'med' / 'min' by the bucket's third; every bucket shuffled and sliced; the batch list shuffled unless curriculum."""
    num_b = b_dict["num_b"]
    order = list(range(num_b))
    if curriculum:
        order = sorted(order)
    else:
        pyrandom.shuffle(order)
    total, out = 0, []
    for b in order:
        if b < num_b // 3:
            size = int(batch_size["max"])
        elif b < (num_b * 2) // 3:
            size = int(batch_size["med"])
        else:
            size = int(batch_size["min"])
        bucket = b_dict["buckets"][b]
        total += len(bucket)
        pyrandom.shuffle(bucket)
        for i in range(0, len(bucket), size):
            out.append((bucket[i:i + size], b))
    if not curriculum:
        pyrandom.shuffle(out)
    return out, total


def default_choice(n_frames, n_drop):
    return np.random.choice(np.arange(n_frames), size=n_drop)


def drop_frames(x, drop_rate, choice=default_choice):
    """dataloader.py:83-93: int(drop_rate * len(x)) frame indices drawn WITH replacement; those frames are multiplied by 0."""
    n = int(drop_rate * len(x))
    if n <= 0:
        return x
    mask = np.ones(len(x), dtype=np.float32)
    mask[np.asarray(choice(len(x), n))] = 0
    return x * mask[:, None]


class RefLoader:
    """data_cfg keys used: zero_input, dec_key, max_pred.  buckets: {set_key: create_buckets(...)}; vocab[dec_key]['w2i']; map[set_key][utt][dec_key]
    = the target words; speech(utt, set_key) -> float32 (T, D)."""

    def __init__(self, data_cfg, buckets, vocab, utt_map, speech, pyrandom=_pyrandom, choice=default_choice):
        self.data_cfg, self.buckets, self.vocab, self.map, self.speech = data_cfg, buckets, vocab, utt_map, speech
        self.pyrandom, self.choice = pyrandom, choice

    def load_speech(self, utt, set_key, max_sp):
        x = np.asarray(self.speech(utt, set_key))[:max_sp]                       # hard truncation (dataloader.py:103, :241)
        if "train" in set_key and self.data_cfg["zero_input"] > 0:
            x = drop_frames(x, self.data_cfg["zero_input"], self.choice)
        return x

    def get_batch(self, batch_size, set_key, train, labels=False):
        bk = self.buckets[set_key]
        num_b, width_b = bk["num_b"], bk["width_b"]
        max_sp = (num_b + 1) * width_b
        batches = []
        for b, bucket in enumerate(bk["buckets"]):
            self.pyrandom.shuffle(bucket)                                        # in place, like the reference
            for i in range(0, len(bucket), batch_size):
                batches.append((bucket[i:i + batch_size], (b + 1) * width_b))
        self.pyrandom.shuffle(batches)
        for utts, _ in batches:
            xs = [self.load_speech(u, set_key, max_sp) for u in utts]
            out = {"utts": list(utts), "X": _pad(xs, np.float32)}
            if labels:
                dec_key, max_pred = self.data_cfg["dec_key"], self.data_cfg["max_pred"]
                ys = []
                for u in utts:
                    ids = [self.vocab[dec_key]["w2i"].get(w, UNK_ID) for w in self.map[set_key][u][dec_key]]
                    ys.append(np.asarray([GO_ID] + ids[:max_pred - 2] + [EOS_ID], dtype=np.int32))
                out["y"] = _pad(ys, np.int32)
            yield out


def _pad(arrays, dtype):
    """F.pad_sequence(xs, padding=0): (len(xs), max length, ...) zero-padded behind every sequence."""
    n = max(len(a) for a in arrays)
    out = np.zeros((len(arrays), n) + tuple(arrays[0].shape[1:]), dtype=dtype)
    for i, a in enumerate(arrays):
        out[i, :len(a)] = a
    return out
