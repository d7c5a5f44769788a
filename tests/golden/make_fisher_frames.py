"""Extracts the DATA the bucketing tests need from the reference-held corpus description (run once in the build container):

    python tests/golden/make_fisher_frames.py      # reads /root/reference/data/fisher/{fisher_20h.info,fisher.vocab}

Output tests/golden/fisher_20h_frames.json: per set of fisher_20h.info the 'sp' frame count and the 'en_w' word count of every
utterance (sorted by utterance id; the ids themselves are not needed and not kept), and the number of types of every vocabulary in fisher.vocab
(dec_vocab_size of the shipped es_en_20h experiment = len(vocab['bpe_w']['w2i']), config.py:24).  The pickles are read with
an unpickler that refuses every global (they hold dicts / lists / bytes / ints only): nothing of the reference is imported or
executed."""
import io
import json
import os
import pickle

REF = "/root/reference/data/fisher"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fisher_20h_frames.json")


class NoGlobals(pickle.Unpickler):
    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"global {module}.{name} refused")


def load(path):
    with open(path, "rb") as f:
        return NoGlobals(io.BytesIO(f.read()), encoding="bytes").load()


def s(x):
    return x.decode() if isinstance(x, bytes) else x


info = load(os.path.join(REF, "fisher_20h.info"))
vocab = load(os.path.join(REF, "fisher.vocab"))
out = {"frames": {}, "en_w": {}, "vocab_types": {}}


def field(rec, name):
    return int(rec[name.encode()] if name.encode() in rec else rec[name])


for set_key, utts in info.items():
    out["frames"][s(set_key)] = [field(utts[u], "sp") for u in sorted(utts)]
    # the English word count of every utterance, same order (round 6: bench.py --histogram sizes each bucket's target length L from it;
    # the BPE-1k token counts themselves live in the LDC-licensed text, which the reference does not ship)
    out["en_w"][s(set_key)] = [field(utts[u], "en_w") for u in sorted(utts)]
for k, v in vocab.items():
    w2i = v[b"w2i"] if b"w2i" in v else v["w2i"]
    out["vocab_types"][s(k)] = len(w2i)
json.dump(out, open(OUT, "w"))
print({k: (len(v), sum(v)) for k, v in out["frames"].items()}, out["vocab_types"])
