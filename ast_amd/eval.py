"""Eval drop-in (eval.py:12-48 of the reference): corpus BLEU-4 of the dev hypotheses against `ref.en<i>` files.

The reference delegates to nltk (`corpus_bleu(..., smoothing_function=SmoothingFunction().method2)`), which is not
installed here; this module restates the published algorithm it calls (Papineni et al. 2002 corpus BLEU; Lin & Och 2004
"add one" smoothing = nltk's method2) so that `train.py`'s per-epoch dev score works offline:
  p_n   = sum over segments of clipped n-gram matches / sum of hypothesis n-gram counts   (micro-average),
  method2: (matches + 1) / (total + 1) for n >= 2,
  BP    = 1 if c > r else exp(1 - r / c), r = sum of the reference lengths closest to each hypothesis (ties: shorter),
  BLEU  = BP * exp(sum_n w_n log p_n); 0 when no unigram matches."""
import math
import os
from collections import Counter


def _ngrams(tokens, n):
    return Counter(tuple(tokens[i:i + n]) for i in range(len(tokens) - n + 1))


def corpus_bleu(list_of_references, hypotheses, weights=(0.25, 0.25, 0.25, 0.25), smooth=True):
    assert len(list_of_references) == len(hypotheses), "one set of references per hypothesis"
    num, den = Counter(), Counter()
    hyp_len = ref_len = 0
    for refs, hyp in zip(list_of_references, hypotheses):
        for n in range(1, len(weights) + 1):
            counts = _ngrams(hyp, n)
            best = Counter()
            for ref in refs:
                for g, c in _ngrams(ref, n).items():
                    if c > best[g]:
                        best[g] = c
            num[n] += sum(min(c, best[g]) for g, c in counts.items())
            den[n] += max(1, sum(counts.values()))
        hyp_len += len(hyp)
        ref_len += min((len(r) for r in refs), key=lambda rl: (abs(rl - len(hyp)), rl))
    if num[1] == 0:
        return 0.0
    logp = 0.0
    for n, w in enumerate(weights, start=1):
        a, b = num[n], den[n]
        if smooth and n > 1:
            a, b = a + 1, b + 1
        if a == 0:
            return 0.0
        logp += w * math.log(a / b)
    bp = 1.0 if hyp_len > ref_len else (0.0 if hyp_len == 0 else math.exp(1.0 - ref_len / hyp_len))
    return bp * math.exp(logp)


class Eval:
    def __init__(self, path, n_evals):
        with open(os.path.join(path, "eval.ids"), "r", encoding="utf-8") as f:
            self.ids = [line.strip() for line in f]
        refs = []
        for i in range(n_evals):
            with open(os.path.join(path, "ref.en{0:d}".format(i)), "r", encoding="utf-8") as f:
                refs.append([line.strip().split() for line in f])
        self.refs = list(zip(*refs))

    def calc_bleu(self, hyps):
        return corpus_bleu(self.refs, [hyps[u] for u in self.ids])

    def write_to_file(self, hyps, fname):
        with open(fname, "w", encoding="utf-8") as f:
            for u in self.ids:
                f.write("{0:s}\n".format(" ".join(hyps[u])))
