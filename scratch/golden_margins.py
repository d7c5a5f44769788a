"""How close the HIP path sits to the full-size golden bounds: python scratch/golden_margins.py CASE -- prints the five largest
err / bound ratios of the sampled gradient entries and of the tensor norms (run it with ASTK_CONV0_DIRECT=0 / 1 etc. to compare paths)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_golden import FULL, _full_inputs
from ast_amd import optimizers as O
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
case = sys.argv[1] if len(sys.argv) > 1 else "cfg5_wide"
c = FULL[case]
P, X, y = _full_inputs(c)
g = SpeechEncoderDecoder(0, c["cfg"]).materialize(c["D"], values=P)
g.inject["use_truth"] = c["flags"]
with using_config("train", True):
    loss = g.forward_loss(torch.from_numpy(X), torch.from_numpy(y), c["teach_ratio"])
    g.cleargrads()
    loss.backward()
    grads = g.arena.to_numpy(grads=True)
print(case, "loss rel err", abs(float(loss.data) - c["loss"]) / abs(c["loss"]))
nmax = max(v["norm"] for v in c["grads"].values())
amax = max(v["absmax"] for v in c["grads"].values())
re, rn = [], []
for k, v in c["grads"].items():
    got = grads[k].astype(np.float64)
    gn = float(np.sqrt((got ** 2).sum()))
    rn.append((abs(gn - v["norm"]) / (3e-4 * max(v["norm"], 1e-3 * nmax)), k))
    err = np.abs(got.ravel()[v["index"]] - np.asarray(v["value"])).max()
    re.append((err / (1e-3 * max(v["absmax"], 1e-3 * amax)), k))
print("entries:", [(round(a, 3), b) for a, b in sorted(re, reverse=True)[:6]])
print("norms:  ", [(round(a, 3), b) for a, b in sorted(rn, reverse=True)[:6]])
