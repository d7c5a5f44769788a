"""One train step of bench.py's model with ASTK_GEMM_LOG=1: every GEMM launch and absolute-maximum region of the step (stderr)."""
import os, sys, copy, random
os.environ["ASTK_GEMM_LOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ast_amd import optimizers as O
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
cfg = copy.deepcopy(bench.MODEL_CFG)
B, T, D, L, V = 32, 800, 80, 40, 1098
m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
opt = O.Adam(alpha=1e-3, amsgrad=True).setup(m); opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
Xh, yh = bench.synth_batch(B, T, D, L, V, 20)
X, y = torch.from_numpy(Xh).cuda(), torch.from_numpy(yh).cuda()
random.seed(1)
with using_config("train", True):
    sys.stderr.write("=== forward\n")
    loss = m.forward_loss(X=X, y=y, teach_ratio=0.8, add_noise=0.25)
    m.cleargrads()
    sys.stderr.write("=== backward\n")
    loss.backward()
    opt.update()
torch.cuda.synchronize()
print("loss", float(loss.data))
