"""How far ahead of the GPU does the host run?  Times the enqueue of N train steps (no sync) and the total until the GPU drains."""
import copy, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from ast_amd import optimizers as O
from oracle.ast_ref import synth_batch
cfg = copy.deepcopy(bench.MODEL_CFG)
B, T, D, L, V = 32, 800, 80, 40, cfg["rnn_config"]["dec_vocab_size"]
m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
opt = O.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(m)
opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
X, y = synth_batch(B, T, D, L, V, 20)
X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
random.seed("seed-ast-20h")
def step():
    with using_config("train", True):
        l = m.forward_loss(X=X, y=y, teach_ratio=0.8, random_out=0, add_noise=0.25)
        m.cleargrads(); l.backward(); opt.update()
for _ in range(5): step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, total {1e3 * (t2 - t0) / N:.2f} ms/step")
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("enqueue of one step into an idle queue:", " ".join(f"{1e3 * t:.2f}" for t in ts), "ms")
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
