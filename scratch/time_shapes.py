"""Step time of the benchmark model at ragged batch shapes (real buckets are not multiples of anything)."""
import copy, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from ast_amd import optimizers as O
from oracle.ast_ref import synth_batch
cfg = copy.deepcopy(bench.MODEL_CFG)
D, V = 80, cfg["rnn_config"]["dec_vocab_size"]
m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
opt = O.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(m)
opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
random.seed("seed-ast-20h")
s = torch.cuda.Stream()
for B, T, L in ((32, 160, 12), (32, 400, 25), (32, 800, 40), (32, 799, 40), (32, 763, 27), (31, 800, 40), (17, 800, 40), (32, 1040, 40), (32, 1120, 40), (32, 1200, 40), (32, 1680, 40), (16, 1680, 40)):
    X, y = synth_batch(B, T, D, L, V, 20)
    X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    def step():
        with torch.cuda.stream(s), using_config("train", True):
            l = m.forward_loss(X=X, y=y, teach_ratio=0.8, random_out=0, add_noise=0.25)
            m.cleargrads(); l.backward(); opt.update()
    torch.cuda.synchronize()
    firsts = []
    for _ in range(3):
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); firsts.append((t1 - t0, time.perf_counter() - t0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"B={B} T={T} L={L}: {dt * 1e3:.2f} ms/step = {B * T / dt / 1e6:.2f} M frames/s", flush=True)
