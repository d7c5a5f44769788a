"""Feasibility: do two half-batch train steps on two HIP streams (latency-bound recurrences of one beside the GEMMs of the other)
finish sooner than one full-batch step?  Two independent model replicas (B/2 each), no coupling (per-replica BatchNorm, own gradients)."""
import copy, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from ast_amd import optimizers as O
from oracle.ast_ref import synth_batch
cfg = copy.deepcopy(bench.MODEL_CFG)
T, D, L, V = 800, 80, 40, cfg["rnn_config"]["dec_vocab_size"]

def make(B, seed):
    m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
    opt = O.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(m)
    opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
    X, y = synth_batch(B, T, D, L, V, seed)
    return m, opt, torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()

def step(m, opt, X, y):
    with using_config("train", True):
        l = m.forward_loss(X=X, y=y, teach_ratio=0.8, random_out=0, add_noise=0.25)
        m.cleargrads(); l.backward(); opt.update()

def timed(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

random.seed(1)
full = make(32, 20)
print(f"one stream, B=32: {timed(lambda: step(*full)):.2f} ms/step")
half = make(16, 21)
print(f"one stream, B=16: {timed(lambda: step(*half)):.2f} ms/step")
reps = [make(16, 21), make(16, 22)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def both():
    for r, s in zip(reps, streams):
        with torch.cuda.stream(s):
            step(*r)
print(f"two streams, 2 x B=16: {timed(both):.2f} ms per pair of steps (= 32 rows)")
