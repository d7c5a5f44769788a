"""Times one train step of a model variant (dec_layers / feat dim) on the GPU: python scratch/time_cfg.py DEC_LAYERS FEAT"""
import copy, random, sys, time
sys.path.insert(0, '.')
import torch
import bench
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from ast_amd import optimizers as O
from oracle.ast_ref import synth_batch
nl, D = int(sys.argv[1]), int(sys.argv[2])
cfg = copy.deepcopy(bench.MODEL_CFG)
cfg["rnn_config"]["dec_layers"] = nl
B, T, L, V = 32, 800, 40, cfg["rnn_config"]["dec_vocab_size"]
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
    return l
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): l = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f"dec_layers={nl} D={D}: {dt*1e3:.2f} ms/step, {B*T/dt:.0f} frames/s, loss {float(l.data):.3f}")
