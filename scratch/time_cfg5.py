"""One fp32 train step at the shape of BASELINE configs[4] (6-layer BiLSTM, hidden 1024 = 512 per direction, BPE-8k): does the path run there?"""
import copy, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from ast_amd import optimizers as O
from oracle.ast_ref import synth_batch
cfg = copy.deepcopy(bench.MODEL_CFG)
cfg["rnn_config"].update(enc_layers=6, hidden_units=1024, attn_units=1024, dec_vocab_size=8004, dec_layers=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
B, T, D, L, V = 32, 800, 80, 40, 8004
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
for _ in range(2): l = step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"params {m.arena.size/1e6:.1f} M; {dt*1e3:.1f} ms/step, {B*T/dt:.0f} frames/s, loss {float(l.data):.3f}, grad norm {opt.last_grad_norm:.3f}")
