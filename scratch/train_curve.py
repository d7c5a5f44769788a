"""Loss curve of 400 updates on a cycle of 4 synthetic batches (bench.py's model and training knobs): used to compare the GEMM operand
schemes (ASTK_GEMM_PREC = default fp16x2 / bf16x3 / f32) over a trajectory where weights and gradients move away from their initial
ranges.  The runs share every random stream; they drift apart like any two float32 evaluations of a chaotic recurrence do, so the
curves are compared, not the digits."""
import copy, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ast_amd import optimizers as O
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
cfg = copy.deepcopy(bench.MODEL_CFG)
B, T, D, L, V = 32, 800, 80, 40, cfg["rnn_config"]["dec_vocab_size"]
model = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
opt = O.Adam(alpha=bench.TRAIN["lr"], beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(model)
opt.add_hook(O.WeightDecay(bench.TRAIN["l2"])); opt.add_hook(O.GradientClipping(bench.TRAIN["grad_clip"]))
random.seed("seed-ast-20h")
batches = []
for k in range(4):
    Xh, yh = bench.synth_batch(B, T, D, L, V, 20 + k)
    batches.append((torch.from_numpy(Xh).cuda(), torch.from_numpy(yh).cuda()))
hist = []
for it in range(400):
    X, y = batches[it % 4]
    with using_config("train", True):
        loss = model.forward_loss(X=X, y=y, teach_ratio=bench.TRAIN["teach_ratio"], random_out=0, add_noise=bench.TRAIN["speech_noise"])
        model.cleargrads(); loss.backward(); opt.update()
    if it % 25 == 24 or it < 3:
        hist.append((it + 1, float(loss.data), opt.last_grad_norm))
print(os.environ.get("ASTK_GEMM_PREC", "fp16x2"), " ".join(f"{i}:{l:.3f}/{g:.2f}" for i, l, g in hist), flush=True)
