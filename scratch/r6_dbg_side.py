"""Debug aid: the side-stream LSTM stack case of tests/test_gpu_ops.py with per-frame errors.  usage: r6_dbg_side.py T B in h nl [knob=value ...]"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ast_amd import _lib
from ast_amd._lib import LstmGrads, LstmParams, LstmStackDesc
from oracle.ast_ref_torch import encoder_torch
import test_gpu_ops as TG
lib = _lib.load()
T, B, in_dim, h, nl = [int(v) for v in sys.argv[1:6]]
_lib.set_tuning("lstm.rows32", 1); _lib.set_tuning("lstm.overlap_chunk", 4)
for kv in sys.argv[6:]:
    _lib.set_tuning(kv.split("=")[0], float(kv.split("=")[1]))
rng = np.random.default_rng(T + B)
P, names = {}, []
for pat in ("L{}_enc", "L{}_rev_enc"):
    n_in = in_dim
    for k in range(nl):
        n = pat.format(k); names.append(n)
        P[n + "/upward/W"] = rng.standard_normal((4 * h, n_in)) / np.sqrt(n_in)
        P[n + "/upward/b"] = rng.standard_normal(4 * h) * 0.3
        P[n + "/lateral/W"] = rng.standard_normal((4 * h, h)) / np.sqrt(h); n_in = h
x = rng.standard_normal((T, B, in_dim))
Pt = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
xt = torch.tensor(x, requires_grad=True)
enc, cT, hT = encoder_torch({"rnn_config": {"enc_layers": nl}}, Pt, xt, None)
g_enc, g_c, g_h = rng.standard_normal(enc.shape), rng.standard_normal(cT.shape), rng.standard_normal(hT.shape)
(enc * torch.tensor(g_enc)).sum().add((cT * torch.tensor(g_c)).sum()).add((hT * torch.tensor(g_h)).sum()).backward()
d = LstmStackDesc(T, B, in_dim, h, nl, 2)
main = torch.cuda.Stream(); side = TG._concurrent_stream(lib, main)
d.side_stream = side.cuda_stream
dev, vp = TG.dev, TG.vp
prm = {k: dev(v) for k, v in P.items()}; grd = {k: torch.zeros_like(v) for k, v in prm.items()}
lp, lg = (LstmParams * (2 * nl))(), (LstmGrads * (2 * nl))()
for i, n in enumerate(names):
    lp[i].Wu, lp[i].b, lp[i].Wl = (prm[n + s].data_ptr() for s in ("/upward/W", "/upward/b", "/lateral/W"))
    lg[i].dWu, lg[i].db, lg[i].dWl = (grd[n + s].data_ptr() for s in ("/upward/W", "/upward/b", "/lateral/W"))
nbytes = lib.astk_lstm_stack_workspace_bytes(C.byref(d))
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
xd = dev(x)
enc_d = torch.zeros(B, T, 2 * h, device="cuda"); cT_d, hT_d = torch.zeros(2, nl, B, h, device="cuda"), torch.zeros(2, nl, B, h, device="cuda")
st = C.c_void_p(main.cuda_stream)
torch.cuda.synchronize()
TG.ok(lib, lib.astk_lstm_stack_fwd(C.byref(d), lp, vp(xd), None, vp(enc_d), vp(cT_d), vp(hT_d), vp(ws), nbytes, st))
torch.cuda.synchronize()
print("enc err", float((enc_d.cpu().double() - enc.detach()).abs().max()))
dx = torch.zeros(T, B, in_dim, device="cuda")
ge_d, gc_d, gh_d = dev(g_enc), dev(g_c), dev(g_h)
TG.ok(lib, lib.astk_lstm_stack_bwd(C.byref(d), lp, lg, vp(xd), None, vp(ge_d), vp(gc_d), vp(gh_d), vp(dx), vp(ws), nbytes, st))
torch.cuda.synchronize()
e = (dx.cpu().double() - xt.grad).abs()
print("dx err per frame", [round(float(e[t].max()), 4) for t in range(T)], "scale", float(xt.grad.abs().max()))
for k in P:
    ref_g = Pt[k].grad if Pt[k].grad is not None else torch.zeros_like(Pt[k])
    print(k, "err", float((grd[k].cpu().double() - ref_g).abs().max()), "scale", float(ref_g.abs().max()))
m = C.c_uint(0); lib.astk_persist_status(C.byref(m), 1); print("status", m.value)
