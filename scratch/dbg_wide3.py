"""debug: two forward calls into two zeroed workspaces in a fresh process; which DecPlan buffers differ?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import test_gpu_ops as T
from ast_amd import _lib
lib = _lib.load()
os.environ["ASTK_DEC_PERSIST"] = "0"
B, L, Tt, H, E, A, V, nl = 32, 4, 263, 1024, 128, 1024, 300, 1
s = T._dec_setup(lib, B, L, Tt, H, E, A, V, nl, False, seed=B + L + 1)
nbytes = lib.astk_decoder_workspace_bytes(C.byref(s["d"]))
dev = T.dev
keep = [dev(s["enc"]), dev(s["c0"]), dev(s["h0"]), dev(s["y"], torch.int32), dev(np.asarray(s["flags"]), torch.int32)]
S, Tp, Vp, XI, CW = L - 1, (Tt + 3) // 4 * 4, (V + 3) // 4 * 4, E + A, 2 * H
names = [("TOK", S * B), ("PRED", S * B), ("X0", S * B * XI), ("G", S * B * 4 * H), ("C", (S + 1) * B * H), ("HR", (S + 1) * B * H), ("HD", S * B * H),
         ("DC0", B * H), ("DC1", B * H), ("WuT", XI * 4 * H), ("WlT", H * 4 * H), ("HDL", 4), ("DLN", B * H), ("DLN2", B * H), ("Q", S * B * H),
         ("ALPHA", S * B * Tp), ("CVH", S * B * CW), ("HT", (S + 1) * B * A), ("LOGITS", S * B * Vp), ("LOSSROWS", S * B)]
offs, off = [], 0
for n, cnt in names:
    off = (off + 255) // 256 * 256
    offs.append((n, off // 4, cnt))
    off += cnt * 4
from oracle.ast_ref_torch import decoder_torch
cfg = {"rnn_config": {"dec_layers": nl, "attn_units": A}}
Pt = {k: torch.tensor(v) for k, v in s["P"].items()}
loss_ref, pred_ref = decoder_torch(cfg, Pt, torch.tensor(s["enc"]), torch.tensor(s["c0"]), torch.tensor(s["h0"]), s["y"], s["flags"], V, None, None)
print("ref", float(loss_ref))
outs = []
for it in range(2):
    ws = torch.zeros(nbytes // 4 + 64, device="cuda")
    loss_d = torch.zeros(1, device="cuda"); pred_d = torch.zeros(S, B, dtype=torch.int32, device="cuda")
    rc = lib.astk_decoder_fwd(C.byref(s["d"]), C.byref(s["dp"]), T.vp(keep[0]), T.vp(keep[1]), T.vp(keep[2]), T.vp(keep[3]),
                              T.vp(keep[4]), None, None, T.vp(loss_d), T.vp(pred_d), T.vp(ws), nbytes, T.stream())
    torch.cuda.synchronize()
    outs.append((float(loss_d), ws.cpu().numpy().copy()))
print("losses", outs[0][0], outs[1][0])
a, b = outs[0][1], outs[1][1]
for n, o, cnt in offs:
    x, y = a[o:o + cnt], b[o:o + cnt]
    bad = np.flatnonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))
    if len(bad):
        print(n, "differs in", len(bad), "of", cnt, "first", bad[:6], "max abs diff", float(np.nanmax(np.abs(x[bad] - y[bad]))), "values", x[bad[:3]], y[bad[:3]])
