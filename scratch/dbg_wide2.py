"""debug: first-call behaviour of the per-launch decoder forward at (32,4,263,1024,128,1024,300); argv[1] = workspace fill (zero|nan|one)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import test_gpu_ops as T
from ast_amd import _lib
from oracle.ast_ref_torch import decoder_torch
lib = _lib.load()
os.environ["ASTK_DEC_PERSIST"] = "0"
B, L, Tt, H, E, A, V, nl = 32, 4, 263, 1024, 128, 1024, 300, 1
s = T._dec_setup(lib, B, L, Tt, H, E, A, V, nl, False, seed=B + L + 1)
cfg = {"rnn_config": {"dec_layers": nl, "attn_units": A}}
Pt = {k: torch.tensor(v) for k, v in s["P"].items()}
loss_ref, pred_ref = decoder_torch(cfg, Pt, torch.tensor(s["enc"]), torch.tensor(s["c0"]), torch.tensor(s["h0"]), s["y"], s["flags"], V, None, None)
nbytes = lib.astk_decoder_workspace_bytes(C.byref(s["d"]))
fill = sys.argv[1] if len(sys.argv) > 1 else "zero"
ws = torch.full((nbytes // 4 + 64,), {"zero": 0.0, "nan": float("nan"), "one": 1.0}[fill], device="cuda")
dev = T.dev
keep = [dev(s["enc"]), dev(s["c0"]), dev(s["h0"]), dev(s["y"], torch.int32), dev(np.asarray(s["flags"]), torch.int32)]
for it in range(3):
    loss_d = torch.zeros(1, device="cuda"); pred_d = torch.zeros(s["S"], B, dtype=torch.int32, device="cuda")
    rc = lib.astk_decoder_fwd(C.byref(s["d"]), C.byref(s["dp"]), T.vp(keep[0]), T.vp(keep[1]), T.vp(keep[2]), T.vp(keep[3]),
                              T.vp(keep[4]), None, None, T.vp(loss_d), T.vp(pred_d), T.vp(ws), nbytes, T.stream())
    torch.cuda.synchronize()
    print(fill, os.environ.get("ASTK_ROW_LONGK"), "call", it, "loss", float(loss_d), "ref", float(loss_ref), "rel", abs(float(loss_d) - float(loss_ref)) / float(loss_ref),
          "pred mismatches", int((pred_d.cpu().numpy() != pred_ref.numpy()).sum()))
