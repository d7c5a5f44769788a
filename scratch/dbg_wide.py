"""debug: per-launch decoder case (32,4,263,1024,128,1024,300) in both GEMM dispatch modes, with and without host flags"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import test_gpu_ops as T
from ast_amd import _lib
from oracle.ast_ref_torch import decoder_torch
lib = _lib.load()
lib.astk_set_gemm_bf16_split_below.restype = C.c_double
os.environ["ASTK_DEC_PERSIST"] = "0"
B, L, Tt, H, E, A, V, nl = 32, 4, 263, 1024, 128, 1024, 300, 1
for mode in (0.0, 3e9):
    for hostflags in (False, True):
        lib.astk_set_gemm_bf16_split_below(C.c_double(mode))
        s = T._dec_setup(lib, B, L, Tt, H, E, A, V, nl, False, seed=B + L + 1)
        if hostflags:
            host = (C.c_int32 * s["S"])(*[int(f) for f in s["flags"]])
            s["d"].use_truth_host = C.cast(host, C.POINTER(C.c_int32))
        cfg = {"rnn_config": {"dec_layers": nl, "attn_units": A}}
        Pt = {k: torch.tensor(v) for k, v in s["P"].items()}
        loss_ref, pred_ref = decoder_torch(cfg, Pt, torch.tensor(s["enc"]), torch.tensor(s["c0"]), torch.tensor(s["h0"]), s["y"], s["flags"], V, None, None)
        nbytes = lib.astk_decoder_workspace_bytes(C.byref(s["d"]))
        ws = torch.zeros(nbytes // 4 + 64, device="cuda")
        dev = T.dev
        loss_d = torch.zeros(1, device="cuda"); pred_d = torch.zeros(s["S"], B, dtype=torch.int32, device="cuda")
        keep = [dev(s["enc"]), dev(s["c0"]), dev(s["h0"]), dev(s["y"], torch.int32), dev(np.asarray(s["flags"]), torch.int32)]
        rc = lib.astk_decoder_fwd(C.byref(s["d"]), C.byref(s["dp"]), T.vp(keep[0]), T.vp(keep[1]), T.vp(keep[2]), T.vp(keep[3]),
                                  T.vp(keep[4]), None, None, T.vp(loss_d), T.vp(pred_d), T.vp(ws), nbytes, T.stream())
        torch.cuda.synchronize()
        print("mode", mode, "hostflags", hostflags, "flags", s["flags"], "rc", rc, "loss", float(loss_d), "ref", float(loss_ref), "rel", abs(float(loss_d) - float(loss_ref)) / float(loss_ref),
              "pred mismatches", int((pred_d.cpu().numpy() != pred_ref.numpy()).sum()), "path", lib.astk_decoder_path(C.byref(s["d"])))
