"""CNN fwd/bwd at a given size against the float64 torch restatement: prints the relative errors of every gradient."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_ops as T
from ast_amd import _lib
lib = _lib.load()
def run(B, Tt, D, c0, c1, perm=False):
    from ast_amd._lib import CnnLayerGrads, CnnLayerParams
    from oracle.ast_ref import init_params
    from oracle.ast_ref_torch import cnn_torch
    cfg = T.tiny_cfg(c0=c0, c1=c1)
    P = init_params(cfg, D, 11, seed=1, dtype=np.float64)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((B, Tt, D))
    if not os.environ.get("NOZERO"): X[B // 2:, Tt // 2:] = 0.0
    if perm: X = X[rng.permutation(B)]
    Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=k.startswith("CNN") and "avg" not in k) for k, v in P.items()}
    out_ref = cnn_torch(cfg, Pt, torch.tensor(X), None)
    gout = np.random.default_rng(1).standard_normal(out_ref.shape)
    out_ref.backward(torch.tensor(gout))
    cd = T._cnn_desc(cfg, B, Tt, D)
    t2, f2, feat = C.c_int(), C.c_int(), C.c_int()
    lib.astk_conv_bn_relu_out_dims(C.byref(cd), C.byref(t2), C.byref(f2), C.byref(feat))
    names = ["CNN_0", "CNN_1"]
    prm = {n + s: T.dev(P[n + s]) for n in names for s in ("/W", "_bn/gamma", "_bn/beta", "_bn/avg_mean", "_bn/avg_var")}
    grd = {k: torch.zeros_like(v) for k, v in prm.items()}
    cp, cg = (CnnLayerParams * 2)(), (CnnLayerGrads * 2)()
    for i, n in enumerate(names):
        cp[i].W, cp[i].gamma, cp[i].beta = prm[n + "/W"].data_ptr(), prm[n + "_bn/gamma"].data_ptr(), prm[n + "_bn/beta"].data_ptr()
        cp[i].avg_mean, cp[i].avg_var = prm[n + "_bn/avg_mean"].data_ptr(), prm[n + "_bn/avg_var"].data_ptr()
        cg[i].dW, cg[i].dgamma, cg[i].dbeta = grd[n + "/W"].data_ptr(), grd[n + "_bn/gamma"].data_ptr(), grd[n + "_bn/beta"].data_ptr()
    nbytes = lib.astk_conv_bn_relu_workspace_bytes(C.byref(cd))
    ws = torch.zeros(nbytes // 4 + 64, device="cuda")
    out = torch.empty(t2.value, B, feat.value, device="cuda")
    xd = T.dev(X)
    s = T.stream()
    assert lib.astk_conv_bn_relu_fwd(C.byref(cd), cp, T.vp(xd), None, T.vp(out), T.vp(ws), nbytes, 1, s) == 0
    g = T.dev(gout)
    assert lib.astk_conv_bn_relu_bwd(C.byref(cd), cp, cg, T.vp(g), T.vp(ws), nbytes, s) == 0
    torch.cuda.synchronize()
    e = {"out": float((out.cpu().double() - out_ref.detach()).abs().max() / out_ref.detach().abs().max())}
    for n in names:
        for sfx in ("/W", "_bn/gamma", "_bn/beta"):
            r = Pt[n + sfx].grad
            e[n + sfx] = float((grd[n + sfx].cpu().double() - r).abs().max() / r.abs().max())
    print(os.environ.get("ASTK_GEMM_PREC", "default"), os.environ.get("ASTK_GEMM_X3_BELOW", ""), (B, Tt, D, c0, c1), "perm" if perm else "", {k: f"{v:.1e}" for k, v in e.items()}, flush=True)
run(32, 800, 13, 128, 512)
run(32, 800, 13, 128, 512, perm=True)
