"""Debug: the (16,400,13,128,512) CNN case under fp16x2 with gemm.forward_pairs 1 vs 0: layer pre-activations and gradients compared."""
import ctypes as C, sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from ast_amd import _lib
from ast_amd._lib import CnnLayerGrads, CnnLayerParams
from conftest import tiny_cfg
from oracle.ast_ref import init_params
import test_gpu_ops as TG
B, T, D, c0, c1 = 16, 400, 13, 128, 512
cfg = tiny_cfg(c0=c0, c1=c1)
P = init_params(cfg, D, 11, seed=1, dtype=np.float64)
rng = np.random.default_rng(0)
for i in range(2):
    P[f"CNN_{i}_bn/gamma"] = 1 + 0.3 * rng.standard_normal(P[f"CNN_{i}_bn/gamma"].shape)
    P[f"CNN_{i}_bn/beta"] = 0.2 * rng.standard_normal(P[f"CNN_{i}_bn/beta"].shape)
X = rng.standard_normal((B, T, D))
with _lib.load_test_hooks() as lib:
    lib.astk_set_gemm_bf16_split_below(C.c_double(0.0)); lib.astk_set_gemm_precision(0)
    cd = TG._cnn_desc(cfg, B, T, D)
    t2, f2, feat = C.c_int(), C.c_int(), C.c_int()
    lib.astk_conv_bn_relu_out_dims(C.byref(cd), C.byref(t2), C.byref(f2), C.byref(feat))
    gout = rng.standard_normal((t2.value, B, feat.value))
    res = {}
    for knob in (1, 0, 1):
        _lib.set_tuning("gemm.forward_pairs", knob)
        names = ["CNN_0", "CNN_1"]
        prm = {n + s: TG.dev(P[n + s]) for n in names for s in ("/W", "_bn/gamma", "_bn/beta", "_bn/avg_mean", "_bn/avg_var")}
        grd = {k: torch.zeros_like(v) for k, v in prm.items()}
        cp, cg = (CnnLayerParams * 2)(), (CnnLayerGrads * 2)()
        for i, n in enumerate(names):
            cp[i].W, cp[i].gamma, cp[i].beta = prm[n + "/W"].data_ptr(), prm[n + "_bn/gamma"].data_ptr(), prm[n + "_bn/beta"].data_ptr()
            cp[i].avg_mean, cp[i].avg_var = prm[n + "_bn/avg_mean"].data_ptr(), prm[n + "_bn/avg_var"].data_ptr()
            cg[i].dW, cg[i].dgamma, cg[i].dbeta = grd[n + "/W"].data_ptr(), grd[n + "_bn/gamma"].data_ptr(), grd[n + "_bn/beta"].data_ptr()
        nbytes = lib.astk_conv_bn_relu_workspace_bytes(C.byref(cd))
        ws = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device="cuda")
        out = torch.empty(t2.value, B, feat.value, device="cuda")
        xd = TG.dev(X)
        st = TG.stream()
        TG.ok(lib, lib.astk_conv_bn_relu_fwd(C.byref(cd), cp, TG.vp(xd), None, TG.vp(out), TG.vp(ws), nbytes, 1, st))
        pre = []
        for layer, rows, ch in ((0, B * 1 * 200, c0), (1, B * 1 * 100, c1)):
            p_ = torch.empty(rows, ch, device="cuda")
            TG.ok(lib, lib.astk_conv_debug_preact(C.byref(cd), TG.vp(ws), nbytes, layer, TG.vp(p_), st))
            pre.append(p_.clone())
        g = TG.dev(gout)
        TG.ok(lib, lib.astk_conv_bn_relu_bwd(C.byref(cd), cp, cg, TG.vp(g), TG.vp(ws), nbytes, st))
        torch.cuda.synchronize()
        res.setdefault(knob, []).append((out.clone(), pre, {k: v.clone() for k, v in grd.items()}))
    a, b, a2 = res[1][0], res[0][0], res[1][1]
    for nm, x, y in (("pairs1 vs pairs0", a, b), ("pairs1 vs pairs1 again", a, a2)):
        print(nm, "out", float((x[0] - y[0]).abs().max()), "pre0", float((x[1][0] - y[1][0]).abs().max()), "pre1", float((x[1][1] - y[1][1]).abs().max()),
              {k: round(float((x[2][k] - y[2][k]).abs().max() / y[2][k].abs().max()), 6) for k in x[2] if "avg" not in k})
    d1 = (a[1][1] - b[1][1]).abs()
    print("pre1 diff > 1e-3:", int((d1 > 1e-3).sum()), "of", d1.numel(), "rows with diff", torch.nonzero((d1 > 1e-3).any(1)).flatten()[:20].tolist(), "cols", torch.nonzero((d1 > 1e-3).any(0)).flatten()[:20].tolist())
