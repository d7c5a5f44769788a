"""Run-to-run determinism of the encoder stack alone: fwd + bwd of the same inputs N times; enc_states, cT / hT and the BIAS gradients (summed
inside the backward recurrence in a fixed order, two commutative atomic adds per element) must be bit-identical from run to run -- any
difference is a stale or torn hand-off inside the persistent kernels.
    python3 scratch/enc_repeat.py [N=2000] [T=200] [B=32] [h=256] [nl=3] [ENV=VALUE ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
args = [a for a in sys.argv[1:] if "=" not in a]
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); os.environ[k] = v
N, T, B, h, nl = (int(args[i]) if len(args) > i else d for i, d in enumerate((2000, 200, 32, 256, 3)))
import numpy as np, torch
from ast_amd import _lib
from ast_amd._lib import LstmGrads, LstmParams, LstmStackDesc
lib = _lib.load()
in_dim = 3072
rng = np.random.default_rng(1)
dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
prm, grd, names = {}, {}, []
for pat in ("L{}_enc", "L{}_rev_enc"):
    n_in = in_dim
    for k in range(nl):
        n = pat.format(k); names.append(n)
        prm[n + "/Wu"] = dev(rng.standard_normal((4 * h, n_in)) / np.sqrt(n_in)); prm[n + "/b"] = dev(rng.standard_normal(4 * h) * 0.3)
        prm[n + "/Wl"] = dev(rng.standard_normal((4 * h, h)) / np.sqrt(h)); n_in = h
for k, v in prm.items(): grd[k] = torch.zeros_like(v)
lp, lg = (LstmParams * (2 * nl))(), (LstmGrads * (2 * nl))()
for i, n in enumerate(names):
    lp[i].Wu, lp[i].b, lp[i].Wl = (prm[n + s].data_ptr() for s in ("/Wu", "/b", "/Wl"))
    lg[i].dWu, lg[i].db, lg[i].dWl = (grd[n + s].data_ptr() for s in ("/Wu", "/b", "/Wl"))
d = LstmStackDesc(T, B, in_dim, h, nl, 2)
nbytes = lib.astk_lstm_stack_workspace_bytes(C.byref(d))
ws = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda")
x = dev(rng.standard_normal((T, B, in_dim)))
mk = dev((rng.random((2, nl, T, B, h)) >= 0.3) / 0.7)
g_enc, g_c, g_h = dev(rng.standard_normal((B, T, 2 * h))), dev(rng.standard_normal((2, nl, B, h))), dev(rng.standard_normal((2, nl, B, h)))
enc = torch.zeros(B, T, 2 * h, device="cuda"); cT = torch.zeros(2, nl, B, h, device="cuda"); hT = torch.zeros(2, nl, B, h, device="cuda")
dx = torch.zeros(T, B, in_dim, device="cuda")
# a SECOND input set, run in between: a hand-off buffer read too early then holds the other set's values, not a copy of the right ones
x2 = dev(rng.standard_normal((T, B, in_dim))); g_enc2 = dev(rng.standard_normal((B, T, 2 * h)))
ref = None
bad = {}
for it in range(N):
    for v in grd.values(): v.zero_()
    if os.environ.get("ALTERNATE", "1") != "0":
        assert lib.astk_lstm_stack_fwd(C.byref(d), lp, vp(x2), vp(mk), vp(enc), vp(cT), vp(hT), vp(ws), nbytes, stream()) == 0
        assert lib.astk_lstm_stack_bwd(C.byref(d), lp, lg, vp(x2), vp(mk), vp(g_enc2), vp(g_c), vp(g_h), vp(dx), vp(ws), nbytes, stream()) == 0
        for v in grd.values(): v.zero_()
    assert lib.astk_lstm_stack_fwd(C.byref(d), lp, vp(x), vp(mk), vp(enc), vp(cT), vp(hT), vp(ws), nbytes, stream()) == 0
    ws_f = ws.clone() if os.environ.get("WS_CHECK") else None
    assert lib.astk_lstm_stack_bwd(C.byref(d), lp, lg, vp(x), vp(mk), vp(g_enc), vp(g_c), vp(g_h), vp(dx), vp(ws), nbytes, stream()) == 0
    cur = {"enc": enc.clone(), "cT": cT.clone(), "hT": hT.clone()}
    if os.environ.get("WS_CHECK"):
        cur["ws_after_bwd"] = ws.clone()
    for n in names: cur[n + "/db"] = grd[n + "/b"].clone()
    if ws_f is not None: cur["ws_after_fwd"] = ws_f
    if ref is None: ref = cur; print("workspace bytes", nbytes); continue
    for k in cur:
        if not torch.equal(cur[k], ref[k]):
            if k.startswith("ws_"):
                idx = (cur[k] != ref[k]).nonzero().flatten()
                e = [int(idx.numel()), int(idx.min()), int(idx.max())]
                zg = T * B * 4 * h * 4
                for nm, st in (("L0 fwd-dir dz", 2048), ("L0 rev-dir dz", 150538240)):       # (layout of this shape: ZG[0][0], ZG[1][0])
                    sel = idx[(idx >= st) & (idx < st + zg)] - st
                    if sel.numel():
                        ts = torch.unique(sel // (B * 4 * h * 4)).tolist()
                        rows = torch.unique((sel // (4 * h * 4)) % B).tolist()
                        cols = torch.unique((sel // 4) % (4 * h) // 4 // 16).tolist()      # unit slices of 16
                        first = max(ts)          # the backward walks t downwards: the LARGEST differing t is where it started
                        s1 = sel[(sel // (B * 4 * h * 4)) == first]
                        e.append((nm, "steps", ts[:12], "rows", [rows[0], rows[-1]], "first differing step", first, "unit slices there",
                                  torch.unique((s1 // 4) % (4 * h) // 4 // 16).tolist(), "rows there", torch.unique((s1 // (4 * h * 4)) % B).tolist(),
                                  "floats there", int(s1.numel()) // 1))
            else:
                e = float((cur[k] - ref[k]).abs().max() / ref[k].abs().max())
            bad.setdefault(k, []).append((it, e))
mask = C.c_uint(0); lib.astk_persist_status(C.byref(mask), 1)
print(f"N {N} T {T} B {B} h {h} nl {nl} path {lib.astk_lstm_stack_path(C.byref(d))} status {mask.value} env {[a for a in sys.argv[1:] if '=' in a]}")
print("mismatching runs:", {k: (len(v), v[:4]) for k, v in bad.items()} if bad else "none: bit-identical")
