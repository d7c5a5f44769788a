"""Isolated accuracy check of specific GEMM shapes against float64 (prec from ASTK_GEMM_PREC)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ast_amd import _lib
lib = _lib.load()
def vp(t): return C.c_void_p(t.data_ptr())
def check(layout, M, N, K, mode=0, ks=1, zero_rows=False, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K))
    if zero_rows:
        if layout == 2: A[:, K // 3: K // 2] = 0; B[:, K // 3: K // 2] = 0
        else: A[M // 3: M // 2] = 0
    A32, B32 = A.astype(np.float32), B.astype(np.float32)
    ref = A32.astype(np.float64) @ B32.astype(np.float64).T
    pad = lambda n: (n + 3) // 4 * 4
    def store(X, tr):
        X = X.T if tr else X
        P = np.zeros((X.shape[0], pad(X.shape[1])), np.float32); P[:, :X.shape[1]] = X
        return torch.from_numpy(P).cuda()
    a, b = store(A32, layout == 2), store(B32, layout != 0)
    c = torch.zeros(M, pad(N), device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.astk_gemm_f32(layout, M, N, K, vp(a), a.shape[1], vp(b), b.shape[1], vp(c), c.shape[1], None, mode, ks, 1, 0, 0, 0, s)
    torch.cuda.synchronize()
    got = c[:, :N].cpu().double().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"prec {os.environ.get('ASTK_GEMM_PREC','default')} layout {layout} {M}x{N}x{K} mode {mode} ks {ks} zero_rows {zero_rows}: rc {rc} rel err {err:.2e}", flush=True)
for zr in (False, True):
    check(0, 12800, 128, 120, zero_rows=zr)
    check(2, 128, 120, 12800, mode=2, ks=8, zero_rows=zr)
    check(2, 128, 120, 12800, mode=2, ks=1, zero_rows=zr)
    check(0, 76800, 128, 120, zero_rows=zr)
    check(2, 128, 120, 76800, mode=2, ks=64, zero_rows=zr)
