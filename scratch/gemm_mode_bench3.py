"""STORE launch against (zero C + ATOMIC launch), same stream, NN 6400x3072x1024 and NT 6400x1024x3072."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ast_amd import _lib
lib = _lib.load()
def vp(t): return C.c_void_p(t.data_ptr())
def run(layout, M, N, K, mode, zero):
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N if layout == 0 else K, K if layout == 0 else N, device='cuda')
    c = torch.zeros(M, N, device='cuda')
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def f():
        if zero: c.zero_()
        lib.astk_gemm_f32(layout, M, N, K, vp(a), K, vp(b), b.shape[1], vp(c), N, None, mode, 1, 1, 0, 0, 0, s)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"layout {layout} {M}x{N}x{K} mode {mode} zero-first {zero}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
for (l, M, N, K) in ((1, 6400, 3072, 1024), (0, 6400, 1024, 3072), (0, 38400, 512, 1152)):
    run(l, M, N, K, 0, False); run(l, M, N, K, 2, True); run(l, M, N, K, 2, False)
