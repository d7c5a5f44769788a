"""Does the operand row stride matter (L2 / memory channel camping)?  4096^3 NT with padded leading dimensions."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ast_amd import _lib
lib = _lib.load()
def vp(t): return C.c_void_p(t.data_ptr())
def run(layout, M, N, K, pad, iters=20):
    ra, ca = (M, K) if layout != 2 else (K, M)
    rb, cb = (N, K) if layout == 0 else (K, N)
    a = torch.randn(ra, ca + pad, device='cuda'); b = torch.randn(rb, cb + pad, device='cuda')
    c = torch.zeros(M, N, device='cuda')
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib.astk_gemm_f32(layout, M, N, K, vp(a), ca + pad, vp(b), cb + pad, vp(c), N, None, 0, 1, 1, 0, 0, 0, s)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"layout {layout} {M}x{N}x{K} pad {pad:4d}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
for pad in (0, 16, 32, 64, 128, 272):
    run(0, 4096, 4096, 4096, pad)
for pad in (0, 32, 272):
    run(0, 6400, 1024, 3072, pad); run(1, 6400, 3072, 1024, pad)
