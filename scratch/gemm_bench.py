import sys, ctypes as C, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from ast_amd import _lib
lib = _lib.load()
def vp(t): return C.c_void_p(t.data_ptr())
def run(layout, M, N, K, iters=20, ks=1, mode=0):
    a = torch.randn(M if layout != 2 else K, K if layout != 2 else M, device='cuda')
    b = torch.randn(N if layout == 0 else K, K if layout == 0 else N, device='cuda')
    c = torch.zeros(M, N, device='cuda')
    lda, ldb = a.shape[1], b.shape[1]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): assert lib.astk_gemm_f32(layout, M, N, K, vp(a), lda, vp(b), ldb, vp(c), N, None, mode, ks, 1, 0, 0, 0, s) == 0, _lib.last_error() if hasattr(_lib, 'last_error') else 'gemm failed'
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): lib.astk_gemm_f32(layout, M, N, K, vp(a), lda, vp(b), ldb, vp(c), N, None, mode, ks, 1, 0, 0, 0, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"layout {layout} M{M} N{N} K{K} ks{ks}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s  tiles {((M+127)//128)*((N+127)//128)*ks}")
import os
print("ASTK_GEMM_G =", os.environ.get("ASTK_GEMM_G"))
run(0, 4096, 4096, 4096)
run(0, 76800, 128, 120); run(0, 38400, 512, 1152); run(0, 6400, 1024, 3072); run(1, 6400, 3072, 1024)
run(2, 1024, 3072, 6400, mode=2, ks=2); run(0, 38400, 128, 2560); run(0, 38400, 128, 2048)
run(2, 1024, 256, 6400, mode=2, ks=32); run(1, 6400, 512, 512); run(2, 2048, 640, 1248, mode=2, ks=3); run(2, 1098, 512, 1248, mode=2, ks=7)
run(2, 128, 120, 76800, mode=2, ks=64); run(0, 1248, 512, 512)
