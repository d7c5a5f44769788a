"""The step's sub-130-TFLOP/s launches, one by one, under forced tile sizes (knob gemm.tile): is the launcher's choice the best one?
   python3 scratch/r6_small_gemms.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ast_amd import _lib
lib = _lib.load()
vp = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [  # layout, M, N, K, mode, batch
    (2, 200, 512, 39, 2, 32), (1, 39, 512, 200, 0, 32), (1, 6400, 512, 512, 0, 1), (0, 1248, 512, 512, 0, 1), (1, 1248, 128, 2048, 0, 1),
    (2, 128, 128, 76800, 2, 1), (2, 512, 1024, 1248, 2, 1), (0, 640, 1024, 3072, 0, 1), (1, 640, 3072, 1024, 0, 1), (2, 1024, 3072, 640, 2, 1)]
for layout, M, N, K, mode, batch in SHAPES:
    a = torch.randn(batch, *((K, M) if layout == 2 else (M, K)), device='cuda')
    b = torch.randn(batch, *((N, K) if layout == 0 else (K, N)), device='cuda')
    ldc = (N + 3) // 4 * 4
    c = torch.zeros(batch, M, ldc, device='cuda')
    out = []
    for tile in (0, 64, 128):
        _lib.set_tuning("gemm.tile", tile)
        call = lambda: lib.astk_gemm_f32(layout, M, N, K, vp(a), a.shape[2], vp(b), b.shape[2], vp(c), ldc, None, mode, 1, batch, a[0].numel(), b[0].numel(), c[0].numel(), s)
        for _ in range(3): assert call() == 0, lib.astk_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): call()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 30 * 1e3)
    _lib.set_tuning("gemm.tile", 0)
    print(f"layout {layout} {M}x{N}x{K} b{batch} mode {mode}: auto {out[0]:7.1f} us  tile64 {out[1]:7.1f}  tile128 {out[2]:7.1f}   ({2*M*N*K*batch/out[0]/1e6:6.1f} TFLOP/s auto)")
