"""one GEMM shape, timed: python3 scratch/gemm_one.py LAYOUT M N K [MODE] [ITERS]  (LAYOUT 0 NT / 1 NN / 2 TN; MODE 0 store / 1 accum / 2 atomic)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ast_amd import _lib
lib = _lib.load()
layout, M, N, K = map(int, sys.argv[1:5])
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
vp = lambda t: C.c_void_p(t.data_ptr())
a = torch.randn(M if layout != 2 else K, K if layout != 2 else M, device='cuda')
b = torch.randn(N if layout == 0 else K, K if layout == 0 else N, device='cuda')
c = torch.zeros(M, N, device='cuda')
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: lib.astk_gemm_f32(layout, M, N, K, vp(a), a.shape[1], vp(b), b.shape[1], vp(c), N, None, mode, 1, 1, 0, 0, 0, s)
for _ in range(3): assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): call()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"layout {layout} M{M} N{N} K{K} mode {mode}: {ms * 1e3:8.1f} us  {2 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
