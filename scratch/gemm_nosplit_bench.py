"""Upper bound of what operands that arrive as fp16 planes would buy: the step's big GEMM shapes, library chosen by ASTK_LIB_PATH."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ast_amd import _lib
lib = _lib.load()
def vp(t): return C.c_void_p(t.data_ptr())
tot = 0.0
def run(layout, M, N, K, mode=0):
    global tot
    mk = (lambda *sh: torch.randn(*sh, device='cuda')) if os.environ.get("DATA") == "randn" else (lambda *sh: torch.rand(*sh, device='cuda') + 0.5)
    if os.environ.get("DATA") == "relu":      # half zeros, like the activations behind a ReLU / a dropout mask
        mk = lambda *sh: torch.relu(torch.randn(*sh, device='cuda'))
    a = mk(K if layout == 2 else M, M if layout == 2 else K)
    b = mk(N if layout == 0 else K, K if layout == 0 else N)
    c = torch.zeros(M, N, device='cuda')
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def f():
        rc = lib.astk_gemm_f32(layout, M, N, K, vp(a), a.shape[1], vp(b), b.shape[1], vp(c), N, None, mode, 1, 1, 0, 0, 0, s)
        assert rc == 0
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print(f"layout {layout} {M}x{N}x{K} mode {mode}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
import os
shapes = ((0, 6400, 1024, 2560, 0), (1, 6400, 2560, 1024, 0), (2, 1024, 2560, 6400, 2), (0, 38400, 512, 1152, 0), (1, 38400, 1152, 512, 0),
          (2, 512, 1152, 38400, 2), (0, 4096, 4096, 4096, 0), (1, 4096, 4096, 4096, 0))
if os.environ.get("SHAPES") == "ksweep":      # same work, different tile counts / k depths
    shapes = ((1, 6400, 3072, 1024, 0), (1, 3200, 3072, 2048, 0), (1, 1600, 3072, 4096, 0), (1, 6144, 3072, 1024, 0), (1, 6400, 3072, 512, 0),
              (0, 6400, 1024, 3072, 0), (0, 3200, 2048, 3072, 0), (0, 6400, 1024, 1024, 0))
if os.environ.get("SHAPES") == "msweep":
    shapes = tuple((1, m, 3072, 1024, 0) for m in (5120, 5632, 6016, 6144, 6272, 6400, 6528, 6656, 7168, 7680, 8192))
for (l, M, N, K, mode) in shapes:
    run(l, M, N, K, mode)
print(f"sum {tot:.1f} us")
