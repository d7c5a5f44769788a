"""Cost of a fork/join between two HIP streams (event record + stream wait both ways) relative to staying on one stream."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ast_amd import _lib
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
x = torch.ones(1024, device="cuda")
def tiny(s): lib.astk_scale_f32(C.c_void_p(x.data_ptr()), 1024, 1.0, C.c_void_p(s.cuda_stream))
def masked(lo, hi):
    m = (C.c_uint32 * 8)()
    for b in range(lo, hi): m[b // 32] |= 1 << (b % 32)
    h = C.c_void_p(); assert hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, m) == 0
    return torch.cuda.ExternalStream(h.value)
a = torch.cuda.Stream()
variants = {"second plain stream": torch.cuda.Stream(), "second stream masked to CUs 0..191": masked(0, 192)}
N = 2000
def run(b):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        tiny(a)
        if b is not None:
            e = torch.cuda.Event(); e.record(a); b.wait_event(e)
            tiny(b)
            e2 = torch.cuda.Event(); e2.record(b); a.wait_event(e2)
        else:
            tiny(a)
        tiny(a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
base = run(None)
print(f"three tiny kernels on one stream: {base:.1f} us")
for name, b in variants.items():
    print(f"middle kernel on a {name}: {run(b):.1f} us (+{run(b) - base:.1f})")
