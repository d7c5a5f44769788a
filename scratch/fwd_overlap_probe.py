"""Upper bound for hiding the tail of the encoder's layer-0 input projection behind the forward recurrence: baseline = both
projection GEMMs (6400x1024x3072) then the recurrence; variant = 80 % of the rows first, then the remaining 20 % on a stream masked
to CUs 192..255 while the recurrence (a stack with a tiny input projection, so the call is the recurrence kernel) runs on CUs 0..191."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ast_amd import _lib
from ast_amd._lib import LstmParams, LstmStackDesc
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
def masked(lo, hi):
    m = (C.c_uint32 * 8)()
    for b in range(lo, hi): m[b // 32] |= 1 << (b % 32)
    h = C.c_void_p(); assert hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, m) == 0
    return torch.cuda.ExternalStream(h.value)
T, B, h, nl, IN = 200, 32, 256, 3, 16
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.05
d = LstmStackDesc(T, B, IN, h, nl, 2)
lp = (LstmParams * (2 * nl))(); keep = []
for i in range(2 * nl):
    ts = [rnd(4 * h, IN if i % nl == 0 else h), rnd(4 * h), rnd(4 * h, h)]; keep += ts
    lp[i].Wu, lp[i].b, lp[i].Wl = (t.data_ptr() for t in ts)
nbytes = lib.astk_lstm_stack_workspace_bytes(C.byref(d))
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
x, enc = rnd(T, B, IN), torch.zeros(B, T, 2 * h, device="cuda")
cT, hT = torch.zeros(2, nl, B, h, device="cuda"), torch.zeros(2, nl, B, h, device="cuda")
X, W, Z = rnd(T * B, 3072), rnd(1024, 3072), torch.zeros(2, T * B, 1024, device="cuda")
vp = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
sp = lambda s: C.c_void_p(s.cuda_stream)
def proj(s, r0, r1):
    for dd in range(2):
        assert lib.astk_gemm_f32(0, r1 - r0, 1024, 3072, vp(X, r0 * 3072 * 4), 3072, vp(W), 3072, vp(Z[dd], r0 * 1024 * 4), 1024, None, 0, 1, 1, 0, 0, 0, sp(s)) == 0
def rec(s): assert lib.astk_lstm_stack_fwd(C.byref(d), lp, vp(x), None, vp(enc), vp(cT), vp(hT), vp(ws), nbytes, sp(s)) == 0
main, lo, hi = torch.cuda.Stream(), masked(0, 192), masked(192, 256)
def baseline():
    proj(main, 0, T * B); rec(main)
def variant(frac):
    r = int(T * frac) * B
    proj(main, 0, r)
    e = torch.cuda.Event(); e.record(main); lo.wait_event(e); hi.wait_event(e)
    proj(hi, r, T * B); rec(lo)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event(); e1.record(lo); e2.record(hi); main.wait_event(e1); main.wait_event(e2)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"baseline: {timed(baseline):.0f} us")
for frac in (0.9, 0.85, 0.8, 0.75):
    print(f"first {frac:.2f} of the rows, rest beside the recurrence: {timed(lambda: variant(frac)):.0f} us")
