"""Can independent GEMM work run beside the encoder's persistent backward recurrence without slowing its chain?
Scenarios: side GEMM on (a) no stream at all, (b) an ordinary second stream, (c) a stream masked to CUs 192..255 with the recurrence
unmasked, (d) as (c) with the whole LSTM backward call on a stream masked to CUs 0..191.  Run under
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scratch/overlap_probe.py run ; python3 scratch/overlap_probe.py join DIR"""
import ctypes as C, csv, glob, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

def run():
    import numpy as np, torch
    from ast_amd import _lib
    from ast_amd._lib import LstmGrads, LstmParams, LstmStackDesc
    lib = _lib.load()
    hip = C.CDLL("libamdhip64.so")
    def masked(lo, hi):
        m = (C.c_uint32 * 8)()
        for b in range(lo, hi): m[b // 32] |= 1 << (b % 32)
        s = C.c_void_p()
        assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, m) == 0
        return s
    def plain():
        s = C.c_void_p(); assert hip.hipStreamCreate(C.byref(s)) == 0; return s
    T, B, in_dim, h, nl = 200, 32, 3072, 256, 3
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.05
    d = LstmStackDesc(T, B, in_dim, h, nl, 2)
    lp, lg = (LstmParams * (2 * nl))(), (LstmGrads * (2 * nl))()
    keep = []
    for i in range(2 * nl):
        fan = in_dim if i % nl == 0 else h
        ts = [rnd(4 * h, fan), rnd(4 * h), rnd(4 * h, h)]
        gs = [torch.zeros_like(t) for t in ts]
        keep += ts + gs
        lp[i].Wu, lp[i].b, lp[i].Wl = (t.data_ptr() for t in ts)
        lg[i].dWu, lg[i].db, lg[i].dWl = (t.data_ptr() for t in gs)
    nbytes = lib.astk_lstm_stack_workspace_bytes(C.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    x, enc = rnd(T, B, in_dim), torch.zeros(B, T, 2 * h, device="cuda")
    cT, hT = torch.zeros(2, nl, B, h, device="cuda"), torch.zeros(2, nl, B, h, device="cuda")
    ge, gc, gh, dx = rnd(B, T, 2 * h), rnd(2, nl, B, h), rnd(2, nl, B, h), torch.zeros(T, B, in_dim, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    main, side_plain, side_hi, lat_lo = plain(), plain(), masked(192, 256), masked(0, 192)
    A, Bm, Cm = rnd(4096, 2048), rnd(1024, 2048), torch.zeros(4096, 1024, device="cuda")
    def fwd(s): assert lib.astk_lstm_stack_fwd(C.byref(d), lp, vp(x), None, vp(enc), vp(cT), vp(hT), vp(ws), nbytes, s) == 0
    def bwd(s): assert lib.astk_lstm_stack_bwd(C.byref(d), lp, lg, vp(x), None, vp(ge), vp(gc), vp(gh), vp(dx), vp(ws), nbytes, s) == 0
    S1, S2 = rnd(2048, 2048), rnd(2048, 2048)
    def side(s):      # a rocBLAS product: its kernel name cannot be confused with the library's own GEMMs in the trace
        with torch.cuda.stream(torch.cuda.ExternalStream(s.value)):
            torch.mm(S1, S2)
    for name, ls, ss in (("a", main, None), ("b", main, side_plain), ("c", main, side_hi), ("d", lat_lo, side_hi)):
        for rep in range(3):
            fwd(main); torch.cuda.synchronize()
            bwd(ls)
            if ss is not None: side(ss)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for st in (main, side_plain, side_hi, lat_lo):
        hip.hipStreamDestroy(st)
    print("done", flush=True)

def join(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    per = [r for r in rows if "lstm_persist_bwd" in r["Kernel_Name"]]
    sides = [r for r in rows if "astk" not in r["Kernel_Name"] and ("Cijk" in r["Kernel_Name"] or "gemm" in r["Kernel_Name"].lower())]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for i, name in enumerate("abcd"):
        p = [dur(r) for r in per[3 * i:3 * i + 3]]
        s = [dur(r) for r in sides[3 * (i - 1):3 * (i - 1) + 3]] if i else []
        ov = []
        if i:
            for a, b in zip(per[3 * i:3 * i + 3], sides[3 * (i - 1):3 * (i - 1) + 3]):
                ov.append(round(max(0, min(int(a["End_Timestamp"]), int(b["End_Timestamp"])) - max(int(a["Start_Timestamp"]), int(b["Start_Timestamp"]))) / 1e3))
        print(name, "persistent bwd us", [round(v) for v in p], "side gemm us", [round(v) for v in s], "overlap us", ov)

if __name__ == "__main__":
    run() if sys.argv[1] == "run" else join(sys.argv[2])
