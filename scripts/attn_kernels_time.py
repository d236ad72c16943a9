"""Attention forward / backward C-ABI calls at the three MSCSA levels (B = 32), kernels only (A/B two builds with
HUPR_LIB_PATH=<other libhupr_hip.so>); the checksums must agree between builds that claim identical arithmetic.
usage: python scripts/attn_kernels_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
for N, C, B in ((4096, 64, 32), (1024, 128, 32), (256, 256, 32)):
    gen = torch.Generator(device="cuda").manual_seed(1)
    k, q, v = (torch.randn(B, N, C, device="cuda", generator=gen).bfloat16() for _ in range(3))
    v32 = v.float(); out = torch.empty(B, N, C, device="cuda"); lse = torch.empty(B, N, device="cuda")
    g = torch.randn(B, N, C, device="cuda", generator=gen).bfloat16(); g32 = g.float()
    dk, dq, dv = (torch.empty(B, N, C, device="cuda") for _ in range(3)); scr = torch.empty(B, N, device="cuda")
    def fwd(): F_.rt.check(L.hupr_attn_fwd_bf16in(F_.rt.ptr(k), F_.rt.ptr(q), F_.rt.ptr(v), F_.rt.ptr(v32), F_.rt.ptr(out), F_.rt.ptr(lse), B, N, C, F_.rt.stream()))
    def bwd(): F_.rt.check(L.hupr_attn_bwd_bf16in(F_.rt.ptr(k), F_.rt.ptr(q), F_.rt.ptr(v), F_.rt.ptr(g), F_.rt.ptr(v32), F_.rt.ptr(out), F_.rt.ptr(g32), F_.rt.ptr(lse), F_.rt.ptr(dk), F_.rt.ptr(dq), F_.rt.ptr(dv), F_.rt.ptr(scr), B, N, C, 1, F_.rt.stream()))
    res = []
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        best = 1e9
        for rnd in range(3):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 10 * 1e3)
        res.append("%s %.0f us" % (name, best))
    cs = [t.double().sum().item() for t in (out, lse, dq, dk, dv)]
    print("N=%d C=%d B=%d: %s | checksums %s" % (N, C, B, ", ".join(res), " ".join("%.9e" % c for c in cs)))
