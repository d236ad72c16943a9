"""Plain vs QS attention kernels (round 5) at the three MSCSA level shapes, B = 32, through the C ABI: forward and backward per call,
TFLOP/s on 4 / 10 N^2 C.  QS = query operand pre-scaled by log2(e), accumulator input = minus deferred maximum / minus log-sum-exp.
usage: python scripts/attn_qs_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L, rt = F_.rt.lib(), F_.rt
for N, C, B in ((4096, 64, 32), (1024, 128, 32), (256, 256, 32)):
    gen = torch.Generator(device="cuda").manual_seed(1)
    k32, q32, v32 = (torch.randn(B, N, C, device="cuda", generator=gen) * s for s in (0.5, 0.5, 1.0))
    k, q, v = k32.bfloat16(), q32.bfloat16(), v32.bfloat16()
    qs = (q32 * 1.4426950408889634).bfloat16()
    out = torch.empty(B, N, C, device="cuda"); lse = torch.empty(B, N, device="cuda")
    g = torch.randn(B, N, C, device="cuda", generator=gen).bfloat16(); g32 = g.float()
    dk, dq, dv = (torch.empty(B, N, C, device="cuda") for _ in range(3)); scr = torch.empty(B, N, device="cuda")
    P = rt.ptr
    calls = {
        "plain": (lambda: rt.check(L.hupr_attn_fwd_bf16in_ld_ws(P(k), C, P(q), C, P(v), P(v32), P(out), P(lse), None, 0, B, N, C, None, 0, rt.stream())),
                  lambda: rt.check(L.hupr_attn_bwd_bf16in_ld(P(k), C, P(q), C, P(v), P(g), C, P(v32), P(out), P(g32), P(lse), P(dk), C, P(dq), C, P(dv), P(scr), B, N, C, 1, 0, rt.stream()))),
        "QS": (lambda: rt.check(L.hupr_attn_fwd_bf16in_ld_ws_qs(P(k), C, P(qs), C, P(v), P(v32), P(out), P(lse), None, 0, B, N, C, None, 0, rt.stream())),
               lambda: rt.check(L.hupr_attn_bwd_bf16in_ld_qs(P(k), C, P(qs), C, P(v), P(g), C, P(v32), P(out), P(g32), P(lse), P(dk), C, P(dq), C, P(dv), P(scr), B, N, C, 1, 0, rt.stream()))),
    }
    for name, (fwd, bwd) in calls.items():
        res = []
        for what, fn, fl in (("fwd", fwd, 4.0), ("bwd", bwd, 10.0)):
            best = 1e9
            for rnd in range(3):
                for _ in range(3): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): fn()
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 10 * 1e3)
            res.append("%s %.1f us = %.0f TF/s (%.3f of 2 500)" % (what, best, fl * N * N * C * B / best / 1e6, fl * N * N * C * B / best / 1e6 / 2500))
        cs = [t.double().sum().item() for t in (out, lse, dq, dk, dv)]
        print("N=%d C=%d B=%d %-5s: %s | checksums %s" % (N, C, B, name, ", ".join(res), " ".join("%.6e" % c for c in cs)), flush=True)
