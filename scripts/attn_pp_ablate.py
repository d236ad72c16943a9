"""Phase ablations of the ping-pong attention forward (timing only; hupr_debug_attn_pingpong(1 | bits << 4)).
usage: python scripts/attn_pp_ablate.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
N, C, B = 4096, 64, 32
gen = torch.Generator(device="cuda").manual_seed(1)
k, q, v = (torch.randn(B, N, C, device="cuda", generator=gen) for _ in range(3))
kb, qb, vb = (k * 0.5).bfloat16(), (q * 0.5).bfloat16(), v.bfloat16()
out = torch.empty(B, N, C, device="cuda"); lse = torch.empty(B, N, device="cuda")
fwd = lambda: F_.rt.check(L.hupr_attn_fwd_bf16in(F_.rt.ptr(kb), F_.rt.ptr(qb), F_.rt.ptr(vb), F_.rt.ptr(v), F_.rt.ptr(out), F_.rt.ptr(lse), B, N, C, F_.rt.stream()))
names = {0: "full kernel", 128: "full kernel, no s_setprio", 256: "full kernel, s_setprio(1) on waves 0-3", 1: "no LDS-DMA", 2: "no fragment reads", 4: "no exponentials", 8: "no maxima", 16: "no MFMAs", 64: "no epilogue stores",
         27: "exponentials + remaining VALU only", 23: "maxima + remaining VALU only", 31: "skeleton (barrier, remaining VALU)",
         127: "skeleton, no barrier, no epilogue"}
for rnd in range(2):
    for bits, name in names.items():
        if name is None:
            continue
        L.hupr_debug_attn_pingpong(1 | (bits << 4))
        best = 1e9
        for _ in range(3):
            for _ in range(2):
                fwd()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                fwd()
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 8 * 1e3)
        print("round %d  abl %2d  %-40s %6.1f us" % (rnd, bits, name, best))
L.hupr_debug_attn_pingpong(1)
