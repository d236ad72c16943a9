"""s_memtime stamps of the ping-pong attention forward (workgroup 0, waves 0 and 4): cycles per segment of a key tile.
stamps per tile: 0 tile start (after the previous tile's barrier), 1 tile done (before its barrier).
usage: python scripts/attn_pp_trace.py"""
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
buf = torch.zeros(3 * 2 * 4096, dtype=torch.int64, device="cuda")
fwd = lambda: F_.rt.check(L.hupr_attn_fwd_bf16in(F_.rt.ptr(kb), F_.rt.ptr(qb), F_.rt.ptr(vb), F_.rt.ptr(v), F_.rt.ptr(out), F_.rt.ptr(lse), B, N, C, F_.rt.stream()))
for _ in range(3):
    fwd()
L.hupr_debug_attn_trace(F_.rt.ptr(buf))
fwd()
torch.cuda.synchronize()
L.hupr_debug_attn_trace(None)
t = buf.cpu().numpy().reshape(3, 2, 4096)[0]
nt = N // 64
for g in (0, 1):
    s = t[g, :2 * nt].reshape(nt, 2).astype("int64")
    d = {"tile": s[:, 1] - s[:, 0], "barrier": s[1:, 0] - s[:-1, 1]}
    print("wave %d: total %d ticks over %d tiles = %.0f per tile" % (4 * g, s[-1, 1] - s[0, 0], nt, (s[-1, 1] - s[0, 0]) / nt))
    for name, a in d.items():
        a = a[8:56]
        print("   %-9s median %6.0f  mean %6.0f  min %6.0f  max %6.0f ticks" % (name, float(sorted(a)[len(a) // 2]), a.mean(), a.min(), a.max()))
print("(s_memtime ticks = shader-clock cycles; the clock itself drops under matrix-pipe load, scripts/probes/valu_issue_probe.hip)")
