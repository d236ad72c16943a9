"""Level-1 attention backward (B = 32, N = 4096, C = 64, bf16 operands): the 512-thread dK / dV kernel (one barrier per query tile,
both tiles prefetched) against the 256-thread kernel (hupr_debug_attn_dkv512(0)) — times of the whole backward (prep + dQ + dK/dV),
interleaved, and bit-identity of dK, dV.  usage (GPU box): python scripts/attn_bwd_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L, rt = F_.rt.lib(), F_.rt
B, N, C = 32, 4096, 64
g = torch.Generator(device="cuda").manual_seed(11)
k, q, v = (torch.randn(B, N, C, device="cuda", generator=g) * s for s in (0.5, 0.5, 1.0))
kb, qb, vb = k.bfloat16(), q.bfloat16(), v.bfloat16()
g32 = torch.randn(B, N, C, device="cuda", generator=g); gb = g32.bfloat16()
out, lse = torch.empty(B, N, C, device="cuda"), torch.empty(B, N, device="cuda")
rt.check(L.hupr_attn_fwd_bf16in(rt.ptr(kb), rt.ptr(qb), rt.ptr(vb), rt.ptr(v), rt.ptr(out), rt.ptr(lse), B, N, C, rt.stream()))
scr = torch.empty(B, N, device="cuda")
res = {}


def bwd(dk, dq, dv):
    rt.check(L.hupr_attn_bwd_bf16in(rt.ptr(kb), rt.ptr(qb), rt.ptr(vb), rt.ptr(gb), rt.ptr(v), rt.ptr(out), rt.ptr(g32), rt.ptr(lse),
                                    rt.ptr(dk), rt.ptr(dq), rt.ptr(dv), rt.ptr(scr), B, N, C, 1, rt.stream()))


for rnd in range(3):
    for mode in (1, 0):
        L.hupr_debug_attn_dkv512(mode)
        dk, dq, dv = (torch.empty(B, N, C, device="cuda") for _ in range(3))
        for _ in range(3):
            bwd(dk, dq, dv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            bwd(dk, dq, dv)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e2
        res[mode] = (dk, dv)
        print("round %d backward (prep + dQ + dK/dV), dK/dV on %s: %.1f us (%.0f TF/s algorithmic)" %
              (rnd, "512 threads, one barrier per tile" if mode else "256 threads, two barriers per tile", t, 10.0 * N * N * C * B / t / 1e6))
L.hupr_debug_attn_dkv512(1)
print("dK identical: %s, dV identical: %s" % (torch.equal(res[0][0], res[1][0]), torch.equal(res[0][1], res[1][1])))
