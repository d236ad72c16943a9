"""Attention forward (C = 64, N = 4096) against the number of workgroups per CU: 66.6 us with one, then ~50 us per additional
workgroup per CU — each workgroup takes ~50 us of its CU whatever shares it: the kernel is throughput-bound on the sum of its
VALU, MFMA and LDS issue, not latency-bound.  usage: python scripts/attn_occupancy.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
N, C = 4096, 64
for B in (2, 4, 8, 16, 24, 32, 48, 64):
    k, q, v = (torch.randn(B, N, C, device="cuda").bfloat16() for _ in range(3))
    v32 = v.float(); out = torch.empty(B, N, C, device="cuda"); lse = torch.empty(B, N, device="cuda")
    def fwd(): F_.rt.check(L.hupr_attn_fwd_bf16in(F_.rt.ptr(k), F_.rt.ptr(q), F_.rt.ptr(v), F_.rt.ptr(v32), F_.rt.ptr(out), F_.rt.ptr(lse), B, N, C, F_.rt.stream()))
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fwd()
    e.record(); torch.cuda.synchronize()
    print("B=%d: %d workgroups (%.2f per CU): %.1f us" % (B, B * 32, B * 32 / 256, s.elapsed_time(e) / 10 * 1e3))
