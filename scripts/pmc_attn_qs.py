"""Level-1 attention forward + backward (C = 64, N = 4096, B = 32, bf16 operands), QS kernels (the round-5 default), 4x for rocprofv3
--pmc / SQ counters:  bash scripts/pmc_sq_run.sh r05_attn scripts/pmc_attn_qs.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L, rt = F_.rt.lib(), F_.rt
B, N, C = 32, 4096, 64
g = torch.Generator(device="cuda").manual_seed(11)
k, q, v = (torch.randn(B, N, C, device="cuda", generator=g) * s for s in (0.5, 0.5, 1.0))
kb, qs, vb = k.bfloat16(), (q * 1.4426950408889634).bfloat16(), v.bfloat16()
g32 = torch.randn(B, N, C, device="cuda", generator=g); gb = g32.bfloat16()
out, lse = torch.empty(B, N, C, device="cuda"), torch.empty(B, N, device="cuda")
dk, dq, dv = (torch.empty(B, N, C, device="cuda") for _ in range(3))
scr = torch.empty(B, N, device="cuda")
P = rt.ptr
for _ in range(4):
    rt.check(L.hupr_attn_fwd_bf16in_ld_ws_qs(P(kb), C, P(qs), C, P(vb), P(v), P(out), P(lse), None, 0, B, N, C, None, 0, rt.stream()))
    rt.check(L.hupr_attn_bwd_bf16in_ld_qs(P(kb), C, P(qs), C, P(vb), P(gb), C, P(v), P(out), P(g32), P(lse), P(dk), C, P(dq), C, P(dv), P(scr), B, N, C, 1, 0, rt.stream()))
torch.cuda.synchronize()
