"""Level-1 attention forward (C = 64, N = 4096, B = 32, bf16 operands) 4x for rocprofv3 --pmc (scripts/pmc_run.sh r04_attn scripts/pmc_attn.py [0|1]):
argv[1] = 0 selects the rounds-1-3 kernel, default the round-4 kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L, rt = F_.rt.lib(), F_.rt
B, N, C = 32, 4096, 64
k, q, v = (torch.randn(B, N, C, device="cuda") for _ in range(3))
kb, qb, vb = (k * 0.5).bfloat16(), (q * 0.5).bfloat16(), v.bfloat16()
out = torch.empty(B, N, C, device="cuda"); lse = torch.empty(B, N, device="cuda")
for _ in range(4):
    rt.check(L.hupr_attn_fwd_bf16in(rt.ptr(kb), rt.ptr(qb), rt.ptr(vb), rt.ptr(v), rt.ptr(out), rt.ptr(lse), B, N, C, rt.stream()))
torch.cuda.synchronize()
