"""One fused MSCSA level (forward + backward) at the bench batch for rocprofv3 --pmc: argv[1] = level (1: C = 64, N = 4096; 2: C = 128,
N = 1024; 3: C = 256, N = 256).  usage: bash scripts/pmc_sq_run.sh r06_attn_l2 scripts/pmc_attn_level.py 2"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
C, H = {1: (64, 64), 2: (128, 32), 3: (256, 16)}[lvl]
B = 32
g = torch.Generator(device="cuda").manual_seed(0)
ra = torch.randn(B, 1, H, H, C, device="cuda", generator=g).requires_grad_(True)
re = torch.randn(B, 1, H, H, C, device="cuda", generator=g).requires_grad_(True)
ws = [(torch.randn(C, C, 1, 1, device="cuda", generator=g) * C ** -0.5).requires_grad_(True) for _ in range(8)]
for _ in range(3):
    (cat,) = F_.MSCSALevelFn.apply(ra, re, 1, *ws)
    cat.backward(torch.randn_like(cat))
torch.cuda.synchronize()
