"""The level-1 streaming temporal-merge kernels a few times, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (scripts/pmc_run.sh tmerge scripts/pmc_tmerge.py):
algorithmic bytes per call = x 134.2 MB (bf16) + merged map / its gradient 33.6 MB (fp32) = 167.8 MB each way."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
x = torch.randn(32, 8, 64, 64, 64, device="cuda").bfloat16().requires_grad_(True)
w = (torch.randn(64, 64, 8, 1, 1, device="cuda") * 0.05).requires_grad_(True)
g = torch.randn(32, 1, 64, 64, 64, device="cuda")
for _ in range(4):
    x.grad = None; w.grad = None
    F_.TemporalMergeFn.apply(x, w).backward(g)
torch.cuda.synchronize()
