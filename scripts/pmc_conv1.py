"""Layer-1 halo convolution (64 -> 64, 3 x 3 x 3, B = 32, bf16 activations) for rocprofv3 --pmc: four launches of each form the step uses —
the forward with fused BatchNorm statistics (<4, 8, 8, 3, 1>), the plain input-gradient form (<4, 8, 8, 3, 0>) and that form with a
prefetched residual (the second input gradient of a block's convolution pair)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
x = torch.randn(32, 8, 64, 64, 64, device="cuda").bfloat16(); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
res = torch.randn(32, 8, 64, 64, 64, device="cuda").bfloat16()
for _ in range(4):
    F_._conv_raw(x, w, 0, None, None, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64), stats=True)
    F_._conv_stats.clear()
for _ in range(4):
    F_._conv_raw(x, w, 1, None, None, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64))
for _ in range(4):
    F_._conv_raw(x, w, 1, None, res, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64))
torch.cuda.synchronize()
