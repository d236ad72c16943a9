"""Layer-1 halo conv (64->64 3x3x3, B=32, bf16 activations) 4x for rocprofv3 --pmc; argv[1] = ablate bits (8 = no XCD map)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
F_.rt.lib().hupr_debug_halo_ablate(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
x = torch.randn(32, 8, 64, 64, 64, device="cuda").bfloat16(); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
for _ in range(4):
    F_._conv_raw(x, w, 0, None, None, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64))
torch.cuda.synchronize()
