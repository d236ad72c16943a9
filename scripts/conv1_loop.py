"""The dominant kernel alone, for a rocprofv3 summary whose average IS the per-shape average: Encoder3D.layer1 64->64 3x3x3 at the
bench batch (B = 32, bf16 activations), 40 launches.  usage: rocprofv3 --kernel-trace --stats -d out -o run -- python scripts/conv1_loop.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
x = torch.randn(32, 8, 64, 64, 64, device="cuda").bfloat16(); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
for _ in range(40):
    F_._conv_raw(x, w, 0, None, None, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64))
torch.cuda.synchronize()
