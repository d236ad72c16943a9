"""Launch the layer-1 weight gradient (64->64 3x3x3, B=32, bf16 activations) 4x with a forced group count so rocprofv3 --pmc
can attribute HBM bytes.  usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o p -- python scripts/pmc_wgrad.py <gw>"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
B, C, D, H, W = 32, 64, 8, 64, 64
x = torch.randn(B, D, H, W, C, device="cuda").bfloat16(); dy = torch.randn(B, D, H, W, C, device="cuda").bfloat16()
dw = torch.empty(C, C, 3, 3, 3, device="cuda")
ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(C, C, 3), x.device)
for _ in range(4):
    F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), B, D, H, W, C, C, C, C, 3, F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
torch.cuda.synchronize()
