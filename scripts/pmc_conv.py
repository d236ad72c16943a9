"""Launch the dominant kernel (halo conv, Encoder3D.layer1 64->64 3x3x3 at B=32) a few times so rocprofv3 --pmc can
attribute HBM bytes to it.  Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o p -- python scripts/pmc_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"      # f32 | bf16 (fp32 activations) | bf16act (bf16 activations)
F_.set_math("f32" if mode == "f32" else "bf16")
B, C, D, H, W = 32, 64, 8, 64, 64
act = torch.bfloat16 if mode == "bf16act" else torch.float32
x = torch.randn(B, D, H, W, C, device="cuda").to(act)
w = torch.randn(C, C, 3, 3, 3, device="cuda") * 0.05
dy = torch.randn(B, D, H, W, C, device="cuda").to(act)
for _ in range(4):
    F_._conv_raw(x, w, 0, None, None, C, (3, 3, 3), (1, 1, 1), (D, H, W))
L = F_.rt.lib()
if mode != "f32":
    fn = L.hupr_conv3x3_wgrad_halo_bf16act if mode == "bf16act" else L.hupr_conv3x3_wgrad_halo_bf16
    ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(C, C, 3), x.device)
    dw = torch.empty_like(w)
    for _ in range(4):
        F_.rt.check(fn(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), B, D, H, W, C, C, C, C, 3,
                                                   F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
torch.cuda.synchronize()
