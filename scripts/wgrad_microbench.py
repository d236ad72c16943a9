"""Weight-gradient kernels of the 3-D encoder layers on bf16-stored activations (the LDS-DMA kernel) vs fp32-stored
(register-staged kernel).  usage: python scripts/wgrad_microbench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
shapes = {"l1 64>64 @8x64x64": (64, 64, 8, 64, 64, 3), "l1.0 32>64 @8x64x64": (32, 64, 8, 64, 64, 3), "l2 128>128 @4x32x32": (128, 128, 4, 32, 32, 3),
          "l3 256>256 @2x16x16": (256, 256, 2, 16, 16, 3), "dec1.0 320>64 @64x64": (320, 64, 1, 64, 64, 1)}
for name, (Ci, Co, D, H, W, kd) in shapes.items():
    res = []
    for act in (torch.float32, torch.bfloat16):
        x = torch.randn(32, D, H, W, Ci, device="cuda").to(act); dy = torch.randn(32, D, H, W, Co, device="cuda").to(act)
        dw = torch.empty(Co, Ci, kd, 3, 3, device="cuda")
        ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), x.device)
        fn = L.hupr_conv3x3_wgrad_halo_bf16act if act == torch.bfloat16 else L.hupr_conv3x3_wgrad_halo_bf16
        run = lambda: F_.rt.check(fn(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), 32, D, H, W, Ci, Ci, Co, Co, kd, F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
        for _ in range(3): run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        res.append("%s %.0f us (%.0f TF/s incl. reduce)" % ("bf16act" if act == torch.bfloat16 else "f32act", us, 2.0 * 32 * D * H * W * Co * Ci * kd * 9 / us / 1e6))
    print(name, " | ".join(res))
