"""A/B of the LDS-DMA weight-gradient kernel's workgroup count per (kz plane, tile pair): the three depth-tap planes of one
spatial tile sequence re-read the same x / dy tiles, so they should share an XCD's L2 (workgroup id -> XCD is id % 8:
a group count that is a multiple of 8 puts planes id, id + gw, id + 2 gw on one XCD).
usage: python scripts/wgrad_groups_ab.py [gw ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
gws = [int(a) for a in sys.argv[1:]] or [0, 85, 80, 72, 64, 88]
shapes = {"l1 64>64 @8x64x64": (64, 64, 8, 64, 64, 3), "l1.0 32>64 @8x64x64": (32, 64, 8, 64, 64, 3), "l2 128>128 @4x32x32": (128, 128, 4, 32, 32, 3),
          "l2.1 64>128 @4x32x32": (64, 128, 4, 32, 32, 3), "dec1.0 320>64 @64x64": (320, 64, 1, 64, 64, 1), "dec2.0 640>128 @32x32": (640, 128, 1, 32, 32, 1)}
for name, (Ci, Co, D, H, W, kd) in shapes.items():
    x = torch.randn(32, D, H, W, Ci, device="cuda").bfloat16(); dy = torch.randn(32, D, H, W, Co, device="cuda").bfloat16()
    dw = torch.empty(Co, Ci, kd, 3, 3, device="cuda")
    ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), x.device)
    run = lambda: F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), 32, D, H, W, Ci, Ci, Co, Co, kd, F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
    out, ref = [], None
    for rnd in range(2):
        for gw in gws:
            L.hupr_debug_wgrad_groups(gw)
            for _ in range(2): run()
            torch.cuda.synchronize()
            if ref is None: ref = dw.clone()
            err = ((dw - ref).abs().max() / ref.abs().max()).item()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): run()
            e.record(); torch.cuda.synchronize()
            if rnd: out.append("gw %d: %.0f us (rel diff %.1e)" % (gw, s.elapsed_time(e) / 10 * 1e3, err))
    L.hupr_debug_wgrad_groups(0)
    print(name, " | ".join(out))
