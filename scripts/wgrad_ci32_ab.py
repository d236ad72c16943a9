"""A/B of the Ci <= 32 mode of the LDS-DMA weight-gradient kernel (K quarters instead of an idle ci quadrant) against the
two-quadrant kernel on the same inputs, plus an fp64 torch check of both.
usage: python scripts/wgrad_ci32_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
shapes = {"l1.0 32>64 @8x64x64 B32": (32, 32, 64, 8, 64, 64, 3), "32>64 @2x16x16 B3": (3, 32, 64, 2, 16, 16, 3), "24>40 @1x16x32 B2 (2-D)": (2, 24, 40, 1, 16, 32, 1),
          "64>64 @8x64x64 B32 (unaffected)": (32, 64, 64, 8, 64, 64, 3)}
for name, (B, Ci, Co, D, H, W, kd) in shapes.items():
    torch.manual_seed(0)
    x = torch.randn(B, D, H, W, Ci, device="cuda").bfloat16(); dy = torch.randn(B, D, H, W, Co, device="cuda").bfloat16()
    dw = torch.empty(Co, Ci, kd, 3, 3, device="cuda")
    ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), x.device)
    run = lambda: F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, kd, F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
    ref = None
    if B <= 3:      # fp64 reference on small shapes
        xs = x.double().permute(0, 4, 1, 2, 3).requires_grad_(False)
        w = torch.zeros(Co, Ci, kd, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv3d(xs, w, padding=(kd // 2, 1, 1))
        y.backward(dy.double().permute(0, 4, 1, 2, 3))
        ref = w.grad
    res, out = {}, []
    for rnd in range(2):
        for mode in (0, 1):
            L.hupr_debug_wgrad_ci32(mode)
            for _ in range(2): run()
            torch.cuda.synchronize()
            res[mode] = dw.clone()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): run()
            e.record(); torch.cuda.synchronize()
            if rnd: out.append("ci32=%d: %.0f us" % (mode, s.elapsed_time(e) / 10 * 1e3))
    L.hupr_debug_wgrad_ci32(1)
    d = ((res[0] - res[1]).abs().max() / res[0].abs().max()).item()
    msg = "%s | %s | rel diff between modes %.1e" % (name, " | ".join(out), d)
    if ref is not None:
        msg += " | vs fp64: " + ", ".join("%.1e" % ((res[m].double() - ref).abs().max() / ref.abs().max()).item() for m in (0, 1))
    print(msg)
