"""Per-layer timing of the convolution kernels at the bench batch (B=32): TF/s for forward (== input-gradient
kernel) and weight gradient, in the fp32 engine, the bf16 implicit-GEMM engine and the bf16 halo kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [  # name, Ci, Co, D, H, W, k, pad
    ("enc.l1 64>64 3^3 @8x64x64", 64, 64, 8, 64, 64, (3, 3, 3), (1, 1, 1)),
    ("enc.l1.0 32>64 3^3", 32, 64, 8, 64, 64, (3, 3, 3), (1, 1, 1)),
    ("enc.l2 128>128 3^3 @4x32x32", 128, 128, 4, 32, 32, (3, 3, 3), (1, 1, 1)),
    ("enc.l3 256>256 3^3 @2x16x16", 256, 256, 2, 16, 16, (3, 3, 3), (1, 1, 1)),
    ("dec1.0 320>64 3x3 @64x64", 320, 64, 1, 64, 64, (1, 3, 3), (0, 1, 1)),
    ("dec3.0 1024>256 3x3 @16x16", 1024, 256, 1, 16, 16, (1, 3, 3), (0, 1, 1)),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for name, Ci, Co, D, H, W, k, pad in SHAPES:
    x = torch.randn(B, D, H, W, Ci, device="cuda")
    w = torch.randn(Co, Ci, *k, device="cuda") * 0.05
    dy = torch.randn(B, D, H, W, Co, device="cuda")
    flop = 2.0 * B * D * H * W * Co * Ci * k[0] * k[1] * k[2]
    row = [name]
    for mode, halo in (("f32", False), ("bf16", False), ("bf16", True)):
        F_.set_math(mode)
        F_.USE_HALO = halo
        t = timeit(lambda: F_._conv_raw(x, w, 0, None, None, Co, k, pad, (D, H, W)))
        row.append("%s%s fwd %.3f ms %.0f TF/s" % (mode, "+halo" if halo else "", t * 1e3, flop / t / 1e12))
    for mode in ("f32", "bf16"):
        F_.set_math(mode)
        ws = F_.workspace(F_.rt.lib().hupr_conv_wgrad_ws_bytes(B, D, H, W, Ci, Co, *k), x.device)
        dw = torch.empty_like(w)
        fn = F_._fn("conv_wgrad")
        t = timeit(lambda: F_.rt.check(fn(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), B, D, H, W, Ci, Ci, D, H, W, Co, Co,
                                         k[0], k[1], k[2], pad[0], pad[1], pad[2], F_.rt.ptr(ws), ws.numel(), F_.rt.stream())))
        row.append("%s wgrad %.3f ms %.0f TF/s" % (mode, t * 1e3, flop / t / 1e12))
    if Ci % 64 == 0:
        L = F_.rt.lib()
        ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, k[0]), x.device)
        dw = torch.empty_like(w)
        t = timeit(lambda: F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), B, D, H, W, Ci, Ci,
                                                                      Co, Co, k[0], F_.rt.ptr(ws), ws.numel(), F_.rt.stream())))
        row.append("bf16+halo wgrad %.3f ms %.0f TF/s" % (t * 1e3, flop / t / 1e12))
    print(" | ".join(row), flush=True)
F_.set_math("f32")
F_.USE_HALO = True
