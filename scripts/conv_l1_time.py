"""Layer-1 shape (64 -> 64 channels, 3 x 3 x 3, 8 x 64 x 64, B = 32) and level-2 / level-3 shapes of the 256-voxel convolution kernel, kernels alone: HIP-event mean of
20 launches + a checksum of the output (A/B two builds with HUPR_LIB_PATH=<other libhupr_hip.so>: builds that claim identical arithmetic must
print identical checksums).  usage (GPU box, repo root): python scripts/conv_l1_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
for B, Ci, Co, D, H, W, res in ((32, 64, 64, 8, 64, 64, False), (32, 64, 64, 8, 64, 64, True), (32, 128, 128, 4, 32, 32, False), (32, 256, 256, 2, 16, 16, False)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, D, H, W, Ci, device="cuda", generator=g).bfloat16()
    w = torch.randn(Co, Ci, 3, 3, 3, device="cuda", generator=g) * (Ci * 27) ** -0.5
    r = torch.randn(B, D, H, W, Co, device="cuda", generator=g).bfloat16() if res else None
    conv = lambda: F_._conv_raw(x, w, 0, None, r, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
    best = 1e9
    for rep in range(3):
        for _ in range(3): conv()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = conv()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    gf = 2 * 27 * Ci * Co * B * D * H * W / 1e9
    print("%3d -> %3d @ %d x %d x %d%s: %.1f us (%.0f TF/s) | checksum %.9e" % (Ci, Co, D, H, W, " + residual" if res else "", best, gf / best * 1e3, y.double().sum().item()))
