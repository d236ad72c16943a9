"""The 256-voxel halo convolution on v_mfma_f32_16x16x32_bf16 (csrc/conv_halo256m_bf16.hip) against the 32 x 32 x 16 kernel:
results (against each other and fp64 on one sample), fused statistics, residual epilogue, times (interleaved).
usage (GPU box): python scripts/halo_m16_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
shapes = {"l1 64>64 @8x64x64 B32": (32, 64, 64, 8, 64, 64), "l2 128>128 @4x32x32 B32": (32, 128, 128, 4, 32, 32), "64>128 @8x32x64 B3": (3, 64, 128, 8, 32, 64),
          "l3 256>256 @2x16x16 B32 (2 x 8 x 16 tile; before: the 128-voxel kernel)": (32, 256, 256, 2, 16, 16), "l3 256>256 @2x16x32 B17": (17, 256, 256, 2, 16, 32),
          "dec1 320>64 @64x64 B32 (1 x 16 x 16 tile, 1x3x3 taps; before: the 128-voxel kernel)": (32, 320, 64, 1, 64, 64),
          "dec1 dgrad 64>320 @64x64 B32": (32, 64, 320, 1, 64, 64), "dec2 640>128 @32x32 B32": (32, 640, 128, 1, 32, 32), "dec2 dgrad 128>640 @32x32 B32": (32, 128, 640, 1, 32, 32)}
for name, (B, Ci, Co, D, H, W) in shapes.items():
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, D, H, W, Ci, device="cuda", generator=g).relu().bfloat16()
    K3, PAD = ((3, 3, 3), (1, 1, 1)) if D > 1 else ((1, 3, 3), (0, 1, 1))
    w = torch.randn(Co, Ci, *K3, device="cuda", generator=g) * (Ci * 9 * K3[0]) ** -0.5
    res = torch.randn(B, D, H, W, Co, device="cuda", generator=g).bfloat16()
    out = {}
    for m16 in (0, 1):
        L.hupr_debug_halo_m16(3 * m16)      # 3: every tile of the 16 x 16 x 32 kernel, the opt-in 1 x 16 x 16 one included
        y = F_._conv_raw(x, w, 0, None, None, Co, K3, PAD, (D, H, W))
        yr = F_._conv_raw(x, w, 0, None, res, Co, K3, PAD, (D, H, W))
        out[m16] = (y.float(), yr.float())
    d = (out[0][0] - out[1][0]).abs()
    ref = torch.nn.functional.conv3d(x[:1].double().permute(0, 4, 1, 2, 3), w.bfloat16().double(), None, 1, PAD).permute(0, 2, 3, 4, 1)
    e0, e1 = ((out[m][0][:1].double() - ref).abs().max().item() for m in (0, 1))
    print("%s: 16x16x32 vs 32x32x16 max-abs %.3e (%.2f %% of the outputs differ, scale %.2f); vs fp64 on sample 0: %.3e / %.3e; with residual max-abs %.3e" %
          (name, d.max().item(), 100.0 * (d > 0).float().mean().item(), out[0][0].abs().max().item(), e1, e0, (out[0][1] - out[1][1]).abs().max().item()))
    t = {0: [], 1: []}
    for rnd in range(3):
        for m16 in (0, 1):
            L.hupr_debug_halo_m16(3 * m16)
            for _ in range(2): F_._conv_raw(x, w, 0, None, None, Co, K3, PAD, (D, H, W))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): F_._conv_raw(x, w, 0, None, None, Co, K3, PAD, (D, H, W))
            e.record(); torch.cuda.synchronize()
            t[m16].append(s.elapsed_time(e) * 100)
    fl = 2.0 * B * D * H * W * Co * Ci * 9 * K3[0]
    print("    time: 32x32x16 %.1f us (%.0f TF/s)   16x16x32 %.1f us (%.0f TF/s)" % (min(t[0]), fl / min(t[0]) / 1e6, min(t[1]), fl / min(t[1]) / 1e6))
L.hupr_debug_halo_m16(1)
