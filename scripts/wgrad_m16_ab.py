"""Weight gradient of the halo convolutions on bf16 activations at the bench batch: the round-5 kernel on v_mfma_f32_16x16x32_bf16
(hupr_k_wgrad_halo_m16) against the rounds-2-4 kernel on v_mfma_f32_32x32x16_bf16 (hupr_debug_wgrad_m16(0)), per call incl. the
split-K reduction, interleaved.   usage: python scripts/wgrad_m16_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L, rt = F_.rt.lib(), F_.rt
shapes = {"l1 64>64 @8x64x64": (64, 64, 8, 64, 64, 3), "l2.0 64>128 @4x32x32": (64, 128, 4, 32, 32, 3), "l2 128>128 @4x32x32": (128, 128, 4, 32, 32, 3),
          "l3.0 128>256 @2x16x16": (128, 256, 2, 16, 16, 3), "l3 256>256 @2x16x16": (256, 256, 2, 16, 16, 3),
          "dec3.0 1024>256 @16x16": (1024, 256, 1, 16, 16, 1), "dec2.0 640>128 @32x32": (640, 128, 1, 32, 32, 1),
          "dec1.0 320>64 @64x64": (320, 64, 1, 64, 64, 1), "dec1 64>64 @64x64": (64, 64, 1, 64, 64, 1)}
tot = {0: 0.0, 1: 0.0}
for name, (Ci, Co, D, H, W, kd) in shapes.items():
    x = torch.randn(32, D, H, W, Ci, device="cuda").bfloat16(); dy = torch.randn(32, D, H, W, Co, device="cuda").bfloat16()
    dw = torch.empty(Co, Ci, kd, 3, 3, device="cuda")
    ws = torch.empty(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), dtype=torch.uint8, device="cuda")
    run = lambda: rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), 32, D, H, W, Ci, Ci, Co, Co, kd, rt.ptr(ws), ws.numel(), rt.stream()))
    best = {0: 1e9, 1: 1e9}
    for rnd in range(3):
        for m16 in (1, 0):
            L.hupr_debug_wgrad_m16(m16)
            for _ in range(3): run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): run()
            e.record(); torch.cuda.synchronize()
            best[m16] = min(best[m16], s.elapsed_time(e) / 10 * 1e3)
    L.hupr_debug_wgrad_m16(1)
    fl = 2.0 * 32 * D * H * W * Co * Ci * kd * 9
    tot[0] += best[0]; tot[1] += best[1]
    print("%-24s 32x32x16 %.1f us (%.0f TF/s) | 16x16x32 %.1f us (%.0f TF/s)  %+.1f %%" % (name, best[0], fl / best[0] / 1e6, best[1], fl / best[1] / 1e6,
                                                                                          100.0 * (best[1] - best[0]) / best[0]), flush=True)
print("sum %.1f -> %.1f us" % (tot[0], tot[1]))
