"""A/B of the two halo conv kernels and phase ablation on the dominant shapes (one process, interleaved rounds)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
shapes = {"l1 64>64 @8x64x64": (64, 64, 8, 64, 64, (3, 3, 3), (1, 1, 1)), "l2 128>128 @4x32x32": (128, 128, 4, 32, 32, (3, 3, 3), (1, 1, 1)),
          "dec1.0 320>64 @64x64": (320, 64, 1, 64, 64, (1, 3, 3), (0, 1, 1)), "dec2.0 640>128 @32x32": (640, 128, 1, 32, 32, (1, 3, 3), (0, 1, 1))}
ACT = torch.bfloat16 if "--bf16act" in sys.argv else torch.float32
for name, (Ci, Co, D, H, W, k, pad) in shapes.items():
    x = torch.randn(32, D, H, W, Ci, device="cuda").to(ACT); w = torch.randn(Co, Ci, *k, device="cuda") * 0.05
    flop = 2.0 * 32 * D * H * W * Co * Ci * k[0] * 9
    res = {}
    for rnd in range(3):
        for variant, bits, label in ((1, 0, "v128 full"), (2, 0, "v256 full"), (2, 256, "v256 old-stage-protocol full"), (2, 512, "v256 with 16x16x32 MFMAs (timing only)"), (2, 512 + 5, "16x16x32 skeleton"), (2, 256 + 5, "old-protocol skeleton"), (2, 2, "v256 immediate-epilogue"), (2, 8, "v256 no-xcd-map"), (2, 1, "v256 no-fill"), (2, 4, "v256 no-store"), (2, 5, "v256 skeleton"), (2, 5 + 16, "skel no-wstage"), (2, 5 + 64, "skel frags-once"), (2, 5 + 112, "skel mfma-only")):
            L.hupr_debug_halo_variant(variant); L.hupr_debug_halo_ablate(bits)
            for _ in range(2): F_._conv_raw(x, w, 0, None, None, Co, k, pad, (D, H, W))
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): F_._conv_raw(x, w, 0, None, None, Co, k, pad, (D, H, W))
            e.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(s.elapsed_time(e) / 10 * 1e3)
    L.hupr_debug_halo_ablate(0); L.hupr_debug_halo_variant(0)
    print(name, " | ".join("%s %.0f us%s" % (k_, min(v), " (%.0f TF/s)" % (flop / min(v) / 1e6) if "full" in k_ else "") for k_, v in res.items()))
