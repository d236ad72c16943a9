"""128- / 256- / 512-voxel halo convolution kernels on the 3-D encoder shapes (B = 32, bf16 activations), interleaved rounds.
HUPR_HALO512_ALL=1 lets the 512-voxel kernel run on every shape of its envelope (the dispatcher uses it for Ci = 32 only)."""
import os, sys
os.environ["HUPR_HALO512_ALL"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
shapes = {"l1.0 32>64 @8x64x64": (32, 64, 8, 64, 64), "l1 64>64": (64, 64, 8, 64, 64), "l2.1 64>128 @4x32x32": (64, 128, 4, 32, 32), "l2 128>128": (128, 128, 4, 32, 32)}
for name, (Ci, Co, D, H, W) in shapes.items():
    x = torch.randn(32, D, H, W, Ci, device="cuda").bfloat16(); w = torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.05
    flop = 2.0 * 32 * D * H * W * Co * Ci * 27
    res = {}
    ys = {}
    for rnd in range(3):
        for variant, label in ((1, "v128"), (2, "v256"), (0, "v512")):
            L.hupr_debug_halo_variant(variant)
            for _ in range(2): y = F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
            torch.cuda.synchronize()
            ys[label] = y.float()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
            e.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(s.elapsed_time(e) / 10 * 1e3)
    L.hupr_debug_halo_variant(0)
    d = (ys["v512"] - ys["v128"]).abs().max().item() / ys["v128"].abs().max().item()
    print(name, " | ".join("%s %.0f us (%.0f TF/s)" % (k, min(v), flop / min(v) / 1e6) for k, v in res.items()), "| v512 vs v128 rel max diff %.2e" % d)
