"""First-layer forward shape (32 -> 64 channels + bias, 8 x 64 x 64, B = 32): the 256-voxel kernel's 64-byte-row form against the 128-voxel kernel
(hupr_debug_halo_tiles(15); until round 6 a 512-voxel kernel of its own took this shape: profiles/r06_conv_ci32_ab.txt); HIP-event mean of 20
launches each, interleaved.  usage (GPU box, repo root): python scripts/conv_ci32_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
B, Ci, Co, D, H, W = 32, 32, 64, 8, 64, 64
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, D, H, W, Ci, device="cuda", generator=g).bfloat16()
w = torch.randn(Co, Ci, 3, 3, 3, device="cuda", generator=g) * (Ci * 27) ** -0.5
bias = torch.randn(Co, device="cuda", generator=g)
def run(mode, n=20):
    L.hupr_debug_halo_tiles(mode)
    for _ in range(3):
        F_._conv_raw(x, w, 0, bias, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        F_._conv_raw(x, w, 0, bias, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gf = 2 * 27 * Ci * Co * B * D * H * W / 1e9
for rep in range(3):
    a, b = run(15), run(31)
    print("128-voxel kernel %.1f us (%.0f TF/s)   256-voxel kernel, 64-byte rows %.1f us (%.0f TF/s)" % (a, gf / a * 1e3, b, gf / b * 1e3))
L.hupr_debug_halo_tiles(31)
