"""First-layer weight gradient (32 -> 64 channels, 8 x 64 x 64, B = 32, bf16-stored activations): K quarters on the 16 x 16 x 32 kernel
(hupr_debug_wgrad_ci32(1), round 6) against the 32 x 32 x 16 kernel (mode 3); kernel + split-K reduction, HIP-event mean of 20 calls each,
interleaved.  usage (GPU box, repo root): python scripts/wgrad_ci32_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
B, Ci, Co, D, H, W, kd = 32, 32, 64, 8, 64, 64, 3
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, D, H, W, Ci, device="cuda", generator=g).bfloat16()
dy = torch.randn(B, D, H, W, Co, device="cuda", generator=g).bfloat16()
dw = torch.empty(Co, Ci, kd, 3, 3, device="cuda")
ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), x.device)
call = lambda: F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, kd,
                                                             F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
def run(mode, n=20):
    L.hupr_debug_wgrad_ci32(mode)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gf = 2 * 27 * Ci * Co * B * D * H * W / 1e9
for rep in range(3):
    a, b = run(3), run(1)
    print("32 x 32 x 16 kernel %.1f us (%.0f TF/s)   16 x 16 x 32 kernel %.1f us (%.0f TF/s)   [incl. the split-K reduction]" % (a, gf / a * 1e3, b, gf / b * 1e3))
L.hupr_debug_wgrad_ci32(1)
