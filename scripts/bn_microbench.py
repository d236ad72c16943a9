"""BatchNorm training passes on the encoder shapes (bf16 activations, B = 32): statistics, apply + ReLU, backward (two-branch tail).
usage: python scripts/bn_microbench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best * 1e3
for name, (D, H, W, C) in {"l1 8x64x64x64": (8, 64, 64, 64), "l2 4x32x32x128": (4, 32, 32, 128), "l3 2x16x16x256": (2, 16, 16, 256)}.items():
    x = torch.randn(32, D, H, W, C, device="cuda").bfloat16()
    bn = torch.nn.BatchNorm3d(C).cuda()
    nbytes = x.numel() * 2
    t_stats = timed(lambda: F_._bn_params(x, bn, True))
    sc, sh, mean, invstd = F_._bn_params(x, bn, True)
    y = torch.empty_like(x)
    L = F_.rt.lib(); rt = F_.rt
    t_apply = timed(lambda: rt.check(L.hupr_scale_shift_act_bf16act(rt.ptr(x), rt.ptr(sc), rt.ptr(sh), None, None, None, rt.ptr(y), x.numel() // C, C, 1, rt.stream())))
    print("%s: statistics %.1f us (%.2f TB/s) | apply+ReLU %.1f us (%.2f TB/s)" % (name, t_stats, nbytes / t_stats / 1e6, t_apply, 2 * nbytes / t_apply / 1e6))
