"""Times the 1x1 key-point head (32 -> 16 channels, B = 32: 131 072 voxels): dedicated fp32 kernels vs the generic fp32 / bf16
convolution path, forward and backward separately.  usage: python scripts/head_microbench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


x = torch.randn(32, 1, 64, 64, 32, device="cuda", requires_grad=True)
w = torch.zeros(16, 32, 1, 1, device="cuda")
w[:14] = torch.randn(14, 32, 1, 1, device="cuda") * 0.2
w.requires_grad_(True)
gy = torch.randn(32, 1, 64, 64, 16, device="cuda")
for name, math, fn in (("dedicated fp32 kernels", "f32", lambda: F_.Head1x1Fn.apply(x, w)),
                       ("generic convolution, fp32 pipe", "f32", lambda: F_.conv(x, w, None, None, (0, 0, 0))),
                       ("generic convolution, bf16 pipe", "bf16", lambda: F_.conv(x, w, None, None, (0, 0, 0)))):
    F_.set_math(math)
    t_f = timeit(lambda: fn())
    y = fn()
    t_b = timeit(lambda: torch.autograd.grad(y, (x, w), gy, retain_graph=True))
    print("%-34s forward %.1f us, backward (dx + dw) %.1f us" % (name, t_f, t_b))
F_.set_math("f32")
