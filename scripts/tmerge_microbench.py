"""Temporal-merge GEMMs (Conv3d (G,1,1) over the frame axis, reference models/layers.py:195-197) on their own: forward, input
gradient, weight gradient at the three encoder levels, with the HBM bytes each must move.
usage: python scripts/tmerge_microbench.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B in [int(a) for a in sys.argv[1:]] or [32, 8]:
    for (G, H, C) in ((8, 64, 64), (4, 32, 128), (2, 16, 256)):
        x = torch.randn(B, G, H, H, C, device="cuda").bfloat16().requires_grad_(True)
        w = (torch.randn(C, C, G, 1, 1, device="cuda") * 0.05).requires_grad_(True)
        y = F_.TemporalMergeFn.apply(x, w)
        g = torch.randn_like(y)
        fwd = timeit(lambda: F_.TemporalMergeFn.apply(x, w))
        def bwd():
            x.grad = None; w.grad = None
            y.backward(g, retain_graph=True)
        both = timeit(bwd)
        xb, yb = x.numel() * 2, y.numel() * 4
        print("B=%d G=%d %dx%d C=%d: fwd %.1f us (%.2f TB/s of %d MB) | dgrad + wgrad %.1f us (%.2f TB/s of %d MB)" %
              (B, G, H, H, C, fwd, (xb + yb) / fwd / 1e6, (xb + yb) >> 20, both, (2 * xb + 2 * yb) / both / 1e6, (2 * xb + 2 * yb) >> 20))
