"""Flash attention kernels at the three MSCSA levels (B = 32).  usage: python scripts/attn_microbench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
for N, C in ((4096, 64), (1024, 128)):
    k, q, v = (torch.randn(32, N, C, device="cuda", requires_grad=True) for _ in range(3))
    g = torch.randn(32, N, C, device="cuda")
    def fwd(): return F_.AttentionFn.apply(k, q, v, True)
    def both():
        o = fwd(); o.backward(g)
    res = []
    for name, fn, flop in (("fwd", fwd, 4.0), ("fwd+bwd", both, 14.0)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        res.append("%s %.0f us (%.0f TF/s)" % (name, us, flop * 32 * N * N * C / us / 1e6))
    print("N=%d C=%d: %s" % (N, C, " | ".join(res)))
