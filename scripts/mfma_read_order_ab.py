"""A/B inside the 256-voxel halo convolution: fragment reads of K-step ks + 1 spread in front of the MFMAs of ks (default) against
the round-1 order (one burst of seven reads, then six MFMAs; compile-time ablation 8 = ablate 128).  Same instructions, same bits.
usage: python scripts/mfma_read_order_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
for name, (Ci, Co, D, H, W) in {"l1 64>64": (64, 64, 8, 64, 64), "l2 128>128": (128, 128, 4, 32, 32)}.items():
    k, pad = (3, 3, 3), (1, 1, 1)
    x = torch.randn(32, D, H, W, Ci, device="cuda").bfloat16(); w = torch.randn(Co, Ci, *k, device="cuda") * 0.05
    res = {}; outs = {}
    for rnd in range(4):
        for bits, label in ((0, "spread"), (128, "burst")):
            L.hupr_debug_halo_ablate(bits)
            for _ in range(2): y = F_._conv_raw(x, w, 0, None, None, Co, k, pad, (D, H, W))
            torch.cuda.synchronize(); outs[label] = y.clone()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): F_._conv_raw(x, w, 0, None, None, Co, k, pad, (D, H, W))
            e.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(s.elapsed_time(e) / 10 * 1e3)
    L.hupr_debug_halo_ablate(0)
    print(name, {k_: "%.1f" % min(v) for k_, v in res.items()}, "equal:", all(torch.equal(outs["spread"], outs[k_]) for k_ in outs))
