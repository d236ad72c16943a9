"""Per-workgroup start / end times of the persistent 256-voxel conv kernel (layer-1 shape): how much of a launch is tail?
usage: python scripts/halo_wg_spread.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
x = torch.randn(32, 8, 64, 64, 64, device="cuda").bfloat16(); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
buf = torch.zeros(4096 + 512, dtype=torch.int64, device="cuda")
for _ in range(3): F_._conv_raw(x, w, 0, None, None, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64))
for rep in range(3):
    buf.zero_()
    L.hupr_debug_halo_trace(F_.rt.ptr(buf))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); F_._conv_raw(x, w, 0, None, None, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64)); e.record()
    torch.cuda.synchronize(); L.hupr_debug_halo_trace(None)
    t = buf[4096:].cpu().numpy().reshape(256, 2).astype(np.float64)
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0            # wall_clock64: 100 MHz -> us
    dur = en - st
    print("launch %.1f us (events) | workgroup start spread %.1f us | end: min %.1f median %.1f max %.1f us | busy time per workgroup: "
          "min %.1f median %.1f max %.1f us | mean idle before the last one ends %.1f us (%.1f %%)" %
          (s.elapsed_time(e) * 1e3, st.max(), en.min(), np.median(en), en.max(), dur.min(), np.median(dur), dur.max(),
           (en.max() - en).mean() + st.mean(), 100 * ((en.max() - en).mean() + st.mean()) / en.max()))
    xcd = np.arange(256) % 8
    print("   per-XCD median end:", " ".join("%.0f" % np.median(en[xcd == c]) for c in range(8)))
