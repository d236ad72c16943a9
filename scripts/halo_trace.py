"""Phase timeline of the 256-voxel halo conv kernel from in-kernel s_memtime stamps (workgroup 0, wave 0).
usage: python scripts/halo_trace.py [--f32act]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
ACT = torch.float32 if "--f32act" in sys.argv else torch.bfloat16
Ci = Co = 64; D, H, W = 8, 64, 64
x = torch.randn(32, D, H, W, Ci, device="cuda").to(ACT); w = torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.05
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
for _ in range(3): F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
L.hupr_debug_halo_trace(F_.rt.ptr(buf))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W)); e.record()
torch.cuda.synchronize(); L.hupr_debug_halo_trace(None)
t = buf.cpu().numpy(); NS = 6
n = int((t != 0).sum()) // NS
t = t[:n * NS].reshape(n, NS).astype(np.float64)
names = ["stage 0 issue work (weights, next halo)", "stage 0 MFMA + barrier", "stages 1..8", "epilogue stores", "next halo -> LDS + barrier", "loop back"]
d = np.diff(np.concatenate([t, np.roll(t[:, :1], -1, 0)], 1), axis=1)[:-1]      # per tile, NS intervals (last = to next tile start)
tot = (t[-1, 0] - t[0, 0]) / (n - 1)
print("kernel %.1f us, %d tiles on workgroup 0, %.0f ticks/tile" % (s.elapsed_time(e) * 1e3, n, tot))
for k in range(NS):
    print("  %-34s mean %8.0f ticks  (%.1f %%)   min %6.0f max %6.0f" % (names[k], d[:, k].mean(), 100 * d[:, k].mean() / tot, d[:, k].min(), d[:, k].max()))
