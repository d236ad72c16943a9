"""The dominant kernel back to back for several seconds (power / clock sampling beside it): Encoder3D.layer1 64->64 3x3x3 at B = 32,
bf16 activations; argv[1] = seconds (default 4), argv[2] = 'relu' for half-zero (post-ReLU-like) activations instead of normal ones."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
x = torch.randn(32, 8, 64, 64, 64, device="cuda")
if len(sys.argv) > 2 and sys.argv[2] == "relu":
    x = x.relu()
x = x.bfloat16(); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
def run(n):
    for _ in range(n):
        F_._conv_raw(x, w, 0, None, None, 64, (3, 3, 3), (1, 1, 1), (8, 64, 64))
run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(200); e1.record(); torch.cuda.synchronize(); n += 200
    print("%.1f s: %.1f us per launch" % (time.perf_counter() - t0, e0.elapsed_time(e1) * 1e3 / 200), flush=True)
