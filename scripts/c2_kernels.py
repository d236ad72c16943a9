"""Config C2 (eval forward, B = 1, bf16): which kernels one forward launches (torch profiler, device side), by count and time.
usage: python scripts/c2_kernels.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.models import HuPRNet
F_.set_math("bf16")
cfg = load_config(); dev = torch.device("cuda", 0)
net = HuPRNet(cfg).to(dev).eval()
h, v = (torch.from_numpy(t).to(dev) for t in synth.model_inputs(1, 5))
with torch.no_grad():
    for _ in range(3): net(h, v)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        net(h, v)
        torch.cuda.synchronize()
agg = collections.OrderedDict()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        k = e.name[:90]
        agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
n = sum(c for c, _ in agg.values()); t = sum(u for _, u in agg.values())
print("%d device kernels / copies in one B=1 eval forward, %.0f us of device time" % (n, t))
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%4d x %7.1f us = %7.0f us  %s" % (c, us / c, us, k))
