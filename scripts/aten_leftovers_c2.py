"""Which torch-native ops still launch kernels in the single-sample eval forward (config C2)?  One eager forward under the kineto tracer."""
import os, sys
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
    n0 = F_.rt.lib().hupr_launch_count()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        net(h, v)
        torch.cuda.synchronize()
    print("library launches per frame:", F_.rt.lib().hupr_launch_count() - n0)
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    print("%-28s n=%-4d dev %.0f us" % (e.key, e.count, e.device_time_total))
