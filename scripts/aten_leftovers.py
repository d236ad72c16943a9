import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.tools.engine import TrainEngine
F_.set_math("bf16"); F_.TWO_STREAMS = False
cfg = load_config(); dev = torch.device("cuda", 0)
eng = TrainEngine(cfg, device=dev, seed=0)
B, G = 32, 8
base_h = torch.from_numpy(synth.adc_cube_int16(10, sensor=0, nframes=16)).to(dev)
adc_h = base_h.repeat(16, 1, 1, 1, 1).contiguous(); adc_v = adc_h.clone()
joints = torch.from_numpy(synth.keypoints(B, 20)).to(dev)
for _ in range(3): eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ev if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    st = [s for s in e.stack if "hupr" in s or "torch/autograd" in s][:3]
    print("%-28s n=%-4d dev %.0f us  | %s" % (e.key, e.count, e.device_time_total, " <- ".join(s.split("/")[-1] for s in st)))
