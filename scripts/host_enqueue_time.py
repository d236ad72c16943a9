import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.tools.engine import TrainEngine
F_.set_math("bf16"); F_.TWO_STREAMS = False
cfg = load_config(); dev = torch.device("cuda", 0)
eng = TrainEngine(cfg, device=dev, seed=0)
base = torch.from_numpy(synth.adc_cube_int16(10, sensor=0, nframes=16)).to(dev)
adc_h = base.repeat(16, 1, 1, 1, 1).contiguous(); adc_v = adc_h.clone()
joints = torch.from_numpy(synth.keypoints(32, 20)).to(dev)
for _ in range(5): eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
torch.cuda.synchronize()
# host enqueue time: issue steps, measure host time before sync
t0 = time.perf_counter()
for _ in range(20): eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("WGS=%s host enqueue %.2f ms/step, total %.2f ms/step" % (os.environ.get("HUPR_WGRAD_SIDE"), (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
