import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.tools.engine import TrainEngine
F_.set_math("bf16")
cfg = load_config(); dev = torch.device("cuda", 0)
def mk():
    eng = TrainEngine(cfg, device=dev, seed=0)
    return eng
B, G = 32, cfg.DATASET.numGroupFrames
base_h = torch.from_numpy(synth.adc_cube_int16(10, sensor=0, nframes=16)).to(dev); base_v = torch.from_numpy(synth.adc_cube_int16(10, sensor=1, nframes=16)).to(dev)
adc_h = base_h.repeat(B * G // 16, 1, 1, 1, 1).contiguous(); adc_v = base_v.repeat(B * G // 16, 1, 1, 1, 1).contiguous()
joints = torch.from_numpy(synth.keypoints(B, 20)).to(dev)
def run(eng, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): loss, _ = eng.train_step_from_adc(adc_h, adc_v, joints)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, float(loss)
e1 = mk()
for _ in range(4): e1.train_step_from_adc(adc_h, adc_v, joints)      # 4 eager steps
ms_e, l_e = run(e1, 6)                                               # steps 5..10
e2 = mk()
for _ in range(2): e2.train_step_from_adc(adc_h, adc_v, joints)
e2.capture(adc_h, adc_v, joints, warmup=2)                          # steps 3,4 eager + capture (=step 5)
ms_g, l_g = run(e2, 5)                                               # steps 6..10
print("eager %.2f ms/step (loss after 10 steps %.6f) | graph %.2f ms/step (loss after 10 steps %.6f)" % (ms_e, l_e, ms_g, l_g))
p1 = torch.cat([p.detach().flatten() for p in e1.model.parameters()]); p2 = torch.cat([p.detach().flatten() for p in e2.model.parameters()])
print("param rel diff after 10 steps: %.3e" % ((p1 - p2).norm() / p1.norm()).item())
