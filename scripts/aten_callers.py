"""Which Python lines of the step still call torch-native kernels (copy_, add_, cat, zero_, fill_, clone, contiguous, zeros)?
One training step with those entry points wrapped; prints call counts and total element counts per caller."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.tools.engine import TrainEngine
F_.set_math("bf16"); F_.TWO_STREAMS = False
cfg = load_config(); dev = torch.device("cuda", 0)
eng = TrainEngine(cfg, device=dev, seed=0)
B = 32
base = torch.from_numpy(synth.adc_cube_int16(10, sensor=0, nframes=16)).to(dev)
adc_h = base.repeat(16, 1, 1, 1, 1).contiguous(); adc_v = adc_h.clone()
joints = torch.from_numpy(synth.keypoints(B, 20)).to(dev)
for _ in range(2): eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
torch.cuda.synchronize()
stats = collections.defaultdict(lambda: [0, 0])
def wrap(owner, name):
    orig = getattr(owner, name)
    def f(*a, **k):
        st = traceback.extract_stack(limit=6)[:-1]
        who = next((s for s in reversed(st) if "hupr" in s.filename), st[-1])
        t = a[0] if a and isinstance(a[0], torch.Tensor) else None
        key = "%s  <- %s:%d %s" % (name, os.path.basename(who.filename), who.lineno, who.name)
        stats[key][0] += 1
        stats[key][1] += t.numel() if t is not None else 0
        return orig(*a, **k)
    setattr(owner, name, f)
for n in ("copy_", "add_", "zero_", "fill_", "clone", "contiguous"):
    wrap(torch.Tensor, n)
for n in ("cat", "zeros", "zeros_like", "empty_like"):
    wrap(torch, n)
eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
torch.cuda.synchronize()
for k, (c, n) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%4d calls %12d elems  %s" % (c, n, k))
