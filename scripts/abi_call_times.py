"""Every C-ABI call of one training step, timed individually (events + synchronise around each call: kernel time without overlap),
summed per entry point.  usage: python scripts/abi_call_times.py [--detail NAME]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hupr_amd import functional as F_, synth, runtime as rt
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
L = rt.lib()
detail = sys.argv[sys.argv.index("--detail") + 1] if "--detail" in sys.argv else None
log = []
def wrap(name):
    orig = getattr(L, name)
    def f(*a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record(); rc = orig(*a); e.record(); torch.cuda.synchronize()
        log.append((name, tuple(x for x in a if isinstance(x, int) and abs(x) < (1 << 31)), s.elapsed_time(e) * 1e3))
        return rc
    f.restype = getattr(orig, "restype", None)
    setattr(L, name, f)
for name, (res, args) in rt.SIGNATURES.items():
    if res is not None and res is not rt.c_size_t and res is not rt.c_char_p and not name.startswith(("hupr_comm", "hupr_version")) and "supported" not in name and "rows" not in name:
        wrap(name)
eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
torch.cuda.synchronize()
agg = collections.OrderedDict()
for n, a, us in log:
    agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += us
tot = sum(v[1] for v in agg.values())
print("C-ABI calls in one step: %d, %.0f us in total" % (len(log), tot))
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d x %7.1f us = %7.0f us (%4.1f %%)  %s" % (c, us / c, us, 100 * us / tot, k))
if detail:
    d = collections.OrderedDict()
    for n, a, us in log:
        if n == detail:
            d.setdefault(a, [0, 0.0]); d[a][0] += 1; d[a][1] += us
    for a, (c, us) in sorted(d.items(), key=lambda kv: -kv[1][1]):
        print("   %3d x %7.1f us  %s" % (c, us / c, a))
