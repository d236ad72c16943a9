import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_trained_gpu as T
from hupr_amd import functional as F_, synth
from hupr_amd.models import HuPRNet
c = T._trained()
h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(32, 123))
with torch.no_grad():
    f1, f2 = c["net"](h, v)
F_.set_math("bf16")
net16 = HuPRNet(c["cfg"]).cuda().eval(); net16.load_state_dict(c["net"].state_dict())
with torch.no_grad():
    b1, b2 = net16(h, v)
F_.set_math("f32")
for name, f, b in (("head", f1, b1), ("gcn", f2, b2)):
    ff, bb = f.reshape(448, -1), b.reshape(448, -1)
    af, ab = ff.argmax(1), bb.argmax(1)
    dx = (af % 64 - ab % 64).abs(); dy = (af // 64 - ab // 64).abs()
    d = torch.maximum(dx, dy).cpu().numpy()
    top = ff.max(1)[0].cpu().numpy()
    gap = (ff.max(1)[0] - ff.gather(1, ab[:, None])[:, 0]).cpu().numpy()      # fp32 map: own max minus value at the bf16 arg-max
    fl = d > 0
    print(name, "identical %.4f, <=1px %.4f, <=2px %.4f, <=4px %.4f; flips: dist %s, fp32 peak %s, fp32 gap at bf16 idx %s" % (
        (d == 0).mean(), (d <= 1).mean(), (d <= 2).mean(), (d <= 4).mean(), d[fl].tolist(), np.round(top[fl], 3).tolist(), np.round(gap[fl], 4).tolist()))
    sat = (ff > 0.98 * ff.max(1, keepdim=True)[0]).sum(1).float()
    print("   pixels within 2%% of the max: median %.0f, max %.0f; peak value median %.3f" % (sat.median().item(), sat.max().item(), np.median(top)))
