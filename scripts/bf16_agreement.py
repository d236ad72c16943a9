"""Arg-max agreement between the bf16 path (bf16 matrix pipe + bf16 activations) and the fp32 parity path on one set of
weights and inputs (eval mode, B = 32: 2 heads x 448 joints), plus the worst near-tie margin of the disagreeing joints.
usage: python scripts/bf16_agreement.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.models import HuPRNet
cfg = load_config()
net = HuPRNet(cfg).cuda().eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hupr_state(1, 1.4).items()})
h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(32, 77))
out = {}
for m in ("f32", "bf16"):
    F_.set_math(m)
    with torch.no_grad():
        p1, p2 = net(h, v)
    out[m] = (p1.reshape(32, 14, -1).float(), p2.reshape(32, 14, -1).float())
F_.set_math("f32")
for hd in (0, 1):
    a, b = out["f32"][hd], out["bf16"][hd]
    ia, ib = a.argmax(-1), b.argmax(-1)
    same = (ia == ib)
    # margin: how far below its own maximum the fp32 map is at the bf16 arg-max (near-tie measure)
    gap = (a.max(-1).values - a.gather(-1, ib[..., None])[..., 0])[~same]
    print("head %d: agreement %.4f (%d/%d)  heat-map max-abs diff %.2e  worst fp32 gap at the bf16 arg-max %.2e" %
          (hd, same.float().mean().item(), int(same.sum()), same.numel(), (a - b).abs().max().item(), gap.max().item() if gap.numel() else 0.0))
