"""which per-region precision switches bring the decoded (PRGCN) head's STRICT arg-max agreement with the fp32 path to
>= 99 % on 2 048 held-out scenes, and what a training step costs with them.  The PRGCN tail (its three layers, the up-sampling and the
sigmoid) already runs on fp32 inputs in every mode (functional.GCN_MATH, PRECISION["head"] = "f32"): what is left is the bf16 rounding
of the decoder maps that feed the head.   usage: python scripts/strict_decoded_ab.py [fit steps]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pose_fit
from hupr_amd import functional as F_, synth
from hupr_amd.tools.engine import TrainEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sd, cfg, log = pose_fit.fit(steps=steps, lr=2e-4, verbose=False, zero_doppler="noise")
print("fit: loss %.4f -> %.4f" % (log[0][1], log[-1][1]), flush=True)
variants = [("library default (dec1b=f32act, head=f32)", {"dec1b": "f32act", "head": "f32"}),
            ("+ dec1a=f32act", {"dec1a": "f32act", "dec1b": "f32act", "head": "f32"}),
            ("+ dec1a, dec2=f32act", {"dec2": "f32act", "dec1a": "f32act", "dec1b": "f32act", "head": "f32"}),
            ("+ dec1a, dec2, dec3=f32act", {"dec3": "f32act", "dec2": "f32act", "dec1a": "f32act", "dec1b": "f32act", "head": "f32"}),
            ("dec1a, dec1b on the fp32 pipe", {"dec1a": "f32", "dec1b": "f32", "head": "f32"})]
NB = 64
ref = []
rng = np.random.default_rng(777); gen = torch.Generator(device="cuda").manual_seed(888)
scenes = []
for _ in range(NB):
    h, v, j = pose_fit.scene_batch(32, rng, gen, torch.device("cuda"), "noise")
    r1, r2 = pose_fit.evaluate(sd, cfg, h, v, "f32")
    ref.append((r1.reshape(32, 14, -1).argmax(-1), r2.reshape(32, 14, -1), r2.reshape(32, 14, -1).argmax(-1)))
    scenes.append((h, v))                            # 64 x 1.07 GB of fp32 inputs: fine on 288 GB
for name, prec in variants:
    same1 = same2 = tie2 = tot = 0
    for (h16, v16), (a1, r2, a2) in zip(scenes, ref):
        b1, b2 = pose_fit.evaluate(sd, cfg, h16, v16, "bf16", precision=prec)
        ab1, ab2 = b1.reshape(32, 14, -1).argmax(-1), b2.reshape(32, 14, -1).argmax(-1)
        same1 += (ab1 == a1).sum().item(); same2 += (ab2 == a2).sum().item()
        gap = r2.max(-1)[0] - r2.gather(-1, ab2[..., None])[..., 0]
        tie2 += ((ab2 == a2) | (gap <= 1e-3)).sum().item()
        tot += 32 * 14
    # step time with the switches
    F_.set_math("bf16"); old = dict(F_.PRECISION); F_.PRECISION.clear(); F_.PRECISION.update(prec)
    eng = TrainEngine(cfg, device="cuda")
    hh, vv, jj = pose_fit.scene_batch(32, rng, gen, torch.device("cuda"), "noise")
    for _ in range(3): eng.train_step(hh, vv, jj)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): eng.train_step(hh, vv, jj)
    torch.cuda.synchronize(); ms = (time.time() - t0) * 100
    eng.close(); del eng
    F_.PRECISION.clear(); F_.PRECISION.update(old); F_.set_math("f32"); F_.invalidate_packed()
    print("%-42s first head identical %.4f | decoded head identical %.4f (%.4f counting ties) | training step (model inputs) %.2f ms" %
          (name, same1 / tot, same2 / tot, tie2 / tot, ms), flush=True)
