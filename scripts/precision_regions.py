"""Where does the bf16 path lose arg-max agreement?  

For two sets of trained weights —
  A  "memorised":  the round-2 fixture (120 fp32 Adam steps on two noise samples; multi-modal eval maps on unseen noise),
  B  "pose scenes": tests/pose_fit.py (a learnable task; uni-modal eval maps on held-out scenes)
— evaluate the full fp32 path (the reference side), the full bf16 path, every leave-one-out variant (one region of
functional.REGIONS on the fp32 pipe with fp32 activations, the rest bf16) and every leave-one-in variant (only that region
bf16), and print the arg-max agreement of both heads at B = 32 (448 joints).  Then time a training step with each switch.

usage: python scripts/precision_regions.py [--steps 600] [--skip-a] [--time]
"""
import argparse
import gc
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pose_fit                                     # noqa: E402
from hupr_amd import functional as F_, synth        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--skip-a", action="store_true")
ap.add_argument("--time", action="store_true")
ap.add_argument("--sets", default="")
ap.add_argument("--brief", action="store_true", help="only the all-bf16 row")
ap.add_argument("--full", action="store_true", help="with --sets: also the leave-one-out / leave-one-in rows")
ap.add_argument("--lrs", default="", help="comma list: fit once per learning rate (brief tables), instead of --lr")
args = ap.parse_args()
DEFAULT = dict(F_.PRECISION)


def parse_set(st):
    """"dec1b=f32act,head" -> {"dec1b": "f32act", "head": "f32"}"""
    return {kv.split("=")[0]: (kv.split("=")[1] if "=" in kv else "f32") for kv in st.split(",") if kv}


def table(tag, sd, cfg, h, v, joints=None):
    r1, r2 = pose_fit.evaluate(sd, cfg, h, v, "f32")
    n = r1.shape[0] * 14
    pk = [(r.reshape(n, -1).max(1)[0] / r.reshape(n, -1).mean(1)).median().item() for r in (r1, r2)]
    print("== %s: fp32 path, %d joints, median max/mean %.1f / %.1f, median peak %.3f / %.3f" %
          (tag, n, pk[0], pk[1], r1.reshape(n, -1).max(1)[0].median().item(), r2.reshape(n, -1).max(1)[0].median().item()))
    for nm, r in (("head", r1), ("gcn", r2)):
        t2 = r.reshape(n, -1).topk(2, dim=1)[0]
        g = (t2[:, 0] - t2[:, 1])
        print("   fp32 %s: top-1 minus top-2 value: median %.2e; below 1e-2 on %.3f of the joints, below 1e-3 on %.3f" %
              (nm, g.median().item(), (g < 1e-2).float().mean().item(), (g < 1e-3).float().mean().item()))
    if joints is not None:
        print("   fp32 arg-max == target centre: %.4f / %.4f ; OKS AP of the decoded head %.4f" %
              (pose_fit.hit_rate(r1, joints), pose_fit.hit_rate(r2, joints), pose_fit.decode_ap(r2, joints)))

    def row(name, prec):
        b1, b2 = pose_fit.evaluate(sd, cfg, h, v, "bf16", prec)
        a1, a2 = pose_fit.agreement(b1, r1), pose_fit.agreement(b2, r2)
        extra = "" if joints is None else "  AP %.4f" % pose_fit.decode_ap(b2, joints)
        print("   %-34s head %.4f (1px %.4f, gap %.1e, err %.1e) | gcn %.4f (1px %.4f, gap %.1e, err %.1e)%s" %
              ((name,) + a1 + a2 + (extra,)), flush=True)
    row("all bf16", {})
    row("library default (%s)" % ",".join("%s=%s" % kv for kv in DEFAULT.items()), dict(DEFAULT))
    if args.brief:
        return
    if args.sets:
        for st in args.sets.split(";"):
            row(st, parse_set(st))
        if not args.full:
            return
    for r in F_.REGIONS:
        row("all bf16 but %s" % r, {r: "f32"})
    for r in F_.REGIONS:
        row("only %s bf16" % r, {q: "f32" for q in F_.REGIONS if q != r})
    row("all regions f32 (bf16 run)", {q: "f32" for q in F_.REGIONS})


if not args.skip_a:
    import test_trained_gpu as T
    c = T._trained()
    h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(32, 123))
    table("A memorised (2 noise samples, 120 fp32 steps), unseen noise", c["net"].state_dict(), c["cfg"], h, v)
    del c, h, v
    T._CACHE.clear()
    torch.cuda.empty_cache()

t0 = time.time()
hn, vn, joints = synth.pose_scenes(32, 1)
print("held-out scenes generated in %.0f s" % (time.time() - t0))
h, v = torch.from_numpy(hn).cuda(), torch.from_numpy(vn).cuda()
for lr in ([float(x) for x in args.lrs.split(",")] if args.lrs else [args.lr]):
    t0 = time.time()
    sd, cfg, log = pose_fit.fit(steps=args.steps, lr=lr)
    print("pose-scene fit: %d steps at lr %g in %.0f s" % (args.steps, lr, time.time() - t0))
    table("B pose scenes (%d bf16 steps, lr %g), held-out scenes" % (args.steps, lr), sd, cfg, h, v, torch.from_numpy(joints))

if args.time:
    from hupr_amd.tools.engine import TrainEngine
    F_.set_math("bf16")
    jt = torch.from_numpy(joints)
    sets = [{}] + ([{r: "f32"} for r in F_.REGIONS] if (args.full or not args.sets) else []) + \
        ([parse_set(s) for s in args.sets.split(";")] if args.sets else [])
    for st in sets:
        F_.PRECISION.clear()
        F_.PRECISION.update(st)
        eng = TrainEngine(cfg, device="cuda", lr=1e-4)
        for _ in range(3):
            eng.train_step(h, v, jt)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            eng.train_step(h, v, jt)
        torch.cuda.synchronize()
        print("   training step (B=32, model inputs) with %-40s %.2f ms" % (",".join("%s=%s" % kv for kv in st.items()) or "all bf16", (time.time() - t0) * 100), flush=True)
        eng.close()
        del eng
        gc.collect()                  # dead engines' weights would otherwise stay in the packed-weight table and be refreshed every step
        torch.cuda.empty_cache()
    F_.PRECISION.clear()
    F_.set_math("f32")
