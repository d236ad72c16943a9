"""Host-side cost of one training step: cProfile over K eager steps (GPU running asynchronously), top entries by
self time, plus the wall time of the enqueue loop with the GPU idle at the start (no back-pressure for the first
steps).  Usage (GPU box): python scripts/host_profile.py [--steps 6] [--batch 32]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from hupr_amd import functional as F_, synth  # noqa: E402
from hupr_amd.config_tree import load_config  # noqa: E402
from hupr_amd.tools.engine import TrainEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = load_config()
    F_.set_math("bf16")
    eng = TrainEngine(cfg, device=dev, seed=0)
    B, G = args.batch, cfg.DATASET.numGroupFrames
    base_h = torch.from_numpy(synth.adc_cube_int16(10, sensor=0, nframes=16)).to(dev)
    base_v = torch.from_numpy(synth.adc_cube_int16(10, sensor=1, nframes=16)).to(dev)
    reps = (B * G + 15) // 16
    adc_h = base_h.repeat(reps, 1, 1, 1, 1)[:B * G].contiguous()
    adc_v = base_v.repeat(reps, 1, 1, 1, 1)[:B * G].contiguous()
    joints = torch.from_numpy(synth.keypoints(B, 20)).to(dev)
    for _ in range(3):
        eng.train_step_from_adc(adc_h, adc_v, joints)
    torch.cuda.synchronize()
    # enqueue-only wall time, one step at a time from an idle GPU
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train_step_from_adc(adc_h, adc_v, joints)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("enqueue %.2f ms, until GPU done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(args.steps):
        eng.train_step_from_adc(adc_h, adc_v, joints)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(args.top)
    # note: the backward pass runs on autograd's own thread and is invisible to cProfile — its cost is the
    # "run_backward" self time above


if __name__ == "__main__":
    main()
