"""Zero-Doppler convention A/B: fit the pose-scene task with slot f = 4 of both inputs holding unit noise
(the reference's normalised rounding residue; the chain's default dither) or zeros (round 3's exact clutter removal), evaluate
each fit on held-out scenes under both conventions AND with a second realisation of the noise (what separates the reference's
residue from the chain's dither), report OKS AP.  python scripts/zero_doppler_ab.py [steps] [scenes]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pose_fit  # noqa: E402


def main(steps=4000, scenes=512):
    res = {}
    for train_conv in ("noise", "zero"):
        sd, cfg, log = pose_fit.fit(steps=steps, lr=2e-4, verbose=False, zero_doppler=train_conv)
        print("fit with slot 4 = %-5s: loss %.4f -> %.4f" % (train_conv, log[0][1], log[-1][1]), flush=True)
        for eval_conv in ("noise", "renoise", "zero"):
            rng = np.random.default_rng(777)
            gen = torch.Generator(device="cuda").manual_seed(888)
            gen4 = torch.Generator(device="cuda").manual_seed(999) if eval_conv == "renoise" else None
            idx, joints, hits = [], [], []
            for _ in range(scenes // 32):
                h, v, j = pose_fit.scene_batch(32, rng, gen, torch.device("cuda"), "noise" if eval_conv == "renoise" else eval_conv, gen4)
                p1, p2 = pose_fit.evaluate(sd, cfg, h, v, "bf16")
                idx.append(p2.reshape(32, 14, -1).argmax(-1).cpu())
                hits.append(pose_fit.hit_rate(p1, j))
                joints.append(j.numpy())
            ap = pose_fit.decode_ap_from_indices(torch.cat(idx).numpy(), np.concatenate(joints))
            res[(train_conv, eval_conv)] = (ap, float(np.mean(hits)), torch.cat(idx))
            print("  evaluated with slot 4 = %-7s: OKS AP %.4f, first head on the target centre %.4f" % (eval_conv, ap, np.mean(hits)), flush=True)
    for tc in ("noise", "zero"):
        for oc in ("noise", "renoise", "zero"):
            if oc == tc:
                continue
            same = (res[(tc, tc)][2] == res[(tc, oc)][2]).float().mean().item()
            print("trained %-5s: AP own convention %.4f, evaluated with %-7s %.4f, difference %.2f AP points; decoded arg-max identical on %.4f" %
                  (tc, res[(tc, tc)][0], oc, res[(tc, oc)][0], 100 * abs(res[(tc, tc)][0] - res[(tc, oc)][0]), same))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
