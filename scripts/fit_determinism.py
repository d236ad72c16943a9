"""Is the training step bit-reproducible?  Runs the pose fit (tests/pose_fit.py) for a few hundred steps several times — inside one
process and, by calling the script repeatedly, across processes — and prints a checksum of the resulting weights and the logged losses.
usage: python scripts/fit_determinism.py [steps] [repeats]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import pose_fit

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in range(reps):
    sd, cfg, log = pose_fit.fit(steps=steps, lr=2e-4, verbose=False, log_every=50)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].detach().float().cpu().numpy().tobytes())
    print("pid %d run %d: weights sha %s  losses %s" % (os.getpid(), r, h.hexdigest()[:16], " ".join("%.9f" % l[1] for l in log)), flush=True)
