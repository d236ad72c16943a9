"""Two-stream forward: which module's output is the first not to repeat bit for bit?  Forward hooks keep a checksum of every module
output of repetition 0; later repetitions are compared in execution order.  usage: python scripts/forward_determinism.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import pose_fit
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.models import HuPRNet

F_.rt.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for a in sys.argv[2:]:                                     # debug toggles of the library: name=value -> hupr_debug_<name>(value)
    k, val = a.split("=")
    getattr(F_.rt.lib(), "hupr_debug_" + k)(int(val))
train = os.environ.get("EVAL", "0") != "1"
F_.set_math("bf16")
cfg = load_config()
net = HuPRNet(cfg).cuda()
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.hupr_state(1).items()})
F_.invalidate_packed()
net.train(train)
dev = torch.device("cuda")
h, v, joints = pose_fit.scene_batch(32, np.random.default_rng(1), torch.Generator(device=dev).manual_seed(2), dev)
log = []


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in flat(x)]
    return []


def hook(name):
    def f(mod, inp, out):
        st = torch.cuda.current_stream()
        log.append((name, [t.detach().clone() for t in flat(out)], st))
    return f


for n, m in net.named_modules():
    if n:
        m.register_forward_hook(hook(n))
first = None
for r in range(reps):
    log.clear()
    with torch.no_grad():
        net(h, v)
    for s in F_.side_streams_in_use(dev):
        torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    cur = [(n, ts) for n, ts, _ in log]
    if first is None:
        first = cur
        print("%d module outputs recorded" % len(cur))
        continue
    bad = []
    byname0 = {}
    for n, ts in first:
        byname0.setdefault(n, []).append(ts)
    seen = {}
    for n, ts in cur:
        i = seen.get(n, 0)
        seen[n] = i + 1
        ref = byname0[n][i]
        if any(not torch.equal(a, b) for a, b in zip(ts, ref)):
            bad.append(n)
    print("rep %d: %d differing module outputs; first: %s" % (r, len(bad), ", ".join(bad[:10])), flush=True)
    if r == 1:
        print("   execution order: " + " ".join(("*" if n in bad else "") + n for n, _ in cur))
