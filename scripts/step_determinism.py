"""Which part of one training step is not bit-reproducible?  Same weights, same batch, forward + backward repeated; compares the
two heads, the loss and every parameter gradient bit for bit against the first repetition.
usage: python scripts/step_determinism.py [reps]      (A/B through the library's environment switches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import pose_fit
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.tools.engine import TrainEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
F_.set_math("bf16")
cfg = load_config()
eng = TrainEngine(cfg, device="cuda", lr=2e-4)
eng.model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.hupr_state(1).items()})
F_.invalidate_packed()
dev = torch.device("cuda")
h, v, joints = pose_fit.scene_batch(32, np.random.default_rng(1), torch.Generator(device=dev).manual_seed(2), dev)
names = [n for n, _ in eng.model.named_parameters()]
first = None
for r in range(reps):
    eng.model.train()
    eng.buckets.prepare(reduce=False)
    preds = eng.model(h, v)
    loss, loss2, _, _ = eng.lossComputer.computeLoss(preds, joints, decode=False)
    loss.backward()
    for s in F_.side_streams_in_use(dev):
        torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    cur = {"head1": preds[0].detach().float().clone(), "head2": preds[1].detach().float().clone(), "loss": loss.detach().clone()}
    for n, p in zip(names, eng.model.parameters()):
        cur["grad:" + n] = p.grad.detach().clone()
    if first is None:
        first = cur
        continue
    bad = [k for k in cur if not torch.equal(cur[k], first[k])]
    print("rep %d: %d of %d tensors differ from rep 0%s" % (r, len(bad), len(cur), (": " + ", ".join(bad[:12])) if bad else ""), flush=True)
    if bad:
        for k in bad[:6]:
            d = (cur[k].double() - first[k].double()).abs().max().item()
            print("    %-60s max|diff| %.3e  (scale %.3e)" % (k, d, first[k].double().abs().max().item()))
