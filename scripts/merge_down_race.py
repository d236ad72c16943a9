"""Two-stream forward: is the level-1 merge / down-sampling pair (F_.MergeDownFn) the victim?  Clones its input and outputs in flight,
then recomputes the pair alone after a device-wide synchronisation and compares."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import pose_fit
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.models import HuPRNet

F_.set_math("bf16")
cfg = load_config()
net = HuPRNet(cfg).cuda().eval()
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.hupr_state(1).items()})
F_.invalidate_packed()
dev = torch.device("cuda")
h, v, joints = pose_fit.scene_batch(32, np.random.default_rng(1), torch.Generator(device=dev).manual_seed(2), dev)
rec = []
orig = F_.MergeDownFn.apply


def patched(x, w, size):
    xin = x.detach().clone()
    merged, down = orig(x, w, size)
    rec.append((xin, w, size, merged.detach().clone(), down.detach().clone(), x.detach().clone(), torch.cuda.current_stream() == F_.side_stream(dev)))
    return merged, down


F_.MergeDownFn.apply = patched
for r in range(4):
    rec.clear()
    with torch.no_grad():
        net(h, v)
    for s in F_.side_streams_in_use(dev):
        torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    for i, (xin, w, size, merged, down, xafter, on_side) in enumerate(rec):
        with torch.no_grad():
            m2, d2 = orig(xin, w, size)
        torch.cuda.synchronize()
        print("rep %d call %d (%s stream, x %s): input changed during the call %s | merged in flight == alone %s | down in flight == alone %s (%d elements differ)" %
              (r, i, "side" if on_side else "main", tuple(xin.shape), not torch.equal(xin, xafter), torch.equal(merged, m2), torch.equal(down, d2),
               (down != d2).sum().item()))
        if not torch.equal(down, d2):
            idx = (down != d2).reshape(-1).nonzero().reshape(-1)
            C = down.shape[-1]
            vox = torch.unique(idx // C)
            print("    differing voxels %d: first %s ... channels of the first: %s" % (vox.numel(), vox[:12].tolist(), (idx[idx // C == vox[0]] % C).tolist()))
            a, b = down.reshape(-1)[idx[:8]].float().tolist(), d2.reshape(-1)[idx[:8]].float().tolist()
            print("    in flight %s\n    alone     %s" % (a, b))
            Bn, Do, Ho, Wo = down.shape[:4]
            v0 = vox[:12]
            t = int(vox[0])
            fl, al = down.reshape(-1, C)[t].float(), d2.reshape(-1, C)[t].float()
            print("    voxel %d in flight:" % t, " ".join("%.4f" % q for q in fl.tolist()))
            print("    voxel %d alone    :" % t, " ".join("%.4f" % q for q in al.tolist()))
            print("    (b, d, h, w) of those voxels:", [(int(t // (Do * Ho * Wo)), int(t // (Ho * Wo) % Do), int(t // Wo % Ho), int(t % Wo)) for t in v0.tolist()])
