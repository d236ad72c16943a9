"""One tiny forward+backward of HuPRNet on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke(); imports oracle/ only as the checker)."""
import numpy as np
import torch


def run():
    from hupr_amd import synth
    from hupr_amd.config_tree import load_config
    from hupr_amd.misc import LossComputer
    from hupr_amd.models import HuPRNet
    from oracle import loss as oloss, model as omodel

    cfg = load_config()
    st = synth.hupr_state(3, gain=1.4)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in st.items()}
    net = HuPRNet(cfg).cuda()
    net.load_state_dict(sd)
    net.train()
    h, v = synth.model_inputs(1, 9)
    gt = synth.keypoints(1, 9)
    p1, p2 = net(torch.from_numpy(h).cuda(), torch.from_numpy(v).cuda())
    loss, _, pred2d, _ = LossComputer(cfg, "cuda").computeLoss((p1, p2), torch.from_numpy(gt))
    loss.backward()
    with torch.no_grad():
        o1, o2 = omodel.forward(sd, torch.from_numpy(h), torch.from_numpy(v), train=True)
    oloss_v, _, opred, _ = oloss.compute_loss((o1, o2), gt)
    e1 = (p1.cpu() - o1).abs().max().item()
    e2 = (p2.cpu() - o2).abs().max().item()
    assert e1 < 1e-3 and e2 < 1e-3, (e1, e2)
    assert np.array_equal(pred2d, opred), "argmax decode differs from the oracle"
    assert abs(loss.item() - oloss_v.item()) < 1e-4
    gn = sum(float(p.grad.double().norm() ** 2) for p in net.parameters()) ** 0.5
    assert np.isfinite(gn) and gn > 0
    print("smoke model ok: heatmap err %.2e / %.2e, loss %.5f, |grad| %.3e" % (e1, e2, loss.item(), gn))
