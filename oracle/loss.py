"""ORACLE (test infrastructure, not product): Gaussian target synthesis, BCE loss on both
heads and arg-max decode, on the CPU.

Follows reference:
  misc/utils.py:6-66     generateTarget  (sigma=2 for 64x64 maps, 13x13 patch, centre = 1,
                         mu = int(joint / 4 + 0.5), joints whose patch is fully outside stay 0)
  misc/losses.py:23-45   computeLoss     (BCELoss(mean) on each head, loss = loss1 + loss2
                         because lossDecay == -1; pred2d decoded from the GCN head only)
  misc/metrics.py:10-38  get_max_preds   (first-max flat index -> (idx % W, idx // W), masked
                         by max > 0)
Parity pinned by tests/golden/loss_*.npz generated from the imported reference.
"""
import numpy as np
import torch
import torch.nn.functional as F


def gaussian_targets(joints, num_kp=14, hsize=64, isize=256, sigma=2):
    """joints: (K,2) integer image coordinates -> (target (K,H,W) float32, centres (K,2))."""
    joints = np.asarray(joints)
    target = np.zeros((num_kp, hsize, hsize), dtype=np.float32)
    centres = np.zeros((num_kp, 2), dtype=np.float64)
    rad = 3 * sigma
    stride = isize / hsize
    ax = np.arange(2 * rad + 1, dtype=np.float32)
    patch = np.exp(-((ax[None, :] - rad) ** 2 + (ax[:, None] - rad) ** 2) / (2 * sigma ** 2))
    for k in range(num_kp):
        mx = int(joints[k][0] / stride + 0.5)
        my = int(joints[k][1] / stride + 0.5)
        x0, y0, x1, y1 = mx - rad, my - rad, mx + rad + 1, my + rad + 1
        if x0 >= hsize or y0 >= hsize or x1 < 0 or y1 < 0:
            continue
        cx0, cy0, cx1, cy1 = max(0, x0), max(0, y0), min(x1, hsize), min(y1, hsize)
        target[k, cy0:cy1, cx0:cx1] = patch[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0]
        centres[k] = (mx, my)
    return target, centres


def batch_targets(gt, **kw):
    tg, ct = zip(*(gaussian_targets(g, **kw) for g in np.asarray(gt)))
    return np.stack(tg), np.stack(ct)


def argmax_decode(heatmaps):
    """heatmaps: ndarray (B,K,H,W) -> (preds (B,K,2) float32 [x,y], maxvals (B,K,1))."""
    B, K, H, W = heatmaps.shape
    flat = heatmaps.reshape(B, K, -1)
    idx = flat.argmax(axis=2)
    mx = flat.max(axis=2)
    preds = np.stack([idx % W, idx // W], axis=2).astype(np.float32)
    preds *= (mx > 0.0)[..., None].astype(np.float32)
    return preds, mx[..., None]


def compute_loss(preds, gt, num_kp=14, hsize=64, isize=256):
    """preds = (heatmap (B,K,1,H,W), gcn_heatmap (B,1,K,H,W)) torch tensors; gt (B,K,2) ints.
    -> (loss, loss2, pred2d ndarray, gt2d ndarray) like LossComputer.computeLoss."""
    targets, _ = batch_targets(np.asarray(gt), num_kp=num_kp, hsize=hsize, isize=isize)
    tgt = torch.from_numpy(targets)
    p1, p2 = preds
    loss1 = F.binary_cross_entropy(p1.reshape(-1, num_kp, hsize, hsize), tgt)
    loss2 = F.binary_cross_entropy(p2.reshape(-1, num_kp, hsize, hsize), tgt)
    loss = loss1 + loss2
    p2m = p2.permute(0, 2, 1, 3, 4).reshape(-1, num_kp, hsize, hsize)
    pred2d, _ = argmax_decode(p2m.detach().cpu().numpy())
    gt2d, _ = argmax_decode(targets)
    return loss, loss2, pred2d, gt2d
