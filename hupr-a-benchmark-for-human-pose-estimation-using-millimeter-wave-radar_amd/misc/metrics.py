"""Arg-max decode on the GPU (reference misc/metrics.py:10-38)."""
import numpy as np
import torch

from .. import functional as F_


def get_max_preds(batch_heatmaps):
    """batch_heatmaps: GPU tensor (B,K,H,W) -> (preds ndarray (B,K,2) float32 [x,y], maxvals ndarray (B,K,1)).
    First maximum wins on ties (np.argmax); joints whose maximum is <= 0 decode to (0,0)."""
    if not isinstance(batch_heatmaps, torch.Tensor) or batch_heatmaps.dim() != 4:
        raise AssertionError("batch_heatmaps should be a 4-dim GPU tensor")
    B, K, H, W = batch_heatmaps.shape
    idx, mx = F_.argmax_rows(batch_heatmaps.reshape(B * K, H * W))
    idx = idx.cpu().numpy().reshape(B, K).astype(np.int64)
    mx = mx.cpu().numpy().reshape(B, K, 1)
    preds = np.stack([idx % W, idx // W], axis=2).astype(np.float32)
    preds *= (mx > 0.0).astype(np.float32)
    return preds, mx
