"""Gaussian target synthesis on the GPU (reference misc/utils.py:6-66)."""
import torch

from .. import functional as F_


def generateTarget(joints, numKeypoints, hSize, iSize, device="cuda"):
    """joints: (K,2) integer image coordinates -> (target (K,H,W) GPU tensor, centres (K,2) tensor).
    sigma = 2 for 64x64 maps / 3 for 128x128, patch radius 3*sigma, centre value 1."""
    sigma = {64: 2, 128: 3}[hSize]
    j = torch.as_tensor(joints, dtype=torch.int64, device=device).reshape(1, numKeypoints, 2)
    target = F_.gaussian_targets(j, hSize, iSize, sigma)[0]
    stride = float(iSize) / float(hSize)
    mu = (j[0].to(torch.float32) / stride + 0.5).to(torch.int64)
    rad = 3 * sigma
    outside = ((mu - rad) >= hSize).any(1) | ((mu + rad + 1) < 0).any(1)
    centres = torch.where(outside[:, None], torch.zeros_like(mu), mu)
    return target, centres
