"""LossComputer (reference misc/losses.py:8-48) with device-side targets, BCE and decode."""

import torch

from .. import functional as F_
from .metrics import get_max_preds

PAIR_BCE = True      # test aid: False = one BCE node per head, combined by torch


class LossComputer():
    def __init__(self, cfg, device):
        self.device = device
        self.cfg = cfg
        self.numFrames = cfg.DATASET.numFrames
        self.numGroupFrames = cfg.DATASET.numGroupFrames
        self.numKeypoints = cfg.DATASET.numKeypoints
        self.heatmapSize = self.width = self.height = cfg.DATASET.heatmapSize
        self.imgSize = self.imgWidth = self.imgHeight = cfg.DATASET.imgSize
        self.lossDecay = cfg.TRAINING.lossDecay
        self.alpha = 0.0
        self.beta = 1.0

    def targets(self, gt):
        sigma = {64: 2, 128: 3}[self.heatmapSize]
        return F_.gaussian_targets(gt.to(self.device), self.heatmapSize, self.imgSize, sigma)

    def computeLoss(self, preds, gt, decode=True):
        """preds = (heatmap (B,K,1,H,W), gcn_heatmap (B,1,K,H,W)); gt (B,K,2) integer joints.
        -> (loss, loss2, pred2d ndarray, gt2d ndarray) — same tuple as the reference."""
        heatmaps = self.targets(gt)
        preds1, preds2 = preds
        K, H, W = self.numKeypoints, self.height, self.width
        a1, a2 = preds1.reshape(-1, K, H, W), preds2.reshape(-1, K, H, W)
        if self.alpha < 1.0:
            self.alpha += self.lossDecay
            self.beta -= self.lossDecay
        if PAIR_BCE and a1.is_cuda and a1.dtype == a2.dtype == heatmaps.dtype == torch.float32 and a1.shape == a2.shape == heatmaps.shape:
            # both losses and their weighted sum as one node (two launches forward, one backward; the same floats)
            w = (self.alpha, self.beta) if self.lossDecay != -1 else (1.0, 1.0)
            loss, loss2 = F_.PairBCEFn.apply(a1, a2, heatmaps, w[0], w[1])
        else:
            loss1 = F_.BCEFn.apply(a1, heatmaps)
            loss2 = F_.BCEFn.apply(a2, heatmaps)
            if self.lossDecay != -1:
                loss = self.alpha * loss1 + self.beta * loss2
            else:
                loss = loss1 + loss2
        if not decode:
            return loss, loss2, None, None
        if decode == "device":
            # the reference decodes both arg-max sets every iteration (misc/losses.py:43-44); here the two decodes are
            # kernels on the step's stream and the (B,K) index / maximum tensors stay on the device (no sync, no D2H)
            return loss, loss2, F_.argmax_rows(preds2.detach().reshape(-1, H * W)), F_.argmax_rows(heatmaps.reshape(-1, H * W))
        pred2d, _ = get_max_preds(preds2.detach().reshape(-1, K, H, W))
        gt2d, _ = get_max_preds(heatmaps)
        return loss, loss2, pred2d, gt2d
