"""PRGCN head (reference models/gcn_networks.py:6-64) on the gfx950 GEMM + epilogue kernels.

Feature maps travel channels-last with the 14 keypoint channels padded to 16 (pad = 0), so the
node-feature tensor (B, 1024, 16) *is* the bilinearly down-sampled heat-map — no permute.
The adjacency is a non-persistent buffer (the reference keeps it as a plain attribute created
with .cuda(), models/layers.py:97-112; either way it is absent from state_dict).
"""
import math

import torch
import torch.nn as nn

from .. import functional as F_


class GCN_layers(nn.Module):
    def __init__(self, in_features, out_features, numKeypoints, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, numKeypoints))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj, relu=False):
        return F_.GCNLayerFn.apply(input, self.weight, self.bias, adj, relu)


class PRGCN(nn.Module):
    def __init__(self, cfg, A):
        super().__init__()
        self.numGroupFrames = cfg.DATASET.numGroupFrames
        self.numFilters = cfg.MODEL.numFilters
        self.width = cfg.DATASET.heatmapSize
        self.height = cfg.DATASET.heatmapSize
        self.numKeypoints = cfg.DATASET.numKeypoints
        self.featureSize = (self.height // 2) * (self.width // 2)
        self.L1 = GCN_layers(self.featureSize, self.featureSize, self.numKeypoints)
        self.L2 = GCN_layers(self.featureSize, self.featureSize, self.numKeypoints)
        self.L3 = GCN_layers(self.featureSize, self.featureSize, self.numKeypoints)
        self.register_buffer("A", A, persistent=False)

    def forward(self, maps16):
        """maps16: channels-last (B,1,H,W,16) logits (channels >= numKeypoints are zero)
        -> (B,1,K,H,W) probabilities, like the reference's PRGCN.forward."""
        B, _, H, W, ld = maps16.shape
        x = F_.interp(maps16, (1, H // 2, W // 2)).reshape(B, self.featureSize, ld)
        x = self.L1(x, self.A, relu=True)
        x = self.L2(x, self.A, relu=True)
        x = self.L3(x, self.A, relu=False)
        hm = F_.interp(x.reshape(B, 1, H // 2, W // 2, ld), (1, H, W))
        p = F_.SigmoidHeadFn.apply(hm.reshape(B, H * W, ld), self.numKeypoints)
        return p.reshape(B, 1, self.numKeypoints, H, W)
