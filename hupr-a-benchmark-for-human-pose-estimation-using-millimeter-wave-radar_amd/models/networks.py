"""HuPRNet — the nn.Module drop-in surface (reference models/networks.py:7-41).

Same constructor (cfg attribute tree), same ``forward(VRDAEmaps_hori, VRDAEmaps_vert)`` signature
with (B,G,F,2,R,A,E) fp32 inputs, same outputs ``(heatmap (B,K,1,H,W), gcn_heatmap (B,1,K,H,W))``
and the same 255 ``state_dict`` entries; all arithmetic runs in hand-written gfx950 kernels
through the C ABI (include/hupr.h).  CUDA/ROCm tensors only — there is no CPU fallback.

Extension used by the fused training loader: ``forward`` also accepts the two inputs already averaged over the elevation
axis (the first thing the reference's forward_chirp does, networks.py:26-27) as planes (B, G, 16, R, A).
"""
import torch
import torch.nn as nn

from .. import functional as F_
from .chirp_networks import MNet
from .layers import Encoder3D, MultiScaleCrossSelfAttentionPRGCN


class HuPRNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.numFrames = cfg.DATASET.numFrames
        self.numFilters = cfg.MODEL.numFilters
        self.rangeSize = cfg.DATASET.rangeSize
        self.heatmapSize = cfg.DATASET.heatmapSize
        self.azimuthSize = cfg.DATASET.azimuthSize
        self.elevationSize = cfg.DATASET.elevationSize
        self.numGroupFrames = cfg.DATASET.numGroupFrames
        self.numKeypoints = cfg.DATASET.numKeypoints
        self.RAchirpNet = MNet(2, self.numFilters, self.numFrames)
        self.REchirpNet = MNet(2, self.numFilters, self.numFrames)
        self.RAradarEncoder = Encoder3D(cfg)
        self.REradarEncoder = Encoder3D(cfg)
        self.radarDecoder = MultiScaleCrossSelfAttentionPRGCN(cfg, batchnorm=False, activation=nn.PReLU)
        # None: the calling thread's mode (functional.set_math); "f32" / "bf16": this model's forward (and, through its autograd
        # nodes, its backward) runs under that mode whatever the thread's is — two models of one process on different pipes
        self.math_mode = None

    def gradient_groups(self):
        """Parameters whose gradient slots should be adjacent in a flat bucket (tools.distributed.GradientBuckets): per MSCSA level
        and map the four 1x1 projection weights in the order of their fused (4C, C) weight-gradient GEMM (functional.MSCSALevelFn)."""
        d = self.radarDecoder
        out = []
        for i in range(len(d.phi_cross_hori)):
            out.append([m[i].weight for m in (d.phi_cross_hori, d.theta_cross_hori, d.phi_self_hori, d.theta_self_hori)])
            out.append([m[i].weight for m in (d.phi_cross_vert, d.theta_cross_vert, d.phi_self_vert, d.theta_self_vert)])
        return out

    def forward_chirp(self, VRDAEmaps_hori, VRDAEmaps_vert):
        return self.RAchirpNet(VRDAEmaps_hori), self.REchirpNet(VRDAEmaps_vert)

    def forward(self, VRDAEmaps_hori, VRDAEmaps_vert):
        if self.math_mode is not None:
            with F_.math_mode(self.math_mode):
                return self._forward(VRDAEmaps_hori, VRDAEmaps_vert)
        return self._forward(VRDAEmaps_hori, VRDAEmaps_vert)

    def _forward(self, VRDAEmaps_hori, VRDAEmaps_vert):
        F_._conv_stats.clear()                  # no fused-statistics hand-over survives a forward pass
        # a model that sat out two or more optimiser epochs (of another model in this process) hands its parameters to the
        # refresh in front of the fork: its packed layouts are no longer among the table pass's candidates (F_.refresh_packed)
        seen = getattr(self, "_pack_epoch_seen", None)
        idle_params = list(self.parameters()) if (seen is None or F_.PACK_EPOCH - seen >= 2) else None
        self._pack_epoch_seen = F_.PACK_EPOCH
        if F_.two_streams_ok(VRDAEmaps_hori):
            # vertical branch on the side stream, horizontal branch on the current one (see functional.TWO_STREAMS)
            dev = VRDAEmaps_hori.device
            capturing = torch.cuda.is_current_stream_capturing()     # graph capture: fork / join become graph edges; the
            if not capturing:                                       # private pool is not recycled, no record_stream needed
                F_.refresh_packed(dev, idle_params)                 # (and the packed-weight cache is bypassed anyway)
            main, side = torch.cuda.current_stream(dev), F_.side_stream(dev)
            side.wait_stream(main)
            if not capturing:
                VRDAEmaps_vert.record_stream(side)
            with torch.cuda.stream(side):
                REl1feat, REl2feat, REfeat = self.REradarEncoder(self.REchirpNet(VRDAEmaps_vert))
            RAl1feat, RAl2feat, RAfeat = self.RAradarEncoder(self.RAchirpNet(VRDAEmaps_hori))
            main.wait_stream(side)
            if not capturing:
                for t in (REl1feat, REl2feat, REfeat):
                    t.record_stream(main)             # allocated on the side stream, consumed by the decoder
        else:
            if VRDAEmaps_hori.is_cuda and not torch.cuda.is_current_stream_capturing():
                F_.refresh_packed(VRDAEmaps_hori.device, idle_params)   # stale cached layouts / derived constants are refilled in place
            RAmaps, REmaps = self.forward_chirp(VRDAEmaps_hori, VRDAEmaps_vert)
            RAl1feat, RAl2feat, RAfeat = self.RAradarEncoder(RAmaps)
            REl1feat, REl2feat, REfeat = self.REradarEncoder(REmaps)
        maps16, gcn_heatmap = self.radarDecoder(RAl1feat, RAl2feat, RAfeat, REl1feat, REl2feat, REfeat)
        B, _, H, W, ld = maps16.shape
        heatmap = F_.SigmoidHeadFn.apply(maps16.reshape(B, H * W, ld), self.numKeypoints)
        return heatmap.reshape(B, self.numKeypoints, 1, H, W), gcn_heatmap
