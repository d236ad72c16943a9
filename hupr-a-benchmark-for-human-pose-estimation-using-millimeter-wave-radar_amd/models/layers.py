"""Encoder / decoder blocks of HuPRNet (reference models/layers.py) on gfx950 kernels.

Module and parameter names replicate the reference so checkpoints interchange; the nn.Conv*/
nn.BatchNorm*/nn.PReLU children are parameter holders only — every forward below goes through
``functional`` (C-ABI kernels) on channels-last (B,D,H,W,C) tensors.
"""
import torch
import torch.nn as nn

from .. import functional as F_
from .gcn_networks import PRGCN


def _conv(x, m, res=None):
    """Run a parameter-holder nn.Conv2d/nn.Conv3d (stride 1) through the implicit-GEMM kernel."""
    p = m.padding
    pad = (0, p[0], p[1]) if len(p) == 2 else tuple(p)
    return F_.conv(x, m.weight, m.bias, res, pad)


class BasicBlock2D(nn.Module):
    """conv-act-conv + conv residual, then act (reference :8-38).  Decoder uses batchnorm=False."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, batchnorm=True,
                 activation=nn.ReLU):
        super().__init__()
        if batchnorm:
            raise NotImplementedError("the reference only instantiates BasicBlock2D with batchnorm=False")
        self.main = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False),
            activation(),
            nn.Conv2d(out_channels, out_channels, kernel_size, stride, padding, bias=False),
        )
        self.downsample = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=False))
        self.relu = activation()
        if not isinstance(self.relu, nn.PReLU):
            raise NotImplementedError("decoder activation must be nn.PReLU (networks.py:21)")

    def forward(self, x):
        p = self.main[0].padding
        pad = (0, p[0], p[1])
        if F_.infer_fast_ok(x) and F_.conv_infer_sliced(tuple(x.shape[:-1]) + (self.main[2].weight.shape[1],), self.main[2].weight, pad):
            # inference: K-sliced convolutions (small grids) hand their partial sums to the PReLU launch that follows them
            # (only where main[2] is sliced: its residual add then happens in that launch, rounded once like the epilogue's)
            res = F_.conv_infer(x, self.downsample[0].weight, pad)
            out = F_.infer_tail(F_.conv_infer(x, self.main[0].weight, pad), prelu=self.main[1].weight)
            return F_.infer_tail(F_.conv_infer(out, self.main[2].weight, pad), res, prelu=self.relu.weight)
        # the two convolutions of x as one node where the halo kernels apply (bf16 math): the second input gradient is
        # accumulated onto the first in the kernel's residual epilogue instead of by a separate add over the widest maps
        out, residual = F_.dual_conv(x, self.main[0].weight, self.downsample[0].weight, (0, p[0], p[1]))
        out = F_.PReLUFn.apply(out, self.main[1].weight)
        out = _conv(out, self.main[2], res=residual)          # main(x) + residual fused in the epilogue
        return F_.PReLUFn.apply(out, self.relu.weight)


class BasicBlock3D(nn.Module):
    """conv-BN-ReLU-conv-BN + (conv-BN) residual, then ReLU (reference :40-70)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, batchnorm=True,
                 activation=nn.ReLU):
        super().__init__()
        if not batchnorm or activation is not nn.ReLU:
            raise NotImplementedError("the reference only instantiates BasicBlock3D with BN + ReLU")
        self.main = nn.Sequential(
            nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding, bias=False),
            nn.BatchNorm3d(out_channels),
            activation(),
            nn.Conv3d(out_channels, out_channels, kernel_size, stride, padding, bias=False),
            nn.BatchNorm3d(out_channels),
        )
        self.downsample = nn.Sequential(
            nn.Conv3d(in_channels, out_channels, 3, 1, 1, bias=False),
            nn.BatchNorm3d(out_channels),
        )
        self.relu = activation()

    def forward(self, x):
        if not self.training and F_.infer_fast_ok(x):
            # inference: K-sliced convolutions (small grids) hand their partial sums to the BatchNorm launch that follows them
            pad = tuple(self.main[0].padding)
            res = F_.conv_infer(x, self.downsample[0].weight, pad)
            out = F_.infer_tail(F_.conv_infer(x, self.main[0].weight, pad), bn_a=self.main[1], relu=True)
            return F_.infer_tail(F_.conv_infer(out, self.main[3].weight, tuple(self.main[3].padding)), res, bn_a=self.main[4],
                                 bn_b=self.downsample[1], relu=True)
        # the two convolutions of x as one node: their input gradients are summed in the second kernel's epilogue
        # every convolution of the block feeds a BatchNorm: in training mode they leave its column sums (stats=True)
        out, res = F_.dual_conv(x, self.main[0].weight, self.downsample[0].weight, tuple(self.main[0].padding),
                                stats=self.training)
        bn1 = self.main[1]
        no_bwd = not torch.is_grad_enabled()
        out = F_.BNActFn.apply(out, bn1.weight, bn1.bias, bn1, self.training, True, no_bwd)
        out = F_.conv(out, self.main[3].weight, None, None, tuple(self.main[3].padding), stats=self.training)
        bn2, bnd = self.main[4], self.downsample[1]
        return F_.BNAddBNReLUFn.apply(out, bn2.weight, bn2.bias, bn2, res, bnd.weight, bnd.bias, bnd, self.training, no_bwd)


class _Resample(nn.Module):
    """Parameter-less stand-in for nn.Upsample(scale_factor, align_corners=True) at the same index
    of the reference's nn.Sequential (keeps the children numbering, e.g. layer2.1 / layer2.2)."""

    def __init__(self, scale_factor, dims):
        super().__init__()
        self.scale_factor, self.dims = scale_factor, dims

    def size_of(self, x):
        B, D, H, W, C = x.shape
        s = self.scale_factor
        return (int(D * s) if self.dims == 3 else D, int(H * s), int(W * s))

    def forward(self, x):
        return F_.interp(x, self.size_of(x))


class MultiScaleCrossSelfAttentionPRGCN(nn.Module):
    def __init__(self, cfg, batchnorm=True, activation=nn.ReLU):
        super().__init__()
        self.numGroupFrames = cfg.DATASET.numGroupFrames
        self.numFilters = nf = cfg.MODEL.numFilters
        self.width = cfg.DATASET.heatmapSize
        self.height = cfg.DATASET.heatmapSize
        self.numKeypoints = cfg.DATASET.numKeypoints
        self.decoderLayer3 = nn.Sequential(
            BasicBlock2D(nf * 8 * 4, nf * 8, 3, 1, 1, batchnorm, activation),
            BasicBlock2D(nf * 8, nf * 4, 3, 1, 1, batchnorm, activation),
            _Resample(2.0, 2),
        )
        self.decoderLayer2 = nn.Sequential(
            BasicBlock2D(nf * 4 * 5, nf * 4, 3, 1, 1, batchnorm, activation),
            BasicBlock2D(nf * 4, nf * 2, 3, 1, 1, batchnorm, activation),
            _Resample(2.0, 2),
        )
        self.decoderLayer1 = nn.Sequential(
            BasicBlock2D(nf * 2 * 5, nf * 2, 3, 1, 1, batchnorm, activation),
            BasicBlock2D(nf * 2, nf, 3, 1, 1, batchnorm, activation),
            nn.Conv2d(nf, self.numKeypoints, 1, 1, 0, bias=False),
        )
        # unnormalised skeleton adjacency with self loops, as the literal matrix of the reference (:97-112)
        edges = {0: (0, 1, 3), 1: (0, 1, 2), 2: (1, 2), 3: (0, 3, 4), 4: (3, 4, 5), 5: (4, 5), 6: (6, 7),
                 7: (6, 7), 8: (6, 8, 9), 9: (8, 9, 10), 10: (9, 10), 11: (6, 11, 12), 12: (11, 12, 13),
                 13: (12, 13)}
        A = torch.zeros(14, 14)
        for r, cols in edges.items():
            A[r, list(cols)] = 1.0
        self.gcn = PRGCN(cfg, A)
        filterList = [nf * 8, nf * 4, nf * 2]
        mk = lambda: nn.ModuleList([nn.Conv2d(i, i, 1, 1, 0, bias=False) for i in filterList])
        self.phi_cross_hori, self.theta_cross_hori = mk(), mk()
        self.phi_cross_vert, self.theta_cross_vert = mk(), mk()
        self.phi_self_hori, self.theta_self_hori = mk(), mk()
        self.phi_self_vert, self.theta_self_vert = mk(), mk()
        self.sigmoid = nn.Sigmoid()

    @staticmethod
    def attention(k, q, maps, residual=False):
        """k, q, maps: channels-last (B,1,H,W,C).  Softmax over keys, value = maps (reference :126-133)."""
        B, _, H, W, C = maps.shape
        out = F_.AttentionFn.apply(k.reshape(B, H * W, C), q.reshape(B, H * W, C), maps.reshape(B, H * W, C),
                                   residual)
        return out.reshape(B, 1, H, W, C)

    def _level(self, i, ra, re, cat_bf16=False):
        if F_.mscsa_level_fused_ok(ra):         # bf16 math: projections + attentions of the level as one autograd node
            w = [m[i].weight for m in (self.phi_cross_hori, self.theta_cross_hori, self.phi_self_hori, self.theta_self_hori,
                                       self.phi_cross_vert, self.theta_cross_vert, self.phi_self_vert, self.theta_self_vert)]
            return list(F_.MSCSALevelFn.apply(ra, re, int(bool(cat_bf16)) | (0 if torch.is_grad_enabled() else 2), *w))
        k_c_h, q_c_v = _conv(ra, self.phi_cross_hori[i]), _conv(re, self.theta_cross_vert[i])
        k_c_v, q_c_h = _conv(re, self.phi_cross_vert[i]), _conv(ra, self.theta_cross_hori[i])
        k_h, q_h = _conv(ra, self.phi_self_hori[i]), _conv(ra, self.theta_self_hori[i])
        k_v, q_v = _conv(re, self.phi_self_vert[i]), _conv(re, self.theta_self_vert[i])
        return [self.attention(k_c_h, q_c_v, ra, residual=True), self.attention(k_h, q_h, ra),
                self.attention(k_c_v, q_c_h, re, residual=True), self.attention(k_v, q_v, re)]

    def forward(self, ral1maps, ral2maps, ramaps, rel1maps, rel2maps, remaps):
        # with bf16 activations (F_.act_bf16()) the BasicBlock2D stacks (3x3 convolutions, PReLU, up-sampling) read and
        # write bf16 too; the attention outputs are cast once on the way in, the 1x1 head gets fp32 back.  Every stage sits in a
        # precision region (F_.region): a region switched to fp32 takes / hands over its maps through casts at its borders.
        d1, d2, d3 = self.decoderLayer1, self.decoderLayer2, self.decoderLayer3
        #         level, regions of the blocks, blocks, the resampling that closes the stage, the level's two maps
        stages = ((0, ("dec3", "dec3"), (d3[0], d3[1]), d3[2], ramaps, remaps), (1, ("dec2", "dec2"), (d2[0], d2[1]), d2[2], ral2maps, rel2maps),
                  (2, ("dec1a", "dec1b"), (d1[0], d1[1]), None, ral1maps, rel1maps))
        prev = None                                    # (previous stage's output before its resampling, that resampling, its region)
        for i, names, blocks, resample, ra, re in stages:
            with F_.region(names[0]):
                bf16_in = F_.act_bf16() and F_.ACT_BF16_DECODER
            B, _, H, W, C = ra.shape
            with F_.region("lvl%d" % i):               # (decided under the LEVEL's precision: a level switched to fp32 is not fused)
                place = bf16_in and F_.level_cat_placement_ok(ra)
            if place and (prev is None or prev[0].dtype == torch.bfloat16):
                # The stage's input [up-sampled previous maps | out1 | out2 | out3 | out4] (reference :166-178) is ONE buffer that its
                # producers fill in place — the previous stage's resampling writes its channel slice, the fused level writes the
                # other — instead of a concatenation copy per stage (F_.JoinFn: the backward hands the slices of the gradient back)
                cprev = 0 if prev is None else prev[0].shape[-1]
                wide = torch.empty((B, 1, H, W, cprev + 4 * C), dtype=torch.bfloat16, device=ra.device)
                parts = []
                if prev is not None:
                    with F_.region(prev[2]):
                        parts.append(F_.interp(prev[0], prev[1].size_of(prev[0]), out=wide[..., :cprev]))
                with F_.region("lvl%d" % i):
                    F_._level_cat_out["out"] = wide[..., cprev:]
                    lv = self._level(i, ra, re, True)
                    F_._level_cat_out.clear()          # (never leaks to another level)
                assert len(lv) == 1, "level_cat_placement_ok() promised a fused level"
                x = F_.JoinFn.apply(wide, *(parts + [lv[0]])) if parts else lv[0]
            else:
                maps = None
                if prev is not None:
                    with F_.region(prev[2]):
                        maps = prev[1](prev[0])
                with F_.region("lvl%d" % i):
                    # bf16 decoder input: a fused level hands back its four maps already concatenated and cast (one tensor)
                    lv = self._level(i, ra, re, bf16_in and F_.CAT_FUSION)
                with F_.region(names[0]):
                    parts = ([] if maps is None else [maps]) + list(lv)
                    parts = [F_.to_act(t, decoder=True) for t in parts]
                    x = torch.cat(parts, 4) if len(parts) > 1 else parts[0]
            with F_.region(names[0]):
                maps = blocks[0](F_.to_act(x, decoder=True))
            with F_.region(names[1]):
                maps = blocks[1](F_.to_act(maps, decoder=True))
            prev = (maps, resample, names[1]) if resample is not None else None
        x = F_.cast(maps, torch.float32)
        # 1x1 head with the 14 output channels zero-padded to 16 so later kernels stay float4-aligned
        with F_.region("head"):
            maps16 = F_.head_conv(x, self.decoderLayer1[2].weight, self.numKeypoints)
        return maps16, self.gcn(maps16)


class Encoder3D(nn.Module):
    def __init__(self, cfg, batchnorm=True, activation=nn.ReLU):
        super().__init__()
        self.numGroupFrames = G = cfg.DATASET.numGroupFrames
        self.numFilters = nf = cfg.MODEL.numFilters
        self.width = cfg.DATASET.heatmapSize
        self.height = cfg.DATASET.heatmapSize
        self.layer1 = nn.Sequential(
            nn.Conv3d(nf, nf * 2, 3, 1, 1),
            BasicBlock3D(nf * 2, nf * 2, 3, 1, 1),
        )
        self.layer2 = nn.Sequential(
            _Resample(0.5, 3),
            BasicBlock3D(nf * 2, nf * 4, 3, 1, 1),
            BasicBlock3D(nf * 4, nf * 4, 3, 1, 1),
        )
        self.layer3 = nn.Sequential(
            _Resample(0.5, 3),
            BasicBlock3D(nf * 4, nf * 8, 3, 1, 1),
            BasicBlock3D(nf * 8, nf * 8, 3, 1, 1),
        )
        self.l1temporalMerge = nn.Conv3d(nf * 2, nf * 2, (G, 1, 1), 1, 0, bias=False)
        self.l2temporalMerge = nn.Conv3d(nf * 4, nf * 4, (G // 2, 1, 1), 1, 0, bias=False)
        self.temporalMerge = nn.Conv3d(nf * 8, nf * 8, (G // 4, 1, 1), 1, 0, bias=False)

    def _merge_and_down(self, x, merge, resample):
        """A level map feeds its temporal merge AND the next level's down-sampling: one autograd node for the pair where the
        fused kernels apply (bf16-stored map, bf16 math), so the two input gradients are summed inside the resampling backward
        instead of by a separate pass (F_.MergeDownFn).  -> (merged fp32, down-sampled in x's storage type)."""
        if F_.merge_down_ok(x):
            return F_.MergeDownFn.apply(x, merge.weight, resample.size_of(x))
        if x.dtype == torch.bfloat16 and F_.MATH != "bf16":          # the "merge" region on the fp32 pipe
            x = F_.cast(x, torch.float32)
        return F_.temporal_merge(x, merge.weight), resample(x)

    def forward(self, maps):
        """maps: channels-last (B, G, R, A, nf) -> three channels-last (B,1,h,w,c) fp32 feature maps.

        With bf16 activations (F_.act_bf16()) everything up to the temporal merges — the 3x3x3 convolutions,
        BatchNorms and resamplings that dominate the step — reads and writes bf16 tensors; the merges read those
        bf16 maps directly and emit fp32 (F_.temporal_merge).  The three levels and the merges are precision regions."""
        with F_.region("enc1"):
            l1maps = self.layer1[1](_conv(F_.to_act(maps), self.layer1[0]))
        with F_.region("merge"):
            l1m, d1 = self._merge_and_down(l1maps, self.l1temporalMerge, self.layer2[0])
        with F_.region("enc2"):
            l2maps = self.layer2[2](self.layer2[1](F_.to_act(d1)))
        with F_.region("merge"):
            l2m, d2 = self._merge_and_down(l2maps, self.l2temporalMerge, self.layer3[0])
        with F_.region("enc3"):
            l3maps = self.layer3[2](self.layer3[1](F_.to_act(d2)))
        with F_.region("merge"):
            if l3maps.dtype == torch.bfloat16 and F_.MATH != "bf16":
                l3maps = F_.cast(l3maps, torch.float32)
            return l1m, l2m, F_.temporal_merge(l3maps, self.temporalMerge.weight)
