"""MNet (reference models/chirp_networks.py:11-21) on the fused gfx950 front-end kernel.

The module keeps the reference's parameter holder ``temporalConvWx1x1`` (an nn.Conv3d whose
forward is never called) so ``state_dict`` keys, shapes and default initialisation are the
reference's; the arithmetic (elevation mean + .view reinterpretation + conv + max-pool) runs in
``hupr_mnet_fwd_f32`` straight from the (B,G,F,2,R,A,E) loader tensor.
"""
import torch
import torch.nn as nn

from .. import functional as F_


class MNet(nn.Module):
    def __init__(self, in_channels, out_channels, numFrames):
        super().__init__()
        self.temporalConvWx1x1 = nn.Conv3d(in_channels, out_channels, (2, 1, 1), (2, 1, 1), (0, 0, 0))
        self.numFrames = numFrames

    def forward(self, VRDAEmaps):
        """VRDAEmaps: (B,G,F,2,R,A,E) fp32 -> channels-last (B, G, R, A, out_channels); stored as bf16 when the
        encoders run on bf16 activations (functional.act_bf16()).  Also accepts the tensor already averaged over its
        elevation axis, as planes (B, G, 16 = 2 f + c, R, A) — what the fused FFT loader emits (fft_chain_loader_means)."""
        with F_.region("mnet"):
            return F_.MNetFn.apply(VRDAEmaps, self.temporalConvWx1x1.weight, self.temporalConvWx1x1.bias,
                                   torch.bfloat16 if F_.act_bf16() else torch.float32)
