"""Loader glue mirrored from the reference's datasets/base.py.

``Normalize`` keeps the reference call shape (a (C,H,W) tensor in, same shape out,
datasets/base.py:13-24) for callers that still iterate slice by slice; the batched, fused
form the training loader uses is ``preprocessing.fft_chain_loader`` /
``preprocessing.loader_normalize``.
"""
import torch

from ..preprocessing.process_iwr1843 import loader_normalize


class Normalize(object):
    def __call__(self, radarData):
        """radarData: GPU tensor (C=8,H=64,W=64) real -> same shape, per-channel (x-mean)/std."""
        if tuple(radarData.shape) != (8, 64, 64):
            raise ValueError("Normalize expects an (8,64,64) elevation-major slice")
        hwc = radarData.permute(1, 2, 0).to(torch.float32).contiguous()
        cube = torch.zeros((1, 16, 64, 64, 8), dtype=torch.complex64, device=radarData.device)
        # put the slice in both the real and imaginary plane of Doppler slot 4 (f = 0)
        cube[0, 4] = torch.complex(hwc, hwc)
        out = loader_normalize(cube)[0, 0, 0]
        return out.permute(2, 0, 1).contiguous()
