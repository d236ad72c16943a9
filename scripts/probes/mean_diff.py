import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hupr_amd import preprocessing, synth
iq = np.concatenate([synth.adc_cube_int16(4, frame=f) for f in range(16)])
dev = torch.from_numpy(iq).cuda()
full = preprocessing.fft_chain_loader(dev); planes = preprocessing.fft_chain_loader_means(dev)
x = full.reshape(16, 16, 64, 64, 8)
want = (((x[..., 0] + x[..., 1]) + (x[..., 2] + x[..., 3])) + ((x[..., 4] + x[..., 5]) + (x[..., 6] + x[..., 7]))) * 0.125
d = (planes - want).abs()
print("max abs diff", d.max().item(), "frac differing", (d > 0).float().mean().item(), "per plane max", d.amax(dim=(0, 2, 3)).cpu().numpy().round(8))
print("value scale", want.abs().max().item())
