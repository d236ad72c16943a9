import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import preprocessing, runtime as rt, synth
n_sf = 512
base = torch.from_numpy(synth.adc_cube_int16(0, nframes=16)).cuda()
iq = base.repeat(n_sf // 16, 1, 1, 1, 1).contiguous()
ws = torch.empty(rt.lib().hupr_fft_chain_ws_bytes(n_sf), dtype=torch.uint8, device="cuda")
out_l = torch.empty((n_sf, 8, 2, 64, 64, 8), dtype=torch.float32, device="cuda")
out_m = torch.empty((n_sf, 16, 64, 64), dtype=torch.float32, device="cuda")
for _ in range(4):
    preprocessing.fft_chain_loader(iq, ws=ws, out=out_l)
    rt.check(rt.lib().hupr_fft_chain_loader_means_f32(rt.ptr(iq), n_sf, rt.ptr(out_m), rt.ptr(ws), ws.numel(), rt.stream()))
torch.cuda.synchronize()
