"""Co-residency check of the FFT chain: the loader chain on a FIXED input, 200 launches per burst, alone and beside
(a) the level-3 convolution hupr_k_conv_halo_bf16<64, 64> on a second stream, (b) all-reduces through the C ABI's RCCL communicator on a
third, (c) both.  Reports how many launches differ from the launch alone, where (the RD intermediate of hupr_k_doppler_range or the
planes of hupr_k_angle) and by how much.  usage: python scripts/fft_race.py [variant bits of hupr_debug_fft_variant]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_, preprocessing, synth
from hupr_amd.tools.distributed import RcclTransport
F_.set_math("bf16")
L, rt = F_.rt.lib(), F_.rt
dev = torch.device("cuda")
if len(sys.argv) > 1:
    L.hupr_debug_fft_variant(int(sys.argv[1]))
n_sf = 32
iq = torch.from_numpy(np.concatenate([synth.adc_cube_int16(31, frame=f) for f in range(n_sf)])).cuda()
nws = L.hupr_fft_chain_ws_bytes(n_sf)


def run():
    ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
    out = preprocessing.fft_chain_loader_means(iq, ws=ws)
    return ws.view(torch.float32), out


ref_ws, ref = run()
torch.cuda.synchronize()
again_ws, again = run()
torch.cuda.synchronize()
print("alone, second launch: RD identical %s, planes identical %s" % (torch.equal(ref_ws, again_ws), torch.equal(ref, again)))
gen = torch.Generator(device=dev).manual_seed(5)
x3 = torch.randn(32, 2, 16, 16, 256, device=dev, generator=gen).bfloat16()
w3 = (torch.randn(256, 256, 3, 3, 3, device=dev, generator=gen) * 0.02).requires_grad_(True)
side, third = F_.side_stream(dev), torch.cuda.Stream(device=dev)
comm = RcclTransport(dev)
bucket = torch.randn(12 << 20, device=dev, generator=gen)
big = torch.randn(64 << 20, device=dev)


def conv():
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(100):
            F_.conv(x3, w3, None, None, (1, 1, 1))


def rccl():
    for _ in range(4):
        comm.all_reduce(bucket, stream=third)


def elementwise():
    with torch.cuda.stream(side):
        for _ in range(40):
            big + 1.0


for name, aggs in (("nothing", ()), ("torch elementwise (256 MB add) on a second stream", (elementwise,)), ("level-3 convolution on a second stream", (conv,)),
                   ("RCCL all-reduce on a third stream", (rccl,)), ("both", (conv, rccl))):
    bad_ws = bad_out = tot = 0
    worst = 0.0
    first = ""
    for _ in range(6):
        side.wait_stream(torch.cuda.current_stream())
        third.wait_stream(torch.cuda.current_stream())
        for a in aggs:
            a()
        outs = [run() for _ in range(100)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.current_stream().wait_stream(third)
        torch.cuda.synchronize()
        for w, o in outs:
            tot += 1
            dw, do = not torch.equal(w, ref_ws), not torch.equal(o, ref)
            bad_ws += dw
            bad_out += do
            if do:
                d = (o - ref).abs()
                worst = max(worst, d.max().item())
                if not first:
                    idx = (o != ref).reshape(-1).nonzero().reshape(-1)
                    pl = sorted(set(((idx // 4096) % 16).tolist()))
                    first = "%d elements, planes %s, max |diff| %.3e, RD differs: %s (%d elements)" % (
                        idx.numel(), pl, d.max().item(), dw, (w != ref_ws).sum().item())
    print("beside %-52s: planes differ in %4d of %d launches, RD in %4d; worst |diff| %.3e%s" %
          (name, bad_out, tot, bad_ws, worst, ("   [first: " + first + "]") if first else ""), flush=True)
comm.close()
