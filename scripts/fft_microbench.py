"""Times the FFT-chain kernels with HIP events (inputs resident in HBM) and prints GB/s against
the algorithmic byte counts of SURVEY.md 8(d).  Usage: python scripts/fft_microbench.py [n_sf]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import preprocessing, runtime as rt, synth  # noqa: E402

n_sf = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = torch.from_numpy(synth.adc_cube_int16(0, nframes=16)).cuda()
iq = base.repeat((n_sf + 15) // 16, 1, 1, 1, 1)[:n_sf].contiguous()
ws = torch.empty(rt.lib().hupr_fft_chain_ws_bytes(n_sf), dtype=torch.uint8, device="cuda")
out_l = torch.empty((n_sf, 8, 2, 64, 64, 8), dtype=torch.float32, device="cuda")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


out_m = torch.empty((n_sf, 16, 64, 64), dtype=torch.float32, device="cuda")
b_c, b_l, b_m = n_sf * 4980736, n_sf * 2883584, n_sf * 1048576          # SURVEY 8(d) algorithmic bytes per sensor-frame
for tag, flag in (("doppler-first K1 (default)", 0), ("range-first K1 (rounds 1-2)", 1)):
    rt.lib().hupr_debug_fft_range_first(flag)
    t_c = timeit(lambda: preprocessing.fft_chain(iq, ws=ws))
    t_l = timeit(lambda: preprocessing.fft_chain_loader(iq, ws=ws, out=out_l))
    t_m = timeit(lambda: rt.check(rt.lib().hupr_fft_chain_loader_means_f32(rt.ptr(iq), n_sf, rt.ptr(out_m), rt.ptr(ws), ws.numel(), rt.stream())))
    print("%s  n_sf=%d\n  c64:    %.1f us  %.0f GB/s (%.3f of 8 TB/s)\n  loader: %.1f us  %.0f GB/s (%.3f)  %.0f sensor-frames/s\n"
          "  loader + elevation mean (the training step's variant): %.1f us  %.0f GB/s (%.3f)  %.0f sensor-frames/s"
          % (tag, n_sf, t_c * 1e6, b_c / t_c / 1e9, b_c / t_c / 8e12, t_l * 1e6, b_l / t_l / 1e9, b_l / t_l / 8e12, n_sf / t_l,
             t_m * 1e6, b_m / t_m / 1e9, b_m / t_m / 8e12, n_sf / t_m))
rt.lib().hupr_debug_fft_range_first(0)

# ---- per-kernel split (events around each C-ABI call are not possible: two kernels per call) and a pure-bandwidth baseline
if len(sys.argv) > 2 and sys.argv[2] == "detail":
    big = torch.empty(n_sf * 524288, dtype=torch.float32, device="cuda")      # same bytes as the loader output
    t_fill = timeit(lambda: big.fill_(1.0))
    src = torch.empty_like(big)
    t_copy = timeit(lambda: big.copy_(src))
    print("baseline: fill %.0f GB/s, copy (r+w) %.0f GB/s" % (big.numel() * 4 / t_fill / 1e9, 2 * big.numel() * 4 / t_copy / 1e9))

# ---- cache-cold check: the training step calls the chain once per sensor per step, with ~25 GB of other traffic in between, so
# its inputs never sit in the 256 MB Infinity Cache and its TLB entries are gone.  Per-call times of the fused-mean variant,
# warm (back to back on the same two buffers) and cold (a 4 GB fill between calls).
if len(sys.argv) > 2 and sys.argv[2] in ("cold", "detail"):
    half = n_sf // 2
    a, b = iq[:half].contiguous(), iq[half:].contiguous()
    om = torch.empty((half, 16, 64, 64), dtype=torch.float32, device="cuda")
    trash = torch.empty(1 << 30, dtype=torch.float32, device="cuda")

    def one(x):
        rt.check(rt.lib().hupr_fft_chain_loader_means_f32(rt.ptr(x), half, rt.ptr(om), rt.ptr(ws), ws.numel(), rt.stream()))

    for variant, cold in [(vv, cc) for vv in (0, 1, 2, 3) for cc in (False, True)]:
        rt.lib().hupr_debug_fft_variant(variant)
        ts = []
        for i in range(12):
            if cold:
                trash.fill_(float(i))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            one(a if i % 2 == 0 else b)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        med = sorted(ts[2:])[len(ts[2:]) // 2]
        print("variant %d (bit 0 temporal ADC loads as in rounds 1-3, bit 1 per-XCD antenna groups) %s, %d sensor-frames per call: per-call us %s  median %.1f us = %.0f GB/s (%.3f of 8 TB/s)" %
              (variant, "cold (4 GB fill between calls)" if cold else "warm (back to back)", half, " ".join("%.0f" % t for t in ts), med,
               half * b_m / n_sf / med / 1e3, half * b_m / n_sf / med / 1e3 / 8000))
    rt.lib().hupr_debug_fft_variant(0)
