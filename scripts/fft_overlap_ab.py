"""Measurement: how much of the angle kernel hides under the (HBM-bound) Doppler/range kernel when the two overlap?

The FFT chain is two launches on one stream: hupr_k_doppler_range (reads the int16 cube once; ~40 us per 256 sensor-frames)
and hupr_k_angle<3> (12.6 MB in from L2, 67 MB out; ~27 us).  Here the batch is cut into chunks and the angle kernel of
chunk c runs on a second stream beside the first kernel of chunk c + 1 (debug bits 2 / 3 of hupr_debug_fft_variant skip one
of the two launches).  Cold timing as in bench.py (1 GiB of unrelated traffic in front of every call).

    python scripts/fft_overlap_ab.py [n_sf]
"""
import sys

import torch

from hupr_amd import runtime as rt
from hupr_amd.preprocessing import process_iwr1843 as P


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda")
    adc = torch.randint(-2048, 2048, (n, 4, 192, 256, 2), dtype=torch.int16, device=dev)
    adc2 = adc.roll(1, 0).contiguous()
    lib = rt.lib()
    out = torch.empty((n, 16, 64, 64), dtype=torch.float32, device=dev)
    ws, nbytes = P._workspace(n, dev)
    per_sf_ws = nbytes // n
    trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(device=dev)

    def call(a, lo, hi, variant, stream):
        lib.hupr_debug_fft_variant(variant)
        rt.check(lib.hupr_fft_chain_loader_means_f32(a[lo:hi].data_ptr(), hi - lo, out[lo:hi].data_ptr(),
                                                     ws.data_ptr() + lo * per_sf_ws, (hi - lo) * per_sf_ws, stream.cuda_stream))

    def serial(a):
        call(a, 0, n, 0, torch.cuda.current_stream())

    def only(variant):
        return lambda a: call(a, 0, n, variant, torch.cuda.current_stream())

    def overlapped(chunks):
        def fn(a):
            main_s = torch.cuda.current_stream()
            step = n // chunks
            for c in range(chunks):
                lo, hi = c * step, (c + 1) * step
                call(a, lo, hi, 8, main_s)                      # first kernel only
                e = torch.cuda.Event()
                e.record(main_s)
                side.wait_event(e)
                call(a, lo, hi, 4, side)                        # angle kernel only, beside the next chunk's first kernel
            main_s.wait_stream(side)
        return fn

    def cold(fn, reps=12):
        ts = []
        for i in range(reps):
            trash.fill_(float(i))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            fn(adc if i % 2 == 0 else adc2)
            ev[1].record()
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    serial(adc)
    ref = out.clone()
    for name, fn in [("serial (the library's two launches)", serial), ("first kernel only", only(8)), ("angle kernel only", only(4))] + \
                    [("%d chunks, angle on a second stream" % c, overlapped(c)) for c in (2, 4, 8, 16)]:
        for _ in range(3):
            fn(adc)
        torch.cuda.synchronize()
        us = cold(fn)
        same = ""
        if "chunks" in name:
            fn(adc)
            torch.cuda.synchronize()
            same = "  bit-identical to serial: %s" % bool(torch.equal(out, ref))
        print("%-44s %7.1f us per %d sensor-frames = %5.0f GB/s on 1 048 576 B/sf%s" % (name, us, n, n * 1048576 / us / 1e3, same),
              flush=True)
    lib.hupr_debug_fft_variant(0)


if __name__ == "__main__":
    main()
