"""Does a same-stream producer -> consumer pair of PLAIN torch kernels read stale data while a second stream is busy?  (No hupr code.)
Main stream: b = a * s (producer), c = b + 1 (consumer), compared with the value computed alone.  Side stream: a stream of large copies /
GEMMs.  usage: python scripts/two_queue_coherence.py [iterations]"""
import sys
import torch
it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
a = torch.randn(64 << 20, device=dev)                  # 256 MB
big = torch.randn(96 << 20, device=dev)
big2 = torch.empty_like(big)
m1 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
bad_total = 0
for mode in ("alone", "copies", "gemms"):
    bad = 0
    for i in range(it):
        s = 1.0 + 0.001 * i
        if mode != "alone":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    if mode == "copies":
                        big2.copy_(big)
                    else:
                        m1 @ m1
        b = a * s
        b.relu_()                                       # in place, like the BatchNorm / activation tails
        c = b[::2] + b[1::2]
        ref_b = torch.relu(a * s)
        if mode != "alone":
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ref = ref_b[::2] + ref_b[1::2]
        n = (c != ref).sum().item()
        bad += n > 0
        bad_total += n
    print("%-7s: %d of %d iterations with stale elements" % (mode, bad, it))
print("total stale elements", bad_total)
