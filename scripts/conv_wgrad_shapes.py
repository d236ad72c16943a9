"""Forward convolution and weight gradient of the encoder / decoder shapes at B = 32 on their own (A/B two builds with
HUPR_LIB_PATH=<other libhupr_hip.so>).  usage: python scripts/conv_wgrad_shapes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, (Ci, Co, D, H, W, kd) in {"l1 64>64": (64, 64, 8, 64, 64, 3), "l2 128>128": (128, 128, 4, 32, 32, 3), "l3 256>256": (256, 256, 2, 16, 16, 3), "dec1 320>64": (320, 64, 1, 64, 64, 1), "dec1 64>320": (64, 320, 1, 64, 64, 1), "l1.0 dgrad 64>32": (64, 32, 8, 64, 64, 3)}.items():
    k, pad = (kd, 3, 3), (kd // 2, 1, 1)
    x = torch.randn(32, D, H, W, Ci, device="cuda").bfloat16(); w = torch.randn(Co, Ci, *k, device="cuda") * 0.05
    dy = torch.randn(32, D, H, W, Co, device="cuda").bfloat16(); dw = torch.empty(Co, Ci, *k, device="cuda")
    ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), x.device)
    conv = t(lambda: F_._conv_raw(x, w, 0, None, None, Co, k, pad, (D, H, W)))
    wg = t(lambda: F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), 32, D, H, W, Ci, Ci, Co, Co, kd, F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))) if Ci % 32 == 0 else float("nan")
    print("%s: conv %.1f us, wgrad %.1f us" % (name, conv, wg))
