import sys, torch
sys.path.insert(0, "/root/repo")
from hupr_amd import functional as F_
F_.set_math("bf16")
for shape, size in (((32, 8, 64, 64, 64), (4, 32, 32)), ((32, 4, 32, 32, 128), (2, 16, 16)), ((32, 1, 32, 32, 128), (1, 64, 64)), ((32, 1, 16, 16, 256), (1, 32, 32))):
    x = torch.randn(shape, device="cuda").bfloat16().requires_grad_(True)
    y = F_.interp(x, size)
    g = torch.randn_like(y)
    for _ in range(3):
        y.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        y.backward(g, retain_graph=True)
    e.record(); torch.cuda.synchronize()
    t_b = s.elapsed_time(e) / 10
    s.record()
    for _ in range(10):
        F_.interp(x, size)
    e.record(); torch.cuda.synchronize()
    print(shape, size, "bwd %.1f us (incl. grad accumulate)  fwd %.1f us" % (t_b * 1e3, s.elapsed_time(e) / 10 * 1e3))
