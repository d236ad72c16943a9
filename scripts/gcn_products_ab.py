"""PRGCN layer (F = 1024, 14 key-points in 16 slots) at the bench batch: the dedicated product kernels (csrc/gcn_products.hip)
against the generic fp32 engine they replace; per-call times of the three products and of the whole layer forward + backward,
max-abs difference of the results.  usage (GPU box): python scripts/gcn_products_ab.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
from oracle.model import adjacency
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F, K = 1024, 14
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.zeros(B, F, 16, device="cuda"); x[..., :K] = torch.randn(B, F, K, device="cuda", generator=g)
w = torch.randn(F, F, device="cuda", generator=g) / 32
b = torch.randn(F, K, device="cuda", generator=g) / 32
A = adjacency().cuda()
dy = torch.zeros(B, F, 16, device="cuda"); dy[..., :K] = torch.randn(B, F, K, device="cuda", generator=g)
L, rt = F_.rt.lib(), F_.rt


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


t = torch.empty_like(x); dw = torch.empty(F, F, device="cuda")
print("B = %d" % B)
print("W.x     new %6.1f us   engine %6.1f us" % (
    timed(lambda: rt.check(L.hupr_gcn_wx_f32(rt.ptr(w), rt.ptr(x), rt.ptr(t), B, F, 16, 0, rt.stream()))),
    timed(lambda: F_.gemm(0, 0, w, x, F, 16, F, F, 16, B, 0, F * 16, math="f32"))))
print("W^T.dt  new %6.1f us   engine %6.1f us" % (
    timed(lambda: rt.check(L.hupr_gcn_wx_f32(rt.ptr(w), rt.ptr(dy), rt.ptr(t), B, F, 16, 1, rt.stream()))),
    timed(lambda: F_.gemm(1, 0, w, dy, F, 16, F, F, 16, B, 0, F * 16, math="f32"))))


def dw_engine():
    dt2 = dy.permute(1, 0, 2).reshape(F, B * 16); x2 = x.permute(1, 0, 2).reshape(F, B * 16)
    return F_.gemm(0, 1, dt2.contiguous(), x2.contiguous(), F, F, B * 16, B * 16, B * 16, 1, 0, 0, math="f32")[0]


print("dW      new %6.1f us   engine (two transposing copies + GEMM) %6.1f us" % (
    timed(lambda: rt.check(L.hupr_gcn_dw_f32(rt.ptr(dy), rt.ptr(x), rt.ptr(dw), B, F, 16, rt.stream()))), timed(dw_engine)))
res = {}
for products in (True, False):
    F_.GCN_PRODUCTS = products
    xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)

    def layer():
        xs.grad = ws.grad = bs.grad = None
        y = F_.GCNLayerFn.apply(xs, ws, bs, A, True)
        y.backward(dy)
        return y
    us = timed(layer, 30)
    y = layer()
    res[products] = (y.detach(), xs.grad.clone(), ws.grad.clone(), us)
    print("layer fwd + bwd, %s: %.1f us" % ("dedicated kernels" if products else "generic engine", us))
for name, i in (("y", 0), ("dx", 1), ("dW", 2)):
    a, c = res[True][i], res[False][i]
    print("%s: max-abs difference %.3e (max-abs value %.3e)" % (name, (a - c).abs().max().item(), c.abs().max().item()))
