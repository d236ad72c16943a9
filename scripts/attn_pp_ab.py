"""Level-1 attention (C = 64, N = 4096, B = 32): the ping-pong kernels (round 4) against the rounds-1-3 kernels, same operands.
Interleaved timing (best of 5 rounds of 10 calls), outputs compared with each other and with fp64 on one sample.
usage: python scripts/attn_pp_ab.py [fwd|all]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
F_.set_math("bf16")
L = F_.rt.lib()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
N, C, B = 4096, 64, 32
gen = torch.Generator(device="cuda").manual_seed(1)
k, q = (torch.randn(B, N, C, device="cuda", generator=gen) * C ** -0.25 * 1.5 for _ in range(2))
v = torch.randn(B, N, C, device="cuda", generator=gen)
kb, qb, vb = k.bfloat16(), q.bfloat16(), v.bfloat16()
g32 = torch.randn(B, N, C, device="cuda", generator=gen); gb = g32.bfloat16()


def run(pp):
    L.hupr_debug_attn_pingpong(pp)
    out = torch.empty(B, N, C, device="cuda"); lse = torch.empty(B, N, device="cuda")
    dk, dq, dv = (torch.empty(B, N, C, device="cuda") for _ in range(3)); scr = torch.empty(B, N, device="cuda")
    fwd = lambda: F_.rt.check(L.hupr_attn_fwd_bf16in(F_.rt.ptr(kb), F_.rt.ptr(qb), F_.rt.ptr(vb), F_.rt.ptr(v), F_.rt.ptr(out), F_.rt.ptr(lse), B, N, C, F_.rt.stream()))
    bwd = lambda: F_.rt.check(L.hupr_attn_bwd_bf16in(F_.rt.ptr(kb), F_.rt.ptr(qb), F_.rt.ptr(vb), F_.rt.ptr(gb), F_.rt.ptr(v), F_.rt.ptr(out), F_.rt.ptr(g32), F_.rt.ptr(lse), F_.rt.ptr(dk), F_.rt.ptr(dq), F_.rt.ptr(dv), F_.rt.ptr(scr), B, N, C, 1, F_.rt.stream()))
    fwd()
    if what == "all":
        bwd()
    torch.cuda.synchronize()
    return dict(fwd=fwd, bwd=bwd, out=out, lse=lse, dk=dk, dq=dq, dv=dv)


def timeit(fn):
    best = 1e9
    for _ in range(5):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e3)
    return best


old, new = run(0), run(1)
names = ("out", "lse") + (("dq", "dk", "dv") if what == "all" else ())
for n in names:
    d = (old[n] - new[n]).abs().max().item()
    print("%-4s ping-pong vs rounds-1-3 kernel: max-abs %.3e (scale %.3e)" % (n, d, old[n].abs().max().item()))
# fp64 on sample 0
s = torch.einsum("jc,kc->jk", kb[0].double(), qb[0].double())
p = torch.softmax(s, 0)
ref = torch.einsum("jc,jk->kc", vb[0].double(), p) + v[0].double()
for tag, r in (("rounds-1-3", old), ("ping-pong", new)):
    print("%-10s vs fp64 (sample 0): out max-abs %.3e, lse max-abs %.3e" % (tag, (r["out"][0].double() - ref).abs().max().item(), (r["lse"][0].double() - torch.logsumexp(s, 0)).abs().max().item()))
for rnd in range(2):
    for pp, r in ((0, old), (1, new)):
        L.hupr_debug_attn_pingpong(pp)
        t = [timeit(r["fwd"])] + ([timeit(r["bwd"])] if what == "all" else [])
        print("round %d %-10s fwd %.1f us (%.0f TF/s)%s" % (rnd, "ping-pong" if pp else "rounds-1-3", t[0], 4.0 * N * N * C * B / t[0] / 1e6,
              ", bwd (prep + dQ + dK/dV) %.1f us (%.0f TF/s algorithmic)" % (t[1], 10.0 * N * N * C * B / t[1] / 1e6) if what == "all" else ""))
L.hupr_debug_attn_pingpong(1)
if what == "all":      # backward kernels: a sample's workgroups on one XCD (round 4) vs plain grid order
    for rnd in range(2):
        for x in (0, 1):
            L.hupr_debug_attn_xcd(x)
            t = timeit(new["bwd"])
            print("round %d backward, workgroups of a sample %s: %.1f us (%.0f TF/s algorithmic)" % (rnd, "on one XCD" if x else "in grid order", t, 10.0 * N * N * C * B / t / 1e6))
    L.hupr_debug_attn_xcd(1)
