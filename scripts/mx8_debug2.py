import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
B, N, C = 1, 256, 64
def run(k, q, v=None):
    v = torch.ones(B, N, C, device="cuda") if v is None else v
    out, lse = F_.attention_mx8(k.cuda(), q.cuda(), v, False)
    S = torch.einsum("bjc,bqc->bjq", k.double().cuda(), q.double().cuda())
    return lse[0], torch.logsumexp(S, 1)[0], out[0]
one = torch.ones(B, N, C)
l, r, o = run(one * 0.25, one * 0.25); print("a) k = q = 0.25: lse", l[:3].tolist(), "expect", r[:3].tolist(), "out", o[0, :2].tolist())
k = one.clone() * 0.25; k[:, :, 32:] = 1.0
l, r, o = run(k, one * 0.25); print("b) k block 1 = 1.0:", l[:3].tolist(), "expect", r[:3].tolist())
k = one.clone() * 0.25; k *= (2.0 ** (torch.arange(N) % 4 - 2)).reshape(1, N, 1)
l, r, o = run(k, one * 0.25); print("c) k rows x 2^(j%4-2):", l[:3].tolist(), "expect", r[:3].tolist())
q = one.clone() * 0.25; q *= (2.0 ** (torch.arange(N) % 4 - 2)).reshape(1, N, 1)
l, r, o = run(one * 0.25, q); print("d) q rows x 2^(q%4-2):", l[:8].tolist(), "expect", r[:8].tolist())
q = one.clone() * 0.25; q[:, :, 32:] = 1.0
l, r, o = run(one * 0.25, q); print("e) q block 1 = 1.0:", l[:3].tolist(), "expect", r[:3].tolist())
k = torch.zeros(B, N, C); k[0, torch.arange(N), torch.arange(N) % C] = 1.0
q = (torch.arange(C).float() / 16).reshape(1, 1, C).expand(B, N, C).contiguous()
l, r, o = run(k, q); print("f) k one-hot channel j%64, q = c/16:", l[:3].tolist(), "expect", r[:3].tolist())
k = torch.randn(B, N, C).bfloat16().float() * 0.5; q = torch.randn(B, N, C).bfloat16().float() * 0.5
l, r, o = run(k, q); print("g) random: max |lse - fp64| %.3e" % (l.double() - r).abs().max().item())
k2 = k.clone(); k2[:, 64:] = 0; q2 = q
l, r, o = run(k2, q2); print("h) random, keys >= 64 zero: max |lse - fp64| %.3e" % (l.double() - r).abs().max().item(), l[:4].tolist(), r[:4].tolist())
