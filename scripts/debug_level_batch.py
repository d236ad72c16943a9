import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from hupr_amd import functional as F_
from test_ops_gpu import rnd
F_.set_math("bf16")
H, C = 64, 64
ra, re = rnd(1, 1, H, H, C, seed=160).cuda(), rnd(1, 1, H, H, C, seed=161).cuda()
ws = [rnd(C, C, 1, 1, seed=162 + i, scale=C ** -0.5).cuda().requires_grad_(True) for i in range(8)]
def run(cat):
    with torch.no_grad():
        return [o.clone() for o in F_.MSCSALevelFn.apply(ra, re, cat | 2, *ws)]
for cat in (1, 0):
    res = []
    for rep in range(3):
        for batch in (True, False):
            F_.ATTN_BATCH = batch
            res.append((batch, rep, run(cat)))
    F_.ATTN_BATCH = True
    base = res[0][2]
    for batch, rep, y in res:
        print("cat", cat, "batch", batch, "rep", rep, [int((a != b).sum().item()) for a, b in zip(y, base)], [float((a.float() - b.float()).abs().max()) for a, b in zip(y, base)])
pa = F_._proj_cat(ws[:4], C)
ref = torch.cat([w.detach().reshape(C, C) for w in ws[:4]], 0)
print("Wc equal", torch.equal(pa[0], ref), "Wq rows", [float((pa[1][j*C:(j+1)*C] / ref[j*C:(j+1)*C]).mean()) for j in range(4)])
