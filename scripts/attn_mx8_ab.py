"""Config 5, block-scaled form (csrc/attention_mx8.hip) against the bf16 flash kernel at the MSCSA level-1 shape (B = 32, N = 4096,
C = 64): kernel times of one attention and of the level's quantisation step, and the error of each against fp64 on a sample.
usage (GPU box): python scripts/attn_mx8_ab.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, C = 4096, 64
L, rt = F_.rt.lib(), F_.rt
g = torch.Generator(device="cuda").manual_seed(1)
ya = (torch.randn(B, N, 4 * C, device="cuda", generator=g) * 0.5).bfloat16()
ye = (torch.randn(B, N, 4 * C, device="cuda", generator=g) * 0.5).bfloat16()
va32 = torch.randn(B, N, C, device="cuda", generator=g)
ve32 = torch.randn(B, N, C, device="cuda", generator=g)
va, ve = va32.bfloat16(), ve32.bfloat16()
out, lse = torch.empty(B, N, C, device="cuda"), torch.empty(B, N, device="cuda")
cat = torch.empty(B, N, 4 * C, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(L.hupr_attn_mx8_ws_bytes(B, N, C), dtype=torch.uint8, device="cuda")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


quant = lambda: rt.check(L.hupr_attn_mx8_quant_level(rt.ptr(ya), rt.ptr(ye), rt.ptr(va), rt.ptr(ve), B, N, C, rt.ptr(ws), ws.numel(), rt.stream()))
mx = lambda: rt.check(L.hupr_attn_mx8_fwd(rt.ptr(ws), 0, 0, 1, 1, 0, rt.ptr(va32), rt.ptr(out), rt.ptr(lse), rt.ptr(cat), 4 * C, B, N, C, ws.numel(), rt.stream()))
bf = lambda: rt.check(L.hupr_attn_fwd_bf16in_ld_ws(ya.data_ptr(), 4 * C, ye.data_ptr() + C * 2, 4 * C, rt.ptr(va), rt.ptr(va32), rt.ptr(out), rt.ptr(lse),
                                                 rt.ptr(cat), 4 * C, B, N, C, None, 0, rt.stream()))
quant()
tq, tm, tb = timed(quant), timed(mx), timed(bf)
fl = 4.0 * N * N * C * B
print("B = %d, N = %d, C = %d" % (B, N, C))
print("level quantisation (2 x (B,N,256) projections + 2 value maps): %.1f us  (%.0f GB/s on %.0f MB read + written)" %
      (tq, (2 * B * N * 320 * 2 + 2 * B * N * 330) / tq / 1e3, (2 * B * N * 320 * 2 + 2 * B * N * 330) / 1e6))
print("one attention forward: mx8 %.1f us (%.0f TF/s)   bf16 %.1f us (%.0f TF/s)" % (tm, fl / tm / 1e6, tb, fl / tb / 1e6))
print("a level's four attentions: mx8 %.1f us incl. quantisation   bf16 %.1f us" % (tq + 4 * tm, 4 * tb))
# accuracy on sample 0 against fp64
k0, q0, v0 = ya[:1, :, :C].double(), ye[:1, :, C:2 * C].double(), va[:1].double()
S = torch.einsum("bjc,bqc->bjq", k0, q0)
ref = torch.einsum("bjq,bjc->bqc", torch.softmax(S, 1), v0)
mx(); torch.cuda.synchronize(); e_mx = ((out[:1].double() - va32[:1].double() - ref).norm() / ref.norm()).item()
bf(); torch.cuda.synchronize(); e_bf = ((out[:1].double() - va32[:1].double() - ref).norm() / ref.norm()).item()
print("rel-L2 of the attention term against fp64: mx8 %.3e   bf16 %.3e" % (e_mx, e_bf))
