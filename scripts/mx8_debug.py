"""Localise a disagreement of the block-scaled fp8 attention: workspace bytes against a torch restatement of the quantiser."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hupr_amd import functional as F_
torch.manual_seed(5)
B, N, C = 2, 1024, 64
L, rt = F_.rt.lib(), F_.rt
k = (torch.randn(B, N, C, device="cuda") * 0.6).bfloat16().float()
q = (torch.randn(B, N, C, device="cuda") * 0.6).bfloat16().float()
v = torch.randn(B, N, C, device="cuda").bfloat16().float()
y = torch.zeros((B, N, 4 * C), dtype=torch.bfloat16, device="cuda"); y[..., :C] = k; y[..., C:2 * C] = q
vb = v.bfloat16().contiguous()
nb = L.hupr_attn_mx8_ws_bytes(B, N, C)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
rt.check(L.hupr_attn_mx8_quant_level(rt.ptr(y), rt.ptr(y), rt.ptr(vb), rt.ptr(vb), B, N, C, rt.ptr(ws), nb, rt.stream()))
torch.cuda.synchronize()
al = lambda n: (n + 255) // 256 * 256
rows = B * N
o_y8, o_ysc = 0, al(rows * 256)
o_vt8 = o_ysc + al(rows * 8)
o_vsc = o_vt8 + al(rows * 64)
y8 = ws[o_y8:o_y8 + rows * 256].view(torch.float8_e4m3fn).float().reshape(rows, 8, 32)
ysc = ws[o_ysc:o_ysc + rows * 8].float().reshape(rows, 8, 1)
yhat = (y8 * torch.exp2(ysc - 127)).reshape(B, N, 256)
print("scale bytes of row 0:", ws[o_ysc:o_ysc + 8].tolist(), " amax per block of row 0:", y[0, 0].float().reshape(8, 32).abs().amax(1).tolist())
print("projection bytes: max-abs error of the dequantised values %.3e (values up to %.2f)" % ((yhat - y.float()).abs().max().item(), y.float().abs().max().item()))
vt8 = ws[o_vt8:o_vt8 + rows * 64].view(torch.float8_e4m3fn).float().reshape(B, N // 64, 64, 2, 32)      # [tile][ch][h][m]
vsc = ws[o_vsc:o_vsc + rows // 64 * 128].float().reshape(B, N // 64, 64, 2, 1)
vsc_m = torch.cat([vsc[..., 0:1, :].expand(-1, -1, -1, 2, 16), vsc[..., 1:2, :].expand(-1, -1, -1, 2, 16)], -1)      # scale of block t = m >> 4
vhat_p = vt8 * torch.exp2(vsc_m - 127)                                     # [b][tile][ch][h][m = 16 t + 4 g + i]
vhat = vhat_p.reshape(B, N // 64, 64, 2, 2, 4, 4).permute(0, 1, 4, 5, 3, 6, 2).reshape(B, N, C)      # key = 32 t + 8 g + 4 h + i
print("value bytes: max-abs error of the dequantised, un-permuted values %.3e" % (vhat - v).abs().max().item())
out, lse = torch.empty(B, N, C, device="cuda"), torch.empty(B, N, device="cuda")
rt.check(L.hupr_attn_mx8_fwd(rt.ptr(ws), 0, 0, 0, 1, 0, None, rt.ptr(out), rt.ptr(lse), None, 0, B, N, C, nb, rt.stream()))
torch.cuda.synchronize()
S = torch.einsum("bjc,bqc->bjq", yhat[..., :C].double(), yhat[..., C:2 * C].double())
print("lse: kernel", lse[0, :4].tolist(), " restated", torch.logsumexp(S, 1)[0, :4].tolist())
ref = torch.einsum("bjq,bjc->bqc", torch.softmax(S, 1), vhat.double())
print("out[0,0,:4]: kernel", out[0, 0, :4].tolist(), " restated", ref[0, 0, :4].tolist())
print("rel-L2 %.3e" % ((out.double() - ref).norm() / ref.norm()).item())
# one key tile only (N = 128 -> 2 tiles) to separate the first tile from the running update
