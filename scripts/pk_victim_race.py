"""Which packed-fp32 operand form is corrupted beside hupr_k_conv_halo_bf16<64, 64>?  Self-checking victims (scripts/probes/pk_victim.hip:
the xor of an even number of evaluations of the same instruction on the same operands must be 0) run alone and beside the level-3
convolution on a second stream.  usage: python scripts/pk_victim_race.py"""
import ctypes, os, subprocess, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from hupr_amd import functional as F_
F_.set_math("bf16")
so = "/tmp/libpk_victim.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(root, "scripts", "probes", "pk_victim.hip")])
V = ctypes.CDLL(so)
V.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(3)
data = torch.randn(1 << 21, device=dev, generator=gen)
x3 = torch.randn(32, 2, 16, 16, 256, device=dev, generator=gen).bfloat16()
w3 = (torch.randn(256, 256, 3, 3, 3, device=dev, generator=gen) * 0.02).requires_grad_(True)
w3s = w3[:64].detach().clone().requires_grad_(True)
side = F_.side_stream(dev)
forms = {0: "v_pk_add_f32 (no operand swizzle)", 6: "v_pk_fma_f32 (no operand swizzle)", 2: "v_pk_fma_f32 op_sel_hi:[1,0,0] (the attention soft-max's form)",
         3: "v_pk_mul_f32 op_sel_hi:[0,1] (complex multiply, first half)", 7: "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]",
         1: "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1] (a - i b)", 4: "v_pk_fma_f32 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo (complex multiply, second half)",
         5: "v_fma_f32 (scalar control)",
         8: "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (swap, no negation)", 9: "v_pk_add_f32 neg_hi:[0,1] (negation, no swap)",
         15: "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1] (a - b)",
         10: "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] (a + i b)", 12: "v_pk_add_f32 op_sel:[0,1]", 13: "v_pk_add_f32 op_sel_hi:[1,0]",
         14: "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]",
         11: "v_pk_fma_f32 d, b, {1,-1}, a op_sel:[1,0,0] op_sel_hi:[0,1,1] (a - i b as an fma)",
         16: "v_pk_add_f32 d, b, a op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0] (a - i b, swizzle on SRC0: the chain's form since round 4)",
         17: "v_pk_add_f32 d, b, a op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] (a + i b, swizzle on SRC0: the chain's form since round 4)"}
for agg_name, wagg in (("alone", None), ("beside conv_halo_bf16<64, 64>", w3)):
    for form, name in forms.items():
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        n = 0
        for _ in range(6):
            side.wait_stream(torch.cuda.current_stream())
            if wagg is not None:
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(100):
                        F_.conv(x3, wagg, None, None, (1, 1, 1))
            for _ in range(50):
                assert V.pk_victim_launch(form, data.data_ptr(), bad.data_ptr(), 2048, 64, torch.cuda.current_stream().cuda_stream) == 0
                n += 1
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        print("%-30s %-112s: %8d of %d threads saw a changed result" % (agg_name, name, bad.item(), n * 2048 * 256), flush=True)
