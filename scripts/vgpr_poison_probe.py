"""Does a kernel read registers it never wrote?  scripts/probes/vgpr_poison.hip fills every VGPR / AGPR of the chip with a pattern and
exits; the kernel under test is launched right behind it on the SAME stream (nothing runs concurrently) and compared with its own
output on clean registers.  Here: the resampling forward as hipcc's SLP vectoriser packs it (v_pk_mul / v_pk_fma_f32) and in its scalar form.
usage: python scripts/vgpr_poison_probe.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hupr_amd import functional as F_

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libvgpr_poison.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "probes", "vgpr_poison.hip"), "-o", so])
P = ctypes.CDLL(so)
P.poison_launch.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
F_.set_math("bf16")
rt = F_.rt
# the kernel under test comes from a copy of csrc/spatial.hip built WITHOUT -fno-slp-vectorize (see scripts/interp_race.py)
root = os.path.dirname(here)
csrc = os.path.join(root, "hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd", "csrc")
slp = "/tmp/libhupr_interp_slp.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(root, "include"),
                       "-I" + csrc, os.path.join(csrc, "spatial.hip"), os.path.join(csrc, "core.hip"), "-o", slp])
L = ctypes.CDLL(slp)
L.hupr_interp_linear_fwd_bf16act.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 10 + [ctypes.c_void_p]
L.hupr_debug_interp_packed.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
sink = torch.zeros(256, dtype=torch.int32, device=dev)
x = torch.randn(32, 4, 32, 32, 128, device=dev).relu().bfloat16()
B, G, H, W, C = x.shape


def interp():
    y = torch.empty((B, 2, 16, 16, C), dtype=x.dtype, device=dev)
    assert L.hupr_interp_linear_fwd_bf16act(rt.ptr(x), rt.ptr(y), B, G, H, W, 2, 16, 16, C, C, C, rt.stream()) == 0
    return y


for packed in (1, 0):
    L.hupr_debug_interp_packed(packed)
    ref = interp()
    torch.cuda.synchronize()
    for name, pat in (("NaN", 0x7fc00000), ("zero", 0), ("1.0f", 0x3f800000), ("-3.0f", 0xc0400000)):
        bad = 0
        worst = 0.0
        for _ in range(50):
            assert P.poison_launch(pat, rt.ptr(sink), rt.stream()) == 0
            y = interp()
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                bad += 1
                d = (y.float() - ref.float())
                worst = max(worst, float("nan") if torch.isnan(d).any() else d.abs().max().item())
        print("%s accumulation, registers pre-filled with %-5s: %2d of 50 launches differ from clean registers (max |diff| %s)" %
              ("packed" if packed else "scalar", name, bad, worst))
L.hupr_debug_interp_packed(0)
