"""Reproducer for round 3's non-reproducible two-stream step: the resampling forward (hupr_k_interp_fwd) AS hipcc's
SLP vectoriser compiled it (op_sel'ed v_pk_mul_f32 / v_pk_fma_f32) returns wrong sums when its launches share the chip with another
stream's hupr_k_conv_halo_bf16<64, 64> (the level-3 convolution at B = 32) — never alone.

The library itself is built with -fno-slp-vectorize and scalar FMAs in that kernel; this script compiles csrc/spatial.hip a second
time WITHOUT that flag into /tmp (hipcc, on the GPU box), takes the victim from that copy (hupr_debug_interp_packed(1)) and the
aggressors from the library.  usage: python scripts/interp_race.py            (PACKED=0: the scalar form of the same copy)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import pose_fit
from hupr_amd import functional as F_, synth
from hupr_amd.config_tree import load_config
from hupr_amd.models import HuPRNet

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd", "csrc")
so = "/tmp/libhupr_interp_slp.so"
# WHOLE_SIMD=1: the victim claims all 512 registers of a lane, so that no other wave can be resident on the SIMD it runs on
extra = ["-DHUPR_PROBE_WHOLE_SIMD"] if os.environ.get("WHOLE_SIMD", "0") == "1" else []
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(root, "include"),
                       "-I" + csrc, os.path.join(csrc, "spatial.hip"), os.path.join(csrc, "core.hip"), "-o", so] + extra)
V = ctypes.CDLL(so)
V.hupr_interp_linear_fwd_bf16act.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 10 + [ctypes.c_void_p]
V.hupr_debug_interp_packed.argtypes = [ctypes.c_int]
packed = int(os.environ.get("PACKED", "1"))
V.hupr_debug_interp_packed(packed)

F_.set_math("bf16")
L, rt = F_.rt.lib(), F_.rt
cfg = load_config()
net = HuPRNet(cfg).cuda().eval()
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.hupr_state(1).items()})
F_.invalidate_packed()
dev = torch.device("cuda")
h, v, joints = pose_fit.scene_batch(32, np.random.default_rng(1), torch.Generator(device=dev).manual_seed(2), dev)
side = F_.side_stream(dev)
x = torch.randn(32, 4, 32, 32, 128, device=dev).relu().bfloat16()
B, G, H, W, C = x.shape


def interp(lib=V):
    y = torch.empty((B, 2, 16, 16, C), dtype=x.dtype, device=dev)
    assert lib.hupr_interp_linear_fwd_bf16act(rt.ptr(x), rt.ptr(y), B, G, H, W, 2, 16, 16, C, C, C, rt.stream()) == 0
    return y


ref = interp()
ref_lib = interp(L)
torch.cuda.synchronize()
print("victim: the %s form of the SLP-enabled copy; alone it differs from the library's kernel in %d elements (max |diff| %.3e)" %
      ("packed" if packed else "scalar", (ref != ref_lib).sum().item(), (ref.float() - ref_lib.float()).abs().max().item()))
m1 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
big = torch.randn(64 << 20, device=dev)
xa = torch.randn(32, 8, 64, 64, 64, device=dev).bfloat16()
wa = (torch.randn(64, 64, 3, 3, 3, device=dev) * 0.02).requires_grad_(True)
x2 = torch.randn(32, 4, 32, 32, 128, device=dev).bfloat16()
w2 = (torch.randn(128, 128, 3, 3, 3, device=dev) * 0.02).requires_grad_(True)
x3 = torch.randn(32, 2, 16, 16, 256, device=dev).bfloat16()
w3 = (torch.randn(256, 256, 3, 3, 3, device=dev) * 0.02).requires_grad_(True)
w3s = w3[:64].detach().clone().requires_grad_(True)
wm = (torch.randn(64, 64, 8, 1, 1, device=dev) * 0.05).requires_grad_(True)


def rep(fn, n):
    def f():
        with torch.no_grad():
            for _ in range(n):
                fn()
    return f


AGG = [
    ("nothing", None),
    ("torch bf16 GEMM 8192^3", rep(lambda: m1 @ m1, 6)),
    ("torch elementwise (256 MB add)", rep(lambda: big + 1.0, 40)),
    ("MNet front end", rep(lambda: net.REchirpNet(v), 10)),
    ("3x3x3 conv level 1 (256-voxel MFMA kernel)", rep(lambda: F_.conv(xa, wa, None, None, (1, 1, 1)), 20)),
    ("3x3x3 conv level 2 (512-voxel MFMA kernel)", rep(lambda: F_.conv(x2, w2, None, None, (1, 1, 1)), 60)),
    ("3x3x3 conv level 3: conv_halo_bf16<64, 64>", rep(lambda: F_.conv(x3, w3, None, None, (1, 1, 1)), 100)),
    ("3x3x3 conv level 3, Co = 64: <32, 64>", rep(lambda: F_.conv(x3, w3s, None, None, (1, 1, 1)), 100)),
    ("temporal merge (LDS-DMA stream kernel)", rep(lambda: F_.TemporalMergeFn.apply(xa, wm), 60)),
    ("the whole RE encoder", rep(lambda: net.REradarEncoder(net.REchirpNet(v)), 1)),
]
if os.environ.get("SPIN", "0") == "1":
    # aggressors that only occupy registers: spin_<N> claims N VGPRs per lane and loops on four v_add_f32 (scripts/probes/vgpr_spin.hip)
    spin_so = "/tmp/libvgpr_spin.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(root, "scripts", "probes", "vgpr_spin.hip"), "-o", spin_so])
    SP = ctypes.CDLL(spin_so)
    SP.spin_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
    sink = torch.zeros(4, device=dev)
    blocks = int(os.environ.get("SPIN_BLOCKS", "2048"))

    def spin(nv):
        def f():
            assert SP.spin_launch(nv, 20000, blocks, rt.ptr(sink), rt.stream()) == 0
        return f
    AGG = [("nothing", None)] + [("a spin kernel holding %3d VGPRs" % nv, spin(nv)) for nv in (56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 160, 192, 224)]
    SP.spin_class_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p]
    srcbuf = torch.arange(64 << 20, device=dev, dtype=torch.int32)

    def cls(which, iters):
        def f():
            assert SP.spin_class_launch(which, iters, blocks, rt.ptr(sink), rt.ptr(srcbuf), srcbuf.numel(), rt.stream()) == 0
        return f
    AGG += [("112 VGPRs, loop of v_mfma_f32_32x32x16_bf16", cls(0, 4000)), ("112 VGPRs + 47 KB LDS, loop of ds_read_b128", cls(1, 20000)),
            ("112 VGPRs + 47 KB LDS, loop of ds_write_b128", cls(2, 20000)), ("112 VGPRs, loop of s_barrier", cls(3, 20000)),
            ("112 VGPRs, loop of global_load_dwordx4", cls(4, 2000)), ("112 VGPRs, loop of v_cvt_pk_bf16_f32 + v_permlane32_swap", cls(5, 20000))]
    if os.environ.get("SPIN_ONLY_CLASSES", "0") == "1":
        AGG = AGG[-6:]
if os.environ.get("ABLATE"):      # narrow the aggressor down: bit0 = no halo fill, bit2 = no epilogue stores (hupr_debug_halo_ablate)
    L.hupr_debug_halo_ablate(int(os.environ["ABLATE"]))
    AGG = [a for a in AGG if "level 3: conv" in a[0]]
for name, agg in AGG:
    bad = tot = 0
    first = ""
    for i in range(12):
        side.wait_stream(torch.cuda.current_stream())
        if agg is not None:
            with torch.cuda.stream(side):
                agg()
        outs = [interp() for _ in range(200)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for o in outs:
            tot += 1
            if not torch.equal(o, ref):
                bad += 1
                if not first:
                    idx = (o != ref).reshape(-1).nonzero().reshape(-1)
                    first = "%d elements, channels %s, got / alone %s" % (
                        idx.numel(), sorted(set((idx % C).tolist()))[:10],
                        " ".join("%.4f/%.4f" % (a, b) for a, b in zip(o.reshape(-1)[idx[:4]].float().tolist(), ref.reshape(-1)[idx[:4]].float().tolist())))
    print("beside %-46s: %4d of %d launches differ from the launch alone%s" % (name, bad, tot, ("   [first: " + first + "]") if bad else ""), flush=True)
