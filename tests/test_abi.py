"""CPU (-m "not gpu"): the C-ABI library builds for gfx950, loads, and exports every symbol
include/hupr.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from hupr_amd import runtime
    return runtime


def _declared(headers=("hupr.h", "hupr_debug.h")):
    names = set()
    for h in headers:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(hupr_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_operator_header_holds_no_debug_entry_points():
    """include/hupr.h is the operator contract only; test / profiling switches live in include/hupr_debug.h."""
    assert not [n for n in _declared(("hupr.h",)) if n.startswith("hupr_debug_")]
    dbg = _declared(("hupr_debug.h",))
    assert dbg and all(n.startswith("hupr_debug_") for n in dbg) and len(dbg) <= 12


def test_every_declared_symbol_is_exported(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    names = _declared()
    assert len(names) >= 6
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n


def test_python_binding_covers_header(built):
    assert sorted(built.SIGNATURES) == _declared()


def test_host_side_argument_errors(built):
    L = built.lib()
    assert L.hupr_version() >= 100
    assert L.hupr_fft_chain_ws_bytes(0) == 0
    assert L.hupr_fft_chain_ws_bytes(3) == 3 * 16 * 64 * 12 * 8
    # empty batch is a no-op, null pointers are rejected before any launch
    assert L.hupr_fft_chain_c64(None, 0, None, None, 0, None) == 0
    assert L.hupr_fft_chain_c64(None, 1, None, None, 0, None) == -1
    assert b"null" in L.hupr_last_error()
    assert L.hupr_fft_chain_c64(None, -1, None, None, 0, None) == -1


def test_no_cpu_fallback(built):
    import torch
    from hupr_amd import preprocessing
    with pytest.raises(built.HuprError):
        preprocessing.fft_chain(torch.zeros((1, 4, 192, 256, 2), dtype=torch.int16))


def test_no_two_source_packed_instruction_swizzles_src1():
    """Round 4: v_pk_add_f32 / v_pk_mul_f32 ... op_sel:[x,1] — a two-source packed instruction whose SRC1 low
    lane reads the high half — returns changed results when its wave shares the chip with hupr_k_conv_halo_bf16<64, 64> on another
    stream (scripts/probes/pk_victim.hip: 11 % of the threads, never alone, never with the swizzle on SRC0).  The build keeps the
    device assembly of every source (csrc/Makefile, --save-temps=obj) and refuses to link such an instruction; this test reads the
    same listings."""
    import glob
    import re
    lst = glob.glob(os.path.join(ROOT, "hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if not lst:
        pytest.skip("no device listings (library not built in this tree)")
    assert len(lst) >= 15
    bad = re.compile(r"v_pk_[a-z0-9_]+ v\[?[0-9:]+\]?, [^,]+, [^,]+ .*op_sel:\[[01],1\]")
    n_packed = 0
    for f in lst:
        for line in open(f):
            if "v_pk_" in line:
                n_packed += 1
                assert not bad.search(line), (os.path.basename(f), line.strip())
    assert n_packed > 5000          # the listings are the real ones (the FFT butterflies alone hold ~6 000 packed instructions)


def test_hot_kernels_keep_their_register_and_lds_budgets():
    """The kernels that hold most of the step must not start spilling vector registers to scratch or lose their occupancy through a
    source change that still compiles and still passes the numerics tests (the 256-voxel convolution is built
    around two waves per SIMD = at most 256 unified registers per wave and 155 KB of LDS; the ring-of-3 variant of round 4 was abandoned
    at 256 VGPRs + 23 spills).  Read from the code-object metadata in the listings the build keeps."""
    import glob
    lst = glob.glob(os.path.join(ROOT, "hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if not lst:
        pytest.skip("no device listings (library not built in this tree)")
    meta = {}
    for f in lst:
        txt = open(f).read()
        for blk in re.findall(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target)", txt, flags=re.S):
            get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk).group(1)
            meta[get("name")] = dict(vgpr=int(get("vgpr_count")), spill=int(get("vgpr_spill_count")),
                                     scratch=int(get("private_segment_fixed_size")), lds=int(get("group_segment_fixed_size")))
    assert len(meta) >= 150
    # (substring of the mangled name, max unified VGPRs as the code object states them, max static LDS bytes).  512-thread kernels
    # with one workgroup per CU run two waves per SIMD: 256 registers each; the FFT kernels are sized for >= 3-4 waves per SIMD
    budgets = [("hupr_k_conv_halo256m_bf16ILi4ELi8ELi8ELi3ELi0E", 256, 160 * 1024),
               ("hupr_k_conv_halo256m_bf16ILi2ELi8ELi16ELi3ELi0E", 256, 160 * 1024),
               ("hupr_k_conv_halo256m_bf16ILi1ELi16ELi16ELi1E", 256, 160 * 1024),
               ("hupr_k_conv_halo256m_bf16ILi8ELi8ELi8ELi3ELi0ELi1E", 256, 160 * 1024),   # 32 output channels: the 8 x 8 x 8 tile (first-layer input gradient)
               ("hupr_k_conv_halo256m_bf16ILi4ELi8ELi8ELi3ELi1E", 256, 160 * 1024),      # fused BatchNorm statistics, one output tile (level 1)
               ("hupr_k_wgrad_halo_m16ILb1E", 256, 160 * 1024), ("hupr_k_wgrad_halo_m16ILb0E", 256, 160 * 1024),
               ("hupr_k_attn_fwd_pp64ILb1E", 256, 160 * 1024), ("hupr_k_attn_bwd_dkvILi256EDF16bLi2ELb1E", 512, 160 * 1024),
               ("hupr_k_attn_fwdILi256EDF16bLb0ELb1E", 512, 160 * 1024), ("hupr_k_attn_bwd_dqILi256EDF16bLb1E", 512, 160 * 1024),
               ("hupr_k_wgrad_halo_gldsILb1ELb0E", 256, 160 * 1024),
               ("hupr_k_attn_fwd_pp64ILb0E", 256, 160 * 1024),
               ("hupr_k_attn_bwd_dkv512", 256, 64 * 1024),
               ("hupr_k_attn_fwdILi128EDF16bLb0ELb1E", 256, 80 * 1024),      # level-2 forward: two workgroups per CU since round 6 (316 registers = one wave per SIMD before)
               ("hupr_k_attn_bwd_dqILi64EDF16bLb", 256, 64 * 1024),      # both forms: plain and QS (round 5)
               ("hupr_k_attn_bwd_dqILi128E", 256, 64 * 1024),           # level-2 dQ on 32-key tiles: two workgroups per CU (358 registers before round 6)
               ("hupr_k_gcn_wxILb0E", 168, 64 * 1024), ("hupr_k_gcn_wxILb1E", 168, 64 * 1024), ("hupr_k_gcn_dw", 168, 64 * 1024),
               ("hupr_k_doppler_rangeILi0ELb1E", 128, 40 * 1024), ("hupr_k_angle", 64, 16 * 1024)]
    for pat, max_vgpr, max_lds in budgets:
        hits = {n: m for n, m in meta.items() if pat in n}
        assert hits, "no kernel matches %s" % pat
        for n, m in hits.items():
            assert m["spill"] == 0 and m["scratch"] == 0, (n, m)
            assert m["vgpr"] <= max_vgpr and m["lds"] <= max_lds, (n, m)
    # the multi-tile / level-3 statistics variants of the convolution sit at 256 registers and park one or two values in scratch in the
    # PROLOGUE, reloaded after the tile loop (checked in the listing: no scratch access between the loop header and its back edge)
    for pat in ("hupr_k_conv_halo256m_bf16ILi4ELi8ELi8ELi3ELi2E", "hupr_k_conv_halo256m_bf16ILi2ELi8ELi16ELi3ELi1E"):
        hits = {n: m for n, m in meta.items() if pat in n}
        assert hits, "no kernel matches %s" % pat
        for n, m in hits.items():
            assert m["spill"] <= 2 and m["vgpr"] <= 256 and m["lds"] <= 160 * 1024, (n, m)
