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


def _declared():
    txt = open(os.path.join(ROOT, "include", "hupr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hupr_[a-z0-9_]+)\s*\(", txt)))


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
