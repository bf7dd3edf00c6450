"""CPU: the C-ABI library loads and exports every symbol include/nlt_hip.h declares, and the
Python binding table covers exactly that set (no compute calls without a GPU)."""
import ctypes
import os
import re

import nlt_amd
from nlt_amd import capi as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'nlt_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return set(re.findall(r'\b(nlt_[a-z0-9_]+)\s*\(', hdr))


def test_library_is_built_in_tree():
    assert os.path.exists(C.LIB_PATH), "run `python __graft_entry__.py` (build()) first"
    assert os.path.dirname(C.LIB_PATH) == os.path.join(ROOT, 'neural-light-transport_amd')


def test_every_header_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 11
    L = ctypes.CDLL(C.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "libnlt_hip.so does not export %s" % s
    assert syms == set(C.SIGNATURES), (syms ^ set(C.SIGNATURES))


def test_metadata_calls_work_without_gpu():
    L = C.lib()
    assert b'gfx950' in L.nlt_version()
    assert L.nlt_status_string(0) == b'ok'
    assert L.nlt_status_string(-2).startswith(b'unsupported')
    # packed sizes: taps * ceil16(c0)+ceil16(c1) chunks * N/16 tiles * 256 floats
    assert C.packed_weight_floats(C.CONV_K2S2, 32, 0, 16) == 4 * 2 * 1 * 256
    assert C.packed_weight_floats(C.DECONV_K2S2, 8, 32, 4) == 1 * 3 * 1 * 256
    assert C.packed_weight_floats(C.CONV1X1, 4, 32, 12) == 1 * 3 * 1 * 256
    assert C.packed_weight_floats(9, 4, 0, 4) == -1


def test_no_cpu_fallback_for_cpu_tensors():
    import pytest
    import torch
    x = torch.zeros(4)
    with pytest.raises(C.NLTError):
        C.mul_forward(x, x) if False else C._ptr(x)
