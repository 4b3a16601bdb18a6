"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports exactly the
symbols include/ofb_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'ofb_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(ofb_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from ofb_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    declared = header_symbols()
    assert declared, 'no symbols parsed from include/ofb_hip.h'
    for s in declared:
        assert hasattr(lib, s), f'{s} declared in include/ofb_hip.h but not exported by libofb_hip.so'
    assert sorted(hip.SYMBOLS) == declared, (set(hip.SYMBOLS) ^ set(declared))


def test_no_cpu_fallback():
    import torch
    from ofb_amd import hip
    a = torch.zeros(4, 4)
    with pytest.raises(hip.OfbError):
        hip.gemm(a, a, a, 4, 4, 4, 4, 4, 4, 1, 1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'once-for-both_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'oracle' not in src.replace('no oracle', ''), f'{f} mentions the oracle'
