"""CPU checks of the boundary: the C-ABI library builds/loads and exports every symbol that
include/lnasr_hip.h declares; the ctypes table covers the same set; no compute call is made."""
import ctypes
import os
import re

import pytest

from conftest import REPO, PKG_ROOT, has_gpu

HEADER = os.path.join(REPO, "include", "lnasr_hip.h")
LIB = os.path.join(PKG_ROOT, "lnasr", "liblnasr_hip.so")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lnasr_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        import importlib.util
        spec = importlib.util.spec_from_file_location("lnasr_build", os.path.join(PKG_ROOT, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return ctypes.CDLL(LIB)


def test_header_symbols_exported(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(built_lib, s)]
    assert not missing, missing


def test_ctypes_table_matches_header(built_lib):
    from lnasr import _ffi
    assert sorted(_ffi.SIGNATURES) == declared_symbols()
    assert _ffi.lib().lnasr_abi_version() == 1


def test_fails_loudly_without_device(built_lib):
    """No GPU -> creating a context raises; nothing falls back to the CPU."""
    if has_gpu():
        pytest.skip("a GPU is present")
    from lnasr import _ffi
    assert _ffi.device_count() == 0
    with pytest.raises(_ffi.LnasrError, match="no HIP device"):
        _ffi.Context(0)
    from lnasr.hmm import HMM
    import numpy as np
    h = HMM(2, 3, np.zeros((2, 2)), np.zeros((2, 3)), np.zeros(2))
    with pytest.raises(_ffi.LnasrError):
        h.decode(np.array([0, 1, 2]))


def test_product_never_imports_oracle():
    """The package must not reference the oracle (test infrastructure only)."""
    for root, _, files in os.walk(PKG_ROOT):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "lnasr_oracle" not in txt and "oracle/" not in txt, os.path.join(root, f)
