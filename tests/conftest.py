"""Shared test plumbing: marker registration, module loading by path, fixtures."""
import importlib.util
import os
import sys

import numpy as np
import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(TESTS)
GOLDEN = os.path.join(TESTS, "golden")
PKG_ROOT = os.path.join(REPO, "ln-asr_amd")

if PKG_ROOT not in sys.path:
    sys.path.insert(0, PKG_ROOT)          # `import lnasr` -> the MI355X drop-in package


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def _load(name, path):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/lnasr_oracle.py header)."""
    return _load("lnasr_oracle", os.path.join(REPO, "oracle", "lnasr_oracle.py"))


@pytest.fixture(scope="session")
def synth():
    return _load("lnasr_synth", os.path.join(PKG_ROOT, "lnasr", "synth.py"))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return get


def has_gpu():
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


import contextlib


def tuned():
    """False when the process runs under fallback switches (tools/fallback_suite.sh): tests then check results only,
    not WHICH kernel produced them."""
    return not any(k.startswith("LNASR_NO_") or k in ("LNASR_F32_GENERIC", "LNASR_VIT", "LNASR_FWD_ONE_MAXU", "LNASR_FWD_LAT_MAXU", "LNASR_CHASE_MAXU", "LNASR_I8_MIN_FRAMES") for k in os.environ)


@contextlib.contextmanager
def env_switch(**kv):
    """Set environment switches for the block and put back what was there before (the fallback-suite runs set
    some of them for the whole process: a test must not clear those)."""
    import os
    from lnasr import _ffi
    before = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    _ffi.reload_switches()                 # the library reads its switches once per process
    try:
        yield
    finally:
        for k, v in before.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _ffi.reload_switches()


def assert_close(a, b, rtol=1e-10, atol=0.0, msg=""):
    """allclose that treats equal infinities (and NaN==NaN) as equal."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s shape %s vs %s" % (msg, a.shape, b.shape)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore"):
        err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = ~same & ~(err <= tol)
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError("%s: %d/%d mismatches, first at %s: %r vs %r" % (
            msg, bad.sum(), bad.size, tuple(i), a[tuple(i)], b[tuple(i)]))
