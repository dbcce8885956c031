"""INTEGRATION.md section 2 -- the ctypes stub a maintainer of the reference would add -- executed VERBATIM, so that the
documented binding cannot rot: the code blocks are read out of the markdown file, the only substitution is the
library's absolute path for the bare file name, and the three documented methods are then run as methods of a
reference-shaped object against the reference's own goldens."""
import os
import re

import numpy as np
import pytest

from conftest import REPO, PKG_ROOT, assert_close

pytestmark = pytest.mark.gpu


def _section2_blocks():
    txt = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = txt[txt.index("## 2. Minimal ctypes stub"):txt.index("## 3. Batches")]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def test_documented_stub_runs_as_written(golden):
    blocks = _section2_blocks()
    assert len(blocks) == 4                                  # the stub module + _map_b + _viterbi + _forward
    lib = os.path.join(PKG_ROOT, "lnasr", "liblnasr_hip.so")
    assert blocks[0].count('"liblnasr_hip.so"') == 1
    ns = {}
    exec(compile(blocks[0].replace('"liblnasr_hip.so"', repr(lib)), "INTEGRATION.md#stub", "exec"), ns)
    methods = {}
    for src in blocks[1:]:
        exec(compile(src, "INTEGRATION.md#method", "exec"), ns, methods)
    assert sorted(methods) == ["_forward", "_map_b", "_viterbi"]
    Ref = type("RefGMMHMM", (), methods)                     # the reference's class, as far as these methods need it

    for tag, prec in (("", np.double), ("_f32", np.float32)):
        g = golden("gmmhmm_small" + tag)
        gh = Ref()
        gh.n, gh.m, gh.d, gh.precision = 5, 3, 2, prec
        gh.A, gh.pi, gh.w, gh.mu, gh.si = (np.ascontiguousarray(g[k], dtype=np.float64) for k in ("A", "pi", "w", "mu", "si"))
        obs = np.ascontiguousarray(g["obs"])
        gh._map_b(obs)
        tol = dict(rtol=2e-5, atol=2e-5) if tag else dict(rtol=1e-12, atol=1e-12)
        assert gh.b.shape == (5, 40) and gh.bm.shape == (5, 3, 40) and gh.b.dtype == prec
        assert_close(gh.b, g["b"], **tol), assert_close(gh.bm, g["bm"], **tol)
        gh.b = g["b"]                                        # trellises on the golden emissions: bit-identical
        v, bt, path = gh._viterbi(obs)
        assert np.array_equal(v, g["v"]) and np.array_equal(bt, g["bt"]) and np.array_equal(path, g["path"])
        alpha = gh._forward(obs)
        assert_close(alpha, g["alpha"], **(dict(rtol=3e-5, atol=3e-5) if tag else dict(rtol=1e-12, atol=1e-12)))
        v2, _, path2 = gh._viterbi(obs, True)                # remap=True: through the stub's own _map_b
        assert np.array_equal(path2, g["path"])
