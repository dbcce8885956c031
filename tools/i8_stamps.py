#!/usr/bin/env python3
"""Developer diagnostic (GPU box): where a wave of the integer matrix-core bank (csrc/gmm_i8.hip) spends a tile.  Builds a
SEPARATE library with -DLNASR_I8_STAMPS (every wave accumulates s_memtime deltas around its two phases and the two barriers of
a tile, and its lifetime in shader cycles and in 100 MHz ticks: the clock the kernel really ran at), runs the metric shard's
emission and prints the per-wave averages the library writes to stderr.

    python tools/i8_stamps.py
"""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "..", "ln-asr_amd")
spec = importlib.util.spec_from_file_location("lnasr_build", os.path.join(PKG, "build.py"))
bld = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bld)
os.environ["LNASR_LIB"] = bld.build(verbose=False, variant="i8stamps", defines=("-DLNASR_I8_STAMPS",))
os.environ["LNASR_I8_STAMPS"] = "1"
sys.path.insert(0, PKG)
import numpy as np  # noqa: E402
from lnasr import _ffi, synth  # noqa: E402

ctx = _ffi.default_ctx()
N, M, D, U, T = 128, 16, 39, 1024, 1000
mdl = synth.synth_model(N, M, D, seed=1)
ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
d_obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
b = ctx.gmm_logprob(d_obs)
for _ in range(3):
    b = ctx.gmm_logprob(d_obs, b_out=b)
ctx.sync()
