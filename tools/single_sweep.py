#!/usr/bin/env python3
"""Single-GMM (N=1) emission kernel: time against the number of frames (development aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
from lnasr import _ffi, synth  # noqa: E402

ctx = _ffi.default_ctx()
lib = _ffi.lib()
M, D, T = 16, 39, 1000
mdl = synth.synth_model(1, M, D, seed=1)
ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
for U in (256, 1024, 4096, 16384):
    obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1))
    b = ctx.empty((U, T, 1), np.float64)
    fn = lambda: _ffi.check(lib.lnasr_gmm_logprob(ctx.handle, obs.ptr, U, T, None, 0, b.ptr, None))
    fn()
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(5):
        fn()
    ctx.sync()
    ctx.prof_enable(False)
    for k, (n, ms) in ctx.prof_report().items():
        fr = U * T
        print("U=%6d %-22s %8.3f ms  %.3e frames/s  %.0f GB/s" % (U, k, ms / n, fr / (ms / n) * 1e3, fr * 320 / (ms / n) / 1e6))
    del obs, b
