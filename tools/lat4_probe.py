#!/usr/bin/env python3
"""Reproduces the one known performance cliff of fwdbwd_lat4 (advisor, round 5; LABLOG 8.7): a frame whose emission LEVEL is set by
a state that holds no mass makes the level-based shift miss by thousands of nats, the e row underflows to zero, and -- the kernel
never rebuilding e from the repaired log-domain row -- every column of every later frame takes the exact path (correct results;
lnasr_debug_fallback_columns counts ~N per frame instead of ~0).  Round 6 built a re-seed three ways; each cost the clean path
15-20 % and none was shipped.      python tools/lat4_probe.py"""
import sys, numpy as np
sys.path.insert(0, "ln-asr_amd")
from lnasr import _ffi
ctx = _ffi.default_ctx()
N, T, t0 = 128, 60, 20
rng = np.random.default_rng(0)
a = rng.random((N, N)) + 1e-3
a[:, N - 1] = 0.0          # state 127 can only be entered from itself
a[N - 1, :] = 1e-3
a[N - 1, N - 1] = 1.0
A = np.log(a / a.sum(1, keepdims=True))
pi = np.log(np.full(N, 1.0 / N))
pi[N - 1] = -2500.0        # ... and starts with (almost) no mass
b = -40.0 * rng.random((1, T, N))
b[0, t0, :N - 1] -= 3000.0   # one frame that only the mass-less state explains: the level of the row says 0, the mass sits 3000 lower
ctx.upload_hmm(A, pi)
ctx.fallback_columns()
alpha, ll = ctx.forward(b)
n_fwd = ctx.fallback_columns()
beta = ctx.backward(b)
n_bwd = ctx.fallback_columns()
print("fallback columns forward", n_fwd, "backward", n_bwd, " (frames after the outlier: %d, columns %d)" % (T - t0, N))
sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("o", "oracle/lnasr_oracle.py"); o = importlib.util.module_from_spec(spec); spec.loader.exec_module(o)
ob = np.ascontiguousarray(b[0].T)
with np.errstate(all="ignore"):
    ra = o.forward(A, pi, ob)
print("alpha max err", np.nanmax(np.abs(alpha[0] - ra) / np.maximum(1, np.abs(ra))), "loglik", ll[0], o.loglik(ra))
