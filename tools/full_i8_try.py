#!/usr/bin/env python3
"""Development aid: the full-covariance bank on the integer matrix cores (csrc/gmm_full_i8.hip) against the fp64 matrix-core
kernel -- values, the model's range, time per 16,000 frames.     python tools/full_i8_try.py [N] [F]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ln-asr_amd"))
from lnasr import _ffi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
M, D = 16, 39
ctx = _ffi.Context(0)
mdl = synth.synth_model(N, M, D, seed=1, full_cov=True)
obs = synth.synth_obs(1, F, D, seed=7)
d_obs = ctx.to_device(obs)


def run(**env):
    for k in ("LNASR_NO_I8EMIT", "LNASR_I8_MIN_FRAMES", "LNASR_I8_RHO_MAX"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    _ffi.reload_switches()
    ctx._model_fp = None
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    rng_ = ctx.model_range()
    b = ctx.gmm_logprob(d_obs)
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(5):
        ctx.gmm_logprob(d_obs, b_out=b)
    ctx.device_sync()
    ctx.prof_enable(False)
    rep = {k: round(m / n, 4) for k, (n, m) in ctx.prof_report().items()}
    return b.to_host()[0], rng_, rep


b8, r8, t8 = run(LNASR_I8_MIN_FRAMES=0)
b64, r64, t64 = run(LNASR_NO_I8EMIT=1)
print("diag model?", ctx.model_is_diag(), "range", r8, "kappa", ctx.model_conditioning())
print("i8  kernels", t8)
print("f64 kernels", t64)
err = np.abs(b8 - b64)
print("max |b_i8 - b_f64| = %.3e  (rel to |b|: %.3e), |b| ~ %.1f" % (err.max(), (err / np.maximum(1, np.abs(b64))).max(), np.abs(b64).mean()))
# long-double direct form on a few frames
x = obs[0, :64].astype(np.longdouble)
mu, si, w = mdl["mu"].astype(np.longdouble), mdl["si"], mdl["w"].astype(np.longdouble)
ref = np.empty((64, N), dtype=np.longdouble)
for j in range(N):
    z = np.empty((64, M), dtype=np.longdouble)
    for m in range(M):
        P = np.linalg.inv(si[j, m]).astype(np.longdouble)
        sign, logdet = np.linalg.slogdet(si[j, m])
        dx = x - mu[j, m]
        z[:, m] = -0.5 * np.einsum("td,de,te->t", dx, P, dx) - 0.5 * (D * np.log(2 * np.longdouble(np.pi)) + np.longdouble(logdet)) + w[j, m]
    mx = z.max(1, keepdims=True)
    ref[:, j] = mx[:, 0] + np.log(np.exp(z - mx).sum(1))
for name, b in (("i8", b8), ("f64", b64)):
    print("%s against the direct form in long double (64 frames): %.3e" % (name, float(np.abs(b[:64].astype(np.longdouble) - ref).max())))
