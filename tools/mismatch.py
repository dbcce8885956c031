#!/usr/bin/env python3
"""Developer check: the scaled linear-domain recursions on data the model does NOT fit -- MFCCs of a test signal against a
random unit-scale model: row maxima of b around -2000 nats that jump by hundreds from frame to frame.  Until round 5 every
forward / backward kernel took its exact log-domain fallback on every frame of such data (35 us a frame instead of 0.5: the
shift of a row lagged a frame and assumed emissions of a steady size); they now add the emission row's level to the shift.
    python tools/mismatch.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
from lnasr import _ffi, synth
from lnasr.mfcc import MFCC
from lnasr.gmmhmm import GMMHMM
ctx = _ffi.default_ctx()
sig = synth.synth_signal(160240, seed=7)
feat = MFCC(ctx=ctx).calc_mfcc(sig)[2]
print("features: mean |x| %.1f, max |x| %.1f" % (np.abs(feat).mean(), np.abs(feat).max()))


def timed(name, fn, reps=5):
    fn(); fn()
    ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    ctx.prof_enable(False)
    rep = ctx.prof_report()
    print("  %-34s %8.3f ms  %s" % (name, dt * 1e3, {k: round(ms / n, 3) for k, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:2]}))


for (N, M) in ((128, 16), (256, 8), (32, 4), (300, 4)):
    D = 39
    mdl = synth.synth_model(N, M, D, seed=1)
    gh = GMMHMM(N, M, D, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"], ctx=ctx)
    gh._map_b(feat)
    b = gh.b
    mx = b.max(axis=0)
    print("N=%d M=%d: per-frame max of b: mean %.0f, std of its frame-to-frame change %.0f" % (N, M, mx.mean(), np.diff(mx).std()))
    timed("calc_prob, one utterance", lambda: gh.calc_prob(feat))
    timed("_backward, one utterance", lambda: gh._backward(feat, True))
    U = 600
    bb = ctx.to_device(np.ascontiguousarray(np.broadcast_to(b.T[None], (U,) + b.T.shape)))
    al = ctx.empty(bb.shape, np.float64)
    ll = ctx.empty((U,), np.float64)
    lib = _ffi.lib()
    T = b.shape[1]
    timed("forward, %d utterances" % U, lambda: _ffi.check(lib.lnasr_forward(ctx.handle, bb.ptr, U, T, None, 0, al.ptr, ll.ptr)), reps=2)
    timed("backward, %d utterances" % U, lambda: _ffi.check(lib.lnasr_backward(ctx.handle, bb.ptr, U, T, None, 0, al.ptr)), reps=2)
    del bb, al, ll
