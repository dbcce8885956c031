#!/usr/bin/env python3
"""Developer check: every dispatch tier that depends on the BATCH size gives the results of the small-batch tier
(which the oracle tests pin).  Sweeps (N, U) for the trellises and (shape, U) for decode / calc_prob / statistics.
    python tools/tier_sweep.py [quick]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
from lnasr import _ffi, synth

quick = len(sys.argv) > 1
ctx = _ffi.default_ctx()
bad = 0


def report(ok, what):
    global bad
    if not ok:
        bad += 1
    print("%s %s" % ("ok  " if ok else "FAIL", what), flush=True)


def close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    fin = np.isfinite(a) & np.isfinite(b)
    if not np.array_equal(np.isfinite(a), np.isfinite(b)) or not np.array_equal(a[~fin], b[~fin]):
        return False
    return bool(np.all(np.abs(a[fin] - b[fin]) <= tol * np.maximum(1.0, np.abs(b[fin]))))


def masked(x, lens):
    x = np.array(x, copy=True)
    for u, L in enumerate(lens):
        x[u, L:] = 0
    return x


Ns = (1, 3, 17, 64, 65, 128, 129, 200, 256, 257, 384, 385, 512, 769, 960, 961, 1024, 1100)
Us = (1, 7, 127, 256, 257, 511, 513, 1030) if not quick else (7, 513)
T = 4
for N in Ns:
    d = synth.synth_discrete(N, 4, 8, seed=N)
    A = d["A"].copy()
    if N > 4:
        A[3, :] = -np.inf
        A[3, 2] = 0.0
        A[:, 4] = -np.inf
    ctx.upload_hmm(A, d["pi"])
    rng = np.random.default_rng(N)
    Umax = max(Us)
    b_all = np.ascontiguousarray(-30.0 * rng.random((Umax, T, N)))
    lens_all = rng.integers(1, T + 1, Umax).astype(np.int32)
    lens_all[0] = T
    # reference: batches of 3 (the per-utterance tiers)
    ra = np.zeros_like(b_all); rb = np.zeros_like(b_all); rv = np.zeros_like(b_all)
    rp = np.zeros((Umax, T), np.uint32); rl = np.zeros(Umax)
    t0 = time.time()
    for s in range(0, Umax, 3):
        sl = slice(s, min(s + 3, Umax))
        a, l = ctx.forward(b_all[sl], lens=lens_all[sl])
        ra[sl], rl[sl] = a, l
        rb[sl] = ctx.backward(b_all[sl], lens=lens_all[sl])
        r = ctx.viterbi(b_all[sl], lens=lens_all[sl], want_bt=False)
        rv[sl], rp[sl] = r["v"], r["path"]
    for U in Us:
        b, lens = b_all[:U], lens_all[:U]
        try:
            a, l = ctx.forward(b, lens=lens)
            _, l2 = ctx.forward(b, lens=lens, want_alpha=False)
            be = ctx.backward(b, lens=lens)
            r = ctx.viterbi(b, lens=lens, want_bt=False)
            ok = (close(masked(a, lens), masked(ra[:U], lens), 1e-12) and close(l, rl[:U], 1e-12)
                  and close(l2, rl[:U], 1e-12)
                  and close(masked(be, lens), masked(rb[:U], lens), 1e-12)
                  and np.array_equal(masked(r["v"], lens), masked(rv[:U], lens))
                  and np.array_equal(masked(r["path"], lens), masked(rp[:U], lens)))
            report(ok, "trellis N=%d U=%d" % (N, U))
        except Exception as e:                                 # noqa: BLE001 -- a sweep reports and goes on
            report(False, "trellis N=%d U=%d: %s" % (N, U, e))

# fused entry points: decode / calc_prob / statistics against batches of 3
shapes = ((128, 16, 39, True), (5, 3, 13, True), (64, 1, 39, True), (256, 4, 20, True), (300, 2, 8, True),
          (6, 4, 13, False), (16, 16, 39, False))
for N, M, D, diag in shapes:
    mdl = synth.synth_model(N, M, D, seed=3, full_cov=not diag)
    cov = mdl["var"] if diag else mdl["si"]
    if diag:
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], cov, diag=True)
    else:
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], cov)
    T = 9
    Umax = 700 if not quick else 130
    obs_all = synth.synth_obs(Umax, T, D, seed=5)
    rng = np.random.default_rng(N + M)
    lens_all = rng.integers(1, T + 1, Umax).astype(np.int32)
    lens_all[0] = T
    rp = np.zeros((Umax, T), np.uint32); rl = np.zeros(Umax)
    for s in range(0, Umax, 3):
        sl = slice(s, min(s + 3, Umax))
        rp[sl] = np.asarray(ctx.decode(obs_all[sl], lens=lens_all[sl]))
        rl[sl] = np.asarray(ctx.calc_prob(obs_all[sl], lens=lens_all[sl]))
    for U in ((1, 63, 64, 65, 129, 511, 512, 700) if not quick else (65, 130)):
        obs, lens = obs_all[:U], lens_all[:U]
        try:
            p = np.asarray(ctx.decode(obs, lens=lens))
            l = np.asarray(ctx.calc_prob(obs, lens=lens))
            ok = np.array_equal(masked(p, lens), masked(rp[:U], lens)) and close(l, rl[:U], 1e-11)
            if N * N * U * T <= 3e8:
                st = np.asarray(ctx.bw_accumulate(obs, lens=lens))
                ref = None
                for s in range(0, U, 3):
                    sl = slice(s, min(s + 3, U))
                    part = np.asarray(ctx.bw_accumulate(obs[sl], lens=lens[sl]))
                    ref = part if ref is None else ref + part
                scale = np.maximum(1.0, np.abs(ref))
                err = np.abs(st - ref) / scale
                fin = np.isfinite(ref)
                ok = ok and np.array_equal(np.isfinite(st), fin) and float(err[fin].max()) <= 1e-9
            report(ok, "fused N=%d M=%d D=%d diag=%s U=%d" % (N, M, D, diag, U))
        except Exception as e:                                 # noqa: BLE001
            report(False, "fused N=%d M=%d D=%d diag=%s U=%d: %s" % (N, M, D, diag, U, e))
print("failures:", bad)
sys.exit(1 if bad else 0)
