#!/usr/bin/env python3
"""Developer check: time per unit of WORK across model sizes, to catch sizes that fall off a cliff (round 5 found one: every N that
is not a tile width had its padding columns walk the exact log-domain fallback on every frame -- 10x, results right).
Prints ps per (frame x N^2) for forward / backward / Viterbi and ps per (frame x N x M x D) for the emission; a size whose figure
is several times its neighbours' is worth a look.      python tools/perf_scan.py [U] [T]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
from lnasr import _ffi, synth

U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = _ffi.default_ctx()
lib = _ffi.lib()


def ms(fn, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


print("U=%d T=%d   (ps per frame x N^2)" % (U, T))
print("%5s %10s %10s %10s %10s | %9s %9s %9s" % ("N", "fwd ms", "bwd ms", "vit ms", "dec ms", "fwd", "bwd", "vit"))
for N in (8, 16, 17, 24, 33, 64, 65, 100, 120, 128, 129, 160, 200, 255, 256, 257, 300, 384, 512):
    d = synth.synth_discrete(N, 4, 8, seed=N)
    ctx.upload_hmm(d["A"], d["pi"])
    b = ctx.to_device(np.ascontiguousarray(-30.0 * np.random.default_rng(N).random((U, T, N))))
    al = ctx.empty((U, T, N), np.float64)
    ll = ctx.empty((U,), np.float64)
    pth = ctx.empty((U, T), np.uint32)
    v = ctx.empty((U, T, N), np.float64)
    f = ms(lambda: _ffi.check(lib.lnasr_forward(ctx.handle, b.ptr, U, T, None, 0, al.ptr, ll.ptr)))
    bw = ms(lambda: _ffi.check(lib.lnasr_backward(ctx.handle, b.ptr, U, T, None, 0, al.ptr)))
    vt = ms(lambda: _ffi.check(lib.lnasr_viterbi(ctx.handle, b.ptr, U, T, None, 0, None, None, pth.ptr, None)))
    w = U * T * N * N
    print("%5d %10.3f %10.3f %10.3f %10s | %9.2f %9.2f %9.2f" % (N, f, bw, vt, "", f * 1e9 / w, bw * 1e9 / w, vt * 1e9 / w), flush=True)
    del b, al, v, pth

print("\nemission, U=%d T=%d   (ps per frame x N x M x D)" % (U, T))
for (N, M, D) in ((128, 16, 39), (120, 16, 39), (128, 12, 39), (128, 16, 13), (100, 5, 20), (64, 32, 39), (256, 32, 39), (33, 7, 40), (128, 16, 64), (8, 2, 13)):
    for full in (False, True):
        if full and N * M > 2048:
            continue
        mdl = synth.synth_model(N, M, D, seed=1, full_cov=full)
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
        Uf = U if not full else max(1, U // 16)
        obs = ctx.to_device(synth.synth_obs(Uf, T, D, seed=2))
        bb = ctx.empty((Uf, T, N), np.float64)
        e = ms(lambda: _ffi.check(lib.lnasr_gmm_logprob(ctx.handle, obs.ptr, Uf, T, None, 0, bb.ptr, None)))
        w = Uf * T * N * M * D * (D if full else 1)
        print("N=%4d M=%3d D=%3d %-5s %9.3f ms  %9.3f ps per unit%s" % (N, M, D, "full" if full else "diag", e, e * 1e9 / w,
              " x D" if full else ""), flush=True)
        del obs, bb
