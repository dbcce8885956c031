#!/usr/bin/env python3
"""Latency of the literal drop-in surface: the reference's own method names on ONE utterance
(hmm.py:276-283 `decode`, :268-274 `calc_prob`, :63-100 `_forward`, :133-172 `_viterbi`, :285-313 `train`) at the
metric shape N=128, M=16, D=39, T=1000 with `si` stored as [N,M,D,D] exactly as the reference stores it.

    python tools/refapi.py            # prints the table; bench.py embeds measure() as its "reference_api" block

Per method: wall ms per call (median), the part of it the device spends in kernels (HIP events on the context's stream,
collected in a second pass so that the events do not sit inside the wall figure) and the remainder (Python, ctypes,
parameter fingerprint, staging, PCIe copies, launch latency) -- beside the reference's 1.90 s per decode() measured
in the survey container."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))

N, M, D, T = 128, 16, 39, 1000
REFERENCE_DECODE_MS = 1900.0            # BASELINE.md section 2 (8-vCPU Xeon 2.1 GHz, numpy 2.2.6)


def _median_ms(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, float(min(ts)) * 1e3


def _device_ms(ctx, fn, reps):
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(reps):
        fn()
    ctx.sync()
    ctx.prof_enable(False)
    rep = ctx.prof_report()
    kern = {k: {"launches_per_call": n / reps, "ms_per_call": ms / reps} for k, (n, ms) in
            sorted(rep.items(), key=lambda kv: -kv[1][1])}
    return sum(v["ms_per_call"] for v in kern.values()), kern


def measure(ctx, reps=20, obs=None, check_path=None):
    """-> dict for the bench line.  obs [T,D] (default: the synthetic utterance of seed 1000); check_path: the path
    decode() must return for it (the checker's answer) -- a mismatch raises and nothing is reported."""
    from lnasr import synth
    from lnasr.gmmhmm import GMMHMM
    mdl = synth.synth_model(N, M, D, seed=1)
    if obs is None:
        obs = synth.synth_obs(1, T, D, seed=1000)[0]
    gh = GMMHMM(N, M, D, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"], ctx=ctx)       # si [N,M,D,D]
    path = gh.decode(obs)
    if check_path is not None and not np.array_equal(path, check_path):
        raise RuntimeError("GMMHMM.decode() path differs from the checker's")
    rows = {}
    calls = {"decode": lambda: gh.decode(obs), "calc_prob": lambda: gh.calc_prob(obs),
             "_viterbi(remap=True)": lambda: gh._viterbi(obs, True), "_forward(remap=True)": lambda: gh._forward(obs, True),
             "_map_b": lambda: gh._map_b(obs)}
    for name, fn in calls.items():
        fn(), fn()
        wall, best = _median_ms(fn, reps)
        dev, kern = _device_ms(ctx, fn, max(3, reps // 4))
        rows[name] = {"wall_ms": wall, "best_ms": best, "device_ms": dev, "host_ms": max(0.0, wall - dev),
                      "kernels": {k: round(v["ms_per_call"], 4) for k, v in list(kern.items())[:5]}}
    # the parameter check every call pays (25 MB of si and friends)
    fp_ms, _ = _median_ms(lambda: ctx._fingerprint("gmm", tuple(np.asarray(x) for x in (gh.A, gh.pi, gh.w, gh.mu, gh.si))), reps)
    # reading the state back afterwards (what decode() no longer does per call)
    gh.decode(obs)
    t0 = time.perf_counter()
    b, bm = gh.b, gh.bm
    read_ms = (time.perf_counter() - t0) * 1e3
    assert b.shape == (N, T) and bm.shape == (N, M, T)
    # one Baum-Welch iteration (train(iters=1)): E-step, M-step with the full [N,M,D,D] Sigma the reference's M-step
    # returns, upload of the new model, likelihood under it
    gt = GMMHMM(N, M, D, mdl["A"].copy(), mdl["pi"].copy(), mdl["w"].copy(), mdl["mu"].copy(), mdl["si"].copy(), ctx=ctx)
    gt.train(obs, iters=1, verbose=False)
    snap = [x.copy() for x in (gt.A, gt.pi, gt.w, gt.mu, gt.si)]

    def one_iter():
        gt.A, gt.pi, gt.w, gt.mu, gt.si = (x.copy() for x in snap)
        gt.train(obs, iters=1, verbose=False)
    wall, best = _median_ms(one_iter, max(3, reps // 4))
    dev, kern = _device_ms(ctx, one_iter, 2)
    rows["train(iters=1)"] = {"wall_ms": wall, "best_ms": best, "device_ms": dev, "host_ms": max(0.0, wall - dev),
                              "kernels": {k: round(v["ms_per_call"], 4) for k, v in list(kern.items())[:5]},
                              "all_kernels": {k: round(v["ms_per_call"], 4) for k, v in kern.items()}}
    # the steady state of a longer run on a problem the data can determine (500 frames per 13-dimensional component; at the
    # metric shape 1000 frames leave half a frame per full 39 x 39 covariance and the second iteration scores a degenerate
    # model, which the forward kernel handles in its exact log-domain fallback at 13 us a frame): parameters stay in HBM between iterations, L(lambda_new | O) comes from the next E-step
    Nw, Mw, Dw, Tw, K = 8, 2, 13, 8000, 6
    mw = synth.synth_model(Nw, Mw, Dw, seed=3)
    ow = synth.synth_obs(1, Tw, Dw, seed=3000)[0]
    gw = GMMHMM(Nw, Mw, Dw, ctx=ctx)

    def k_iters(k):
        gw.A, gw.pi, gw.w, gw.mu, gw.si = (mw[x].copy() for x in ("A", "pi", "w", "mu", "si"))
        gw.train(ow, iters=k, eps=0.0, verbose=False)
    k_iters(K)
    w1, _ = _median_ms(lambda: k_iters(1), max(3, reps // 4))
    wK, _ = _median_ms(lambda: k_iters(K), max(3, reps // 4))
    dev, kern = _device_ms(ctx, lambda: k_iters(K), 2)
    rows["train(iters=%d), N=%d M=%d D=%d T=%d" % (K, Nw, Mw, Dw, Tw)] = {
        "wall_ms": wK, "best_ms": wK, "device_ms": dev, "host_ms": max(0.0, wK - dev), "first_iteration_ms": w1,
        "ms_per_further_iteration": (wK - w1) / (K - 1),
        "kernels": {k: round(v["ms_per_call"], 4) for k, v in list(kern.items())[:6]}}
    # the front end through the reference's method names: MFCC.calc_mfcc on 10 s of audio (160,240 samples -> 1000 frames;
    # the reference: 0.207 s) and the whole acoustic path PCM -> MFCC -> decode (Recognizer.recognize;
    # upstream's is `pass`), model file loaded the reference's way
    import tempfile
    from lnasr.mfcc import MFCC
    from lnasr.recognizer import Recognizer
    sig = synth.synth_signal(160240, seed=7)
    pcm = np.clip(sig, -32768, 32767).astype(np.int16)
    mf = MFCC(ctx=ctx)
    for name, fn in (("MFCC.calc_mfcc(10 s, float64)", lambda: mf.calc_mfcc(sig)),
                     ("MFCC.calc_mfcc(10 s, int16)", lambda: mf.calc_mfcc(pcm))):
        fn(), fn()
        wall, best = _median_ms(fn, reps)
        dev, kern = _device_ms(ctx, fn, max(3, reps // 4))
        rows[name] = {"wall_ms": wall, "best_ms": best, "device_ms": dev, "host_ms": max(0.0, wall - dev),
                      "reference_ms": 207.0, "kernels": {k: round(v["ms_per_call"], 4) for k, v in list(kern.items())[:5]}}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "am.npz")
        gh.save(path)
        rec = Recognizer({"am": path, "lm": None})
    rec.am.gmmhmm._ctx = ctx
    rec.am.mfcc._ctx = ctx
    fn = lambda: rec.recognize(pcm)
    fn(), fn()
    wall, best = _median_ms(fn, reps)
    dev, kern = _device_ms(ctx, fn, max(3, reps // 4))
    rows["Recognizer.recognize(10 s int16 PCM)"] = {
        "wall_ms": wall, "best_ms": best, "device_ms": dev, "host_ms": max(0.0, wall - dev),
        "kernels": {k: round(v["ms_per_call"], 4) for k, v in list(kern.items())[:6]}}
    toy = toy_scripts(ctx)
    return {"toy_scripts": toy,
            "what": "the reference's own method names on one utterance through the mirror classes (lnasr.gmmhmm.GMMHMM): "
                    "N=%d M=%d D=%d T=%d, si as [N,M,D,D], numpy arrays in and out; wall = median of %d calls, "
                    "device = kernel time of the same call (HIP events, separate pass), host = the rest" % (N, M, D, T, reps),
            "methods": rows, "parameter_fingerprint_ms": fp_ms, "read_b_and_bm_afterwards_ms": read_ms,
            "reference_decode_ms": REFERENCE_DECODE_MS,
            "reference_decode_source": "BASELINE.md section 2: the unmodified reference, survey container",
            "decode_frames_per_s": T / (rows["decode"]["wall_ms"] * 1e-3)}


def toy_scripts(ctx):
    """The unsupervised-training halves of the reference's own test scripts, at their own (toy) sizes -- where launch and
    transfer latency, not arithmetic, is what a GPU path could lose on: test/hmm-test.py:41-51 (discrete HMM n=4, m=5, T=190,
    reset('random'), train(obs, 100): 48 Baum-Welch iterations until eps in the survey's run, 1.26 s for the whole script)
    and test/gmmhmm-test.py:10-20 (GMMHMM n=5, m=3, d=2, T=40, train(obs, 10): 0.09 s for the whole script)."""
    import contextlib
    import io
    from lnasr.gmmhmm import GMMHMM
    from lnasr.hmm import HMM
    out = {}
    obs = np.array((0, 1, 1, 2, 0, 4, 2, 3, 4, 2, 0, 2, 3, 4, 4, 2, 1, 0, 1) * 10)
    for rep in range(2):                                   # second run: library warm
        np.random.seed(0)
        h = HMM(4, 5, ctx=ctx)
        h.reset("random")
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            h.train(obs, 100)
        dt = time.perf_counter() - t0
    iters = buf.getvalue().count("Iter:")
    out["hmm-test.py train(obs, 100)"] = {"wall_ms": dt * 1e3, "iterations": iters, "ms_per_iteration": dt * 1e3 / max(1, iters),
                                         "reference_script_ms": 1260.0}
    for rep in range(2):
        np.random.seed(0)
        x = np.array(0.6 * np.random.random_sample((40, 2)) - 0.3, dtype=np.double)
        gh = GMMHMM(5, 3, 2, ctx=ctx)
        gh.reset("random")
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            gh.train(x, 10)
        dt = time.perf_counter() - t0
    iters = buf.getvalue().count("Iter:")
    out["gmmhmm-test.py train(obs, 10)"] = {"wall_ms": dt * 1e3, "iterations": iters, "ms_per_iteration": dt * 1e3 / max(1, iters),
                                           "reference_script_ms": 90.0}
    return out


def main():
    from lnasr import _ffi
    ctx = _ffi.default_ctx()
    r = measure(ctx)
    print(r["what"])
    print("%-38s %9s %9s %9s %9s" % ("method", "wall ms", "best ms", "device ms", "host ms"))
    for k, v in r["methods"].items():
        print("%-38s %9.3f %9.3f %9.3f %9.3f   %s" % (k, v["wall_ms"], v["best_ms"], v["device_ms"], v["host_ms"], v["kernels"]))
    print("train(iters=1) kernels:", r["methods"]["train(iters=1)"]["all_kernels"])
    for k, v in r["methods"].items():
        if "ms_per_further_iteration" in v:
            print("%s: first iteration %.3f ms, %.3f ms per further iteration" % (k, v["first_iteration_ms"], v["ms_per_further_iteration"]))
    print("parameter fingerprint (per call): %.3f ms; reading b + bm back afterwards: %.3f ms; reference decode(): %.0f ms"
          % (r["parameter_fingerprint_ms"], r["read_b_and_bm_afterwards_ms"], r["reference_decode_ms"]))
    for k, v in r["toy_scripts"].items():
        print("%-32s %8.2f ms for %d iterations (%.3f ms each); the reference's whole script: %.0f ms"
              % (k, v["wall_ms"], v["iterations"], v["ms_per_iteration"], v["reference_script_ms"]))


if __name__ == "__main__":
    main()
