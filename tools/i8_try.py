#!/usr/bin/env python3
"""Development aid: the integer matrix-core bank (gmm_i8.hip) against the fp64 one -- values and time.

    python tools/i8_try.py            a few shapes
    python tools/i8_try.py big        the metric shard (1024 x 1000 frames, N=128, M=16, D=39)
    python tools/i8_try.py sweep      batch sizes 1 .. 256 utterances and other shapes (D = 14 .. 39, M = 8 / 16 / 32)
    python tools/i8_try.py small      banks of 32 .. 96 states (the dispatch threshold)
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
import numpy as np
from lnasr import _ffi, synth

def run(N, M, D, U, T, reps=5):
    ctx = _ffi.default_ctx()
    mdl = synth.synth_model(N, M, D, seed=5)
    obs = synth.synth_obs(U, T, D, seed=6)
    d_obs = ctx.to_device(obs)
    out = {}
    for name, env in (("i8", {"LNASR_I8_MIN_FRAMES": "0"}), ("f64", {"LNASR_NO_I8EMIT": "1"})):
        for k in ("LNASR_I8_MIN_FRAMES", "LNASR_NO_I8EMIT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        _ffi.lib().lnasr_reload_switches()
        ctx._model_fp = None
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
        b = ctx.gmm_logprob(d_obs)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            b = ctx.gmm_logprob(d_obs, b_out=b)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        out[name] = (b.to_host(), dt)
    bi, bf = out["i8"][0], out["f64"][0]
    err = np.abs(bi - bf)
    print("N=%d M=%d D=%d U=%d T=%d: i8 %.3f ms  f64 %.3f ms   max |i8 - f64| = %.3e (rel %.2e)  nan %d/%d"
          % (N, M, D, U, T, out["i8"][1] * 1e3, out["f64"][1] * 1e3, np.nanmax(err), np.nanmax(err / np.abs(bf)),
             np.isnan(bi).sum(), np.isnan(bf).sum()))
    return bi, bf

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        run(128, 16, 39, 1024, 1000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        for U in (1, 2, 4, 8, 16, 32, 64, 128, 256):
            run(128, 16, 39, U, 1000, reps=20)
        run(256, 32, 39, 64, 2000)
        run(32, 16, 39, 512, 1000)
        run(128, 16, 26, 1024, 1000)
        run(128, 16, 32, 1024, 1000)
        run(128, 16, 14, 1024, 1000)
        run(128, 8, 39, 1024, 1000)
        run(256, 8, 26, 512, 1000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        for N, M in ((64, 16), (64, 8), (96, 16), (32, 32), (48, 16), (64, 32)):
            run(N, M, 39, 512, 1000, reps=10)
        sys.exit(0)
    run(16, 16, 39, 4, 100)
    run(16, 16, 39, 3, 77)
    run(128, 16, 39, 64, 1000)
    run(128, 16, 39, 1024, 1000)
