#!/usr/bin/env python3
"""Developer micro-bench: Viterbi / forward beyond 256 states, frame-parallel against streaming kernels.
    python tools/bign.py [T]      (LNASR_FRAMEPAR_MAXU=4096 forces the frame-parallel kernels, LNASR_NO_FRAMEPAR=1 the streaming ones)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
from lnasr import _ffi, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = _ffi.default_ctx()
for N in (320, 512, 1024):
    for U in (64, 256, 512, 1024):
        mdl = synth.synth_model(N, 1, 2, seed=1)
        ctx.upload_hmm(mdl["A"], mdl["pi"])
        rng = np.random.default_rng(N + U)
        b = rng.normal(-10.0, 3.0, (U, T, N))
        d_b = ctx.to_device(b)
        res = []
        for name, fn in (("viterbi", lambda: ctx.viterbi(d_b, want_bt=False)),
                         ("forward", lambda: ctx.forward(d_b))):
            try:
                fn()
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(3):
                    fn()
                ctx.sync()
                res.append("%s %8.3f ms" % (name, (time.perf_counter() - t0) / 3 * 1e3))
            except TypeError as e:
                res.append("%s n/a (%s)" % (name, e))
        print("N=%4d U=%4d T=%d  %s" % (N, U, T, "  ".join(res)), flush=True)
        del d_b
