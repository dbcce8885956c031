#!/usr/bin/env python3
"""What does the Viterbi kernel's time depend on?  (viterbi_reg1b_n128 went 1.26 -> 1.34 ms between rounds 3
and 5 while its code gained.)  One process, one box, the metric shard (N=128, M=16, D=39, 1024 x 1000), HIP events around every
launch (lnasr_prof_*), the cases interleaved three times:

  after_i8     decode(): slice_features + emit_diag_i8 (integer matrix cores) -> viterbi -> backtrace        [the bench step]
  after_f64    decode() under LNASR_NO_I8EMIT: emit_diag_mfma (fp64 matrix cores) -> viterbi -> backtrace    [round 4's step]
  alone        lnasr_viterbi on a resident b, launches back to back
  alone_idle   the same with the GPU left idle for 20 ms before every launch
  alone_cold_b the same after a 1 GB memset that evicts b from the L2 / MALL

    python tools/vit_after.py [reps]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ln-asr_amd"))
from lnasr import _ffi, synth  # noqa: E402

N, M, D, T, U = 128, 16, 39, 1000, 1024
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def prof(ctx, fn, reps):
    ctx.device_sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(reps):
        fn()
    ctx.device_sync()
    ctx.prof_enable(False)
    return {k: m / n for k, (n, m) in ctx.prof_report().items()}


def main():
    ctx = _ffi.Context(0)
    mdl = synth.synth_model(N, M, D, seed=1)
    d_obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
    d_path = ctx.empty((U, T), np.uint32)
    d_score = ctx.empty((U,), np.float64)
    scratch = ctx.empty((1 << 27,), np.float64)                       # 1 GB

    def upload(**env):
        for k in ("LNASR_NO_I8EMIT",):
            os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in env.items()})
        _ffi.reload_switches()
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)

    upload()
    d_b = ctx.gmm_logprob(d_obs)                                      # resident emissions for the stand-alone launches
    out = {}

    def add(name, res):
        for k, v in res.items():
            if k.startswith("viterbi_reg") or k.startswith("emit_diag") or k.startswith("backtrace"):
                out.setdefault(name, {}).setdefault(k, []).append(v)

    def vit_alone():
        ctx.viterbi(d_b, want_v=False, want_bt=False)

    def vit_idle():
        ctx.device_sync()
        time.sleep(0.02)
        ctx.viterbi(d_b, want_v=False, want_bt=False)

    def vit_cold():
        _ffi.check(_ffi.lib().lnasr_memset(ctx.handle, scratch.ptr, 0, scratch.nbytes))
        ctx.viterbi(d_b, want_v=False, want_bt=False)

    for rnd in range(3):
        upload()
        add("after_i8", prof(ctx, lambda: ctx.decode(d_obs, path_out=d_path, score_out=d_score), REPS))
        upload(LNASR_NO_I8EMIT=1)
        add("after_f64", prof(ctx, lambda: ctx.decode(d_obs, path_out=d_path, score_out=d_score), REPS))
        upload()
        add("alone", prof(ctx, vit_alone, REPS))
        add("alone_idle", prof(ctx, vit_idle, REPS))
        add("alone_cold_b", prof(ctx, vit_cold, max(3, REPS // 3)))
    for name, ks in out.items():
        print("%-13s %s" % (name, "  ".join("%s %s" % (k, "/".join("%.3f" % x for x in v)) for k, v in sorted(ks.items()))))
    ctx.close()


if __name__ == "__main__":
    main()
