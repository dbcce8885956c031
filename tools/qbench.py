#!/usr/bin/env python3
"""Developer micro-bench: per-kernel HIP-event times of the decode path on one GPU.
    python tools/qbench.py [U] [T] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
from lnasr import _ffi, synth

U = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, M, D = 128, 16, 39
ctx = _ffi.default_ctx()
print(ctx.device_info())
mdl = synth.synth_model(N, M, D, seed=1)
ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
obs = synth.synth_obs(U, T, D, seed=1000)
d_obs = ctx.to_device(obs)
d_path = ctx.empty((U, T), np.uint32)
d_score = ctx.empty((U,), np.float64)
for _ in range(2):
    ctx.decode(d_obs, path_out=d_path, score_out=d_score)
ctx.sync()
ctx.prof_enable(True)
ctx.prof_reset()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.decode(d_obs, path_out=d_path, score_out=d_score)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
ctx.prof_enable(False)
rep = ctx.prof_report()
frames = U * T
print("decode (profiled): %.3f ms/step  %.3e frames/s" % (dt * 1e3, frames / dt))
for k, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s %4d launches  %9.3f ms avg  (%.3e frames/s)" % (k, n, ms / n, frames / (ms / n * 1e-3)))
t0 = time.perf_counter()
for _ in range(reps):
    ctx.decode(d_obs, path_out=d_path, score_out=d_score)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print("decode (unprofiled): %.3f ms/step  %.3e frames/s  (%.1f%% of 2.886e8 fp64 roof)" % (dt * 1e3, frames / dt, 100 * frames / dt / 2.886e8))
p = d_path.to_host()
print("path checksum", int(p.astype(np.int64).sum()), "score[0]", d_score.to_host()[0])
