#!/usr/bin/env python3
"""Developer diagnostic (GPU box): when every block of the two headline kernels lived.  Builds a SEPARATE library with
-DLNASR_BLOCK_TIMES (thread 0 of every block records its first and last 100 MHz tick), runs one decode step of the
bench's shape and prints start-up skew, the distribution of block lifetimes and the tail.

    python tools/block_times.py
"""
import ctypes as C
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "..", "ln-asr_amd")
OUT = "/tmp/lnasr_btimes"
os.makedirs(OUT, exist_ok=True)
lib = os.path.join(OUT, "liblnasr_hip_btimes.so")
objs = []
for src in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))):
    o = os.path.join(OUT, os.path.basename(src)[:-4] + ".o")
    objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
                           "-I/opt/rocm/include", "-Wno-unused-variable", "-Wno-unused-function", "-DLNASR_BLOCK_TIMES", "-c", src, "-o", o])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs +
                      ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
os.environ["LNASR_LIB"] = lib
sys.path.insert(0, PKG)
import numpy as np  # noqa: E402
from lnasr import _ffi, synth  # noqa: E402

ctx = _ffi.default_ctx()
N, M, D, T, U = 128, 16, 39, 1000, 1024
mdl = synth.synth_model(N, M, D, seed=1)
ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
d_obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
path = ctx.empty((U, T), np.uint32)
for _ in range(3):
    ctx.decode(d_obs, path_out=path)
ctx.sync()
raw = _ffi.lib()
buf = (C.c_ulonglong * (2 * 16384))()


def report(name, fn, nblocks, per_cu):
    assert fn(buf) == 0
    t = np.array(buf[:2 * nblocks], dtype=np.float64).reshape(nblocks, 2) / 100.0        # us
    t0 = t[:, 0].min()
    start, end = t[:, 0] - t0, t[:, 1] - t0
    dur = end - start
    print("%s: %d blocks (%d per CU)" % (name, nblocks, per_cu))
    print("   starts: median %.1f us, last %.1f us;  ends: first %.1f, median %.1f, last %.1f us" %
          (np.median(start), start.max(), end.min(), np.median(end), end.max()))
    print("   lifetimes: min %.1f  median %.1f  max %.1f us;  sum of lifetimes / (slots x span) = %.3f" %
          (dur.min(), np.median(dur), dur.max(), dur.sum() / (256 * per_cu * end.max())))
    hist, edges = np.histogram(end, bins=10)
    print("   blocks ending per tenth of the span:", hist.tolist())


report("emit_diag_mfma", raw.lnasr_dbg_block_times_gmm, 256, 1)
report("viterbi_reg1b", raw.lnasr_dbg_block_times_trellis, 512, 2)

# config 5: the statistics pass (N=256, M=32, T=2000, 64 utterances): the moments kernels
N, M, T, U = 256, 32, 2000, 64
mdl = synth.synth_model(N, M, D, seed=1)
ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
d_obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
stats = ctx.empty((ctx.bw_stats_len(),), np.float64)
for prec, name, fn in ((np.float64, "mix_mfma", raw.lnasr_dbg_block_times_bwstats), (np.float32, "mix_mfma_f32", raw.lnasr_dbg_block_times_f32stats)):
    for _ in range(2):
        ctx.bw_accumulate(d_obs, stats_out=stats, precision=prec)
    ctx.sync()
    report(name, fn, 512, 2)
report("emit_diag_mfma(_bm) at config 5", raw.lnasr_dbg_block_times_gmm, 256, 1)
