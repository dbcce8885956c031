#!/usr/bin/env python3
"""Development aid: block lifetimes and in-kernel shader clock of emit_full_i8 (diagnostic build, -DLNASR_BLOCK_TIMES).
    python tools/full_i8_clock.py"""
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
os.environ["LNASR_I8_MIN_FRAMES"] = "0"
sys.path.insert(0, PKG)
import numpy as np  # noqa: E402
from lnasr import _ffi, synth  # noqa: E402

ctx = _ffi.default_ctx()
N, M, D, F = 128, 16, 39, 16000
mdl = synth.synth_model(N, M, D, seed=1, full_cov=True)
ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
d_obs = ctx.to_device(synth.synth_obs(1, F, D, seed=7))
b = ctx.gmm_logprob(d_obs)
for _ in range(3):
    ctx.gmm_logprob(d_obs, b_out=b)
ctx.sync()
raw = _ffi.lib()
nblocks = 16384                                        # (every slot; the blocks that ran have a later end than start)
tbuf = (C.c_ulonglong * (2 * 16384))()
cbuf = (C.c_ulonglong * 16384)()
assert raw.lnasr_dbg_block_times_full_i8(tbuf) == 0 and raw.lnasr_dbg_block_times_full_i8_cycles(cbuf) == 0
t = np.array(tbuf[:2 * nblocks], dtype=np.float64).reshape(nblocks, 2) / 100.0
cyc = np.array(cbuf[:nblocks], dtype=np.float64)
ok = t[:, 1] > t[:, 0]
life = (t[:, 1] - t[:, 0])[ok]
ghz = cyc[ok] / life / 1e3
ksteps = 250 * 32 * 13 / max(1, ok.sum())              # 8000 tiles of 13 k-steps, shared by the blocks that ran
print("blocks %d  span %.3f ms  life mean %.2f us (min %.2f max %.2f)  clock mean %.3f GHz (min %.3f max %.3f)  cycles per block %.0f = %.0f per k-step" % (
    ok.sum(), (t[ok, 1].max() - t[ok, 0].min()) / 1e3, life.mean(), life.min(), life.max(), ghz.mean(), ghz.min(), ghz.max(), cyc[ok].mean(), cyc[ok].mean() / ksteps))
