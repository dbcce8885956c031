#!/usr/bin/env python3
"""Shader clock the Viterbi kernel actually runs at, by what ran before it.  Builds the diagnostic library
(-DLNASR_BLOCK_TIMES: thread 0 of every block records the 100 MHz real-time counter and the shader-clock cycle counter at its
first and last instruction), then per case prints the kernel's span, the blocks' mean lifetime and mean clock:

  after_i8    decode(): emit_diag_i8 (integer matrix cores, power-limited) -> viterbi
  after_f64   decode() under LNASR_NO_I8EMIT: emit_diag_mfma -> viterbi
  alone       lnasr_viterbi on resident emissions, launches back to back (the GPU idles between them)

    python tools/vit_clock.py
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
d_obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
path = ctx.empty((U, T), np.uint32)
raw = _ffi.lib()
tbuf = (C.c_ulonglong * (2 * 16384))()
cbuf = (C.c_ulonglong * 16384)()


def upload(**env):
    os.environ.pop("LNASR_NO_I8EMIT", None)
    os.environ.update({k: str(v) for k, v in env.items()})
    _ffi.reload_switches()
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)


def report(name, nblocks=512):
    assert raw.lnasr_dbg_block_times_trellis(tbuf) == 0 and raw.lnasr_dbg_block_times_trellis_cycles(cbuf) == 0
    t = np.array(tbuf[:2 * nblocks], dtype=np.float64).reshape(nblocks, 2) / 100.0        # us
    cyc = np.array(cbuf[:nblocks], dtype=np.float64)
    life = t[:, 1] - t[:, 0]
    ghz = cyc / life / 1e3
    print("%-10s span %.3f ms  block life mean %.3f ms  clock mean %.3f GHz (min %.3f max %.3f)  cycles per block %.3e" % (
        name, (t[:, 1].max() - t[:, 0].min()) / 1e3, life.mean() / 1e3, ghz.mean(), ghz.min(), ghz.max(), cyc.mean()))


upload()
d_b = ctx.gmm_logprob(d_obs)
for rnd in range(3):
    upload()
    for _ in range(4):
        ctx.decode(d_obs, path_out=path)
    ctx.sync()
    report("after_i8")
    upload(LNASR_NO_I8EMIT=1)
    for _ in range(4):
        ctx.decode(d_obs, path_out=path)
    ctx.sync()
    report("after_f64")
    upload()
    for _ in range(4):
        ctx.viterbi(d_b, want_v=False, want_bt=False)
    ctx.sync()
    report("alone")
