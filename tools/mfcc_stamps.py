#!/usr/bin/env python3
"""Developer diagnostic (GPU box): where the cycles of one group of four frames go in the fft-512 MFCC frame kernel.
Builds a SEPARATE library with -DLNASR_MFCC_STAMPS (lane 0 of wave 0 of block 0 accumulates s_memtime deltas per
phase), runs the config-2 batch and prints cycles per group and phase.

    python tools/mfcc_stamps.py [i16]
"""
import ctypes as C
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "..", "ln-asr_amd")
OUT = "/tmp/lnasr_stamps"
os.makedirs(OUT, exist_ok=True)
lib = os.path.join(OUT, "liblnasr_hip_stamps.so")
objs = []
for src in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))):
    o = os.path.join(OUT, os.path.basename(src)[:-4] + ".o")
    objs.append(o)
    flags = ["-DLNASR_MFCC_STAMPS"] if src.endswith("mfcc.hip") else []
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
                           "-I/opt/rocm/include", "-Wno-unused-variable", "-c", src, "-o", o] + flags)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs +
                      ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
os.environ["LNASR_LIB"] = lib
sys.path.insert(0, PKG)
import numpy as np  # noqa: E402
from lnasr import _ffi, synth  # noqa: E402
from lnasr.mfcc import MFCC  # noqa: E402

ctx = _ffi.default_ctx()
U, n = 256, 160240
batch = np.tile(synth.synth_signal(n, seed=7), U)
if "i16" in sys.argv:
    batch = np.clip(batch, -32768, 32767).astype(np.int16)
offs = (np.arange(U + 1) * n).astype(np.int64)
cfg, win, bank = MFCC()._cfg()
d_sig = ctx.to_device(batch)
fo = ctx.empty((U * 1000, 39), np.float64)
raw = _ffi.lib()
raw.lnasr_dbg_mfcc_stamps.argtypes = [C.c_void_p, C.c_void_p]
buf = (C.c_ulonglong * 16)()
ctx.mfcc(d_sig, offs, cfg, win, bank, False, False, fo)
raw.lnasr_dbg_mfcc_stamps(ctx.handle, buf)
reps = 5
for _ in range(reps):
    ctx.mfcc(d_sig, offs, cfg, win, bank, False, False, fo)
raw.lnasr_dbg_mfcc_stamps(ctx.handle, buf)
groups = reps * 32.0                     # groups walked by one wave per launch: 64,000 groups / 2048 waves
names = ["loop top -> stamp", "pre-emphasis (waits for the prefetched samples) + next group's whereabouts",
         "window + fft16 #1 + twiddle", "exchange through the LDS (4 phases)", "fft16 #2",
         "mirror shuffles + real-input split + power stores + energy", "request next samples",
         "mel/log/DCT tail (whole)", "  mel taps", "  neighbour sums + log + store", "  DCT products", "  DCT reduce + store"]
tot = sum(buf[i] for i in range(8))
print("cycles per group of four frames (lane 0 of one wave, %s samples): total %.0f" % ("int16" if "i16" in sys.argv else "f64",
                                                                                         tot / groups))
for i, nm in enumerate(names):
    print("  %8.0f  %s" % (buf[i] / groups, nm))
print("  %8.0f  kernel start -> first group (tables, lane plan, first loads), cycles per LAUNCH" % (buf[12] / reps))
