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
groups = float(buf[15])                  # groups the stamped wave walked (waves draw chunks from a ticket counter)
names = ["loop top -> stamp", "pre-emphasis (waits for the prefetched samples) + next group's whereabouts",
         "window + fft16 #1 + twiddle", "exchange through the LDS (4 phases)", "fft16 #2",
         "mirror shuffles + real-input split + power stores + energy", "request next samples",
         "mel/log/DCT tail (whole)", "  mel taps", "  neighbour sums + log + store", "  DCT products", "  DCT reduce + store"]
tot = sum(buf[i] for i in range(8))
print("cycles per group of four frames (lane 0 of one wave, %s samples): total %.0f" % ("int16" if "i16" in sys.argv else "f64",
                                                                                         tot / groups))
for i, nm in enumerate(names):
    print("  %8.0f  %s" % (buf[i] / groups, nm))
life, real = buf[13] / reps, buf[14] / reps
print("  wave lifetime %.0f shader cycles = %.1f us of the 100 MHz counter -> in-kernel clock %.2f GHz" % (life, real / 100.0, life / (real * 10.0)))
print("  %8.0f  kernel start -> first group (tables, lane plan, first loads), cycles per LAUNCH" % (buf[12] / reps))

# when each wave of the LAST launch lived (100 MHz ticks): how much of the kernel's duration is start-up skew and tail
wt = (C.c_ulonglong * 8192)()
raw.lnasr_dbg_mfcc_wave_times.argtypes = [C.c_void_p, C.c_void_p]
raw.lnasr_dbg_mfcc_wave_times(ctx.handle, wt)
t = np.array(wt[:], dtype=np.float64).reshape(4096, 2)[:2048] / 100.0          # us
t0 = t[:, 0].min()
start, end = t[:, 0] - t0, t[:, 1] - t0
dur = end - start
print("waves: first start 0, last start %.1f us; first end %.1f, last end %.1f us; lifetime min / median / max %.1f / %.1f / %.1f us"
      % (start.max(), end.min(), end.max(), dur.min(), np.median(dur), dur.max()))
blk = dur.reshape(256, 8)
print("per block (8 waves): median of block maxima %.1f us; slowest blocks %s" % (np.median(blk.max(axis=1)), np.round(np.sort(blk.max(axis=1))[-5:], 1)))
print("starts by block: min %.1f median %.1f max %.1f us" % (start.reshape(256, 8).min(axis=1).min(), np.median(start.reshape(256, 8).min(axis=1)), start.reshape(256, 8).min(axis=1).max()))
for bi in (0, 1, 7, 100, 255):
    print("block %3d wave lifetimes (us): %s" % (bi, np.round(blk[bi], 1)))
order = np.argsort(dur)
print("fastest waves:", order[:8], "slowest waves:", order[-8:])
print("mean lifetime by wave slot (w = 0..7):", np.round(blk.mean(axis=0), 1))
