import os, sys, numpy as np
sys.path.insert(0, "/root/repo/ln-asr_amd")
from lnasr import _ffi, synth
from lnasr.mfcc import MFCC
ctx = _ffi.default_ctx()
U, n = 256, 160240
sig = synth.synth_signal(n, seed=7)
batch = np.tile(sig, U)
offs = (np.arange(U + 1) * n).astype(np.int64)
m = MFCC(); cfg, win, bank = m._cfg()
d_sig = ctx.to_device(batch)
for skip in ("0", "1", "2", "4", "8", "16", "3", "15", "31"):
    os.environ["LNASR_MFCC_SKIP"] = skip
    fn = lambda: ctx.mfcc(d_sig, offs, cfg, win, bank, False, False)
    fn(); ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(3): fn()
    ctx.sync(); ctx.prof_enable(False)
    rep = ctx.prof_report()
    print("skip=%s" % skip, {k: round(ms / c, 3) for k, (c, ms) in rep.items() if 'frame' in k})
