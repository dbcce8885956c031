#!/usr/bin/env python3
"""Developer aid: where one Baum-Welch iteration of the reference's test/hmm-test.py model (n=4, m=5, T=190) spends its time --
each step of HMM.train timed alone through the mirror classes / the ctypes layer.  It showed that the forward and backward
KERNELS (0.38 ms each: two barriers per frame whatever N), not the transfers, made the iteration 1.6 ms, which led to
fwdbwd_tiny_kernel.      python tools/steptime.py"""
import sys, time, numpy as np
sys.path.insert(0, "ln-asr_amd")
from lnasr.hmm import HMM
from lnasr import _ffi
obs = np.array((0, 1, 1, 2, 0, 4, 2, 3, 4, 2, 0, 2, 3, 4, 4, 2, 1, 0, 1) * 10)
np.random.seed(0)
h = HMM(4, 5); h.reset("random")
ctx = h.ctx
def T(label, fn, n=200):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    print("%-34s %8.1f us" % (label, dt)); return r
T("_map_b", lambda: h._map_b(obs))
T("_upload (hmm)", lambda: h._upload())
b = h._b_dev()
al = T("ctx.forward dev->dev", lambda: ctx.forward(b, want_alpha=True, want_loglik=False)[0])
be = T("ctx.backward dev->dev", lambda: ctx.backward(b))
xg = T("ctx.baumwelch dev", lambda: ctx.baumwelch(b, al, be))
T("estimate_discrete", lambda: ctx.estimate_discrete(obs.astype(np.int32), 5, xg[0].view((190,4,4)), xg[1].view((190,4))))
T("rows_to_host", lambda: al.rows_to_host(189, 1, 4))
T("calc_prob", lambda: h.calc_prob(obs))
T("_forward host out", lambda: h._forward(obs))
T("DeviceArray alloc+free 3KB", lambda: ctx.empty((190, 4)))
T("ctx.sync", lambda: ctx.sync())
from lnasr.utils import lse
x = np.random.rand(4)
T("utils.lse(4)", lambda: lse(x))
T("discrete_map_b only", lambda: ctx.discrete_map_b(np.asarray(h.B, dtype=np.float64), obs.astype(np.int32), np.float64, device_out=True, b_out=b))
