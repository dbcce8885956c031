#!/usr/bin/env python3
"""Per-row measurements of the other BASELINE.json configs on one GPU (development / DESIGN.md numbers;
the driver-facing benchmark is bench.py).  Inputs resident in HBM, HIP-event kernel times.

    python tools/bench_rows.py [mfcc] [e2e] [seg] [gmm] [fwd] [fwdsweep] [btsweep] [bw] [refapi] [nanfix] [all]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ln-asr_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lnasr import _ffi, synth  # noqa: E402
from lnasr.mfcc import MFCC  # noqa: E402

what = set(sys.argv[1:]) or {"all"}
ctx = _ffi.default_ctx()
print(ctx.device_info())


def timed(label, fn, units, unit_name, reps=5, algo_bytes=None, algo_flop=None):
    fn()
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    ctx.prof_enable(False)
    rep = ctx.prof_report()
    extra = ""
    if algo_bytes:
        extra += "  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (algo_bytes / dt / 1e9, algo_bytes / dt / 8e12 * 100)
    if algo_flop:
        extra += "  %.2f TFLOP/s algorithmic (%.1f%% of 78.6)" % (algo_flop / dt / 1e12, algo_flop / dt / 78.6e12 * 100)
    print("%-34s %9.3f ms  %.3e %s/s%s" % (label, dt * 1e3, units / dt, unit_name, extra))
    for k, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:6]:
        print("      %-28s %5.1f launches/step %9.3f ms avg" % (k, n / reps, ms / n))


if what & {"mfcc", "all"}:
    # config 2: 16 kHz, 25/10 ms, 40 mel, 13 cepstra -> D=39; n = 160,240 samples -> 1000 frames; U utterances
    U, n = 256, 160240
    sig = synth.synth_signal(n, seed=7)
    batch = np.tile(sig, U)
    offs = (np.arange(U + 1) * n).astype(np.int64)
    m = MFCC()
    cfg, win, bank = m._cfg()
    d_sig = ctx.to_device(batch)
    frames = U * 1000
    fo = ctx.empty((frames, 39), np.float64)
    timed("mfcc f64 samples (U=%d)" % U, lambda: ctx.mfcc(d_sig, offs, cfg, win, bank, False, False, fo), frames, "frames",
          algo_bytes=frames * 1592.0)
    d_i16 = ctx.to_device(np.clip(batch, -32768, 32767).astype(np.int16))
    timed("mfcc int16 PCM (U=%d)" % U, lambda: ctx.mfcc(d_i16, offs, cfg, win, bank, False, False, fo), frames, "frames",
          algo_bytes=frames * 632.0)

if what & {"e2e", "all"}:
    # PCM -> state paths: int16 samples resident in HBM -> MFCC (D=39) -> emission -> Viterbi -> backtrace,
    # features and trellis never leave the device; 1024 utterances x 1000 frames, N=128, M=16
    U, n, N, M, D = 1024, 160240, 128, 16, 39
    sig = np.clip(synth.synth_signal(n, seed=7), -32768, 32767).astype(np.int16)
    d_pcm = ctx.to_device(np.tile(sig, U))
    offs = (np.arange(U + 1) * n).astype(np.int64)
    cfg, win, bank = MFCC()._cfg()
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    feat = ctx.empty((U * 1000, D), np.float64)
    path = ctx.empty((U, 1000), np.uint32)

    def e2e():
        ctx.mfcc(d_pcm, offs, cfg, win, bank, False, False, feat)
        ctx.decode(feat.view((U, 1000, D)), path_out=path)

    timed("int16 PCM -> MFCC -> decode (U=%d)" % U, e2e, U * 1000, "frames", algo_bytes=U * 1000 * (632.0 + 4))
    del d_pcm, feat, path

if what & {"seg", "all"}:
    # scripts/seg/seg.py shape: BMES segmentation HMM, n = 4 states, m = 65536 symbols, one long
    # sequence through the reference-shaped API (host arrays in, host path out)
    from lnasr.hmm import HMM
    n, m, T = 4, 65536, 30000
    d = synth.synth_discrete(n, m, T, seed=3)
    h = HMM(n, m, d["A"], d["B"], d["pi"])
    timed("HMM.decode n=4 m=65536 T=%d" % T, lambda: h.decode(d["obs"]), T, "frames")

if what & {"refapi", "all"}:
    # the reference's own method names on one utterance, metric shape (tools/refapi.py; bench.py's reference_api block)
    import refapi
    r = refapi.measure(ctx)
    for k, v in r["methods"].items():
        print("refapi %-22s wall %8.3f ms  device %8.3f ms  host %8.3f ms   %s" % (k, v["wall_ms"], v["device_ms"], v["host_ms"], v["kernels"]))
    print("refapi parameter fingerprint %.3f ms per call; b + bm read back afterwards %.3f ms; reference decode() %.0f ms"
          % (r["parameter_fingerprint_ms"], r["read_b_and_bm_afterwards_ms"], r["reference_decode_ms"]))

if what & {"nanfix", "all"}:
    # What the non-finite-data checks cost on CLEAN data (the parity suite asserts launch counts only):
    # decode (the emission kernel tested the features it staged: one launch that reads a word), lnasr_viterbi on a dense
    # model (the back-trace kernel looked at v), and lnasr_viterbi on a LEFT-TO-RIGHT model, whose -inf transitions put
    # -inf cells into v on perfectly clean data: the suspect word is raised and the repair kernel scans b for nothing.
    U, T, N, M, D = 64, 1000, 128, 16, 39
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    obs = ctx.to_device(synth.synth_obs(U, T, D, seed=5))
    timed("nanfix: decode, clean data (U=%d)" % U, lambda: ctx.decode(obs), U * T, "frames")
    b = ctx.gmm_logprob(obs)
    timed("nanfix: viterbi, dense model", lambda: ctx.viterbi(b, want_v=False, want_bt=False), U * T, "frames")
    A = mdl["A"].copy()
    A[np.tril_indices(N, -1)] = -np.inf                  # left-to-right: no way back, every path starts in state 0
    pi0 = np.full(N, -np.inf)
    pi0[0] = 0.0
    ctx.upload_hmm(A, pi0)
    timed("nanfix: viterbi, left-to-right model", lambda: ctx.viterbi(b, want_v=False, want_bt=False), U * T, "frames")
    b1 = ctx.to_device(b.to_host()[:1])
    timed("nanfix: viterbi, left-to-right, U=1", lambda: ctx.viterbi(b1, want_v=False, want_bt=False), T, "frames")
    ctx.upload_hmm(mdl["A"], mdl["pi"])
    timed("nanfix: viterbi, dense, U=1", lambda: ctx.viterbi(b1, want_v=False, want_bt=False), T, "frames")
    del b, b1, obs

if what & {"gmm", "all"}:
    # config 3: T=1000, M=16, D=39, U=1024; single GMM (N=1) and the 128-state bank
    U, T, M, D = 1024, 1000, 16, 39
    obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
    for N in (1, 128):
        mdl = synth.synth_model(N, M, D, seed=1)
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
        b = ctx.empty((U, T, N), np.float64)
        lib = _ffi.lib()
        fn = lambda: _ffi.check(lib.lnasr_gmm_logprob(ctx.handle, obs.ptr, U, T, None, 0, b.ptr, None))
        timed("gmm log-likelihood N=%d" % N, fn, U * T, "frames", algo_bytes=U * T * (312.0 + 8 * N),
              algo_flop=U * T * 3.0 * N * M * D)

if what & {"fwd", "all"}:
    U, T, N, M, D = 1024, 1000, 128, 16, 39
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
    b = ctx.gmm_logprob(obs)
    al = ctx.empty((U, T, N), np.float64)
    ll = ctx.empty((U,), np.float64)
    lib = _ffi.lib()
    timed("forward N=128 (alpha + loglik)", lambda: _ffi.check(lib.lnasr_forward(ctx.handle, b.ptr, U, T, None, 0, al.ptr, ll.ptr)),
          U * T, "frames", algo_flop=U * T * 3.0 * N * N)
    timed("backward N=128", lambda: _ffi.check(lib.lnasr_backward(ctx.handle, b.ptr, U, T, None, 0, al.ptr)), U * T, "frames",
          algo_flop=U * T * 3.0 * N * N)
    timed("calc_prob (emission + forward)", lambda: _ffi.check(lib.lnasr_calc_prob(ctx.handle, obs.ptr, U, T, None, ll.ptr)),
          U * T, "frames")

if what & {"fwd1"}:
    # one utterance, N = 128: the latency tier of forward / backward (for tools/pmc_rows.sh)
    T, N, M, D = 1000, 128, 16, 39
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    obs = ctx.to_device(synth.synth_obs(1, T, D, seed=1000))
    b = ctx.gmm_logprob(obs)
    al = ctx.empty((1, T, N), np.float64)
    ll = ctx.empty((1,), np.float64)
    lib = _ffi.lib()
    timed("forward U=1", lambda: _ffi.check(lib.lnasr_forward(ctx.handle, b.ptr, 1, T, None, 0, al.ptr, ll.ptr)), T, "frames", reps=10)
    timed("backward U=1", lambda: _ffi.check(lib.lnasr_backward(ctx.handle, b.ptr, 1, T, None, 0, al.ptr)), T, "frames", reps=10)

if what & {"vit1"}:
    # one utterance, N = 128: the latency tier of Viterbi + back-trace (for tools/pmc_rows.sh)
    T, N, M, D = 1000, 128, 16, 39
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    obs = ctx.to_device(synth.synth_obs(1, T, D, seed=1000))
    timed("decode U=1", lambda: ctx.decode(obs), T, "frames", reps=10)

if what & {"midn"}:
    # 17 ... 64 states on ONE utterance (no BASELINE configuration; the reference's API takes any n)
    from lnasr.hmm import HMM
    for N in (24, 32, 64):
        T = 4000
        d = synth.synth_discrete(N, 4, T, seed=N)
        ctx.upload_hmm(d["A"], d["pi"])
        b = ctx.to_device(np.ascontiguousarray(-30.0 * np.random.default_rng(N).random((1, T, N))))
        al = ctx.empty((1, T, N), np.float64)
        ll = ctx.empty((1,), np.float64)
        lib = _ffi.lib()
        timed("forward U=1 N=%d T=%d" % (N, T), lambda: _ffi.check(lib.lnasr_forward(ctx.handle, b.ptr, 1, T, None, 0, al.ptr, ll.ptr)), T, "frames")
        timed("backward U=1 N=%d" % N, lambda: _ffi.check(lib.lnasr_backward(ctx.handle, b.ptr, 1, T, None, 0, al.ptr)), T, "frames")
        timed("viterbi U=1 N=%d" % N, lambda: ctx.viterbi(b, want_v=False, want_bt=False), T, "frames")

if what & {"fwdsweep"}:
    # dispatch::fwdbwd_one_per_block (csrc/trellis.h): one utterance per four-wave block against two, N = 128, by batch size
    import contextlib
    T, N, M, D = 1000, 128, 16, 39
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    lib = _ffi.lib()
    for U in (1, 64, 256, 384, 512, 768):
        obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
        b = ctx.gmm_logprob(obs)
        al = ctx.empty((U, T, N), np.float64)
        ll = ctx.empty((U,), np.float64)
        for maxu, maxl, tag in ((0, 0, "two per block"), (1 << 20, 0, "one per block"), (0, 1 << 20, "one per block, log/exp off the chain")):
            os.environ["LNASR_FWD_ONE_MAXU"] = str(maxu)
            os.environ["LNASR_FWD_LAT_MAXU"] = str(maxl)
            _ffi.reload_switches()
            timed("forward U=%d, %s" % (U, tag), lambda: _ffi.check(lib.lnasr_forward(ctx.handle, b.ptr, U, T, None, 0, al.ptr, ll.ptr)),
                  U * T, "frames")
            timed("backward U=%d, %s" % (U, tag), lambda: _ffi.check(lib.lnasr_backward(ctx.handle, b.ptr, U, T, None, 0, al.ptr)),
                  U * T, "frames")
        del obs, b, al, ll
    os.environ.pop("LNASR_FWD_ONE_MAXU", None)
    os.environ.pop("LNASR_FWD_LAT_MAXU", None)
    _ffi.reload_switches()

if what & {"btsweep"}:
    # dispatch::backtrace_chase (csrc/trellis.h): back-pointer chase against the arg-max back-trace, N = 128, by batch size
    T, N, M, D = 1000, 128, 16, 39
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    for U in (1, 4, 8, 16, 32, 64, 128):
        obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
        ref = None
        for maxu, tag in ((0, "arg-max back-trace"), (1 << 20, "back-pointer chase")):
            os.environ["LNASR_CHASE_MAXU"] = str(maxu)
            _ffi.reload_switches()
            timed("decode U=%d, %s" % (U, tag), lambda: ctx.decode(obs), U * T, "frames")
            p = ctx.decode(obs).to_host()
            assert ref is None or np.array_equal(ref, p)
            ref = p
        del obs
    os.environ.pop("LNASR_CHASE_MAXU", None)
    _ffi.reload_switches()

if what & {"bw", "all"}:
    # config 5: N=256, M=32, T=2000, D=39, 64 utterances per GPU
    U, T, N, M, D = 64, 2000, 256, 32, 39
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
    stats = ctx.empty((ctx.bw_stats_len(),), np.float64)
    lib = _ffi.lib()
    timed("Baum-Welch statistics N=256 M=32", lambda: _ffi.check(lib.lnasr_bw_accumulate(ctx.handle, obs.ptr, U, T, None, stats.ptr)),
          U * T, "frames", reps=2)
    timed("Baum-Welch statistics, precision=float32", lambda: _ffi.check(lib.lnasr_bw_accumulate_p(ctx.handle, obs.ptr, U, T, None, 1, stats.ptr)),
          U * T, "frames", reps=2)
    # the two numbers that price the "recompute bm inside the moments kernel" alternative: the emission
    # WITHOUT the per-mixture output (what a fused pass would still run before the trellises), and with it (what runs now)
    bb = ctx.empty((U, T, N), np.float64)
    timed("  emission, b only (config-5 shape)", lambda: _ffi.check(lib.lnasr_gmm_logprob(ctx.handle, obs.ptr, U, T, None, 0, bb.ptr, None)),
          U * T, "frames", reps=2, algo_flop=U * T * 3.0 * N * M * D)
    bm = ctx.empty((U, T, N, M), np.float64)
    timed("  emission, b + bm (config-5 shape)", lambda: _ffi.check(lib.lnasr_gmm_logprob(ctx.handle, obs.ptr, U, T, None, 0, bb.ptr, bm.ptr)),
          U * T, "frames", reps=2, algo_flop=U * T * 3.0 * N * M * D)
    del bm, bb

if what & {"full", "all"}:
    # full-covariance bank (the reference's own M-step produces full Sigma): N=128, M=16, D=39
    U, T, N, M, D = 16, 1000, 128, 16, 39
    mdl = synth.synth_model(N, M, D, seed=1, full_cov=True)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    obs = ctx.to_device(synth.synth_obs(U, T, D, seed=1000))
    b = ctx.empty((U, T, N), np.float64)
    lib = _ffi.lib()
    timed("gmm log-likelihood FULL cov N=128", lambda: _ffi.check(lib.lnasr_gmm_logprob(ctx.handle, obs.ptr, U, T, None, 0, b.ptr, None)),
          U * T, "frames", reps=2, algo_flop=U * T * 2.0 * N * M * (D * (D + 1) / 2 + D + 1))
