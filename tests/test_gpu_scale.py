"""GPU, BASELINE.json full sizes: size-independent properties of the decode path (determinism,
path-score identity, spot checks against the oracle), and the section-8f 'next' rows
(character-segmentation sized discrete HMM: n=4, m=65536, long T)."""
import os

import numpy as np
import pytest

from conftest import assert_close, env_switch, tuned

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def test_metric_config_full_size(synth, oracle):
    """N=128, M=16, D=39, T=1000, U=1024 (one GPU's shard of config 4), inputs resident in HBM."""
    from lnasr import _ffi
    ctx = _ffi.default_ctx()
    N, M, D, T, U = 128, 16, 39, 1000, 1024
    mdl = synth.synth_model(N, M, D, seed=1)
    obs = synth.synth_obs(U, T, D, seed=1000)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    d_obs = ctx.to_device(obs)
    ctx.prof_enable(True)
    ctx.prof_reset()
    p1, s1 = ctx.decode(d_obs, want_score=True)
    names = set(ctx.prof_report())
    ctx.prof_enable(False)
    # (round 5: at this size the emission is the integer matrix-core bank, csrc/gmm_i8.hip -- every check below, the exhaustive
    # comparison with the oracle included, is a check of that kernel)
    assert "emit_diag_i8" in names or not tuned(), names
    p2, s2 = ctx.decode(d_obs, want_score=True)
    path, score = p1.to_host(), s1.to_host()
    assert np.array_equal(path, p2.to_host()) and np.array_equal(score, s2.to_host())      # deterministic
    assert path.max() < N
    # path-score identity: the Viterbi score equals the score accumulated along the decoded path
    b = ctx.gmm_logprob(d_obs).to_host()
    uu = np.arange(U)[:, None]
    tt = np.arange(T)[None, :]
    emis = b[uu, tt, path]
    trans = mdl["A"][path[:, :-1], path[:, 1:]]
    along = mdl["pi"][path[:, 0]] + emis.sum(axis=1) + trans.sum(axis=1)
    assert_close(along, score, 1e-11)
    # optimality spot check: no single-state substitution improves the path score (local optimality)
    rng = np.random.default_rng(0)
    for _ in range(200):
        u, t = int(rng.integers(U)), int(rng.integers(1, T - 1))
        cur = path[u, t]
        alt = int(rng.integers(N))
        d_cur = mdl["A"][path[u, t - 1], cur] + b[u, t, cur] + mdl["A"][cur, path[u, t + 1]]
        d_alt = mdl["A"][path[u, t - 1], alt] + b[u, t, alt] + mdl["A"][alt, path[u, t + 1]]
        assert d_alt <= d_cur + 1e-9
    # exact agreement with the oracle's plain restatement on a sample of utterances ...
    for u in rng.choice(U, 6, replace=False):
        ob, _ = oracle.gmm_map_b_diag(obs[u], mdl["w"], mdl["mu"], mdl["var"])
        ref = oracle.viterbi(mdl["A"], mdl["pi"], ob)[2]
        assert np.array_equal(path[u], ref), int(u)
    # ... and EXHAUSTIVELY: all 1024 x 1000 decoded states against the oracle (north_star: "Viterbi paths bit-identical").
    # The oracle runs in a fresh process that never touches the GPU (a fork pool may not follow a HIP context).
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "paths.npy")
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_oracle_paths.py"), out, str(U), "1000",
                            str(min(16, len(os.sched_getaffinity(0))))], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        want = np.load(out)
    assert want.shape == path.shape and want.dtype == np.uint32
    mism = np.flatnonzero((want != path).any(axis=1))
    assert mism.size == 0, "utterances whose path differs from the oracle's: %s" % mism[:20]
    # forward/backward at full size: sum_j exp(alpha+beta) is frame independent and equals the likelihood
    d_b = ctx.to_device(b[:64])
    alpha, ll = ctx.forward(d_b)
    beta = ctx.backward(d_b)
    tot = oracle.lse(alpha.to_host() + beta.to_host(), axis=2)
    assert np.abs(tot - ll.to_host()[:, None]).max() < 1e-7


def test_single_gmm_config_full_size(synth, oracle):
    """BASELINE configs[2]: one 16-mixture diagonal GMM (N = 1), D = 39, T = 1000, 1024 utterances resident in HBM --
    the HBM-bound kernel at its full size: spot checks against the oracle, b = lse_m(w + bm) everywhere, determinism."""
    from lnasr import _ffi
    ctx = _ffi.default_ctx()
    M, D, T, U = 16, 39, 1000, 1024
    mdl = synth.synth_model(1, M, D, seed=12)
    obs = synth.synth_obs(U, T, D, seed=1000)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    d_obs = ctx.to_device(obs)
    ctx.prof_enable(True)
    ctx.prof_reset()
    b1 = ctx.gmm_logprob(d_obs).to_host()
    names = set(ctx.prof_report())
    ctx.prof_enable(False)
    assert "emit_single_direct" in names or not tuned()
    assert np.array_equal(b1, ctx.gmm_logprob(d_obs).to_host())
    assert np.isfinite(b1).all() and b1.shape == (U, T, 1)
    rng = np.random.default_rng(3)
    for u in rng.choice(U, 5, replace=False):
        ob, _ = oracle.gmm_map_b_diag(obs[u], mdl["w"], mdl["mu"], mdl["var"])
        assert_close(b1[u, :, 0], ob[0], 1e-12)
    # the per-mixture output of a slice (bm for everything would be 131 MB: fine, but a slice says the same)
    sl = ctx.to_device(obs[:64])
    b2, bm2 = ctx.gmm_logprob(sl, want_bm=True)
    b2, bm2 = b2.to_host(), bm2.to_host()
    assert np.array_equal(b2, b1[:64])
    assert_close(oracle.lse(mdl["w"][0][None, None, :] + bm2[:, :, 0, :], axis=2), b2[:, :, 0], 1e-13)


def test_segmentation_sized_discrete_hmm(oracle):
    """scripts/seg/seg.py shape: n = 4 (BMES), m = 65536 symbols, one long character sequence."""
    from lnasr.hmm import HMM
    rng = np.random.default_rng(4)
    n, m, T = 4, 65536, 30000
    a = 1.0 - rng.random((n, n))
    bb = 1.0 - rng.random((n, m))
    p = 1.0 - rng.random(n)
    A, B, pi = np.log(a / a.sum(1, keepdims=True)), np.log(bb / bb.sum(1, keepdims=True)), np.log(p / p.sum())
    obs = rng.integers(0, m, T).astype(np.int32)
    h = HMM(n, m, A, B, pi)
    path = h.decode(obs)
    ref = oracle.viterbi(A, pi, oracle.map_b(B, obs))[2]
    assert np.array_equal(path, ref)
    ll = h.calc_prob(obs[:2000])
    ref_ll = oracle.loglik(oracle.forward(A, pi, oracle.map_b(B, obs[:2000])))
    assert abs(ll - ref_ll) < 1e-10 * abs(ref_ll)


def test_host_buffers_are_uploaded_in_a_pipeline(synth):
    """Observations handed over as numpy arrays are uploaded in chunks from a helper thread while the emission
    kernel already runs (>= 128 utterances); the result is the one a device-resident batch gives, ragged or not."""
    from lnasr import _ffi
    ctx = _ffi.default_ctx()
    U, T, N, M, D = 333, 24, 16, 4, 39
    mdl = synth.synth_model(N, M, D, seed=9)
    obs = synth.synth_obs(U, T, D, seed=91)
    lens = np.random.default_rng(5).integers(1, T + 1, U).astype(np.int32)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    dev = ctx.decode(ctx.to_device(obs), lens=lens).to_host()
    piped = np.asarray(ctx.decode(obs, lens=lens))
    ll_piped = np.asarray(ctx.calc_prob(obs, lens=lens))
    with env_switch(LNASR_NO_PIPELINE=1):
        plain = np.asarray(ctx.decode(obs, lens=lens))
        ll_plain = np.asarray(ctx.calc_prob(obs, lens=lens))
    for u in range(U):
        assert np.array_equal(piped[u, :lens[u]], dev[u, :lens[u]]) and np.array_equal(plain[u, :lens[u]], dev[u, :lens[u]])
    assert np.array_equal(ll_piped, ll_plain)


def test_rccl_single_rank_collectives():
    """The two collectives on a 1-rank RCCL communicator (all a 1-GPU box can run): the path gather with and
    without the byte packing gives back the input, the statistics all-reduce leaves a single contribution
    unchanged."""
    from lnasr import _ffi
    ctx = _ffi.Context(0)
    try:
        ctx.comm_init(_ffi.comm_unique_id(), 0, 1)
        rng = np.random.default_rng(3)
        p = rng.integers(0, 128, (37, 40)).astype(np.uint32)          # 1480 entries: a multiple of 4
        assert np.array_equal(np.asarray(ctx.allgather_u32(p))[0], p)
        assert np.array_equal(np.asarray(ctx.allgather_paths(p, 128))[0], p)
        assert np.array_equal(ctx.allgather_paths(ctx.to_device(p), 256).to_host()[0], p)
        q = rng.integers(0, 1000, (3, 7)).astype(np.uint32)           # 21 entries, 1000 states: plain u32 route
        assert np.array_equal(np.asarray(ctx.allgather_paths(q, 1000))[0], q)
        assert np.array_equal(np.asarray(ctx.allgather_paths(q, 200))[0], q)
        s = rng.standard_normal(1000)
        assert np.array_equal(np.asarray(ctx.allreduce_f64(s.copy())), s)
    finally:
        ctx.comm_destroy()


def test_pcm_to_paths_on_device(synth):
    """int16 PCM -> MFCC -> decode with the features never leaving the device (DeviceArray.view) gives the paths of
    the host round trip (Recognizer-style: calc_mfcc to numpy, then decode)."""
    from lnasr import _ffi
    from lnasr.mfcc import MFCC
    ctx = _ffi.default_ctx()
    U, n, N, M, D = 6, 16240, 12, 3, 39                     # 100 frames per utterance
    rng = np.random.default_rng(3)
    pcm = rng.integers(-4000, 4000, U * n).astype(np.int16)
    offs = (np.arange(U + 1) * n).astype(np.int64)
    mdl = synth.synth_model(N, M, D, seed=2)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    m = MFCC()
    cfg, win, bank = m._cfg()
    _, _, feat, foff = ctx.mfcc(ctx.to_device(pcm), offs, cfg, win, bank, False, False)
    T = int(foff[1])
    assert np.all(np.diff(foff) == T)
    path_dev = np.asarray(ctx.decode(feat.view((U, T, D))).to_host())
    host_feat = np.stack([m.calc_mfcc(pcm[offs[u]:offs[u + 1]])[2] for u in range(U)])
    assert np.array_equal(feat.to_host().reshape(U, T, D), host_feat)
    assert np.array_equal(path_dev, np.asarray(ctx.decode(host_feat)))
    with pytest.raises(ValueError):
        feat.view((U, T, D + 1))


def test_ragged_pcm_to_paths_on_device(synth):
    """Variable-length utterances batched by offsets: int16 PCM -> MFCC (packed [F, 39]) -> pad_rows ->
    decode with lens, all on the device, equals decoding every utterance on its own."""
    from lnasr import _ffi
    from lnasr.mfcc import MFCC
    ctx = _ffi.default_ctx()
    N, M, D = 12, 3, 39
    rng = np.random.default_rng(4)
    ns = [16240, 4000, 700, 9999, 16241, 2500]
    pcm = [rng.integers(-4000, 4000, n).astype(np.int16) for n in ns]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    mdl = synth.synth_model(N, M, D, seed=2)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    m = MFCC()
    cfg, win, bank = m._cfg()
    _, _, feat, foff = ctx.mfcc(ctx.to_device(np.concatenate(pcm)), offs, cfg, win, bank, False, False)
    padded, lens = ctx.pad_rows(feat, foff)
    assert padded.shape == (len(ns), int(np.diff(foff).max()), D) and np.array_equal(lens, np.diff(foff))
    paths = ctx.decode(padded, lens=lens).to_host()
    host = padded.to_host()
    for u, x in enumerate(pcm):
        fu = m.calc_mfcc(x)[2]
        assert np.array_equal(host[u, :lens[u]], fu) and not host[u, lens[u]:].any()
        assert np.array_equal(paths[u, :lens[u]], np.asarray(ctx.decode(fu[None]))[0])
    # host arrays take the same route; a pad shorter than an utterance is refused
    p2, _ = ctx.pad_rows(feat.to_host(), foff, pad_rows=int(lens.max()) + 3)
    assert p2.shape[1] == lens.max() + 3 and np.array_equal(p2[:, :lens.max()], host)
    with pytest.raises(_ffi.LnasrError):
        ctx.pad_rows(feat, foff, pad_rows=int(lens.max()) - 1)


def test_back_to_back_pageable_batches_keep_their_own_observations(synth, ctx):
    """Two decode calls in a row on DIFFERENT pageable batches with device outputs and no lens: the first call returns
    when its last uploads are enqueued, not finished, and the second call's host threads refill the pinned staging ring
    at once.  A ring slot must not be rewritten before the copy that last read it has completed (per-slot event), or
    the first batch silently scores the second batch's observations.  Both results equal the resident runs."""
    N, M, D, U, T = 128, 16, 39, 1024, 160
    mdl = synth.synth_model(N, M, D, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    batches = [synth.synth_obs(U, T, D, seed=s) for s in (41, 42, 43)]
    want = [ctx.decode(ctx.to_device(x)).to_host() for x in batches]
    assert not np.array_equal(want[0], want[1])
    outs = [ctx.empty((U, T), np.uint32) for _ in batches]
    for x, o in zip(batches, outs):                   # no host-side wait between the calls
        ctx.decode(x, path_out=o)
    ctx.sync()
    for k, o in enumerate(outs):
        assert np.array_equal(o.to_host(), want[k]), "batch %d decoded another batch's observations" % k
    # the statistics pass takes the same upload route
    st = [ctx.empty((ctx.bw_stats_len(),), np.float64) for _ in batches[:2]]
    ref = [ctx.bw_accumulate(ctx.to_device(x[:256])).copy() for x in batches[:2]]
    for x, o in zip(batches[:2], st):
        ctx.bw_accumulate(x[:256], stats_out=o)
    ctx.sync()
    for k, o in enumerate(st):
        assert_close(o.to_host(), ref[k], 1e-12, 1e-12, msg="statistics of batch %d" % k)
