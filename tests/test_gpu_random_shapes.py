"""GPU parity over seeded random shapes: every dispatch branch (single mixture, matrix-core bank with and without
virtual states, scalar kernels, on-chip / register / streaming trellises, statistics tile counts) meets shapes it was
not hand-picked for.  Emission, forward and backward against the oracle to 1e-10; Viterbi on the GPU's own emissions
bit-identical to the oracle's recursion on the same numbers; statistics to 1e-8."""
import numpy as np
import pytest

from conftest import assert_close, env_switch

pytestmark = pytest.mark.gpu

# LNASR_TEST_SEEDS="a:b" widens (or moves) the seed range of both sweeps for a soak run; the default is what the
# suite runs every time
import os
_lo, _hi = (int(x) for x in os.environ.get("LNASR_TEST_SEEDS", "0:28").split(":"))
_SEEDS_DIAG = range(_lo, _hi)
_SEEDS_FULL = range(_lo, _lo + max(1, (_hi - _lo) * 12 // 28))


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def _shape(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 31, 64, 65, 100, 128, 129, 160, 256, 257, 300]))
    M = int(rng.choice([1, 2, 3, 4, 7, 8, 15, 16, 17, 32, 48]))
    D = int(rng.choice([1, 2, 3, 5, 12, 13, 20, 26, 27, 38, 39, 40, 50, 63, 64]))
    # keep the oracle's work (N * M * T * D, and N^2 * T for the trellises) bounded
    T = int(rng.integers(1, max(2, min(40, 400000 // (N * M * D) + 2, 3000000 // (N * N) + 2))))
    U = int(rng.integers(1, 5))
    return rng, N, M, D, T, U


@pytest.mark.parametrize("seed", _SEEDS_DIAG)
def test_random_shape(seed, synth, oracle, ctx):
    rng, N, M, D, T, U = _shape(seed)
    mdl = synth.synth_model(N, M, D, seed=seed + 7)
    w = mdl["w"].copy()
    A = mdl["A"].copy()
    if seed % 3 == 0 and M > 1:                            # a dead component
        w[0, M - 1] = -np.inf
        w[0] -= oracle.lse(w[0])
    if seed % 4 == 1 and N > 2:                            # a forbidden transition and an unreachable state
        A[0, 1] = -np.inf
        A[:, N - 1] = -np.inf
        A -= oracle.lse(A, axis=1)[:, None]
    obs = synth.synth_obs(U, T, D, seed=seed + 50)
    lens = rng.integers(1, T + 1, U).astype(np.int32)
    lens[0] = T
    ctx.upload_gmmhmm(A, mdl["pi"], w, mdl["mu"], mdl["var"], diag=True)
    tag = "seed=%d N=%d M=%d D=%d T=%d U=%d" % (seed, N, M, D, T, U)

    b, bm = ctx.gmm_logprob(obs, want_bm=True)
    r = ctx.viterbi(b, lens=lens)
    alpha, ll = ctx.forward(b, lens=lens)
    beta = ctx.backward(b, lens=lens)
    path_fused = np.asarray(ctx.decode(obs, lens=lens))
    for u in range(U):
        L = int(lens[u])
        with np.errstate(divide="ignore", invalid="ignore"):
            ob, obm = oracle.gmm_map_b_diag(obs[u, :L], w, mdl["mu"], mdl["var"])
        assert_close(b[u, :L], ob.T, 1e-10, 1e-10, msg="b " + tag)
        assert_close(bm[u, :L], np.transpose(obm, (2, 0, 1)), 1e-10, 1e-10, msg="bm " + tag)
        gb = np.ascontiguousarray(b[u, :L].T)                 # the trellises are checked on the GPU's own b
        v, bt, path = oracle.viterbi(A, mdl["pi"], gb)
        assert np.array_equal(r["v"][u, :L], v), "v " + tag
        assert np.array_equal(r["bt"][u, :L], bt) and np.array_equal(r["path"][u, :L], path), "path " + tag
        assert np.array_equal(path_fused[u, :L], path), "decode " + tag
        with np.errstate(divide="ignore", invalid="ignore"):
            ra, rb = oracle.forward(A, mdl["pi"], gb), oracle.backward(A, gb)
        assert_close(alpha[u, :L], ra, 1e-10, 1e-10, msg="alpha " + tag)
        assert_close(beta[u, :L], rb, 1e-10, 1e-10, msg="beta " + tag)
        assert abs(ll[u] - oracle.loglik(ra)) <= 1e-10 * max(1.0, abs(ll[u])), "loglik " + tag

    # precision=np.float32 storage on the same dispatch branch: v / bt / path bit-identical to the oracle's float32 run, alpha and
    # beta rounded at every store like the reference's float32 arrays (at most one ulp where an fp64 sum straddles a boundary)
    b32 = np.asarray(b).astype(np.float32)
    r32 = ctx.viterbi(b32, lens=lens)
    a32, _ = ctx.forward(b32, lens=lens)
    be32 = ctx.backward(b32, lens=lens)
    for u in range(U):
        L = int(lens[u])
        gb = np.ascontiguousarray(b32[u, :L].T)
        with np.errstate(divide="ignore", invalid="ignore"):
            v, bt, path = oracle.viterbi(A, mdl["pi"], gb, precision=np.float32)
            ra, rb = oracle.forward(A, mdl["pi"], gb, precision=np.float32), oracle.backward(A, gb, precision=np.float32)
        assert r32["v"].dtype == np.float32 and np.array_equal(r32["v"][u, :L], v), "v f32 " + tag
        assert np.array_equal(r32["bt"][u, :L], bt) and np.array_equal(r32["path"][u, :L], path), "path f32 " + tag
        for got, want, what in ((a32[u, :L], ra, "alpha"), (be32[u, :L], rb, "beta")):
            assert got.dtype == np.float32
            assert_close(got, want, 1.3e-7, 0.0, msg="%s f32 %s" % (what, tag))
            off = int((got != want).sum()) - int((np.isnan(got) & np.isnan(want)).sum())
            assert off <= max(2, want.size // 5000), "%s f32 %s: %d of %d values off by an ulp" % (what, tag, off, want.size)

    if N * N * T <= 400000 and N * M * D * T <= 200000:         # statistics: the oracle materialises xi [T,N,N]
        st = ctx.bw_accumulate(obs, lens=lens)
        with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
            ref = oracle.bw_stats_diag([obs[u, :lens[u]] for u in range(U)], A, mdl["pi"], w, mdl["mu"], mdl["var"])
        o = 0
        for key, n in (("xi_sum", N * N), ("gamma_sum", N), ("gamma0", N), ("mix", N * M), ("mix_x", N * M * D),
                       ("mix_xx", N * M * D)):
            assert_close(st[o:o + n], np.asarray(ref[key]).ravel(), 1e-8, 1e-10, msg=key + " " + tag)
            o += n
        assert abs(st[o] - ref["loglik"]) <= 1e-9 * max(1.0, abs(ref["loglik"]))


@pytest.mark.parametrize("seed", _SEEDS_DIAG)
def test_random_shape_bad_features(seed, synth, oracle, ctx):
    """The same sweep with a NaN or an infinite feature dropped into one utterance (inside or behind its end): on every
    dispatch branch the emissions get the reference's NaN rows, decode and the Viterbi on those emissions return the
    oracle's v / bt / path NaN for NaN, the likelihood is NaN exactly where the oracle's is, and the statistics block is
    all NaN or untouched."""
    rng, N, M, D, T, U = _shape(seed)
    rng = np.random.default_rng(5000 + seed)
    mdl = synth.synth_model(N, M, D, seed=seed + 7)
    if seed % 3 == 0 and M > 1:                            # a dead component
        mdl["w"][0, M - 1] = -np.inf
        mdl["w"][0] -= oracle.lse(mdl["w"][0])
    if seed % 4 == 1 and N > 2:                            # a forbidden transition and an unreachable state: -inf cells in v
        mdl["A"][0, 1] = -np.inf
        mdl["A"][:, N - 1] = -np.inf
        mdl["A"] -= oracle.lse(mdl["A"], axis=1)[:, None]
    obs = synth.synth_obs(U, T, D, seed=seed + 50)
    lens = rng.integers(1, T + 1, U).astype(np.int32)
    lens[0] = T
    ub, tb, db = int(rng.integers(0, U)), int(rng.integers(0, T)), int(rng.integers(0, D))
    obs[ub, tb, db] = [np.nan, np.inf, -np.inf][seed % 3]
    inside = tb < lens[ub]
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    tag = "seed=%d N=%d M=%d D=%d T=%d U=%d bad=(%d,%d,%d) inside=%s" % (seed, N, M, D, T, U, ub, tb, db, inside)
    b = ctx.gmm_logprob(obs)
    # whole NaN rows where a feature is not finite -- except that with D = 1 there is no zero off-diagonal for an infinity
    # to meet (gmm.py:63-68): the reference's density is then -inf, as the oracle's direct form says
    rows = ~np.isfinite(obs).all(axis=2) if D > 1 else np.isnan(obs).any(axis=2)
    assert np.array_equal(np.isnan(b).all(axis=2), rows) and np.array_equal(np.isnan(b).any(axis=2), rows), "b " + tag
    if D == 1:
        assert np.array_equal(np.isneginf(b).all(axis=2), np.isinf(obs).any(axis=2)), "b -inf rows " + tag
    r = ctx.viterbi(b, lens=lens, want_score=True)
    paths, score = ctx.decode(obs, lens=lens, want_score=True)
    ll = ctx.calc_prob(obs, lens=lens)
    alpha, ll_fw = ctx.forward(b, lens=lens)
    beta = ctx.backward(b, lens=lens)
    for u in range(U):
        L = int(lens[u])
        gb = np.ascontiguousarray(b[u, :L].T)
        with np.errstate(invalid="ignore"):
            v, bt, path = oracle.viterbi(mdl["A"], mdl["pi"], gb)
        assert np.array_equal(r["v"][u, :L], v, equal_nan=True), "v " + tag
        assert np.array_equal(r["bt"][u, :L], bt) and np.array_equal(r["path"][u, :L], path), "path " + tag
        assert np.array_equal(np.asarray(paths)[u, :L], path), "decode " + tag
        with np.errstate(divide="ignore", invalid="ignore"):
            ra, rb = oracle.forward(mdl["A"], mdl["pi"], gb), oracle.backward(mdl["A"], gb)
        for got, want, nm in ((alpha[u, :L], ra, "alpha "), (beta[u, :L], rb, "beta ")):
            assert np.array_equal(np.isnan(got), np.isnan(want)), nm + tag
            ok = np.isfinite(want)
            assert np.allclose(got[ok], want[ok], rtol=1e-10, atol=1e-10), nm + tag
        assert np.isnan(ll_fw[u]) == np.isnan(oracle.loglik(ra)), "forward loglik " + tag
        hit = bool(inside and u == ub) and bool(np.isnan(gb).any())           # (D = 1 with an infinity: -inf, no NaN)
        assert np.isnan(r["score"][u]) == hit and np.isnan(score[u]) == hit and np.isnan(ll[u]) == hit, "score " + tag
    st = ctx.bw_accumulate(obs, lens=lens)
    n_sums = st.size - 2                                   # everything but the two counts
    assert np.isnan(st[:n_sums]).all() if inside else not np.isnan(st).any(), "stats " + tag
    assert st[-2] == U and st[-1] == lens.sum()


@pytest.mark.parametrize("seed", _SEEDS_DIAG)
def test_random_shape_real_feature_scale(seed, synth, oracle, ctx):
    """The same sweep at real-feature scale: a common offset per dimension (0, 5 or 50 units), means spread over 1, 3
    or 15 units, variances down to 0.02 -- whichever side of the conditioning gate the model falls on (centred matrix
    cores or direct form), emissions match the reference's direct (x - mu) arithmetic to 1e-11 and the decoded path is
    the oracle's on those emissions; under precision=float32 the statistics stay within float accuracy of the fp64
    pass."""
    rng, N, M, D, T, U = _shape(seed)
    rng = np.random.default_rng(3000 + seed)
    mdl = synth.synth_model(N, M, D, seed=seed + 11)
    offset = float(rng.choice([0.0, 5.0, 50.0])) * rng.standard_normal(D)
    spread = float(rng.choice([1.0, 3.0, 15.0]))
    vscale = float(rng.choice([1.0, 0.2, 0.04]))
    mu = offset + spread * rng.standard_normal((N, M, D))
    var = vscale * mdl["var"]
    pick_n, pick_m = rng.integers(0, N, (U, T)), rng.integers(0, M, (U, T))
    obs = mu[pick_n, pick_m] + np.sqrt(var[pick_n, pick_m]) * rng.standard_normal((U, T, D))
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mu, var, diag=True)
    kappa, direct = ctx.model_conditioning()
    tag = "seed=%d N=%d M=%d D=%d T=%d U=%d offset=%.0f spread=%.0f vscale=%.2f kappa=%.3g direct=%s" % (
        seed, N, M, D, T, U, np.abs(offset).max(), spread, vscale, kappa, direct)
    assert direct == (not kappa <= 4096.0), tag
    b, bm = ctx.gmm_logprob(obs, want_bm=True)
    paths = np.asarray(ctx.decode(obs))
    for u in range(U):
        ob, obm = oracle.gmm_map_b_diag_direct(obs[u], mdl["w"], mu, var)
        assert_close(b[u], ob.T, 1e-11, 1e-11, msg="b " + tag)
        assert_close(bm[u], np.transpose(obm, (2, 0, 1)), 1e-11, 1e-11, msg="bm " + tag)
        gb = np.ascontiguousarray(b[u].T)
        assert np.array_equal(paths[u], oracle.viterbi(mdl["A"], mdl["pi"], gb)[2]), "decode " + tag
    if T >= 2:
        st64 = ctx.bw_accumulate(obs)
        st32 = ctx.bw_accumulate(obs, precision=np.float32)
        scale = np.abs(st64).max()
        assert np.isfinite(st32).all() and np.abs(st32 - st64).max() <= 2e-4 * scale, "float32 pass " + tag


@pytest.mark.parametrize("seed", _SEEDS_FULL)
def test_random_shape_full_covariance(seed, synth, oracle, ctx):
    """Full covariance: emission against the oracle's (x-mu)^T Sigma^-1 (x-mu) form, the matrix-core statistics
    against the scalar statistics kernel."""
    import os
    rng = np.random.default_rng(2000 + seed)
    N = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 20]))
    M = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 17]))
    D = int(rng.choice([1, 2, 3, 5, 8, 13, 20, 39]))
    T = int(rng.integers(2, max(3, min(24, 60000 // (N * M * D * D) + 3))))
    U = int(rng.integers(1, 4))
    mdl = synth.synth_model(N, M, D, seed=seed + 3, full_cov=True)
    obs = synth.synth_obs(U, T, D, seed=seed + 70)
    lens = rng.integers(2, T + 1, U).astype(np.int32)
    lens[0] = T
    tag = "seed=%d N=%d M=%d D=%d T=%d U=%d" % (seed, N, M, D, T, U)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    b, bm = ctx.gmm_logprob(obs, want_bm=True)
    for u in range(U):
        ob, obm = oracle.gmm_map_b(obs[u], mdl["w"], mdl["mu"], mdl["si"])
        assert_close(b[u], ob.T, 1e-9, 1e-9, msg="b " + tag)
        assert_close(bm[u], np.transpose(obm, (2, 0, 1)), 1e-9, 1e-9, msg="bm " + tag)
    if D > 1 and not ctx.model_is_diag():
        st = ctx.bw_accumulate(obs, lens=lens)
        with env_switch(LNASR_NO_MFMA_STATS=1):
            ref = ctx.bw_accumulate(obs, lens=lens)
        assert_close(st, ref, 1e-8, 1e-10, msg="stats " + tag)
    # the same shapes with a NaN / +-inf feature.  A NaN makes the frame's densities NaN; an INFINITY under a genuinely full
    # covariance meets no zeros in the inverse and the reference's quadratic form comes out +inf or NaN component by
    # component (inf - inf where signs disagree): the bad frames are re-scored in the reference's own arithmetic, so b and
    # bm hold NaN and -inf exactly where the oracle's do, decode returns the oracle's path, the likelihood is NaN or -inf
    # as the oracle's, and the statistics are all NaN
    ub, tb, db = int(rng.integers(0, U)), int(rng.integers(0, T)), int(rng.integers(0, D))
    bad = obs.copy()
    bad[ub, tb, db] = [np.nan, np.inf, -np.inf][seed % 3]
    inside = tb < lens[ub]
    bb, bbm = ctx.gmm_logprob(bad, want_bm=True)
    paths = np.asarray(ctx.decode(bad, lens=lens))
    ll = ctx.calc_prob(bad, lens=lens)
    for u in range(U):
        L = int(lens[u])
        with np.errstate(all="ignore"):
            ob, obm = oracle.gmm_map_b(bad[u], mdl["w"], mdl["mu"], mdl["si"])
            _, _, path = oracle.viterbi(mdl["A"], mdl["pi"], ob[:, :L])
            oll = oracle.loglik(oracle.forward(mdl["A"], mdl["pi"], ob[:, :L]))
        for got, want, nm in ((bb[u], ob.T, "b "), (bbm[u], np.transpose(obm, (2, 0, 1)), "bm ")):
            assert np.array_equal(np.isnan(got), np.isnan(want)), nm + "NaN pattern " + tag
            assert np.array_equal(np.isneginf(got), np.isneginf(want)), nm + "-inf pattern " + tag
            ok = np.isfinite(want)
            assert np.allclose(got[ok], want[ok], rtol=1e-9, atol=1e-9), nm + tag
        assert np.array_equal(paths[u, :L], path), "decode with a bad feature " + tag
        assert np.isnan(ll[u]) == np.isnan(oll) and (np.isnan(oll) or ll[u] == oll or abs(ll[u] - oll) <= 1e-9 * abs(oll)), \
            "loglik " + tag
    if D > 1 and not ctx.model_is_diag():
        st = ctx.bw_accumulate(bad, lens=lens)
        assert np.isnan(st[:st.size - 2]).all() if inside else not np.isnan(st).any(), "stats with a bad feature " + tag
