"""GPU parity: forward / backward / Baum-Welch posteriors / M-steps / train loops / statistics,
against golden vectors from the reference and against the CPU oracle."""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import assert_close, env_switch, tuned

pytestmark = pytest.mark.gpu

# north_star tolerance: forward/backward log-likelihoods within 1e-4 relative; we hold 1e-11.
LL_RTOL = 1e-11


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def test_toy_hmm_full(golden):
    from lnasr.hmm import HMM
    g = golden("toy_hmm")
    h = HMM(2, 3, g["A"], g["B"], g["pi"])
    alpha = h._forward(g["obs"], True)
    beta = h._backward(g["obs"])
    assert_close(alpha, g["alpha"], 1e-14), assert_close(beta, g["beta"], 1e-14)
    assert_close(np.exp(alpha), [[.32, .02], [.0404, .069], [.023496, .005066]], 1e-12)
    xi, gamma = h._baumwelch(g["obs"], alpha, beta)
    assert_close(xi, g["xi"], 1e-13), assert_close(gamma, g["gamma"], 1e-13)
    assert np.isneginf(gamma[-1]).all() and np.isneginf(xi[-1]).all()
    est = h._estimate(g["obs"], alpha, beta, xi, gamma)
    assert_close(est["A"], g["est_A"], 1e-13), assert_close(est["B"], g["est_B"], 1e-13)
    assert_close(est["pi"], g["est_pi"], 1e-13)
    assert np.isneginf(est["B"][:, 1]).all()
    p = h.calc_prob(g["obs"])
    assert abs(p - g["calc_prob"]) < 1e-14 and abs(np.exp(p) - 0.028562) < 1e-9


def test_hmm_train_matches_reference_log(golden):
    from lnasr.hmm import HMM
    g = golden("hmm_train_4x5")
    h = HMM(4, 5, g["A0"], g["B0"], g["pi0"])
    alpha = h._forward(g["obs"], True)
    beta = h._backward(g["obs"])
    assert_close(alpha, g["alpha"], 1e-12), assert_close(beta, g["beta"], 1e-12)
    xi, gamma = h._baumwelch(g["obs"], alpha, beta)
    assert_close(xi, g["xi"], 1e-11, 1e-12), assert_close(gamma, g["gamma"], 1e-11, 1e-12)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        h.train(g["obs"], 3, 1e-4, True)
    assert buf.getvalue() == str(g["train3_log"])
    assert_close(h.A, g["train3_A"], 1e-10)
    # reset('random') consumes the global numpy RNG exactly like the reference
    np.random.seed(0)
    h2 = HMM(4, 5)
    h2.reset("random")
    assert np.array_equal(h2.A, g["A0"]) and np.array_equal(h2.B, g["B0"]) and np.array_equal(h2.pi, g["pi0"])
    h2.reset("bogus")                      # silent no-op, as in the reference
    assert np.array_equal(h2.A, g["A0"])


def test_discrete_mid(golden):
    from lnasr.hmm import HMM
    g = golden("hmm_discrete_16x24")
    h = HMM(16, 24, g["A"], g["B"], g["pi"])
    alpha = h._forward(g["obs"], True)
    beta = h._backward(g["obs"])
    assert_close(alpha, g["alpha"], 1e-12), assert_close(beta, g["beta"], 1e-12)
    xi, gamma = h._baumwelch(g["obs"], alpha, beta)
    assert_close(gamma, g["gamma"], 1e-11)
    assert_close(xi[0], g["xi_t0"], 1e-11), assert_close(xi[-2], g["xi_tlast"], 1e-11)
    est = h._estimate(g["obs"], alpha, beta, xi, gamma)
    assert_close(est["A"], g["est_A"], 1e-10), assert_close(est["B"], g["est_B"], 1e-10)
    assert_close(est["pi"], g["est_pi"], 1e-10)
    assert abs(h.calc_prob(g["obs"]) - g["calc_prob"][0]) < LL_RTOL * abs(g["calc_prob"][0])


@pytest.mark.parametrize("tag", ["", "_f32"])
def test_gmmhmm_small_bw(golden, tag):
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_small" + tag)
    prec = np.float32 if tag else np.float64
    tol = dict(rtol=3e-5, atol=3e-5) if tag else dict(rtol=1e-10, atol=1e-11)
    gh = GMMHMM(5, 3, 2, g["A"], g["pi"], g["w"], g["mu"], g["si"], precision=prec)
    gh._map_b(g["obs"])
    gh.b, gh.bm = g["b"], g["bm"]                       # trellises on the golden emissions
    alpha = gh._forward(g["obs"])
    beta = gh._backward(g["obs"])
    assert alpha.dtype == prec
    assert_close(alpha, g["alpha"], **tol), assert_close(beta, g["beta"], **tol)
    xi, gamma = gh._baumwelch(g["obs"], g["alpha"], g["beta"])
    assert_close(xi, g["xi"], **tol), assert_close(gamma, g["gamma"], **tol)
    est = gh._estimate(g["obs"], g["alpha"], g["beta"], g["xi"], g["gamma"])
    big = dict(rtol=50 * tol["rtol"], atol=50 * tol["atol"])
    for k in ("A", "pi", "w", "mu", "si"):
        assert_close(est[k], g["est_" + k], msg=k, **big)
    assert est["si"].shape == (5, 3, 2, 2)
    if not tag:
        gh = GMMHMM(5, 3, 2, g["A"], g["pi"], g["w"], g["mu"], g["si"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            gh.train(g["obs"], 3, 0.0, True)
        assert buf.getvalue() == str(g["train_log"])
        for k in ("A", "pi", "w", "mu", "si"):
            assert_close(getattr(gh, k), g["train3_" + k], 1e-8, 1e-10, msg=k)
        assert abs(gh.calc_prob(g["obs"]) - g["train3_loglik"][0]) < 1e-9
        assert np.array_equal(gh.decode(g["obs"]), g["train3_decode"])
        np.random.seed(0)
        _ = np.array(0.6 * np.random.random_sample((40, 2)) - 0.3)
        g2 = GMMHMM(5, 3, 2)
        g2.reset("random")
        assert np.array_equal(g2.A, g["A"]) and np.array_equal(g2.mu, g["mu"]) and np.array_equal(g2.si, g["si"])


def test_train_fused_equals_stepwise(synth):
    """GMMHMM.train takes the on-device E+M step when no step is overridden; a subclass that overrides one walks
    the reference's call sequence instead.  Same iterates either way (and a diagonal start still yields the
    reference's full Sigma)."""
    from lnasr.gmmhmm import GMMHMM

    class Stepwise(GMMHMM):
        def _estimate(self, obs, alpha, beta, xi, gamma):
            return super()._estimate(obs, alpha, beta, xi, gamma)

    mdl = synth.synth_model(6, 3, 5, seed=21)                # si is diagonal here
    obs = synth.synth_obs(1, 80, 5, seed=22)[0]
    a = GMMHMM(6, 3, 5, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    b = Stepwise(6, 3, 5, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    assert a._fused_train_ok() and not b._fused_train_ok()
    out_a, out_b = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out_a):
        a.train(obs, 3, 0.0, True)
    with contextlib.redirect_stdout(out_b):
        b.train(obs, 3, 0.0, True)
    assert out_a.getvalue() == out_b.getvalue() and out_a.getvalue().count("Iter:") == 3
    for k in ("A", "pi", "w", "mu", "si"):
        assert_close(getattr(a, k), getattr(b, k), 1e-8, 1e-10, msg=k)
    assert a.si.shape == (6, 3, 5, 5) and np.abs(a.si[0, 0] - np.diag(np.diag(a.si[0, 0]))).max() > 0
    assert a.b.shape == (6, 80) and a.bm.shape == (6, 3, 80)


@pytest.mark.parametrize("kind", ["diag", "full"])
def test_gmmhmm_d39_bw(golden, kind):
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_d39_" + kind)
    N, M, D = g["mu"].shape
    if kind == "diag":
        si = np.zeros((N, M, D, D))
        si[:, :, np.arange(D), np.arange(D)] = g["var"]
    else:
        si = g["si"]
    gh = GMMHMM(N, M, D, g["A"], g["pi"], g["w"], g["mu"], si)
    alpha = gh._forward(g["obs"], True)
    beta = gh._backward(g["obs"])
    assert_close(alpha, g["alpha"], 1e-12), assert_close(beta, g["beta"], 1e-12)
    ll = gh.calc_prob(g["obs"])
    assert abs(ll - g["loglik"][0]) < LL_RTOL * abs(ll)
    xi, gamma = gh._baumwelch(g["obs"], alpha, beta)
    assert_close(gamma, g["gamma"], 1e-10)
    est = gh._estimate(g["obs"], alpha, beta, xi, gamma)
    for k in ("A", "pi", "w", "mu", "si"):
        assert_close(est[k], g["est_" + k], 1e-9, 1e-11, msg=k)


def test_forward_backward_metric_shape(golden, synth, oracle, ctx):
    """N=128: log-likelihood of the first 32 frames against the reference; alpha/beta against the oracle;
    sum_j exp(alpha+beta) is the same likelihood at every frame (size-independent property)."""
    g = golden("gmmhmm_metric_slice")
    N, M, D, T, U = (int(x) for x in g["shape"])
    mdl = synth.synth_model(N, M, D, seed=1)
    obs = synth.synth_obs(U, T, D, seed=1000)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    ll32 = ctx.calc_prob(obs[:, :32])
    assert_close(ll32, g["loglik32"], LL_RTOL)
    b = ctx.gmm_logprob(obs[:, :200])
    alpha, ll = ctx.forward(b)
    beta = ctx.backward(b)
    ob = np.ascontiguousarray(b[0].T)
    assert_close(alpha[0], oracle.forward(mdl["A"], mdl["pi"], ob), 1e-12)
    assert_close(beta[0], oracle.backward(mdl["A"], ob), 1e-12)
    tot = oracle.lse(alpha + beta, axis=2)
    assert np.abs(tot - ll[:, None]).max() < 1e-8
    # ragged batch
    lens = np.array([200, 57], dtype=np.int32)
    _, ll_r = ctx.forward(b, lens=lens)
    assert abs(ll_r[0] - ll[0]) < 1e-9
    assert abs(ll_r[1] - oracle.loglik(oracle.forward(mdl["A"], mdl["pi"], np.ascontiguousarray(b[1, :57].T)))) < 1e-8


def test_forward_underflow_fallback(ctx, oracle):
    """Left-to-right model with huge emission gaps: the linear-domain sums underflow and the exact
    log-domain fallback must take over; -inf transitions stay -inf."""
    N, T = 6, 40
    A = np.full((N, N), -np.inf)
    for i in range(N):
        A[i, i] = np.log(0.6)
        if i + 1 < N:
            A[i, i + 1] = np.log(0.4)
    A[N - 1, N - 1] = 0.0
    pi = np.full(N, -np.inf)
    pi[0] = 0.0
    rng = np.random.default_rng(3)
    b = -2000.0 * rng.random((1, T, N))          # gaps of hundreds of nats between states
    b[0, 5, 2] = -np.inf
    ctx.upload_hmm(A, pi)
    alpha, ll = ctx.forward(b)
    beta = ctx.backward(b)
    ob = np.ascontiguousarray(b[0].T)
    with np.errstate(invalid="ignore"):
        ra, rb = oracle.forward(A, pi, ob), oracle.backward(A, ob)
    assert_close(alpha[0], ra, 1e-12), assert_close(beta[0], rb, 1e-12)
    assert np.isfinite(ll[0]) and abs(ll[0] - oracle.loglik(ra)) < 1e-9 * abs(ll[0])
    xi, gamma = ctx.baumwelch(b, alpha, beta)
    rxi, rga = oracle.baumwelch(A, ob, ra, rb)
    assert_close(gamma[0], rga, 1e-10, 1e-12), assert_close(xi[0], rxi, 1e-10, 1e-12)


def test_bw_statistics(golden, synth, oracle, ctx):
    """Multi-utterance sufficient statistics: U=1 reproduces the reference's single-sequence M-step
    (SURVEY 8e), U>1 equals the oracle's sums, and accumulation is additive over batch splits."""
    g = golden("gmmhmm_d39_diag")
    ctx.upload_gmmhmm(g["A"], g["pi"], g["w"], g["mu"], g["var"], diag=True)
    st = ctx.bw_accumulate(g["obs"][None])
    m = ctx.bw_mstep(st)
    D = g["mu"].shape[2]
    assert_close(m["A"], g["est_A"], 1e-9), assert_close(m["pi"], g["est_pi"], 1e-9)
    assert_close(m["w"], g["est_w"], 1e-9), assert_close(m["mu"], g["est_mu"], 1e-9)
    assert_close(m["var"], g["est_si"][:, :, np.arange(D), np.arange(D)], 1e-9)
    # U = 3 ragged, against the oracle's per-utterance sums
    mdl = synth.synth_model(8, 4, 6, seed=4)
    obs = synth.synth_obs(3, 50, 6, seed=55)
    lens = np.array([50, 21, 37], dtype=np.int32)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    st = ctx.bw_accumulate(obs, lens=lens)
    ref = oracle.bw_stats_diag([obs[u, :lens[u]] for u in range(3)], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])
    N, M, Dd = 8, 4, 6
    o = 0
    for key, n in (("xi_sum", N * N), ("gamma_sum", N), ("gamma0", N), ("mix", N * M), ("mix_x", N * M * Dd),
                   ("mix_xx", N * M * Dd)):
        assert_close(st[o:o + n], np.asarray(ref[key]).ravel(), 1e-9, 1e-12, msg=key)
        o += n
    assert abs(st[o] - ref["loglik"]) < 1e-9 * abs(ref["loglik"]) and st[o + 1] == 3 and st[o + 2] == lens.sum()
    st_a = ctx.bw_accumulate(obs[:2], lens=lens[:2])
    st_b = ctx.bw_accumulate(obs[2:], lens=lens[2:])
    assert_close(st_a + st_b, st, 1e-10, 1e-13)
    m2 = ctx.bw_mstep(st)
    mo = oracle.mstep_from_stats(ref)
    for k in ("A", "pi", "w", "mu", "var"):
        assert_close(m2[k], mo[k], 1e-9, 1e-12, msg=k)


def _check_stats(st, ref, N, M, D, rtol=1e-9, atol=1e-12):
    o = 0
    for key, n in (("xi_sum", N * N), ("gamma_sum", N), ("gamma0", N), ("mix", N * M), ("mix_x", N * M * D),
                   ("mix_xx", N * M * D)):
        assert_close(st[o:o + n], np.asarray(ref[key]).ravel(), rtol, atol, msg=key)
        o += n
    assert abs(st[o] - ref["loglik"]) <= 1e-9 * abs(ref["loglik"])


def test_bw_statistics_extreme_range(oracle, ctx):
    """A frame whose xi needs more than e^690 of dynamic range (a state hopeless given the past but forced by
    the future) cannot be factored into exp(alpha - c) * exp(beta + c); the statistics kernels must notice
    and take those frames through the exact log-domain path instead of dropping the term."""
    with np.errstate(divide="ignore"):
        A = np.log(np.array([[0.5, 0.5], [0.0, 1.0]]))
        pi = np.log(np.array([0.5, 0.5]))
        w = np.log(np.ones((2, 1)))
    mu = np.array([[[0.0]], [[40.0]]])
    var = np.ones((2, 1, 1))
    # o_0 = 40 makes state 0 e^-800 less likely than state 1, but o_1 = 0 can only come from state 0,
    # and state 1 is absorbing: the path 0 -> 0 carries a third of the mass
    obs = [np.array([[40.0], [0.0], [0.1], [39.5], [40.2]]), np.array([[0.3], [40.0], [39.0]])]
    ctx.upload_gmmhmm(A, pi, w, mu, var, diag=True)
    T = 5
    batch = np.zeros((2, T, 1))
    lens = np.array([5, 3], dtype=np.int32)
    for u, o in enumerate(obs):
        batch[u, :len(o)] = o
    st = ctx.bw_accumulate(batch, lens=lens)
    with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
        ref = oracle.bw_stats_diag(obs, A, pi, w, mu, var)
    assert ref["xi_sum"][0, 0] > 0.3                      # the term in question is not negligible
    _check_stats(st, ref, 2, 1, 1)


def test_bw_statistics_wide(synth, oracle, ctx):
    """Shapes that exercise every tile path of the matrix-core statistics kernels: N not a multiple of 8 or
    16, M = 17..32 (two mixture tiles), D = 39 (five feature tiles), ragged lengths, -inf weights."""
    for N, M, D, U, T, seed in ((19, 20, 39, 3, 70, 1), (130, 3, 5, 2, 40, 2), (8, 32, 13, 4, 33, 3), (5, 2, 20, 2, 37, 4),
                                (3, 17, 28, 1, 66, 5), (2, 2, 1, 2, 9, 6), (6, 48, 13, 2, 31, 7),
                                (3, 70, 5, 2, 40, 8), (9, 33, 39, 1, 21, 9), (260, 2, 3, 2, 5, 10)):
        mdl = synth.synth_model(N, M, D, seed=seed)
        w = mdl["w"].copy()
        w[1, 0] = -np.inf                                   # a dead component
        w[1] -= oracle.lse(w[1])
        obs = synth.synth_obs(U, T, D, seed=60 + seed)
        lens = np.array([T, T // 2, 1, T - 3][:U], dtype=np.int32)
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], w, mdl["mu"], mdl["var"], diag=True)
        st = ctx.bw_accumulate(obs, lens=lens)
        with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
            ref = oracle.bw_stats_diag([obs[u, :lens[u]] for u in range(U)], mdl["A"], mdl["pi"], w, mdl["mu"],
                                       mdl["var"])
        _check_stats(st, ref, N, M, D, 1e-9, 1e-11)
        assert st[N * N + 2 * N + M] == 0.0                 # mix[1, 0]: exact zero stays exact


@pytest.mark.parametrize("route", ["frame-parallel", "streaming"])
def test_trellises_beyond_256_states(route, synth, oracle, ctx):
    """N > 256: the frame-parallel kernels (one launch per frame over the batch; the default for small batches) and
    the per-utterance streaming kernels (large batches; forced here with LNASR_NO_FRAMEPAR) against the oracle."""
    rng = np.random.default_rng(23)
    with env_switch(**({"LNASR_NO_FRAMEPAR": 1} if route == "streaming" else {})):
        for N in (257, 520):
            d = synth.synth_discrete(N, 4, 8, seed=N)
            A = d["A"].copy()
            A[3, :] = -np.inf
            A[3, 5] = 0.0
            A[:, 7] = -np.inf
            ctx.upload_hmm(A, d["pi"])
            T = 11
            b = np.ascontiguousarray(-30.0 * rng.random((4, T, N)))
            lens = np.array([T, 1, 6, 2], dtype=np.int32)
            alpha, ll = ctx.forward(b, lens=lens)
            _, ll_only = ctx.forward(b, want_alpha=False, lens=lens)
            beta = ctx.backward(b, lens=lens)
            r = ctx.viterbi(b, lens=lens)
            for u in range(4):
                L = lens[u]
                ob = np.ascontiguousarray(b[u, :L].T)
                with np.errstate(invalid="ignore"):
                    ra, rb = oracle.forward(A, d["pi"], ob), oracle.backward(A, ob)
                assert_close(alpha[u, :L], ra, 1e-12, msg="alpha N=%d u=%d" % (N, u))
                assert_close(beta[u, :L], rb, 1e-12, msg="beta N=%d u=%d" % (N, u))
                assert abs(ll[u] - oracle.loglik(ra)) <= 1e-11 * abs(ll[u]) and ll_only[u] == ll[u]
                v, bt, path = oracle.viterbi(A, d["pi"], ob)
                assert np.array_equal(r["v"][u, :L], v) and np.array_equal(r["path"][u, :L], path)
                assert np.array_equal(r["bt"][u, :L], bt)


@pytest.mark.parametrize("N", [512, 1000, 1024])
def test_trellises_beyond_256_states_large_batch(N, synth, oracle, ctx):
    """N > 256 with a batch large enough for the blocked tiers (several utterances per streaming block -- two when
    four do not fit the LDS, N > 960 -- and the wide-tile frame-parallel Viterbi): a sample of utterances against
    the oracle, the rest against the same utterance decoded in a small batch."""
    rng = np.random.default_rng(N)
    d = synth.synth_discrete(N, 4, 8, seed=N)
    ctx.upload_hmm(d["A"], d["pi"])
    U, T = 600, 4
    b = np.ascontiguousarray(-30.0 * rng.random((U, T, N)))
    lens = rng.integers(1, T + 1, U).astype(np.int32)
    lens[0] = T
    for maxu in ("256", "4096"):                          # streaming, then frame-parallel for this batch
        with env_switch(LNASR_FRAMEPAR_MAXU=maxu):
            alpha, ll = ctx.forward(b, lens=lens)
            beta = ctx.backward(b, lens=lens)
            r = ctx.viterbi(b, lens=lens, want_bt=False)
        for u in (0, 1, 2, 3, 298, 597, 598, 599):
            L = lens[u]
            ob = np.ascontiguousarray(b[u, :L].T)
            ra, rb = oracle.forward(d["A"], d["pi"], ob), oracle.backward(d["A"], ob)
            assert_close(alpha[u, :L], ra, 1e-12, msg="alpha N=%d u=%d" % (N, u))
            assert_close(beta[u, :L], rb, 1e-12, msg="beta N=%d u=%d" % (N, u))
            assert abs(ll[u] - oracle.loglik(ra)) <= 1e-11 * abs(ll[u])
            v, _, path = oracle.viterbi(d["A"], d["pi"], ob)
            assert np.array_equal(r["v"][u, :L], v) and np.array_equal(r["path"][u, :L], path)
        sub = slice(100, 104)                              # small batch: the per-utterance tier
        a2, ll2 = ctx.forward(b[sub], lens=lens[sub])
        r2 = ctx.viterbi(b[sub], lens=lens[sub], want_bt=False)
        for k, u in enumerate(range(100, 104)):
            L = lens[u]
            assert_close(alpha[u, :L], a2[k, :L], 1e-12, msg="alpha vs small batch u=%d" % u)
            assert np.array_equal(r["v"][u, :L], r2["v"][k, :L]) and np.array_equal(r["path"][u, :L], r2["path"][k, :L])


@pytest.mark.parametrize("N", [65, 100, 128])
def test_forward_backward_column_owned_kernel(N, synth, oracle, ctx):
    """64 < N <= 128: the column-owned forward / backward kernels (row-group sums exchanged with permlane swaps) --
    two utterances per four-wave block (batches), one per block and the chain-shortened kernel (small batches) -- against the oracle on
    a sample of utterances and against the row-sliced kernel (LNASR_NO_COLFWD) on all of them; an odd number of
    utterances (a half-empty last block), ragged lengths, a forbidden transition, an unreachable state, an utterance
    whose likelihood underflows the linear domain (exact fallback)."""
    rng = np.random.default_rng(N)
    d = synth.synth_discrete(N, 4, 8, seed=N)
    A = d["A"].copy()
    A[3, :] = -np.inf
    A[3, 5] = 0.0
    A[:, 7] = -np.inf
    ctx.upload_hmm(A, d["pi"])
    U, T = 4 * ctx.device_info()["cu_count"] + 7, 7
    b = np.ascontiguousarray(-30.0 * rng.random((U, T, N)))
    b[5, 3] -= 2000.0                                      # a frame far below the shift: column sums underflow
    b[6, 2, : N // 2] = -np.inf
    lens = rng.integers(1, T + 1, U).astype(np.int32)
    lens[:8] = T
    alpha, ll = ctx.forward(b, lens=lens)
    _, ll_only = ctx.forward(b, want_alpha=False, lens=lens)
    beta = ctx.backward(b, lens=lens)
    with env_switch(LNASR_NO_COLFWD=1):
        alpha2, ll2 = ctx.forward(b, lens=lens)
        beta2 = ctx.backward(b, lens=lens)
    with env_switch(LNASR_FWD_ONE_MAXU=1 << 20, LNASR_FWD_LAT_MAXU=0):   # ONE utterance per four-wave block: batches up to 2 x CUs (round 5)
        alpha4, ll4 = ctx.forward(b, lens=lens)
        beta4 = ctx.backward(b, lens=lens)
    with env_switch(LNASR_FWD_LAT_MAXU=1 << 20):           # log / exp off the frame's chain (fwdbwd_lat4): batches up to the CU count
        alpha6, ll6 = ctx.forward(b, lens=lens)
        beta6 = ctx.backward(b, lens=lens)
    for u in range(U):                                     # e = sum * G instead of exp(log(sum) + ...): one rounding apart
        L = lens[u]
        assert_close(alpha6[u, :L], alpha[u, :L], 1e-13, msg="alpha, chain-shortened kernel u=%d" % u)
        assert_close(beta6[u, :L], beta[u, :L], 1e-13, msg="beta, chain-shortened kernel u=%d" % u)
    assert_close(ll6, ll, 1e-13)
    with env_switch(LNASR_FWD_ONE_MAXU=0, LNASR_FWD_LAT_MAXU=0):   # ... and two per block for a batch that would take one
        alpha5, ll5 = ctx.forward(b[:9], lens=lens[:9])
        beta5 = ctx.backward(b[:9], lens=lens[:9])
    assert np.array_equal(ll_only, ll)
    # same products, same swap network, same order of additions: the column-owned kernels agree to the bit
    for u in range(U):
        L = lens[u]
        assert np.array_equal(alpha[u, :L], alpha4[u, :L]) and np.array_equal(beta[u, :L], beta4[u, :L]), u
    for u in range(9):
        L = lens[u]
        assert np.array_equal(alpha[u, :L], alpha5[u, :L]) and np.array_equal(beta[u, :L], beta5[u, :L]), u
    assert np.array_equal(ll, ll4) and np.array_equal(ll[:9], ll5)
    for u in range(U):
        L = lens[u]
        assert_close(alpha[u, :L], alpha2[u, :L], 1e-12, msg="alpha vs row-sliced kernel u=%d" % u)
        assert_close(beta[u, :L], beta2[u, :L], 1e-12, msg="beta vs row-sliced kernel u=%d" % u)
    assert_close(ll, ll2, 1e-12)
    for u in (0, 1, 2, 3, 4, 5, 6, 7, U // 2, U - 3, U - 2, U - 1):
        L = lens[u]
        ob = np.ascontiguousarray(b[u, :L].T)
        with np.errstate(invalid="ignore", divide="ignore"):
            ra, rb = oracle.forward(A, d["pi"], ob), oracle.backward(A, ob)
        assert_close(alpha[u, :L], ra, 1e-12, msg="alpha N=%d u=%d" % (N, u))
        assert_close(beta[u, :L], rb, 1e-12, msg="beta N=%d u=%d" % (N, u))
        assert abs(ll[u] - oracle.loglik(ra)) <= 1e-11 * abs(ll[u])


def test_bw_statistics_full_covariance_wide(synth, ctx):
    """Full covariance with two mixture tiles (M = 20) and with more than 32 mixtures (groups of 32), N not a
    multiple of 8, ragged lengths: the matrix-core statistics kernel against the scalar kernel that the golden
    vectors pin (LNASR_NO_MFMA_STATS)."""
    for N, M, D, U, T in ((9, 20, 39, 3, 45), (3, 40, 13, 2, 30)):
        mdl = synth.synth_model(N, M, D, seed=12, full_cov=True)
        obs = synth.synth_obs(U, T, D, seed=77)
        lens = np.array([T, 17, T - 1][:U], dtype=np.int32)
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
        assert not ctx.model_is_diag()
        st = ctx.bw_accumulate(obs, lens=lens)
        with env_switch(LNASR_NO_MFMA_STATS=1):
            ref = ctx.bw_accumulate(obs, lens=lens)
        assert st.shape == ref.shape == (N * N + 2 * N + N * M + N * M * D + N * M * D * D + 3,)
        assert_close(st, ref, 1e-9, 1e-11)
        si = ctx.bw_mstep(st)["si"]
        assert np.allclose(si, np.swapaxes(si, -1, -2), rtol=0, atol=1e-12)      # symmetric second moments


def test_forward_backward_large_n(synth, oracle, ctx):
    """N > 128 (streaming kernels): alpha / beta / loglik / posteriors against the oracle, ragged batch."""
    rng = np.random.default_rng(17)
    for N in (129, 200, 256):
        d = synth.synth_discrete(N, 4, 8, seed=N)
        A = d["A"].copy()
        A[3, :] = -np.inf
        A[3, 5] = 0.0                                   # a deterministic transition
        A[:, 7] = -np.inf                               # an unreachable state
        ctx.upload_hmm(A, d["pi"])
        T = 21
        b = np.ascontiguousarray(-30.0 * rng.random((5, T, N)))
        lens = np.array([T, 1, 9, 2, 20], dtype=np.int32)
        alpha, ll = ctx.forward(b, lens=lens)
        beta = ctx.backward(b, lens=lens)
        r = ctx.viterbi(b, lens=lens)
        for u in range(5):
            L = lens[u]
            ob = np.ascontiguousarray(b[u, :L].T)
            with np.errstate(invalid="ignore"):
                ra, rb = oracle.forward(A, d["pi"], ob), oracle.backward(A, ob)
            assert_close(alpha[u, :L], ra, 1e-12, msg="alpha N=%d u=%d" % (N, u))
            assert_close(beta[u, :L], rb, 1e-12, msg="beta N=%d u=%d" % (N, u))
            assert abs(ll[u] - oracle.loglik(ra)) <= 1e-11 * abs(ll[u])
            v, bt, path = oracle.viterbi(A, d["pi"], ob)
            assert np.array_equal(r["v"][u, :L], v) and np.array_equal(r["path"][u, :L], path)
            assert np.array_equal(r["bt"][u, :L], bt)


def test_bw_statistics_full_covariance(golden, ctx):
    """Full-covariance statistics: U=1 reproduces the reference's M-step (old-mean Sigma + min_std I);
    accumulation is additive."""
    g = golden("gmmhmm_d39_full")
    ctx.upload_gmmhmm(g["A"], g["pi"], g["w"], g["mu"], g["si"])
    assert not ctx.model_is_diag()
    N, M, D = g["mu"].shape
    st = ctx.bw_accumulate(g["obs"][None])
    assert st.shape == (N * N + 2 * N + N * M + N * M * D + N * M * D * D + 3,)
    m = ctx.bw_mstep(st)
    assert_close(m["A"], g["est_A"], 1e-9), assert_close(m["pi"], g["est_pi"], 1e-9)
    assert_close(m["w"], g["est_w"], 1e-9), assert_close(m["mu"], g["est_mu"], 1e-9, 1e-12)
    assert m["si"].shape == (N, M, D, D)
    assert_close(m["si"], g["est_si"], 1e-9, 1e-12)
    two = np.stack([g["obs"], g["obs"][::-1].copy()])
    st2 = ctx.bw_accumulate(two)
    st_b = ctx.bw_accumulate(two[1:])
    assert_close(st + st_b, st2, 1e-10, 1e-12)


def test_onchip256_hard_inputs(synth, oracle, ctx):
    """The on-chip 129..256-state forward / backward against the oracle on ragged lengths, a forbidden transition and
    the thousand-nat emission gaps of the underflow test (sums leave the safe range: exact log-domain path)."""
    for N, T, U in ((256, 37, 3), (200, 20, 2), (129, 9, 4)):
        d = synth.synth_discrete(N, 4, 8, seed=N)
        A = d["A"].copy()
        A[3, :] = -np.inf
        A[3, 2] = 0.0
        rng = np.random.default_rng(N)
        b = -60.0 * rng.random((U, T, N))
        b[0, T // 2, : N // 2] -= 2000.0                     # sums leave the safe range: exact path
        lens = rng.integers(1, T + 1, U).astype(np.int32)
        lens[0] = T
        ctx.upload_hmm(A, d["pi"])
        a0, l0 = ctx.forward(b, lens=lens)
        be0 = ctx.backward(b, lens=lens)
        for u in range(U):
            L = int(lens[u])
            with np.errstate(divide="ignore", invalid="ignore"):
                ra, rb = oracle.forward(A, d["pi"], b[u, :L].T), oracle.backward(A, b[u, :L].T)
            assert_close(a0[u, :L], ra, 1e-11, 1e-11), assert_close(be0[u, :L], rb, 1e-11, 1e-11)
            assert_close(l0[u], oracle.loglik(ra), 1e-12)


def _ulps32(a, b):
    """Distance in float32 units in the last place (equal infinities: 0)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_float32_forward_backward_round_at_every_store(golden, synth, oracle):
    """precision=np.float32 (hmm.py:93-99, 124-130): the reference stores every alpha[t] / beta[t] into a float32 array
    and reads it back for the next frame.  The float-storage kernels do the same -- fp64 sum, one rounding at the store,
    recursion continued from the rounded value -- so alpha and beta equal the reference's float32 arrays bit for bit up
    to the rare value whose fp64 sum lies within ~1e-15 of a float rounding boundary (then one ulp, never more, and
    never the drift of a recursion that rounds only once at the end).  Checked against the reference's own arrays
    (goldens gmmhmm_f32_mid N=32 T=512, gmmhmm_small_f32) and, for the other kernel families, against the oracle run
    with precision=np.float32: N=128 (column-owned kernel), N=200 (on-chip kernel), N=300 (generic kernel)."""
    from lnasr import _ffi
    from lnasr.gmmhmm import GMMHMM
    ctx = _ffi.default_ctx()

    def check(got, want, what, max_off=2):
        assert got.dtype == np.float32 and got.shape == want.shape, what
        d = _ulps32(got, want)
        assert d.max() <= 1 and int((d > 0).sum()) <= max_off, (what, int(d.max()), int((d > 0).sum()), d.size)

    g = golden("gmmhmm_f32_mid")
    N, M, D, T = (int(x) for x in g["shape"])
    mdl = synth.synth_model(N, M, D, seed=int(g["seeds"][0]))
    obs = synth.synth_obs(1, T, D, seed=int(g["seeds"][1]))[0]
    gh = GMMHMM(N, M, D, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"], precision=np.float32, ctx=ctx)
    gh.b = g["b"]                                            # the reference's own float32 emissions
    check(gh._forward(obs), g["alpha"], "f32_mid alpha")
    check(gh._backward(obs), g["beta"], "f32_mid beta")
    ll = gh.ctx.forward(gh._b_dev(), want_alpha=False)[1][0]
    assert abs(ll - float(g["loglik"][0])) <= 2e-6 * abs(float(g["loglik"][0]))      # the reference folds this one in float32
    gs = golden("gmmhmm_small_f32")
    g2 = GMMHMM(5, 3, 2, gs["A"], gs["pi"], gs["w"], gs["mu"], gs["si"], precision=np.float32, ctx=ctx)
    g2.b = gs["b"]
    check(g2._forward(gs["obs"]), gs["alpha"], "small_f32 alpha", 0)
    check(g2._backward(gs["obs"]), gs["beta"], "small_f32 beta", 0)
    # the other families, against the oracle's float32 recursion; ragged batch, an unreachable state, a -inf emission
    for N, T, U in ((128, 300, 3), (64, 90, 2), (200, 260, 3), (256, 120, 2), (300, 40, 2)):
        d = synth.synth_discrete(N, 4, 8, seed=N)
        A = d["A"].copy()
        A[:, 3] = -np.inf
        rng = np.random.default_rng(N + 1)
        b = (-25.0 * rng.random((U, T, N))).astype(np.float32)
        b[0, T // 2, 5] = -np.inf
        lens = np.array([T, T - 7, T - 1][:U], dtype=np.int32)
        ctx.upload_hmm(A, d["pi"])
        al, _ = ctx.forward(b, lens=lens)
        be = ctx.backward(b, lens=lens)
        for u in range(U):
            L = int(lens[u])
            with np.errstate(divide="ignore", invalid="ignore"):
                oa = oracle.forward(A, d["pi"], b[u, :L].T, precision=np.float32)
                ob = oracle.backward(A, b[u, :L].T, precision=np.float32)
            check(al[u, :L], oa, "N=%d alpha u=%d" % (N, u), max_off=max(2, oa.size // 20000))
            check(be[u, :L], ob, "N=%d beta u=%d" % (N, u), max_off=max(2, ob.size // 20000))


def test_float32_reference_arrays_on_every_kernel_family(golden, synth):
    """The reference itself run with precision=np.float32 at N = 12 (one-wave kernels), N = 128 (register-resident Viterbi,
    column-owned forward / backward) and N = 200 (on-chip kernel), golden hmm_f32_families: v, bt and path bit for bit; alpha
    and beta bit for bit up to the rare value whose fp64 sum straddles a float rounding boundary (one ulp, at most two values).
    Through the mirror class, i.e. HMM(precision=np.float32)._forward / _backward / _viterbi on the resident float32 b."""
    from lnasr.hmm import HMM
    g = golden("hmm_f32_families")
    for N, T in g["shapes"]:
        N, T = int(N), int(T)
        d = synth.synth_discrete(N, 8, T, seed=40 + N)
        A = d["A"].copy()
        A[2, :] = -np.inf
        A[2, 5] = 0.0
        A[:, 7] = -np.inf
        tag = "n%d_" % N
        assert np.array_equal(d["obs"], g[tag + "obs"])
        h = HMM(N, 8, A, d["B"], d["pi"], precision=np.float32)
        alpha = h._forward(d["obs"], True)
        beta = h._backward(d["obs"])
        v, bt, path = h._viterbi(d["obs"])
        assert alpha.dtype == np.float32 and beta.dtype == np.float32 and v.dtype == np.float32 and h.b.dtype == np.float32
        assert np.array_equal(v, g[tag + "v"]) and np.array_equal(bt, g[tag + "bt"]) and np.array_equal(path, g[tag + "path"]), N
        for got, want, what in ((alpha, g[tag + "alpha"], "alpha"), (beta, g[tag + "beta"], "beta")):
            dd = _ulps32(got, want)
            assert dd.max() <= 1 and int((dd > 0).sum()) <= 2, (N, what, int(dd.max()), int((dd > 0).sum()))
        assert np.array_equal(h.decode(d["obs"]), g[tag + "path"])


def test_float32_train_tracks_the_reference(golden):
    """GMMHMM(precision=np.float32).train, three iterations (golden gmmhmm_small_f32, the imported reference): the float32 model
    walks the reference's call sequence (the fused step is float64 only); every printed likelihood agrees to float32 accuracy (the
    reference folds lse(alpha[-1]) in float32 arithmetic, np.logaddexp on float32 scalars: not reproducible bit for bit), the
    re-estimated parameters to 1e-5, the decoded path exactly."""
    import re
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_small_f32")
    gh = GMMHMM(5, 3, 2, g["A"], g["pi"], g["w"], g["mu"], g["si"], precision=np.float32)
    assert not gh._fused_train_ok()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gh.train(g["obs"], 3, 0.0, True)
    num = r"[-+]?\\d\\.\\d+e[-+]\\d+"
    got = [[float(x) for x in re.findall(num, line)] for line in buf.getvalue().splitlines()]
    want = [[float(x) for x in re.findall(num, line)] for line in str(g["train_log"]).splitlines()]
    assert len(got) == len(want) == 3
    for a, b in zip(got, want):
        assert_close(np.array(a), np.array(b), 3e-6)
    for k in ("A", "pi", "w", "mu", "si"):
        assert np.asarray(getattr(gh, k)).dtype == g["train3_" + k].dtype, k
        assert_close(getattr(gh, k), g["train3_" + k], 1e-5, 1e-5, msg=k)
    assert abs(gh.calc_prob(g["obs"]) - g["train3_loglik"][0]) < 1e-5 * abs(g["train3_loglik"][0])
    assert np.array_equal(gh.decode(g["obs"]), g["train3_decode"])


@pytest.mark.parametrize("N", [17, 64, 65, 100, 128, 129, 200, 256])
def test_back_pointer_chase_equals_the_argmax_backtrace(N, synth, oracle, ctx):
    """Small batches (round 5): every back-pointer first (bytes), then path[t-1] = bt[t][path[t]] as a chain of LDS reads
    (backtrace_chase_kernel) -- against the recomputing back-trace kernels (LNASR_CHASE_MAXU=0) on ragged batches (lengths 1, 2,
    a multiple of the chunk, one past it), ties, forbidden transitions and an unreachable state, both storages; and against the
    oracle."""
    rng = np.random.default_rng(N)
    d = synth.synth_discrete(N, 4, 8, seed=N)
    A = d["A"].copy()
    A[3, :] = -np.inf
    A[3, 5] = 0.0
    A[:, 7] = -np.inf
    A[1, :] = A[0, :]                                     # equal rows: ties between source states 0 and 1
    ctx.upload_hmm(A, d["pi"])
    U, T = 9, 300
    b = np.ascontiguousarray(-30.0 * rng.random((U, T, N)))
    b[:, :, 1] = b[:, :, 0]
    lens = np.array([T, 1, 2, 64, 65, 128, 129, 257, T - 1], dtype=np.int32)
    for dt in (np.float64, np.float32):
        bb = b.astype(dt)
        with env_switch(LNASR_CHASE_MAXU=1 << 20):
            r1 = ctx.viterbi(bb, want_v=True, want_bt=False, want_score=True, lens=lens)
        with env_switch(LNASR_CHASE_MAXU=0):
            r0 = ctx.viterbi(bb, want_v=True, want_bt=False, want_score=True, lens=lens)
        for u in range(U):
            L = lens[u]
            assert np.array_equal(r1["path"][u, :L], r0["path"][u, :L]), (N, dt, u)
            assert np.array_equal(r1["v"][u, :L], r0["v"][u, :L])
        assert np.array_equal(r1["score"], r0["score"])
        if dt == np.float64:
            for u in (0, 3, 7):
                L = lens[u]
                with np.errstate(invalid="ignore", divide="ignore"):
                    _, _, want = oracle.viterbi(A, d["pi"], np.ascontiguousarray(b[u, :L].T))
                assert np.array_equal(r1["path"][u, :L], want), (N, u)


@pytest.mark.parametrize("scale", [1.0, 20.0, 60.0])
def test_chain_shortened_forward_backward_at_any_emission_size(scale, synth, oracle, ctx):
    """fwdbwd_lat4 (one utterance per block, batches up to the CU count) expresses the e row of a frame in a shift it extrapolates
    from the maxima two and three frames back: emissions of -30, -600 and -1800 per frame (and one frame 4000 below its
    neighbours, where the extrapolation overshoots and the exact log-domain column takes over) against the oracle."""
    N, U, T = 128, 3, 40
    rng = np.random.default_rng(int(scale))
    d = synth.synth_discrete(N, 4, 8, seed=5)
    ctx.upload_hmm(d["A"], d["pi"])
    b = np.ascontiguousarray(-30.0 * scale * rng.random((U, T, N)))
    b[1, 17] -= 4000.0
    b[2, 5, : N // 2] = -np.inf
    alpha, ll = ctx.forward(b)
    beta = ctx.backward(b)
    for u in range(U):
        ob = np.ascontiguousarray(b[u].T)
        with np.errstate(invalid="ignore", divide="ignore"):
            ra, rb = oracle.forward(d["A"], d["pi"], ob), oracle.backward(d["A"], ob)
        assert_close(alpha[u], ra, 1e-12, msg="alpha scale=%g u=%d" % (scale, u))
        assert_close(beta[u], rb, 1e-12, msg="beta scale=%g u=%d" % (scale, u))
        assert_close(ll[u], oracle.loglik(ra), 1e-12)


@pytest.mark.parametrize("N,U", [(128, 3), (128, 600), (32, 5), (32, 600), (96, 300), (200, 4), (300, 3), (300, 40), (1100, 2)])
def test_emission_levels_that_jump_stay_in_the_linear_domain(N, U, synth, oracle, ctx):
    """Data a model does not fit (round 5, tools/mismatch.py): the row maximum of b around -2200 nats, jumping by +-800 from
    frame to frame.  The shift of a linear-domain row used to be the lagged log-domain maximum alone, every column of every
    frame then left the exponent range and was redone in the exact form -- right, and 65 times slower.  With the level of the
    emission row in the shift NO column does (lnasr_debug_fallback_columns counts them; every kernel family of the scaled
    recursion and both batch tiers); a model with an unreachable state still takes the exact form for that state's column,
    which shows that the counter counts."""
    T = 50
    rng = np.random.default_rng(U + N)
    d = synth.synth_discrete(N, 4, 8, seed=9)
    ctx.upload_hmm(d["A"], d["pi"])
    level = -2200.0 + 800.0 * rng.standard_normal((U, T, 1))
    b = np.ascontiguousarray(level - 40.0 * rng.random((U, T, N)))
    ctx.fallback_columns()
    alpha, ll = ctx.forward(b)
    beta = ctx.backward(b)
    if tuned():
        assert ctx.fallback_columns() == 0
    for u in (0, U // 2, U - 1):
        ob = np.ascontiguousarray(b[u].T)
        ra, rb = oracle.forward(d["A"], d["pi"], ob), oracle.backward(d["A"], ob)
        assert_close(alpha[u], ra, 1e-12, msg="alpha u=%d" % u)
        assert_close(beta[u], rb, 1e-12, msg="beta u=%d" % u)
        assert_close(ll[u], oracle.loglik(ra), 1e-12)
    A = d["A"].copy()
    A[:, 7] = -np.inf                                    # state 7 cannot be entered: its column sum is 0 in every frame
    ctx.upload_hmm(A, d["pi"])
    ctx.forward(b[:1])
    if tuned():
        assert ctx.fallback_columns() >= T - 1
