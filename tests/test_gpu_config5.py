"""GPU, BASELINE.json configs[4] ("config 5"): Baum-Welch re-estimation at N=256 states, M=32 mixtures,
D=39 -- the multi-utterance statistics pass (gmmhmm.py:69-156 + hmm.py:63-131,174-200 restated per
utterance and summed) and the M-step every rank runs on the all-reduced statistics.

(a) the kernel combination of that shape (on-chip 256-state trellises, 32-mixture emission as two virtual
    16-mixture states, two-tile mixture moments, xi GEMM) against the oracle at a size the oracle finishes
    in seconds;
(b) the full size (T=2000, 64 utterances per GPU) through size-independent properties.
"""
import numpy as np
import pytest

from conftest import assert_close, env_switch, tuned

pytestmark = pytest.mark.gpu

N, M, D = 256, 32, 39


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def _split(st):
    o, out = 0, {}
    for key, n in (("xi_sum", N * N), ("gamma_sum", N), ("gamma0", N), ("mix", N * M), ("mix_x", N * M * D),
                   ("mix_xx", N * M * D), ("loglik", 1), ("n_utt", 1), ("n_frames", 1)):
        out[key] = st[o:o + n]
        o += n
    assert o == st.size
    return out


def test_config5_statistics_against_oracle(synth, oracle, ctx):
    mdl = synth.synth_model(N, M, D, seed=5)
    lens = np.array([24, 17, 9], dtype=np.int32)
    U, T = 3, 24
    obs = synth.synth_obs(U, T, D, seed=505)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    st = ctx.bw_accumulate(obs, lens=lens)
    with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
        ref = oracle.bw_stats_diag([obs[u, :lens[u]] for u in range(U)], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"],
                                   mdl["var"])
    s = _split(st)
    for key in ("xi_sum", "gamma_sum", "gamma0", "mix", "mix_x", "mix_xx"):
        assert_close(s[key], np.asarray(ref[key]).ravel(), 1e-9, 1e-12, msg=key)
    assert abs(s["loglik"][0] - ref["loglik"]) <= 1e-10 * abs(ref["loglik"])
    assert s["n_utt"][0] == U and s["n_frames"][0] == lens.sum()
    m = ctx.bw_mstep(st)
    mo = oracle.mstep_from_stats(ref)
    for k in ("A", "pi", "w", "mu", "var"):
        assert_close(m[k], mo[k], 1e-8, 1e-10, msg=k)
    # U = 1 is the reference's single-sequence E-step (hmm.py:174-200): xi / gamma sums of the oracle's baumwelch
    st1 = _split(ctx.bw_accumulate(obs[:1]))
    b, _ = oracle.gmm_map_b_diag(obs[0], mdl["w"], mdl["mu"], mdl["var"])
    alpha, beta = oracle.forward(mdl["A"], mdl["pi"], b), oracle.backward(mdl["A"], b)
    xi, gamma = oracle.baumwelch(mdl["A"], b, alpha, beta)
    assert_close(st1["xi_sum"], np.exp(xi[:-1]).sum(axis=0).ravel(), 1e-9, 1e-13, msg="xi")
    assert_close(st1["gamma0"], np.exp(gamma[0]), 1e-9, 1e-13, msg="gamma0")


def test_config5_full_size_properties(synth, ctx):
    U, T = 64, 2000
    mdl = synth.synth_model(N, M, D, seed=1)
    obs = synth.synth_obs(U, T, D, seed=1000)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    d_obs = ctx.to_device(obs)
    st = ctx.bw_accumulate(d_obs)
    s = _split(st)
    assert np.isfinite(st).all()
    assert s["n_utt"][0] == U and s["n_frames"][0] == U * T
    # posteriors are normalised per frame: gamma covers frames 0..T-2 of every utterance, the mixture
    # posteriors (alpha + beta based, gmmhmm.py:98-100) every frame
    assert abs(s["gamma_sum"].sum() - U * (T - 1)) < 1e-7 * U * T
    assert abs(s["xi_sum"].sum() - U * (T - 1)) < 1e-7 * U * T
    assert_close(s["xi_sum"].reshape(N, N).sum(axis=1), s["gamma_sum"], 1e-9, msg="sum_j xi = gamma")
    assert abs(s["gamma0"].sum() - U) < 1e-9 * U
    assert abs(s["mix"].sum() - U * T) < 1e-7 * U * T
    # sum_m mix[j, m] = sum over ALL frames of the state posterior = gamma_sum[j] + the last frame's posterior
    b = ctx.gmm_logprob(d_obs)
    alpha, ll = ctx.forward(b)
    ll = ll.to_host()
    last = np.exp(alpha.to_host()[:, T - 1, :] - ll[:, None]).sum(axis=0)
    assert_close(s["mix"].reshape(N, M).sum(axis=1), s["gamma_sum"] + last, 1e-9, msg="sum_m mix = sum_t posterior")
    # the log-likelihood in the statistics is calc_prob's
    cp = ctx.calc_prob(d_obs).to_host()
    assert_close(cp, ll, 1e-13)
    assert abs(s["loglik"][0] - cp.sum()) <= 1e-12 * abs(cp.sum())
    # additive over batch splits (what the all-reduce over ranks relies on)
    st_a = ctx.bw_accumulate(obs[:40])
    st_b = ctx.bw_accumulate(obs[40:])
    assert_close(st_a + st_b, st, 1e-10, 1e-9)
    # a second run over the same batch reproduces the statistics (split-K sums meet in fp64 atomics, so the
    # order of the last additions may differ: equal to rounding, not necessarily to the bit)
    st2 = ctx.bw_accumulate(d_obs)
    assert_close(st2, st, 1e-12, 1e-12)
    # the M-step yields normalised rows and the variance floor
    m = ctx.bw_mstep(st)
    # (log-likelihoods of 2000-frame utterances are ~ -1e5, one ulp of which is 1.5e-11: the posteriors
    # exp(alpha + beta - loglik) carry that much relative error, and so do sums that should be exactly 1)
    assert np.abs(np.exp(m["A"]).sum(axis=1) - 1).max() < 1e-12
    assert np.abs(np.exp(m["w"]).sum(axis=1) - 1).max() < 1e-12
    assert abs(np.exp(m["pi"]).sum() - 1) < 1e-9
    assert (m["var"] >= 0.01).all() and np.isfinite(m["mu"]).all()
    # weighted mean of the new means = weighted mean of the data, per state (first moments are consistent)
    tot_x = (s["mix_x"].reshape(N, M, D)).sum(axis=(0, 1))
    assert_close(tot_x, obs.reshape(-1, D).sum(axis=0), 1e-8, 1e-6, msg="sum of first moments = sum of data")


def _kernels_of(ctx, fn):
    ctx.prof_enable(True)
    ctx.prof_reset()
    out = fn()
    names = set(ctx.prof_report())
    ctx.prof_enable(False)
    return out, names


def _rel(a, b):
    """max |a - b| relative to the largest magnitude of b (statistics of one kind share a scale)."""
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_config5_float32_pass_against_oracle(synth, oracle, ctx):
    """`precision=np.float32` (config 5: "fp32 match to numpy"): component log-densities and posterior-weighted
    moments on the fp32 matrix cores, float hand-off between them; statistics and the M-step agree with the fp64
    oracle to float32 accuracy."""
    mdl = synth.synth_model(N, M, D, seed=5)
    lens = np.array([24, 17, 9], dtype=np.int32)
    U, T = 3, 24
    obs = synth.synth_obs(U, T, D, seed=505)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    st, names = _kernels_of(ctx, lambda: ctx.bw_accumulate(obs, lens=lens, precision=np.float32))
    if tuned():
        assert "emit_diag_mfma_f32" in names and "mix_mfma_f32" in names and "emit_diag_mfma_bm" not in names
    with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
        ref = oracle.bw_stats_diag([obs[u, :lens[u]] for u in range(U)], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"],
                                   mdl["var"])
    s = _split(st)
    for key in ("xi_sum", "gamma_sum", "gamma0", "mix", "mix_x", "mix_xx"):
        assert _rel(s[key], np.asarray(ref[key]).ravel()) < 2e-4, key
    assert abs(s["loglik"][0] - ref["loglik"]) <= 1e-6 * abs(ref["loglik"])
    m = ctx.bw_mstep(st)
    mo = oracle.mstep_from_stats(ref)
    assert _rel(np.exp(m["A"]), np.exp(mo["A"])) < 2e-4 and _rel(np.exp(m["w"]), np.exp(mo["w"])) < 2e-4
    assert np.abs(m["mu"] - mo["mu"]).max() < 2e-3 and _rel(m["var"], mo["var"]) < 2e-3
    # and against the fp64 pass of the same library: float32 accuracy, not more
    st64 = _split(ctx.bw_accumulate(obs, lens=lens))
    for key in ("mix", "mix_x", "mix_xx", "xi_sum"):
        assert _rel(s[key], st64[key]) < 2e-4 and (_rel(s[key], st64[key]) > 0 or not tuned()), key


@pytest.mark.parametrize("n,m,d", [(16, 4, 13), (8, 8, 39), (12, 16, 20), (4, 32, 5), (20, 16, 39)])
def test_float32_pass_shapes(n, m, d, synth, oracle, ctx):
    """Every tile path of the fp32 kernels: 4 / 8 / 16 / 32 mixtures, one to five feature tiles, ragged lengths,
    a dead component; and the models it must refuse (padding needed, ill-conditioned, full covariance) take the
    fp64 pass."""
    mdl = synth.synth_model(n, m, d, seed=n + m)
    w = mdl["w"].copy()
    w[1, 0] = -np.inf
    w[1] -= oracle.lse(w[1])
    U, T = 3, 70
    obs = synth.synth_obs(U, T, d, seed=9)
    lens = np.array([T, T // 2, 1], dtype=np.int32)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], w, mdl["mu"], mdl["var"], diag=True)
    st, names = _kernels_of(ctx, lambda: ctx.bw_accumulate(obs, lens=lens, precision=np.float32))
    assert "mix_mfma_f32" in names or not tuned()
    with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
        ref = oracle.bw_stats_diag([obs[u, :lens[u]] for u in range(U)], mdl["A"], mdl["pi"], w, mdl["mu"], mdl["var"])
    o = 0
    for key, cnt in (("xi_sum", n * n), ("gamma_sum", n), ("gamma0", n), ("mix", n * m), ("mix_x", n * m * d),
                     ("mix_xx", n * m * d)):
        assert _rel(st[o:o + cnt], np.asarray(ref[key]).ravel()) < 2e-4, key
        o += cnt
    assert st[n * n + 2 * n + m] == 0.0                    # the dead component's mass: exactly zero
    if (n, m) == (16, 4):
        # N = 15 needs padding: fp64 pass, silently (more precise than asked for)
        m15 = synth.synth_model(15, m, d, seed=1)
        ctx.upload_gmmhmm(m15["A"], m15["pi"], m15["w"], m15["mu"], m15["var"], diag=True)
        _, names = _kernels_of(ctx, lambda: ctx.bw_accumulate(obs, lens=lens, precision=np.float32))
        assert "mix_mfma_f32" not in names and ("mix_mfma" in names or not tuned())


def test_train_batch_float32_stops_on_eps(synth, ctx):
    """under precision=np.float32 both likelihoods of the stopping rule (hmm.py:301-313) must come from the
    same arithmetic -- with L(lambda|O) from the float32 pass and L(lambda_new|O) from an fp64 scoring pass their
    difference has a floor of ~1e-6 |L| and a converged model never stops on eps.  Now the loop stops, the last
    iteration's scoring pass (lnasr_calc_prob_p) agrees with what the statistics pass reports for the same model, and
    the float32 run follows the fp64 run."""
    from lnasr.gmmhmm import GMMHMM
    n, m, d, U, T = 8, 4, 6, 8, 40
    mdl = synth.synth_model(n, m, d, seed=21)
    obs = synth.synth_obs(U, T, d, seed=22) * 1.3 + 0.2
    runs = {}
    for prec in (np.float64, np.float32):
        gh = GMMHMM(n, m, d, mdl["A"].copy(), mdl["pi"].copy(), mdl["w"].copy(), mdl["mu"].copy(), mdl["var"].copy(),
                    precision=prec, ctx=ctx)
        hist = gh.train_batch(obs, iters=400, eps=0.05, verbose=False)
        assert 5 < len(hist) < 400, (prec, len(hist))                       # stopped on eps, not on the iteration cap
        assert abs(hist[-1][1] - hist[-1][0]) < 0.05 and abs(hist[-2][1] - hist[-2][0]) >= 0.05
        runs[prec] = (gh, hist)
        # the scoring pass of a LAST iteration = the statistics pass's likelihood of the same model
        gh.upload()
        ll_stats = float(np.asarray(gh.accumulate(obs))[-3])
        ll_score = float(np.asarray(ctx.calc_prob(obs, precision=prec)).sum())
        assert abs(ll_stats - ll_score) <= 1e-11 * abs(ll_stats), prec
    h64, h32 = runs[np.float64][1], runs[np.float32][1]
    assert abs(len(h64) - len(h32)) <= 3
    k = min(len(h64), len(h32)) - 1
    assert abs(h64[k][1] - h32[k][1]) < 2e-4 * abs(h64[k][1])


def test_float32_pass_conditioning_gate(synth, oracle, ctx):
    """the float32 expanded quadratic cancels terms of size kappa with a float's 6e-8, so the fp32 pass has
    its own gate (kappa <= 128); a model beyond it (here: the unit-scale model with its means spread 10x wider, kappa
    in the hundreds, still far below the fp64 gate of 4096) takes the fp64 kernels -- more precise than asked for,
    never less -- and a moderately conditioned one (5x, kappa ~ 50-100) stays within the documented 2e-4."""
    n, m, d, U, T = 16, 8, 39, 3, 60
    base = synth.synth_model(n, m, d, seed=31)
    obs0 = synth.synth_obs(U, T, d, seed=32)
    for scale, want_f32 in ((5.0, True), (10.0, False)):
        mu = base["mu"] * scale
        rng = np.random.default_rng(int(scale))
        obs = obs0 + mu[rng.integers(0, n, (U, T)), rng.integers(0, m, (U, T))]      # frames near the component means
        ctx.upload_gmmhmm(base["A"], base["pi"], base["w"], mu, base["var"], diag=True)
        kappa, direct = ctx.model_conditioning()
        assert not direct and ((kappa <= 128.0) == want_f32), (scale, kappa)
        st, names = _kernels_of(ctx, lambda: ctx.bw_accumulate(obs, precision=np.float32))
        assert (("mix_mfma_f32" in names) == want_f32) or not tuned(), (scale, kappa, sorted(names))
        with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
            ref = oracle.bw_stats_diag(list(obs), base["A"], base["pi"], base["w"], mu, base["var"])
        o = 0
        for key, cnt in (("xi_sum", n * n), ("gamma_sum", n), ("gamma0", n), ("mix", n * m), ("mix_x", n * m * d),
                         ("mix_xx", n * m * d)):
            assert _rel(st[o:o + cnt], np.asarray(ref[key]).ravel()) < (2e-4 if want_f32 else 1e-9), (scale, key)
            o += cnt


def test_float32_pass_full_size_and_mirror(synth, ctx):
    """Config 5 at full size under precision=float32: statistics within float32 accuracy of the fp64 pass; the
    mirror class routes `accumulate` by its `precision=`."""
    from lnasr.gmmhmm import GMMHMM
    U, T = 64, 2000
    mdl = synth.synth_model(N, M, D, seed=1)
    obs = synth.synth_obs(U, T, D, seed=1000)
    si = np.zeros((N, M, D, D))
    si[:, :, np.arange(D), np.arange(D)] = mdl["var"]
    gh32 = GMMHMM(N, M, D, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], si, precision=np.float32, ctx=ctx)
    d_obs = ctx.to_device(obs)
    gh32.upload()
    st32, names = _kernels_of(ctx, lambda: gh32.accumulate(d_obs))
    # (at this size the emission is the integer bank's four-plane float hand-off, csrc/gmm_i8.hip: same b + q32)
    assert ("emit_diag_i8_q32_m32" in names and "mix_mfma_f32" in names) or not tuned()
    st64 = ctx.bw_accumulate(d_obs)
    a, b = _split(st32), _split(st64)
    for key in ("xi_sum", "gamma_sum", "gamma0", "mix", "mix_x", "mix_xx"):
        assert _rel(a[key], b[key]) < 1e-3, key
    assert abs(a["loglik"][0] - b["loglik"][0]) < 1e-6 * abs(b["loglik"][0])
    assert abs(a["mix"].sum() - U * T) < 1e-4 * U * T


@pytest.mark.parametrize("prec", [np.float64, np.float32])
def test_train_batch_em_iterations(prec, synth, oracle, ctx, capsys):
    """`GMMHMM.train_batch` (the config-5 loop as one call: accumulate -> [all-reduce] -> M-step -> new likelihood):
    the first M-step is the oracle's on the summed statistics, the likelihood never decreases (EM), the printed line
    has the reference's format, and a one-rank communicator changes nothing."""
    from lnasr.gmmhmm import GMMHMM
    n, m, d, U, T = 16, 4, 13, 6, 50
    mdl = synth.synth_model(n, m, d, seed=21)
    obs = synth.synth_obs(U, T, d, seed=22) * 1.3 + 0.2
    lens = np.array([50, 31, 50, 12, 44, 50], dtype=np.int32)
    gh = GMMHMM(n, m, d, mdl["A"].copy(), mdl["pi"].copy(), mdl["w"].copy(), mdl["mu"].copy(), mdl["var"].copy(),
                precision=prec, ctx=ctx)
    hist = gh.train_batch(obs, lens=lens, iters=4, eps=0.0, verbose=True)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(hist) == 4 and len(out) == 4 and out[0].startswith("Iter:   0, L(lambda|O) = ")
    tol = 1e-9 if prec == np.float64 else 2e-4
    with np.errstate(divide="ignore", invalid="ignore", under="ignore"):
        ref = oracle.bw_stats_diag([obs[u, :lens[u]] for u in range(U)], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])
    assert abs(hist[0][0] - ref["loglik"]) <= max(tol, 1e-9) * abs(ref["loglik"])
    for k in range(4):
        assert hist[k][1] >= hist[k][0] - 1e-6 * abs(hist[k][0])                 # EM: never worse
        if k:
            # old of k IS new of k - 1: the E-step of iteration k produced both (one arithmetic, whatever the precision)
            assert hist[k][0] == hist[k - 1][1]
    # first M-step against the oracle
    gh1 = GMMHMM(n, m, d, mdl["A"].copy(), mdl["pi"].copy(), mdl["w"].copy(), mdl["mu"].copy(), mdl["var"].copy(),
                 precision=prec, ctx=ctx)

    class OneRank:                                            # what lnasr.dist.RcclComm does with one rank
        def allreduce_sum_f64(self, buf):
            return buf
    gh1.train_batch(obs, lens=lens, iters=1, verbose=False, comm=OneRank())
    mo = oracle.mstep_from_stats(ref)
    for name, got in (("A", gh1.A), ("w", gh1.w), ("mu", gh1.mu), ("var", gh1.si)):
        assert np.abs(np.asarray(got) - mo[name]).max() <= (1e-8 if prec == np.float64 else 5e-3), name
