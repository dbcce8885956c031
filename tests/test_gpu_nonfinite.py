"""NaN and infinite data: the reference's propagation, reproduced (ln-asr_amd/csrc/nanfix.hip).

numpy's max / argmax / logaddexp carry a NaN through every later frame (hmm.py:97-99,162-171) and the reference's
product with the full inverse matrix turns a frame with a NaN OR an infinite feature into NaN densities (gmm.py:63-68).
The tuned kernels use IEEE maxNum; a repair pass rewrites their outputs to the reference's closed form.  Everything here
is compared with the golden `gmmhmm_nonfinite` (generated from the imported reference) or with the numpy oracle, NaN for
NaN and path for path."""
import numpy as np
import pytest

from conftest import env_switch, tuned

pytestmark = pytest.mark.gpu


def _eq_nan(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def _host(x):
    return x.to_host() if hasattr(x, "to_host") else np.asarray(x)


def test_reference_golden(golden, synth, ctx):
    g = golden("gmmhmm_nonfinite")
    N, M, D, T = (int(x) for x in g["shape"])
    mdl = synth.synth_model(N, M, D, seed=int(g["seeds"][0]))
    names = [str(x) for x in g["names"]]
    obs = np.stack([g[n + "_obs"] for n in names])                       # one utterance per case
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    path, score = ctx.decode(obs, want_score=True)
    ll = ctx.calc_prob(obs)
    for prec, tag, tol in ((np.float64, "", 1e-12), (np.float32, "_f32", 2e-6)):
        b, bm = ctx.gmm_logprob(obs, want_bm=True, precision=prec)
        for u, n in enumerate(names):
            k = n + tag
            gb, gbm = g[k + "_b"].T, g[k + "_bm"].transpose(2, 0, 1)     # [T,N], [T,N,M]
            assert np.array_equal(np.isnan(b[u]), np.isnan(gb)), k
            assert np.array_equal(np.isnan(bm[u]), np.isnan(gbm)), k
            ok = ~np.isnan(gb)
            assert np.allclose(b[u][ok], gb[ok], rtol=tol, atol=tol), k
            # the trellis on the reference's own emissions
            ctx.upload_hmm(mdl["A"], mdl["pi"])
            r = ctx.viterbi(np.ascontiguousarray(gb[None]), want_score=True)
            assert _eq_nan(r["v"][0], g[k + "_v"]), k
            assert np.array_equal(r["bt"][0], g[k + "_bt"]), k
            assert np.array_equal(r["path"][0], g[k + "_path"]), k
            assert np.isnan(r["score"][0])
            # forward / backward on the same emissions: np.logaddexp's propagation (hmm.py:97-99,128-130)
            tol_t = 1e-10 if not tag else 2e-5
            alpha, ll1 = ctx.forward(np.ascontiguousarray(gb[None]))
            beta = ctx.backward(np.ascontiguousarray(gb[None]))
            for got, want, nm in ((alpha[0], g[k + "_alpha"], "alpha"), (beta[0], g[k + "_beta"], "beta")):
                assert np.array_equal(np.isnan(got), np.isnan(want)), (k, nm)
                okc = ~np.isnan(want)
                assert np.allclose(got[okc], want[okc], rtol=tol_t, atol=tol_t), (k, nm)
            assert np.isnan(ll1[0]) and np.isnan(g[k + "_loglik"][0])
            ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    for u, n in enumerate(names):
        assert np.array_equal(path[u], g[n + "_path"]), n
        assert np.isnan(score[u]) and np.isnan(ll[u]), n
    # full-covariance route (the matrices as the reference stores them)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"], force_full=True)
    path2 = ctx.decode(obs)
    b2 = ctx.gmm_logprob(obs)
    for u, n in enumerate(names):
        assert np.array_equal(path2[u], g[n + "_path"]), n
        assert np.array_equal(np.isnan(b2[u]), np.isnan(g[n + "_b"].T)), n


@pytest.mark.parametrize("N,T,dtype", [(3, 40, np.float64), (6, 70, np.float64), (16, 33, np.float32), (40, 24, np.float64),
                                       (128, 40, np.float64), (128, 40, np.float32), (200, 12, np.float64),
                                       (256, 20, np.float64), (256, 20, np.float32), (300, 9, np.float64),
                                       (520, 6, np.float64)])
def test_viterbi_families_propagate_nan(synth, oracle, ctx, N, T, dtype):
    """Every Viterbi kernel family (tiny, register, on-chip 256, streamed, per-frame; both storages): NaN cells in
    caller-supplied emissions -- first frame, last frame, a single cell, a whole row, two frames, behind the utterance's
    end (ignored) -- give the reference's v, bt, path and score."""
    d = synth.synth_discrete(N, 4, 8, seed=N)
    rng = np.random.default_rng(N + T)
    U = 7
    b = (-30.0 * rng.random((U, T, N))).astype(dtype)
    lens = np.array([T, T, T, T - 2, T, max(1, T // 2), T], dtype=np.int32)
    b[0, 0, N // 2] = np.nan                       # frame 0
    b[1, T - 1, N - 1] = np.nan                    # last frame, last state
    b[2, T // 2, 0] = np.nan                       # a single cell mid-sequence
    b[3, T // 3, :] = np.nan                       # a whole row, ragged utterance
    b[4, T // 4, 1:] = np.nan                      # two frames: the earlier one decides
    b[4, T // 2, 0] = np.nan
    b[5, lens[5]:, :] = np.nan                     # only behind the end: a clean utterance
    ctx.upload_hmm(d["A"], d["pi"])
    r = ctx.viterbi(b, want_score=True, lens=lens)
    for u in range(U):
        L = int(lens[u])
        with np.errstate(invalid="ignore"):
            ov, obt, opath = oracle.viterbi(d["A"], d["pi"], b[u, :L].T, dtype)
        assert _eq_nan(r["v"][u, :L], ov), (N, u)
        assert np.array_equal(r["bt"][u, :L], obt), (N, u)
        assert np.array_equal(r["path"][u, :L], opath), (N, u)
        want = np.max(ov[-1].astype(np.float64))
        assert (np.isnan(r["score"][u]) and np.isnan(want)) or r["score"][u] == want, (N, u)
    assert np.isnan(r["score"][:5]).all() and not np.isnan(r["score"][5:]).any()


@pytest.mark.parametrize("N,M,D,U,T", [(128, 16, 39, 130, 48), (32, 8, 39, 20, 64), (12, 1, 13, 9, 50), (5, 3, 2, 4, 30)])
def test_decode_with_bad_features(synth, oracle, ctx, N, M, D, U, T):
    """decode / calc_prob on observations with NaN / +-inf features, device-resident and from host memory (U = 130 takes
    the chunked upload), for the matrix-core kernel, the single-Gaussian kernel and the generic one: paths equal the
    oracle's for every utterance, clean utterances are untouched."""
    mdl = synth.synth_model(N, M, D, seed=7)
    obs = synth.synth_obs(U, T, D, seed=77)
    lens = np.full(U, T, dtype=np.int32)
    lens[1] = T - 5
    bad = {0: (T // 2, 3 % D, np.nan), 1: (T - 2, 0, np.nan), 2: (0, D - 1, np.inf), 3: (T - 1, 1, -np.inf)}
    if U > 100:
        bad[U - 1] = (7, 5, np.nan)
        bad[70] = (T // 3, 2, np.inf)
    for u, (t, dd, val) in bad.items():
        obs[u, t, dd] = val
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    want = []
    with np.errstate(all="ignore"):
        for u in range(U):
            want.append(oracle.decode_diag(obs[u, :lens[u]], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"]))
    for src in (obs, ctx.to_device(obs)):
        path, score = ctx.decode(src, want_score=True, lens=lens)
        path, score, ll = (_host(x) for x in (path, score, ctx.calc_prob(src, lens=lens)))
        for u in range(U):
            assert np.array_equal(path[u, :lens[u]], want[u]), (N, u)
            # utterance 1's bad frame lies behind its end: clean
            hit = u in bad and bad[u][0] < lens[u]
            assert np.isnan(score[u]) == hit and np.isnan(ll[u]) == hit, (N, u)
    # emissions: whole NaN rows exactly where a feature is not finite
    b = ctx.gmm_logprob(obs)
    rows = ~np.isfinite(obs).all(axis=2)
    assert np.array_equal(np.isnan(b).all(axis=2), rows) and np.array_equal(np.isnan(b).any(axis=2), rows)


def test_clean_data_launches_only_the_checks(synth, ctx):
    """On clean data the repair costs decode ONE launch that reads a word and returns (the emission kernel tested the
    features it staged), and nothing changes in the results."""
    mdl = synth.synth_model(128, 16, 39, seed=1)
    obs = ctx.to_device(synth.synth_obs(64, 50, 39, seed=5))
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    ctx.decode(obs)
    ctx.prof_enable(True)
    ctx.prof_reset()
    p1 = ctx.decode(obs).to_host()
    rep = ctx.prof_report()
    ctx.prof_enable(False)
    if tuned():                                            # under fallback switches other kernels run, and some of them scan
        # launch counts are the claim (the check read one word and returned: no scan, no locate pass); what the launch
        # costs is a measurement, reported by tools/bench_rows.py `nanfix`, never asserted in the parity suite
        assert "viterbi_nan_repair" in rep and "nan_rows" not in rep and "nan_locate" not in rep, rep
        assert rep["viterbi_nan_repair"][0] == 1, rep
    assert not np.isnan(ctx.calc_prob(obs).to_host()).any()
    assert p1.max() < 128


def test_caller_emissions_clean_or_minus_inf(synth, oracle, ctx):
    """lnasr_viterbi on caller-supplied emissions: the back-trace kernel flags NaN / -inf cells of v while it reads them,
    so a dense model never pays a scan of b (one launch reads a word and returns); a model with -inf transitions or
    emissions raises the flag for nothing -- the repair kernel walks its utterances, finds no NaN, and the results stay
    the oracle's."""
    N, U, T = 128, 16, 60
    d = synth.synth_discrete(N, 4, 8, seed=5)
    rng = np.random.default_rng(9)
    b = -30.0 * rng.random((U, T, N))
    ctx.upload_hmm(d["A"], d["pi"])
    db = ctx.to_device(b)
    ctx.viterbi(db)
    ctx.prof_enable(True)
    ctx.prof_reset()
    r = ctx.viterbi(db, want_bt=False)
    rep = ctx.prof_report()
    ctx.prof_enable(False)
    if tuned():                                            # the back-trace kernel looked: no scan, one launch that returns
        assert "nan_locate" not in rep and rep["viterbi_nan_repair"][0] == 1, rep
    assert np.array_equal(r["path"].to_host()[0], oracle.viterbi(d["A"], d["pi"], b[0].T)[2])
    # left-to-right model and some impossible emissions: -inf cells everywhere, no NaN
    A = d["A"].copy()
    A[np.tril_indices(N, -1)] = -np.inf
    b2 = b.copy()
    b2[:, ::7, ::5] = -np.inf
    ctx.upload_hmm(A, d["pi"])
    r = ctx.viterbi(b2, want_score=True)
    for u in (0, 7, 15):
        with np.errstate(invalid="ignore"):
            ov, obt, opath = oracle.viterbi(A, d["pi"], b2[u].T)
        assert np.array_equal(r["v"][u], ov) and np.array_equal(r["bt"][u], obt) and np.array_equal(r["path"][u], opath)
    assert not np.isnan(r["score"]).any()


@pytest.mark.parametrize("precision", [np.float64, np.float32])
def test_statistics_of_a_batch_with_a_bad_feature(synth, oracle, ctx, precision):
    """Baum-Welch statistics (gmmhmm.py:69-156): alpha is NaN from the bad frame on, beta before it, so every posterior
    of that utterance is NaN and every sum it enters -- the oracle's statistics are all NaN (the counts are not), and
    so are these; the same batch without the bad feature is untouched by the check."""
    N, M, D, U, T = 16, 4, 39, 6, 40
    mdl = synth.synth_model(N, M, D, seed=3)
    obs = synth.synth_obs(U, T, D, seed=4)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    def split(st):
        st = _host(st)
        o, out = 0, {}
        for key, n in (("xi_sum", N * N), ("gamma_sum", N), ("gamma0", N), ("mix", N * M), ("mix_x", N * M * D),
                       ("mix_xx", N * M * D), ("loglik", 1), ("n_utt", 1), ("n_frames", 1)):
            out[key] = st[o:o + n]
            o += n
        assert o == st.size
        return out

    clean = split(ctx.bw_accumulate(obs, precision=precision))
    assert all(not np.isnan(x).any() for x in clean.values())
    bad = obs.copy()
    bad[4, 17, 20] = np.inf
    with np.errstate(all="ignore"):
        want = oracle.bw_stats_diag(bad, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])
    for src in (bad, ctx.to_device(bad)):
        got = split(ctx.bw_accumulate(src, precision=precision))
        for k in ("xi_sum", "gamma_sum", "gamma0", "mix", "mix_x", "mix_xx", "loglik"):
            assert np.isnan(np.asarray(want[k])).all() and np.isnan(np.asarray(got[k])).all(), k
        assert got["n_utt"][0] == U and got["n_frames"][0] == U * T
    lens = np.full(U, T, dtype=np.int32)
    lens[4] = 17                                       # the bad frame now lies behind the utterance's end
    got = split(ctx.bw_accumulate(bad, lens=lens, precision=precision))
    assert all(not np.isnan(x).any() for x in got.values())


def test_mfcc_propagates_bad_samples(synth, oracle):
    """MFCC of a signal with a NaN and an infinite sample (mfcc.py:108-175 is plain arithmetic: no max, nothing to
    repair): the same frames and the same delta rows are NaN as in the oracle, every other value agrees -- on the
    single-utterance path, in a ragged batch and for int16-free f64 samples through the 512-point fast path."""
    from lnasr.mfcc import MFCC
    s = synth.synth_signal(16000 + 240, seed=3)
    bad = s.copy()
    bad[4321] = np.nan
    bad[9000] = np.inf
    with np.errstate(all="ignore"):
        op, oc, of = oracle.mfcc(bad)
    p, c, f = MFCC().calc_mfcc(bad)
    for got, want, name in ((p, op, "power"), (c, oc, "cepstrum"), (f, of, "features")):
        # power: a NaN sample makes its frames' bins NaN here as there; an INFINITE one leaves inf in some bins and NaN
        # (inf - inf) in the others, and which is which follows the FFT's butterfly order (numpy's pocketfft: 40-72 inf
        # bins per frame, the radix-4/16 kernel here: 1) -- the bins that are not finite are the same, and one logarithm
        # later both are NaN
        bad_got, bad_want = (~np.isfinite(got), ~np.isfinite(want)) if name == "power" else (np.isnan(got), np.isnan(want))
        assert np.array_equal(bad_got, bad_want), name
        assert bad_want.any()
        ok = ~bad_want
        assert np.allclose(got[ok], want[ok], rtol=1e-9, atol=1e-6), name
    nan_only = s.copy()
    nan_only[4321] = np.nan
    with np.errstate(all="ignore"):
        assert np.array_equal(np.isnan(MFCC().calc_mfcc(nan_only)[0]), np.isnan(oracle.mfcc(nan_only)[0]))
    # the six frames that see a bad sample are NaN in the power spectrum; the cepstral rows pass through a sum over ALL
    # frames (mfcc.py:150-160), so every cepstrum and feature row is NaN -- in the reference and here
    # (the log-energy column comes from each frame's own power and stays finite elsewhere)
    assert (~np.isfinite(op)).any(axis=1).sum() == 6 and np.isnan(of[:, :12]).all() and not np.isnan(of[:, 12]).all()
    feat, foff = MFCC().calc_mfcc_batch([s[:5000], bad, s[:3000]])
    assert np.array_equal(feat[foff[1]:foff[2]], f, equal_nan=True)
    assert not np.isnan(feat[:foff[1]]).any() and not np.isnan(feat[foff[2]:]).any()


@pytest.mark.parametrize("N,T,dtype", [(3, 40, np.float64), (16, 33, np.float32), (40, 24, np.float64), (64, 30, np.float64),
                                       (100, 26, np.float64), (128, 40, np.float64), (128, 40, np.float32),
                                       (200, 12, np.float64), (256, 20, np.float64), (300, 9, np.float64),
                                       (520, 6, np.float64)])
def test_forward_backward_families_propagate_nan(synth, oracle, ctx, N, T, dtype):
    """Every forward / backward kernel family on caller-supplied emissions with NaN cells (the placements of the Viterbi
    test above): alpha is NaN in the NaN states of the first bad frame and everywhere after it, beta everywhere before
    the LAST bad frame, the likelihood NaN -- cell for cell as the oracle's logaddexp folds leave them -- and the
    posteriors of lnasr_baumwelch are NaN for every frame whose alpha / emission / beta slice holds one."""
    d = synth.synth_discrete(N, 4, 8, seed=N)
    rng = np.random.default_rng(N + T)
    U = 7
    b = (-30.0 * rng.random((U, T, N))).astype(dtype)
    lens = np.array([T, T, T, T - 2, T, max(2, T // 2), T], dtype=np.int32)
    b[0, 0, N // 2] = np.nan
    b[1, T - 1, N - 1] = np.nan
    b[2, T // 2, 0] = np.nan
    b[3, T // 3, :] = np.nan
    b[4, T // 4, 1:] = np.nan
    b[4, T // 2, 0] = np.nan
    b[5, lens[5]:, :] = np.nan
    ctx.upload_hmm(d["A"], d["pi"])
    alpha, ll = ctx.forward(b, lens=lens)
    _, ll_only = ctx.forward(b, want_alpha=False, lens=lens)
    beta = ctx.backward(b, lens=lens)
    xi, gamma = ctx.baumwelch(b, alpha, beta, lens=lens)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    for u in range(U):
        L = int(lens[u])
        ob = np.ascontiguousarray(b[u, :L].T)
        with np.errstate(invalid="ignore", divide="ignore"):
            ra, rb = oracle.forward(d["A"], d["pi"], ob, dtype), oracle.backward(d["A"], ob, dtype)
            rxi, rga = oracle.baumwelch(d["A"], ob, ra, rb, dtype)
        for got, want, nm in ((alpha[u, :L], ra, "alpha"), (beta[u, :L], rb, "beta"), (gamma[u, :L], rga, "gamma"),
                              (xi[u, :L], rxi, "xi")):
            assert np.array_equal(np.isnan(got), np.isnan(want)), (N, u, nm)
            ok = ~np.isnan(want) & np.isfinite(want)
            assert np.allclose(got[ok], want[ok], rtol=tol, atol=tol), (N, u, nm)
        want_ll = oracle.loglik(ra)
        assert np.isnan(ll[u]) == np.isnan(want_ll) and np.isnan(ll_only[u]) == np.isnan(want_ll), (N, u)
    assert np.isnan(ll[:5]).all() and not np.isnan(ll[5:]).any()


def test_gaussian_free_functions_with_bad_points(golden, oracle):
    """gmm.py's free functions (gmm.py:16-100) are plain arithmetic: NaN and infinite points come out as the oracle's
    NaN / 0 / -inf, cell for cell, in the linear and the log forms, for a diagonal-looking and a genuinely full Sigma."""
    from lnasr import gmm
    g = golden("gmm_funcs")

    def same(got, want, tol=1e-12):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        assert got.shape == want.shape
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(np.sign(got[np.isinf(got)]),
                                                                                np.sign(want[np.isinf(want)]))
        ok = np.isfinite(want)
        assert np.allclose(got[ok], want[ok], rtol=tol, atol=tol)

    x1 = np.array(g["x1"], dtype=np.float64).copy()
    x1[[1, 5, 7]] = [np.nan, np.inf, -np.inf]
    with np.errstate(all="ignore"):
        same(gmm.gaussian_distribution(x1, 0, 4), oracle.gaussian(x1, 0, 4))
        same(gmm.gaussian_distribution_log(x1, 0, 4), oracle.gaussian_log(x1, 0, 4))
        for Sigma in (np.diag([2.0, 0.5]), np.asarray(g["Sigma"], dtype=np.float64)):
            x2 = np.array(g["x2"], dtype=np.float64).copy()
            x2[2, 0], x2[4, 1], x2[6, 0], x2[8] = np.nan, np.inf, -np.inf, [np.inf, np.inf]
            mu0 = np.zeros(2)
            same(gmm.gaussian_multi_distribution(x2, mu0, Sigma), oracle.gaussian_multi(x2, mu0, Sigma))
            same(gmm.gaussian_multi_distribution_log(x2, mu0, Sigma), oracle.gaussian_multi_log(x2, mu0, Sigma))
        x3 = np.array(g["x3"], dtype=np.float64).copy()
        x3[3, 0], x3[9, 1], x3[20] = np.nan, np.inf, [-np.inf, np.inf]
        same(gmm.gaussian_mixture_distribution(g["c"], x3, g["mu3"], g["Sigma3"]),
             oracle.gaussian_mixture(g["c"], x3, g["mu3"], g["Sigma3"]))
        same(gmm.gaussian_mixture_distribution_log(np.log(g["c"]), x3, g["mu3"], g["Sigma3"]),
             oracle.gaussian_mixture_log(np.log(g["c"]), x3, g["mu3"], g["Sigma3"]))


def test_lse_propagates_nan(oracle):
    """utils.lse / lse2 (utils.py:140-204) fold with np.logaddexp: a NaN anywhere along the reduced axis is a NaN result,
    -inf entries are neutral, an all -inf line is -inf -- along every axis form the reference accepts."""
    from lnasr import utils
    rng = np.random.default_rng(3)
    x = -20.0 * rng.random((5, 7, 33))
    x[1, 2, 5] = np.nan
    x[3, :, 8] = -np.inf
    x[4, 6, :] = np.nan
    x[0, 0, :] = -np.inf
    for axis in (0, 1, 2, (1, 2), (0, 1, 2), None):
        with np.errstate(all="ignore"):
            want, want2 = oracle.lse(x, axis), oracle.lse2(x, axis)
        for got, ref in ((utils.lse(x, axis), want), (utils.lse2(x, axis), want2)):
            got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
            assert got.shape == ref.shape, axis
            assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref)), axis
            ok = np.isfinite(ref)
            assert np.allclose(got[ok], ref[ok], rtol=1e-13, atol=1e-13), axis


def test_all_nan_row_with_the_plain_backtrace(synth, oracle, ctx):
    """Regression for the argmax of an all-NaN row of v (fix 32a9ea1): under LNASR_NO_LDSBT the plain back-trace kernel
    runs, whose wave arg-max started from index 0x7fffffff and never claimed one when every cell compared false -- it
    then read A^T at that index.  An emission row of NaN makes v[t] all NaN from that frame on; path, v and bt must be
    the oracle's (np.argmax of an all-NaN row is 0), for one utterance and inside a batch."""
    N, U, T = 128, 5, 40
    d = synth.synth_discrete(N, 4, 8, seed=11)
    rng = np.random.default_rng(12)
    b = -20.0 * rng.random((U, T, N))
    b[0, 17, :] = np.nan                      # a whole row
    b[2, 0, :] = np.nan                       # from the first frame
    b[3, T - 1, 5] = np.nan                   # one cell of the last frame
    ctx.upload_hmm(d["A"], d["pi"])
    with np.errstate(invalid="ignore"):
        want = [oracle.viterbi(d["A"], d["pi"], b[u].T) for u in range(U)]
    for switches in (dict(LNASR_NO_LDSBT=1), dict(LNASR_NO_LDSBT=1, LNASR_NO_REGVIT=1), {}):
        with env_switch(**switches):
            for src in (b, ctx.to_device(b), b[:1]):
                r = ctx.viterbi(src, want_v=True, want_bt=True, want_score=True)
                r = {k: _host(x) for k, x in r.items()}
                for u in range(len(r["path"])):
                    ov, obt, opath = want[u]
                    assert np.array_equal(r["path"][u], opath), (switches, u)
                    assert np.array_equal(r["v"][u], ov, equal_nan=True), (switches, u)
                    assert np.array_equal(r["bt"][u], obt), (switches, u)
                    assert np.isnan(r["score"][u]) == bool(np.isnan(ov[-1]).any()), (switches, u)
