"""GPU parity of the model-upload kernels of round 5: the register-resident Gauss-Jordan inversion (lu_invert_reg_kernel:
rows never move, the pivot search runs over logical row labels, the permutation is undone at the write) and the tiled
coefficient packer (pack_wfull_tile_kernel), against numpy's inverse / the oracle's direct (x - mu)^T Sigma^-1 (x - mu) and
against the LDS kernels they replace (LNASR_NO_REGLU)."""
import numpy as np
import pytest

from conftest import assert_close, env_switch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def _model(N, M, D, seed, kind):
    rng = np.random.default_rng(seed)
    a = rng.random((N, N)) + 0.1
    w = rng.random((N, M)) + 0.1
    pi = rng.random(N) + 0.1
    mu = 0.6 * rng.random((N, M, D)) - 0.3
    si = np.zeros((N, M, D, D))
    for n in range(N):
        for m in range(M):
            L = 0.4 * (rng.random((D, D)) - 0.5)
            S = L @ L.T + np.diag(0.5 + rng.random(D))
            if kind == "nonsym":                       # test/gmm-test.py:42-43: Sigma need not be symmetric (LU semantics)
                S = S + 0.05 * (rng.random((D, D)) - 0.5)
            if kind == "pivot":                        # small leading entries: partial pivoting has to exchange rows
                P = rng.permutation(D)
                S = S[P]                               # a row permutation of an SPD matrix: still invertible, det sign flips
                if np.linalg.det(S) < 0:               # log det must stay real for the density
                    S[[0, 1]] = S[[1, 0]]
            si[n, m] = S
    return (np.log(a / a.sum(1, keepdims=True)), np.log(pi / pi.sum()), np.log(w / w.sum(1, keepdims=True)), mu, si)


@pytest.mark.parametrize("D,kind", [(1, "spd"), (2, "nonsym"), (5, "pivot"), (8, "spd"), (9, "pivot"), (13, "nonsym"),
                                    (16, "pivot"), (24, "spd"), (25, "pivot"), (33, "nonsym"), (39, "pivot"), (40, "spd"),
                                    (41, "spd")])
def test_register_lu_against_numpy(D, kind, oracle, ctx):
    """Every template size of the register kernel (8 ... 40 columns), sizes in between, and D = 41 (the LDS kernel): per-mixture
    log-densities of the direct form through the C-ABI against the oracle's np.linalg on the same Sigma."""
    N, M, T = 4, 3, 11
    A, pi, w, mu, si = _model(N, M, D, 100 + D, kind)
    obs = np.random.default_rng(D).standard_normal((1, T, D))
    with env_switch(LNASR_KAPPA_MAX=0):                # direct (x - mu)^T P (x - mu): P itself is what is tested
        ctx.upload_gmmhmm(A, pi, w, mu, si, force_full=(D == 1))     # (a 1 x 1 Sigma has no off-diagonal: forced onto the LU road)
        assert not ctx.model_is_diag()
        b, bm = ctx.gmm_logprob(obs, want_bm=True)
    with env_switch(LNASR_KAPPA_MAX=0, LNASR_NO_REGLU=1):
        ctx._model_fp = None                           # same bytes: the fingerprint would skip the upload otherwise
        ctx.upload_gmmhmm(A, pi, w, mu, si, force_full=(D == 1))
        b0, bm0 = ctx.gmm_logprob(obs, want_bm=True)
    want = np.empty((T, N, M))
    for n in range(N):
        for m in range(M):
            want[:, n, m] = oracle.gaussian_multi_log(obs[0], mu[n, m], si[n, m])
    assert_close(bm[0], want, 1e-11, 1e-11, msg="register LU vs numpy, D=%d %s" % (D, kind))
    assert_close(bm[0], bm0[0], 1e-12, 1e-12, msg="register LU vs LDS LU")
    assert_close(b[0], b0[0], 1e-12, 1e-12)


def test_register_lu_declines_singular_and_nonfinite_components(ctx):
    """A zero pivot, an all-zero matrix, a NaN and an infinity: the register kernel flags the component and the LDS kernel's
    answer stands -- whatever it is, it is the answer LNASR_NO_REGLU gives; the other components are untouched."""
    N, M, D, T = 2, 4, 7, 5
    A, pi, w, mu, si = _model(N, M, D, 7, "spd")
    si[0, 1, 3, :] = si[0, 1, 2, :]                    # two equal rows: singular
    si[0, 2] = 0.0
    si[1, 0, 4, 4] = np.nan
    si[1, 3, 0, 1] = np.inf
    obs = np.random.default_rng(3).standard_normal((1, T, D))
    outs = []
    for sw in ({}, {"LNASR_NO_REGLU": 1}):
        with env_switch(LNASR_KAPPA_MAX=0, **sw):
            ctx._model_fp = None
            with np.errstate(all="ignore"):
                ctx.upload_gmmhmm(A, pi, w, mu, si)
                outs.append(ctx.gmm_logprob(obs, want_bm=True)[1][0])
    a, b = outs
    good = [(0, 0), (0, 3), (1, 1), (1, 2)]
    for n, m in good:
        assert np.isfinite(a[:, n, m]).all()
        assert_close(a[:, n, m], b[:, n, m], 1e-12, 1e-12)
    for n, m in [(0, 1), (0, 2), (1, 0), (1, 3)]:
        assert np.array_equal(a[:, n, m], b[:, n, m], equal_nan=True), (n, m)


@pytest.mark.parametrize("N,M,D", [(8, 4, 39), (4, 16, 13), (12, 8, 20), (5, 3, 6)])
def test_tiled_packer_writes_the_same_fragments(N, M, D, synth, ctx):
    """pack_wfull_tile_kernel (one block per MFMA tile row group, each precision matrix read once) against the per-value packer
    (LNASR_NO_TILEPACK): the expanded-quadratic emission of a full-covariance model must not change by a bit -- same expressions,
    same summation order.  Shapes with and without padding (N not a multiple of 4, M not in {4, 8, 16})."""
    mdl = synth.synth_model(N, M, D, seed=5 + N, full_cov=True)
    obs = synth.synth_obs(2, 37, D, seed=8)
    res = []
    for sw in ({}, {"LNASR_NO_TILEPACK": 1}):
        with env_switch(**sw):
            ctx._model_fp = None
            ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
            kappa, direct = ctx.model_conditioning()
            assert not direct                          # the expanded form (the packed coefficients) scores this model
            res.append(ctx.gmm_logprob(obs, want_bm=True))
    (b1, bm1), (b0, bm0) = res
    assert np.array_equal(b1, b0) and np.array_equal(bm1, bm0)
