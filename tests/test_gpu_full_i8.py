"""The FULL-covariance GMM bank on the integer matrix cores (csrc/gmm_full_i8.hip, round 6): the K = D (D + 1) / 2 + D
quadratic-feature GEMM of gmm.py:59-68 / gmmhmm.py:46-67 as a 56-bit fixed-point product of digit planes -- the reference's
values to the same 1e-12 as the fp64 matrix-core kernel wherever the model's range (lnasr_model_range) admits it."""
import numpy as np
import pytest

from conftest import assert_close, env_switch, tuned

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def _kernels_of(ctx, fn):
    ctx.prof_enable(True)
    ctx.prof_reset()
    out = fn()
    names = set(ctx.prof_report())
    ctx.prof_enable(False)
    return out, names


def _upload(ctx, mdl):
    ctx._model_fp = None
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])


@pytest.mark.parametrize("N,M,D,U,T", [(16, 16, 39, 2, 77), (128, 16, 39, 1, 2500), (8, 32, 20, 2, 40), (6, 10, 13, 3, 21),
                                       (4, 16, 5, 1, 1), (12, 16, 39, 1, 130), (20, 16, 26, 2, 64)])
def test_full_integer_bank_equals_the_reference(N, M, D, U, T, synth, oracle, ctx):
    """b from the sliced integer GEMM against the fp64 matrix-core kernel (LNASR_NO_I8EMIT) and the oracle's reference
    arithmetic: 13 k-steps at D = 39 and fewer below, frames not a multiple of 64, one frame, M = 32 as two virtual states
    and M = 10 / N = 6 on the padded copy, state counts that leave slabs of an XCD empty, a zero-weight mixture; 2500 frames
    of the metric's shape take the bank by default."""
    mdl = synth.synth_model(N, M, D, seed=N + M + D, full_cov=True)
    mdl["w"][1, 3] = -np.inf
    obs = synth.synth_obs(U, T, D, seed=T)
    forced = {} if U * T >= 2048 else dict(LNASR_I8_MIN_FRAMES=0)
    with env_switch(**forced):
        _upload(ctx, mdl)
        assert not ctx.model_is_diag()
        rho, bank = ctx.model_range()
        b, names = _kernels_of(ctx, lambda: ctx.gmm_logprob(obs))
    assert (bank and 0 < rho < 100) or not tuned(), (rho, bank)         # (planes are not packed under LNASR_NO_I8EMIT)
    assert "emit_full_i8" in names or not tuned(), names
    with env_switch(LNASR_NO_I8EMIT=1):
        _upload(ctx, mdl)
        b64, names64 = _kernels_of(ctx, lambda: ctx.gmm_logprob(obs))
    assert "emit_full_i8" not in names64
    assert float((np.abs(b - b64) / np.maximum(1.0, np.abs(b64))).max()) <= 1e-13
    t = min(T, 24)
    with np.errstate(divide="ignore"):
        ob, _ = oracle.gmm_map_b(obs[U - 1, :t], mdl["w"], mdl["mu"], mdl["si"])
    assert_close(b[U - 1, :t].T, ob, 1e-12)
    # b + bm is the fp64 kernel's (the integer bank writes b only)
    with env_switch(**forced):
        _upload(ctx, mdl)
        (b2, bm2), names2 = _kernels_of(ctx, lambda: ctx.gmm_logprob(obs, want_bm=True))
    assert "emit_full_i8" not in names2
    assert_close(b2, b64, 1e-13)


def test_full_integer_bank_across_its_frame_chunks(synth, ctx):
    """The feature planes are 5.8 KB per frame at D = 39, so a batch is scored in chunks of 131,072 frames: 140,000 frames of a
    small model cross that boundary (and end in a ragged tile); every row equals the fp64 kernel's."""
    N, M, D, T = 4, 16, 13, 140000
    mdl = synth.synth_model(N, M, D, seed=77, full_cov=True)
    obs = synth.synth_obs(1, T, D, seed=78)
    _upload(ctx, mdl)
    b, names = _kernels_of(ctx, lambda: ctx.gmm_logprob(obs))
    assert "emit_full_i8" in names or not tuned()
    with env_switch(LNASR_NO_I8EMIT=1):
        _upload(ctx, mdl)
        b64 = ctx.gmm_logprob(obs)
    assert float((np.abs(b - b64) / np.maximum(1.0, np.abs(b64))).max()) <= 1e-13
    assert np.isfinite(b[0, 131072 - 70:131072 + 70]).all() and np.isfinite(b[0, -40:]).all()


def test_decode_and_bad_frames_over_the_full_integer_bank(synth, ctx):
    """Emission -> Viterbi -> back-trace over the integer bank gives the fp64 route's paths; a NaN and an infinite feature
    give the rows the fp64 route gives (the pass over non-finite frames re-scores them in the reference's arithmetic)."""
    N, M, D, U, T = 32, 16, 39, 4, 600
    mdl = synth.synth_model(N, M, D, seed=5, full_cov=True)
    obs = synth.synth_obs(U, T, D, seed=6)
    obs[1, 7, 3] = np.nan
    obs[2, 9, 0] = np.inf
    out = {}
    for name, env in (("i8", {}), ("f64", dict(LNASR_NO_I8EMIT=1))):
        with env_switch(**env):
            _upload(ctx, mdl)
            (b, names) = _kernels_of(ctx, lambda: ctx.gmm_logprob(obs))
            out[name] = (b, names, np.asarray(ctx.decode(obs[:1])), np.asarray(ctx.calc_prob(obs[3:])))
    assert "emit_full_i8" in out["i8"][1] or not tuned()
    bi, bf = out["i8"][0], out["f64"][0]
    assert np.array_equal(np.isnan(bi), np.isnan(bf)) and np.array_equal(np.isneginf(bi), np.isneginf(bf))
    ok = np.isfinite(bf)
    assert float((np.abs(bi[ok] - bf[ok]) / np.maximum(1.0, np.abs(bf[ok]))).max()) <= 1e-13
    assert np.array_equal(out["i8"][2], out["f64"][2])
    assert_close(out["i8"][3], out["f64"][3], 1e-12)


@pytest.mark.parametrize("floor,accepted", [(1.0, True), (0.05, True), (0.01, True), (0.002, False)])
def test_full_range_gate(floor, accepted, ctx):
    """What the reference's own M-step produces (gmmhmm.py:138-147): the scatter of the few frames a component owns plus a
    floor on the diagonal (0.01 there) -- rank-deficient covariances whose inverse has entries of 1 / floor.  The smaller the floor the
    larger the range the fixed point must hold: rho is measured at upload, past kFullI8RhoMax the fp64 kernel scores the
    model, and either way the default route returns b to 1e-12 max(1, |b|) of the direct form in long double at frames
    drawn around the components."""
    N, M, D, T = 8, 16, 20, 2304
    rng = np.random.default_rng(int(floor * 1000))
    mu = rng.uniform(-0.3, 0.3, (N, M, D))
    pts = rng.standard_normal((N, M, 12, D))                       # twelve frames per component: rank 12 < D
    si = np.einsum("nmpd,nmpe->nmde", pts, pts) / 12.0 + floor * np.eye(D)
    w = np.log(np.full((N, M), 1.0 / M))
    a = rng.random((N, N)) + 1e-3
    mdl = {"A": np.log(a / a.sum(1, keepdims=True)), "pi": np.log(np.full(N, 1.0 / N)), "w": w, "mu": mu, "si": si}
    comp = rng.integers(0, N * M, T)
    L = np.linalg.cholesky(si.reshape(-1, D, D))
    obs = (mu.reshape(-1, D)[comp] + np.einsum("tde,te->td", L[comp], rng.standard_normal((T, D))))[None]
    # the direct form in long double, on 96 frames
    t = 96
    x = obs[0, :t].astype(np.longdouble)
    ref = np.empty((t, N), dtype=np.longdouble)
    for j in range(N):
        z = np.empty((t, M), dtype=np.longdouble)
        for m in range(M):
            P = np.linalg.inv(si[j, m]).astype(np.longdouble)
            logdet = np.longdouble(np.linalg.slogdet(si[j, m])[1])
            dx = x - mu[j, m].astype(np.longdouble)
            z[:, m] = -0.5 * np.einsum("td,de,te->t", dx, P, dx) - 0.5 * (D * np.log(2 * np.longdouble(np.pi)) + logdet) + np.longdouble(w[j, m])
        mx = z.max(1, keepdims=True)
        ref[:, j] = mx[:, 0] + np.log(np.exp(z - mx).sum(1))
    out = {}
    for name, env in (("default", {}), ("i8", dict(LNASR_I8_MIN_FRAMES=0, LNASR_I8_RHO_MAX=1e300)), ("f64", dict(LNASR_NO_I8EMIT=1))):
        with env_switch(**env):
            _upload(ctx, mdl)
            rho, bank = ctx.model_range()
            kappa, direct = ctx.model_conditioning()
            b, names = _kernels_of(ctx, lambda: ctx.gmm_logprob(obs))
        rel = float((np.abs(b[0, :t].astype(np.longdouble) - ref) / np.maximum(1.0, np.abs(ref))).max())
        out[name] = dict(rel=rel, rho=rho, bank=bank, direct=direct, i8="emit_full_i8" in names)
    tag = "floor %g cond %.0f: %r" % (floor, np.linalg.cond(si[0, 0]), out)
    print(tag)
    if out["default"]["direct"]:
        pytest.skip("the conditioning gate took this model: " + tag)
    assert out["default"]["rel"] <= 1e-12 and out["f64"]["rel"] <= 1e-12, tag
    if tuned():
        assert out["default"]["bank"] == accepted and out["default"]["i8"] == accepted, tag
    if accepted:
        assert out["i8"]["rel"] <= 1e-12, tag


import os
_lo, _hi = (int(x) for x in os.environ.get("LNASR_TEST_SEEDS", "0:16").split(":"))


@pytest.mark.parametrize("seed", range(_lo, _lo + max(1, (_hi - _lo) * 12 // 16)))
def test_full_integer_bank_over_seeded_shapes(seed, ctx):
    """Seeded shapes (N 2 .. 40, M = 16 / 32 / 10 / 12, D 2 .. 39: one to thirteen k-steps, padded and virtual states, slabs
    that leave XCDs idle), covariances = a scaled scatter of a few frames + a floor (conditioning 2 .. ~1000), a common
    offset, frame counts that end in ragged tiles: the forced integer bank against the fp64 kernel to 1e-12 max(1, |b|)
    whenever the model is admitted, decode paths identical."""
    rng = np.random.default_rng(50000 + seed)
    N = int(rng.integers(2, 41))
    M = int(rng.choice([16, 16, 32, 10, 12]))
    D = int(rng.integers(2, 40))
    U, T = int(rng.integers(1, 3)), int(rng.integers(1, 200))
    floor = float(rng.choice([1.0, 0.3, 0.05, 0.01]))
    npts = int(rng.integers(max(2, D // 2), 2 * D + 2))
    offset = float(rng.choice([0.0, 0.0, 10.0]))
    mu = offset + rng.uniform(-0.5, 0.5, (N, M, D))
    pts = rng.standard_normal((N, M, npts, D))
    si = np.einsum("nmpd,nmpe->nmde", pts, pts) / npts + floor * np.eye(D)
    w = rng.random((N, M)) + 1e-3
    w = np.log(w / w.sum(1, keepdims=True))
    a = rng.random((N, N)) + 1e-3
    mdl = {"A": np.log(a / a.sum(1, keepdims=True)), "pi": np.log(np.full(N, 1.0 / N)), "w": w, "mu": mu, "si": si}
    comp = rng.integers(0, N * M, (U, T))
    L = np.linalg.cholesky(si.reshape(-1, D, D))
    obs = mu.reshape(-1, D)[comp] + np.einsum("utde,ute->utd", L[comp], rng.standard_normal((U, T, D)))
    out = {}
    for name, env in (("i8", dict(LNASR_I8_MIN_FRAMES=0)), ("f64", dict(LNASR_NO_I8EMIT=1))):
        with env_switch(**env):
            _upload(ctx, mdl)
            rho, bank = ctx.model_range()
            kappa, direct = ctx.model_conditioning()
            b, names = _kernels_of(ctx, lambda: ctx.gmm_logprob(obs))
            out[name] = (b, names, np.asarray(ctx.decode(obs)), rho, bank, direct)
    tag = "seed=%d N=%d M=%d D=%d U=%d T=%d floor=%g npts=%d offset=%g rho=%.0f bank=%d direct=%d i8=%d" % (
        seed, N, M, D, U, T, floor, npts, offset, out["i8"][3], out["i8"][4], out["i8"][5], "emit_full_i8" in out["i8"][1])
    bi, bf = out["i8"][0], out["f64"][0]
    assert float((np.abs(bi - bf) / np.maximum(1.0, np.abs(bf))).max()) <= 1e-12, tag
    assert np.array_equal(out["i8"][2], out["f64"][2]), tag
    if tuned() and out["i8"][4] and not out["i8"][5] and M in (16, 32):
        assert "emit_full_i8" in out["i8"][1], tag
