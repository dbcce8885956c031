"""The reference's own method names as the drop-in surface (SURVEY 8b "State"; hmm.py:276-283, gmmhmm.py:46-67):
`decode` / `calc_prob` / `_viterbi` / `_forward` keep the emissions in HBM between `_map_b` and the trellis, and
`self.b` / `self.bm` still read as the reference leaves them -- the arrays of the golden vectors -- whenever somebody
looks, assigns, or edits them in place."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lnasr import _ffi
    return _ffi.default_ctx()


def _diag_model(g):
    N, M, D = g["mu"].shape
    si = np.zeros((N, M, D, D))
    si[:, :, np.arange(D), np.arange(D)] = g["var"]
    return N, M, D, si


def test_b_and_bm_read_as_golden_after_decode(golden):
    """decode() moves only the path; reading gh.b / gh.bm AFTERWARDS still yields the reference's arrays."""
    from lnasr.gmmhmm import GMMHMM
    for name in ("gmmhmm_d39_diag", "gmmhmm_d39_full", "gmmhmm_small"):
        g = golden(name)
        if "si" in g.files:
            (N, M, D), si = g["mu"].shape, g["si"]
        else:
            N, M, D, si = _diag_model(g)
        gh = GMMHMM(N, M, D, g["A"], g["pi"], g["w"], g["mu"], si)
        assert np.array_equal(gh.decode(g["obs"]), g["path"]), name
        # nothing came back but the path (the parameters the caller handed over are host arrays, as given)
        assert set(gh._resident) == {"b", "bm"} and set(gh._host) == set(GMMHMM._PARAMS) | {"B"}
        assert_close(gh.b, g["b"], 1e-12, 1e-12, msg=name + " b")
        assert_close(gh.bm, g["bm"], 1e-12, 1e-12, msg=name + " bm")
        assert gh.b.shape == g["b"].shape and gh.bm.shape == g["bm"].shape
        assert not gh._resident                                           # read once: the host copies are the state now
        # ... and the remap=False calls go on from them
        v, bt, path = gh._viterbi(g["obs"])
        assert np.array_equal(path, g["path"]) and np.array_equal(bt, g["bt"])
        assert_close(gh.calc_prob(g["obs"]), g["loglik"][0], 1e-12)


def test_remap_false_reads_the_resident_b(golden):
    """_map_b, then _forward / _backward / _viterbi / _baumwelch / _estimate with remap=False: the emissions never
    leave HBM, results as the golden ones."""
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_small")
    gh = GMMHMM(5, 3, 2, g["A"], g["pi"], g["w"], g["mu"], g["si"])
    gh._map_b(g["obs"])
    alpha, beta = gh._forward(g["obs"]), gh._backward(g["obs"])
    v, bt, path = gh._viterbi(g["obs"])
    xi, gamma = gh._baumwelch(g["obs"], alpha, beta)
    est = gh._estimate(g["obs"], alpha, beta, xi, gamma)
    assert set(gh._resident) == {"b", "bm"} and set(gh._host) == set(GMMHMM._PARAMS) | {"B"}    # (the parameters: host arrays as given)
    assert_close(alpha, g["alpha"], 1e-12), assert_close(beta, g["beta"], 1e-12)
    assert np.array_equal(path, g["path"]) and np.array_equal(bt, g["bt"])
    assert_close(v, g["v"], 1e-12), assert_close(xi, g["xi"], 1e-11, 1e-11), assert_close(gamma, g["gamma"], 1e-11, 1e-11)
    for k in ("A", "pi", "w", "mu", "si"):
        assert_close(est[k], g["est_" + k], 1e-9, 1e-9, msg=k)


def test_host_copy_is_authoritative_once_read_or_assigned(golden, oracle):
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_d39_diag")
    N, M, D, si = _diag_model(g)
    gh = GMMHMM(N, M, D, g["A"], g["pi"], g["w"], g["mu"], si)
    gh._map_b(g["obs"])
    b = gh.b                                           # read: from now on this array is the state
    assert gh.b is b
    rng = np.random.default_rng(3)
    b += rng.normal(size=b.shape)                      # the caller edits it in place ...
    v, bt, path = gh._viterbi(g["obs"])                # ... and remap=False must see the edit (hmm.py:155-157)
    rv, rbt, rpath = oracle.viterbi(g["A"], g["pi"], b)
    assert np.array_equal(v, rv) and np.array_equal(bt, rbt) and np.array_equal(path, rpath)
    assert_close(gh._forward(g["obs"]), oracle.forward(g["A"], g["pi"], b), 1e-12)
    gh.b = g["b"]                                      # assignment of a caller's array
    v, bt, path = gh._viterbi(g["obs"])
    assert np.array_equal(v, g["v"]) and np.array_equal(bt, g["bt"]) and np.array_equal(path, g["path"])
    # remap=True replaces both again
    assert np.array_equal(gh.decode(g["obs"]), g["path"])
    assert_close(gh.b, g["b"], 1e-12)


def test_in_place_parameter_edits_are_noticed_between_calls(golden, oracle):
    """The reference re-reads self.mu / self.si / self.A on every call: an in-place edit between two decode() calls must
    reach the device (fingerprint of the parameter bytes, lnasr_hash128), an untouched model must not be uploaded again."""
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_d39_diag")
    N, M, D, si = _diag_model(g)
    gh = GMMHMM(N, M, D, g["A"].copy(), g["pi"].copy(), g["w"].copy(), g["mu"].copy(), si)
    assert np.array_equal(gh.decode(g["obs"]), g["path"])
    fp = gh.ctx._model_fp
    assert np.array_equal(gh.decode(g["obs"]), g["path"]) and gh.ctx._model_fp == fp
    for what in ("mu", "si", "A", "w", "pi"):
        arr = getattr(gh, what)
        if what == "si":
            arr[1, 0, 2, 2] *= 3.0
        elif what in ("A", "w", "pi"):
            arr[...] = np.roll(arr, 1, axis=-1)
        else:
            arr[0, 1, :] += 0.75
        var = gh.si[:, :, np.arange(D), np.arange(D)]
        got = gh.decode(g["obs"])
        assert gh.ctx._model_fp != fp, what
        fp = gh.ctx._model_fp
        assert np.array_equal(got, oracle.decode_diag(g["obs"], gh.A, gh.pi, gh.w, gh.mu, var)), what
        rb, _ = oracle.gmm_map_b_diag(g["obs"], gh.w, gh.mu, var)
        assert_close(gh.calc_prob(g["obs"]), oracle.loglik(oracle.forward(gh.A, gh.pi, rb)), 1e-11, msg=what)


def test_overridden_steps_are_walked(golden):
    """A subclass that replaces a step keeps the reference's call sequence (decode -> _viterbi(obs, True) -> _map_b;
    calc_prob -> _forward(obs, True)), as scripts/seg/seg.py's Seg(HMM) relies on."""
    from lnasr.hmm import HMM
    g = golden("toy_hmm")
    calls = []

    class Traced(HMM):
        def _viterbi(self, obs, remap=False):
            calls.append(("viterbi", remap))
            return super()._viterbi(obs, remap)

        def _forward(self, obs, remap=False):
            calls.append(("forward", remap))
            return super()._forward(obs, remap)

        def _map_b(self, obs):
            calls.append(("map_b",))
            super()._map_b(obs)

    h = Traced(2, 3, g["A"], g["B"], g["pi"])
    assert np.array_equal(h.decode(g["obs"]), g["decode"])
    assert_close(h.calc_prob(g["obs"]), g["calc_prob"], 1e-14)
    assert calls == [("viterbi", True), ("map_b",), ("forward", True), ("map_b",)]
    assert np.array_equal(h.b, g["b"])
    plain = HMM(2, 3, g["A"], g["B"], g["pi"])
    assert np.array_equal(plain.decode(g["obs"]), g["decode"]) and np.array_equal(plain.b, g["b"])
    assert_close(plain.calc_prob(g["obs"]), g["calc_prob"], 1e-14)
    with pytest.raises(AttributeError):
        HMM(2, 3, g["A"], g["B"], g["pi"]).b              # as in the reference before any _map_b


def test_metric_shape_single_utterance_surface(golden, synth):
    """N=128, M=16, D=39, T=1000 with `si` as [N,M,D,D] -- the literal drop-in call (hmm.py:276-283): the golden
    metric slice's paths, likelihood on the first 32 frames, and b read back afterwards."""
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_metric_slice")
    mdl = synth.synth_model(128, 16, 39, seed=1)
    obs = synth.synth_obs(2, 1000, 39, seed=1000)
    gh = GMMHMM(128, 16, 39, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    for u in range(2):
        assert np.array_equal(gh.decode(obs[u]), g["path"][u])
        b, bm = gh.b, gh.bm                                   # [N,T], [N,M,T] as the reference stores them
        assert b.shape == (128, 1000) and bm.shape == (128, 16, 1000)
        assert_close(b[g["sample_j"], g["sample_t"]], g["b_sample"][u], 1e-12)
        assert_close(bm[g["sample_j"], g["sample_m"], g["sample_t"]], g["bm_sample"][u], 1e-12)
        assert abs(b.sum() - g["b_sum"][u]) < 1e-6
    assert_close(gh.calc_prob(obs[0][:32]), g["loglik32"][0], 1e-11)


def test_parameter_check_runs_beside_the_kernels(golden, oracle, monkeypatch):
    """decode() / calc_prob() on a context whose model this object uploaded last: the fingerprint of the parameters is
    computed in the background (lnasr_hash128_begin) while the kernels run on the model the device holds, and the answer
    is kept only if it confirms.  Second call: no synchronous check, no upload; after an in-place edit: the
    speculative answer is dropped and the call repeated on the new parameters; another object using the context in
    between: no speculation at all."""
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_d39_diag")
    N, M, D, si = _diag_model(g)
    gh = GMMHMM(N, M, D, g["A"].copy(), g["pi"].copy(), g["w"].copy(), g["mu"].copy(), si)
    ctx = gh.ctx
    calls = {"upload": 0, "job": 0}
    up, job = ctx.upload_gmmhmm, ctx.gmm_fingerprint_job
    monkeypatch.setattr(ctx, "upload_gmmhmm", lambda *a, **k: (calls.__setitem__("upload", calls["upload"] + 1), up(*a, **k))[1])
    monkeypatch.setattr(ctx, "gmm_fingerprint_job", lambda *a, **k: (calls.__setitem__("job", calls["job"] + 1), job(*a, **k))[1])
    assert np.array_equal(gh.decode(g["obs"]), g["path"])
    assert calls == {"upload": 1, "job": 0} and ctx.model_owner == id(gh)
    assert np.array_equal(gh.decode(g["obs"]), g["path"])
    assert_close(gh.calc_prob(g["obs"]), g["loglik"][0], 1e-12)
    assert calls == {"upload": 1, "job": 2}                      # both answers came from speculative runs that were confirmed
    gh.mu[3, 1, :] -= 0.5                                        # in place: the next speculative answer must be thrown away
    var = gh.si[:, :, np.arange(D), np.arange(D)]
    want = oracle.decode_diag(g["obs"], gh.A, gh.pi, gh.w, gh.mu, var)
    assert not np.array_equal(want, g["path"])
    assert np.array_equal(gh.decode(g["obs"]), want)
    assert calls == {"upload": 2, "job": 3}
    rb, _ = oracle.gmm_map_b_diag(g["obs"], gh.w, gh.mu, var)
    assert_close(gh.b, rb, 1e-12)                                # the state left behind is the repeated call's, not the stale one
    other = GMMHMM(N, M, D, g["A"], g["pi"], g["w"], g["mu"], si, ctx=ctx)
    assert np.array_equal(other.decode(g["obs"]), g["path"])     # uploads its own model: gh no longer owns the device copy
    assert np.array_equal(gh.decode(g["obs"]), want)
    assert calls == {"upload": 4, "job": 3}
    # a call that fails for the caller's own reasons still fails, speculating or not
    with pytest.raises(ValueError, match="D="):
        gh.decode(g["obs"][:, :5])


def test_train_leaves_the_new_parameters_in_hbm_until_somebody_reads_them(golden):
    """GMMHMM.train on the fused route (round 5): the M-step's A, pi, w, mu, Sigma stay in HBM and are installed as the next
    model device to device; the attributes are numpy arrays the moment they are read -- the golden ones (gmmhmm_small: the
    imported reference after three iterations) -- and from then on host arrays rule again: an in-place edit reaches the device."""
    import contextlib
    import io
    from lnasr.gmmhmm import GMMHMM
    from lnasr import _ffi
    g = golden("gmmhmm_small")
    gh = GMMHMM(5, 3, 2, g["A"].copy(), g["pi"].copy(), g["w"].copy(), g["mu"].copy(), g["si"].copy())    # the model before training
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gh.train(g["obs"], 3, 0.0, True)
    assert buf.getvalue() == str(g["train_log"])
    assert set(GMMHMM._PARAMS) <= set(gh._resident)                       # nothing but likelihoods came back
    assert all(isinstance(gh._resident[k][0], _ffi.DeviceArray) for k in GMMHMM._PARAMS)
    # scoring and decoding go on from the resident model (no host copy appears)
    assert abs(gh.calc_prob(g["obs"]) - g["train3_loglik"][0]) < 1e-9
    assert np.array_equal(gh.decode(g["obs"]), g["train3_decode"])
    assert set(GMMHMM._PARAMS) <= set(gh._resident)
    # reading makes them numpy arrays: the reference's values
    for k in GMMHMM._PARAMS:
        assert_close(getattr(gh, k), g["train3_" + k], 1e-8, 1e-10, msg=k)
        assert isinstance(getattr(gh, k), np.ndarray) and k not in gh._resident
    assert abs(gh.calc_prob(g["obs"]) - g["train3_loglik"][0]) < 1e-9       # the checked (host) road gives the same
    # an in-place edit of a parameter that was read reaches the device
    before = gh.calc_prob(g["obs"])
    gh.mu[...] += 0.05
    assert gh.calc_prob(g["obs"]) != before
    # a second training run starts from the edited host arrays and ends resident again, in the spare buffers
    with contextlib.redirect_stdout(io.StringIO()):
        gh.train(g["obs"], 2, 0.0, True)
    assert set(GMMHMM._PARAMS) <= set(gh._resident)


def test_train_stops_on_eps_with_the_state_calc_prob_leaves(synth):
    """The fused loop takes L(lambda_new | O) from the next iteration's E-step; when it stops on eps right after one, self.b /
    self.bm are those of the final model, as after the reference's closing calc_prob (hmm.py:304)."""
    from lnasr.gmmhmm import GMMHMM
    N, M, D, T = 6, 2, 3, 200
    mdl = synth.synth_model(N, M, D, seed=11)
    obs = synth.synth_obs(1, T, D, seed=12)[0]

    class Stepwise(GMMHMM):                          # an overridden step: the reference's call sequence, calc_prob included
        def _estimate(self, *a):
            return super()._estimate(*a)
    runs = []
    for cls in (GMMHMM, Stepwise):
        gh = cls(N, M, D, mdl["A"].copy(), mdl["pi"].copy(), mdl["w"].copy(), mdl["mu"].copy(), mdl["si"].copy())
        import contextlib, io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            gh.train(obs, 60, 0.5, True)
        runs.append((buf.getvalue(), gh))
    (log_a, a), (log_b, b) = runs
    assert a._fused_train_ok() and not b._fused_train_ok()
    assert log_a == log_b and 1 < log_a.count("Iter:") < 60                 # stopped on eps, same iteration, same lines
    # (two routes to the same M-step -- packed statistics with split-K atomics against the per-sequence estimate -- agree to
    # ~1e-12 per iteration; a dozen EM iterations amplify that, hence the tolerances.  What is pinned: the state is the FINAL
    # model's, i.e. b / bm equal what _map_b gives for the parameters the object now holds)
    assert_close(a.b, b.b, 1e-7, 1e-7), assert_close(a.bm, b.bm, 1e-7, 1e-7)
    for k in GMMHMM._PARAMS:
        assert_close(getattr(a, k), getattr(b, k), 1e-6, 1e-8, msg=k)
    b_state, bm_state = a.b.copy(), a.bm.copy()
    a._map_b(obs)
    assert np.array_equal(a.b, b_state) and np.array_equal(a.bm, bm_state)


def test_discrete_train_keeps_the_loop_in_hbm(golden, synth):
    """HMM.train on the fused route (round 5): observations up once, alpha / beta / xi / gamma and the M-step's A, B, pi stay in
    HBM, L(lambda_new | O) comes from the forward pass the next iteration needs anyway.  Pinned: the printed lines and the
    parameters are the imported reference's (hmm_train_4x5), a subclass that overrides a step walks the reference's call
    sequence to the same lines and the same state, and the resident parameters behave as arrays the moment they are read."""
    import contextlib
    import io
    from lnasr.hmm import HMM
    from lnasr import _ffi
    g = golden("hmm_train_4x5")
    h = HMM(4, 5, g["A0"].copy(), g["B0"].copy(), g["pi0"].copy())
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        h.train(g["obs"], 3, 1e-4, True)
    assert buf.getvalue() == str(g["train3_log"])
    assert set(HMM._PARAMS) <= set(h._resident) and all(isinstance(h._resident[k][0], _ffi.DeviceArray) for k in HMM._PARAMS)
    p_res = h.calc_prob(g["obs"])                       # scoring and decoding go on from the resident model
    path_res = h.decode(g["obs"])
    assert set(HMM._PARAMS) <= set(h._resident)
    assert_close(h.A, g["train3_A"], 1e-10)             # reading makes them numpy arrays: the reference's values
    for k in HMM._PARAMS:
        assert isinstance(getattr(h, k), np.ndarray) and k not in h._resident
    assert h.calc_prob(g["obs"]) == p_res and np.array_equal(h.decode(g["obs"]), path_res)   # the checked (host) road
    h.B[:, 0] -= 0.3                                    # an in-place edit of a parameter that was read reaches the device
    assert h.calc_prob(g["obs"]) != p_res
    with contextlib.redirect_stdout(io.StringIO()):
        h.train(g["obs"], 2, 0.0, True)                 # starts from the edited host arrays, ends resident again
    assert set(HMM._PARAMS) <= set(h._resident)

    class Stepwise(HMM):                                # an overridden step: the reference's call sequence
        def _baumwelch(self, *a, **k):
            return super()._baumwelch(*a, **k)
    d = synth.synth_discrete(6, 9, 300, seed=31)
    runs = []
    for cls in (HMM, Stepwise):
        m = cls(6, 9, d["A"].copy(), d["B"].copy(), d["pi"].copy())
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            m.train(d["obs"], 300, 1e-2, True)
        runs.append((buf.getvalue(), m))
    (log_a, a), (log_b, b) = runs
    assert "A" in a._resident and "A" not in b.__dict__.get("_resident", {})
    assert log_a == log_b and 1 < log_a.count("Iter:") < 300                 # stopped on eps, same iteration, same lines
    assert_close(a.b, b.b, 1e-9, 1e-9)                                        # the final model's emissions, as calc_prob leaves them
    for k in HMM._PARAMS:
        assert_close(getattr(a, k), getattr(b, k), 1e-8, 1e-10, msg=k)
