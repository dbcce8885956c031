"""GPU: error behaviour of the boundary -- bad arguments come back as LnasrError with the library's
message (the reference raises ordinary numpy exceptions in the same situations), nothing crashes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_argument_errors(synth):
    from lnasr import _ffi
    ctx = _ffi.Context(0)
    try:
        b = np.zeros((1, 4, 3))
        with pytest.raises(_ffi.LnasrError, match="no HMM transition model"):
            ctx.viterbi(b)
        d = synth.synth_discrete(3, 4, 8, seed=1)
        ctx.upload_hmm(d["A"], d["pi"])
        with pytest.raises(_ffi.LnasrError, match="no GMM-HMM model"):
            ctx.decode(np.zeros((1, 4, 2)))
        with pytest.raises(_ffi.LnasrError, match="lens"):
            ctx.viterbi(b, lens=np.array([9], dtype=np.int32))
        with pytest.raises(_ffi.LnasrError, match="outside the symbol set"):
            ctx.discrete_map_b(d["B"], np.array([[0, 1, 7]], dtype=np.int32))
        with pytest.raises(ValueError):
            ctx.upload_hmm(d["A"][:, :2], d["pi"])
        mdl = synth.synth_model(4, 2, 5, seed=1)
        with pytest.raises(ValueError):
            ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])      # diag flag missing
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
        with pytest.raises(ValueError, match="D="):
            ctx.gmm_logprob(np.zeros((1, 3, 7)))
        # empty batches are no-ops
        assert ctx.decode(np.zeros((0, 5, 5))).shape == (0, 5)
        # the context still works after errors
        assert ctx.decode(synth.synth_obs(2, 6, 5, seed=3)).shape == (2, 6)
        # singular covariance: det = 0 -> log det = -inf -> +inf density constant, as numpy would give inf/nan
        var = mdl["var"].copy()
        var[0, 0, 0] = 0.0
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], var, diag=True)
        out = ctx.gmm_logprob(synth.synth_obs(1, 3, 5, seed=3))
        assert not np.isfinite(out[0, :, 0]).all()
    finally:
        ctx.close()


def test_two_contexts_and_reupload(synth, oracle):
    """Independent contexts; re-uploading a model of a different shape reuses the context."""
    from lnasr import _ffi
    a, b = _ffi.Context(0), _ffi.Context(0)
    try:
        m1 = synth.synth_model(6, 3, 4, seed=1)
        m2 = synth.synth_model(128, 16, 39, seed=2)
        a.upload_gmmhmm(m1["A"], m1["pi"], m1["w"], m1["mu"], m1["var"], diag=True)
        b.upload_gmmhmm(m2["A"], m2["pi"], m2["w"], m2["mu"], m2["var"], diag=True)
        o1, o2 = synth.synth_obs(2, 9, 4, seed=5), synth.synth_obs(2, 9, 39, seed=6)
        p1, p2 = a.decode(o1), b.decode(o2)
        assert np.array_equal(p1[0], oracle.decode_diag(o1[0], m1["A"], m1["pi"], m1["w"], m1["mu"], m1["var"]))
        assert np.array_equal(p2[1], oracle.decode_diag(o2[1], m2["A"], m2["pi"], m2["w"], m2["mu"], m2["var"]))
        a.upload_gmmhmm(m2["A"], m2["pi"], m2["w"], m2["mu"], m2["var"], diag=True)
        assert np.array_equal(a.decode(o2), p2)
    finally:
        a.close()
        b.close()


def test_launch_helpers_get_device_pointers_only(synth):
    """LNASR_CHECK_PTRS=1: every pointer handed to a kernel launcher (`*_device`, `*_nan_repair`, emit_*) is verified
    with hipPointerGetAttributes first.  Host-array calls of every entry point of the path pass -- the entry points
    stage them -- with host, pinned and device inputs and outputs mixed; and the guard does fire on a heap pointer."""
    from conftest import env_switch
    from lnasr import _ffi
    from lnasr.gmmhmm import GMMHMM
    with env_switch(LNASR_CHECK_PTRS=1):
        ctx = _ffi.default_ctx()
        heap = np.zeros(16)
        assert _ffi.lib().lnasr_debug_require_device_ptr(heap.ctypes.data) == -1
        assert _ffi.lib().lnasr_debug_require_device_ptr(ctx.empty((4,)).ptr) == 0
        for N, M, D, U, T in ((128, 16, 39, 130, 24), (256, 32, 39, 3, 24), (5, 3, 2, 2, 9), (320, 2, 4, 2, 6)):
            mdl = synth.synth_model(N, M, D, seed=2)
            obs = synth.synth_obs(U, T, D, seed=3)
            ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
            lens = np.full(U, T, dtype=np.int32)
            lens[-1] = T - 2
            for src in (obs, ctx.pinned_copy(obs), ctx.to_device(obs)):
                p = ctx.decode(src, lens=lens)
                ll = ctx.calc_prob(src, lens=lens)
                st = ctx.bw_accumulate(src, lens=lens)
                st32 = ctx.bw_accumulate(src, lens=lens, precision=np.float32)
                b, bm = ctx.gmm_logprob(src, want_bm=True)
                r = ctx.viterbi(b, want_score=True, lens=lens)
                al, l2 = ctx.forward(b, lens=lens)
                be = ctx.backward(b, lens=lens)
                xi, ga = ctx.baumwelch(b, al, be, lens=lens)
            ctx.sync()
            # mixed placement: device emissions, host outputs (the mirror classes' route)
            db = ctx.to_device(np.asarray(b.to_host() if hasattr(b, "to_host") else b))
            r = ctx.viterbi(db, host_out=True)
            assert isinstance(r["path"], np.ndarray)
        mdl = synth.synth_model(5, 3, 2, seed=4)
        g = GMMHMM(5, 3, 2, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
        o = synth.synth_obs(1, 30, 2, seed=5)[0]
        g.train(o, iters=2, verbose=False)
        g32 = GMMHMM(5, 3, 2, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"], precision=np.float32)
        g32.train(o, iters=2, verbose=False)
        assert g.decode(o).shape == (30,)


def test_clock_probe_reads_a_plausible_shader_clock():
    """lnasr_debug_clock: a one-wave kernel counts shader-clock cycles over 20 us of the 100 MHz counter, directly behind whatever
    the stream ran last.  Any MI355X clock lies between 0.3 and 3 GHz; two probes in a row agree to a few per cent."""
    from lnasr import _ffi
    ctx = _ffi.default_ctx()
    a, b = ctx.debug_clock(), ctx.debug_clock()
    assert 0.3 < a < 3.0 and 0.3 < b < 3.0, (a, b)
    assert abs(a - b) < 0.3 * max(a, b), (a, b)
