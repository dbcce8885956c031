"""GPU parity: numeric utils (lse/lse2/split_frames), Gaussian free functions and the MFCC chain."""
import os

import numpy as np
import pytest

from conftest import assert_close, GOLDEN

pytestmark = pytest.mark.gpu


def test_lse_cases(golden):
    from lnasr.utils import lse, lse2
    g = golden("lse_cases")
    a = g["a"]
    i = np.array([0, 0, 1])
    assert_close(lse2(a), g["lse2_all"], 1e-14), assert_close(lse2(a, axis=0), g["lse2_0"], 1e-14)
    assert_close(lse2(a, axis=2), g["lse2_2"], 1e-14), assert_close(lse2(a, axis=(0, 2)), g["lse2_t02"], 1e-14)
    assert_close(lse2(a[:, 2 == i, :], axis=1), g["lse2_sel2"], 1e-14)
    assert_close(lse2(a[:, 0 == i, :], axis=1), g["lse2_sel0"], 1e-14)
    assert_close(lse(a), g["lse_all"], 1e-14)
    for ax, key in ((0, "lse_0"), (1, "lse_1"), (2, "lse_2"), ((0, 2), "lse_t02"), ((0, 1, 2), "lse_t012")):
        assert_close(lse(a, axis=ax), g[key], 1e-14, msg=key)
    r = g["r"]
    for ax, key in ((0, "r_lse0"), (1, "r_lse1"), (2, "r_lse2"), (None, "r_lse_all"), ((1, 2), "r_lse_t12")):
        assert_close(lse(r, axis=ax), g[key], 1e-13, msg=key)
    assert_close(lse2(r, axis=1), g["r_lse2_1"], 1e-13)
    assert np.isneginf(lse(r, axis=1)[2, 1]) and np.isneginf(lse(r, axis=2)[3, 5])
    out = lse(np.zeros((3, 0, 4)), axis=1)
    assert out.shape == tuple(g["empty_shape"]) and np.isneginf(out).all()
    assert np.array_equal(lse(g["one"], axis=1), g["one_lse1"])


def test_frames(golden):
    from lnasr.utils import split_frames, create_hamming
    g = golden("frames")
    assert np.array_equal(split_frames(np.arange(30), 3, 2), g["f30"])
    assert np.array_equal(create_hamming(400), g["ham400"]) and np.array_equal(create_hamming(7), g["ham7"])
    for n, nf, L in g["shapes"]:
        f = split_frames(np.arange(n, dtype=np.float64), 400, 160)
        assert f.shape == (nf, L)
        key = "f_n%d" % n
        if key in g.files:
            assert np.array_equal(f, g[key])
    assert np.array_equal(split_frames(np.arange(23.0), 5, 5), g["f_L5S5"])
    assert np.array_equal(split_frames(np.arange(9.0), 4, 1), g["f_L4S1"])


def test_gmm_funcs(golden):
    from lnasr import gmm
    g = golden("gmm_funcs")
    assert_close(gmm.gaussian_distribution(g["x1"], 0, 4), g["g1"], 1e-14)
    assert_close(gmm.gaussian_distribution_log(g["x1"], 0, 4), g["g1_log"], 1e-14)
    mu0 = np.zeros(2)
    assert_close(gmm.gaussian_multi_distribution(g["x2"], mu0, g["Sigma"]), g["g2"], 1e-13)
    assert_close(gmm.gaussian_multi_distribution_log(g["x2"], mu0, g["Sigma"]), g["g2_log"], 1e-13)
    gm = gmm.gaussian_mixture_distribution(g["c"], g["x3"], g["mu3"], g["Sigma3"])
    assert gm.shape == (1, 169)
    assert_close(gm, g["gm"], 1e-13)
    assert_close(gmm.gaussian_mixture_distribution_log(np.log(g["c"]), g["x3"], g["mu3"], g["Sigma3"]), g["gm_log"], 1e-13)
    assert_close(gmm.gaussian_mixture_distribution_log(g["c39"], g["x39"], g["mu39"], g["Sigma39"]), g["gm39_log"], 1e-12)


def test_mfcc_data_raw(golden):
    from lnasr.mfcc import MFCC
    from lnasr.utils import read_pcm
    g = golden("mfcc_data_raw")
    sig = read_pcm(os.path.join(GOLDEN, "data.raw"))
    m = MFCC()
    assert np.array_equal(m.create_filter_bank(), g["fbank"])
    assert_close(m.calc_mel(np.array([0.0, 700.0, 8000.0])), g["mel_pts"], 1e-15)
    assert_close(m.calc_mel_inv(np.array([0.0, 1000.0, 2595.0])), g["mel_inv_pts"], 1e-15)
    assert np.array_equal(m.calc_high_pass_filter(sig.astype(np.float64))[:8], g["hpf_head"])
    power, cep, feat = m.calc_mfcc(sig.astype(np.float64))
    assert feat.shape == (135, 39) and cep.shape == (135, 40) and power.shape == (135, 257)
    assert_close(power[:3], g["power_head"], 1e-10, 1e-7)
    assert_close(power.sum(axis=1), g["power_rowsum"], 1e-11)
    assert_close(cep, g["cepstrum"], 1e-9, 1e-9)
    assert_close(feat, g["features"], 1e-9, 1e-9)
    assert np.allclose(cep[:, 1:13].mean(axis=0), -1e-8, atol=1e-11)
    assert np.array_equal(feat[0, 13:26], feat[1, :13])
    # int16 PCM straight to the device gives the same features
    _, _, feat_i16 = m.calc_mfcc(sig)
    assert np.array_equal(feat_i16, feat)


def test_mfcc_synth_and_batch(golden, synth, oracle):
    from lnasr.mfcc import MFCC
    g = golden("mfcc_synth_small")
    p, c, f = MFCC().calc_mfcc(g["signal"])
    assert_close(p, g["power"], 1e-9, 1e-6), assert_close(c, g["cepstrum"], 1e-9, 1e-9)
    assert_close(f, g["features"], 1e-9, 1e-9)
    g2 = golden("mfcc_synth")
    s = synth.synth_signal(int(g2["n"][0]), seed=int(g2["seed"][0]))
    _, _, f = MFCC().calc_mfcc(s)
    assert f.shape == (1000, 39)
    assert_close(f, g2["features"], 1e-9, 1e-9)
    g3 = golden("mfcc_alt_params")
    m = MFCC(fs=8000, frame_T=32e-3, frame_stride=16e-3, hpf_alpha=0.95, fft_N=256, mfbank_num=26, cepstrum_num=12)
    assert np.array_equal(m.create_filter_bank(), g3["fbank"])
    p, c, f = m.calc_mfcc(g3["signal"])
    assert_close(c, g3["cepstrum"], 1e-9, 1e-9), assert_close(f, g3["features"], 1e-9, 1e-9)
    # ragged batch == per-utterance results
    sigs = [g["signal"], s[:5000], g["signal"][:1000]]
    feat, foff = MFCC().calc_mfcc_batch(sigs)
    for u, x in enumerate(sigs):
        _, _, fu = MFCC().calc_mfcc(x)
        assert np.array_equal(feat[foff[u]:foff[u + 1]], fu)
        assert_close(fu, oracle.mfcc(x)[2], 1e-9, 1e-9)
    # a single frame has no delta row in the reference (IndexError there): refused loudly here
    from lnasr._ffi import LnasrError
    with pytest.raises(LnasrError):
        MFCC().calc_mfcc(np.ones(300))


def test_recognizer_wiring(tmp_path, synth):
    """Model persistence + Recognizer shell: PCM -> MFCC -> decode (parity unpinned upstream)."""
    from lnasr.gmmhmm import GMMHMM
    from lnasr.recognizer import Recognizer
    mdl = synth.synth_model(4, 2, 39, seed=8)
    gh = GMMHMM(4, 2, 39, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    path = str(tmp_path / "gh.npz")
    gh.save(path)
    rec = Recognizer({"am": path, "lm": None})
    assert (rec.am.gmmhmm.n, rec.am.gmmhmm.m, rec.am.gmmhmm.d) == (4, 2, 39)
    pcm = np.clip(synth.synth_signal(8000, seed=1) * 5, -32768, 32767).astype(np.int16)
    states = rec.recognize(pcm)
    assert states.dtype == np.uint32 and states.shape == (49,) and states.max() < 4


def _model_on_features(feats, N, M, seed):
    """A GMM-HMM whose components sit on frames of the clip itself (means = features of chosen frames, one shared
    per-dimension variance): states compete for real, unlike a model far away from the data."""
    rng = np.random.default_rng(seed)
    D = feats.shape[1]
    pick = rng.choice(len(feats), size=N * M, replace=len(feats) < N * M)
    mu = feats[pick].reshape(N, M, D) + 0.05 * rng.standard_normal((N, M, D))
    var = np.broadcast_to(0.25 * feats.var(axis=0) + 1e-3, (N, M, D)).copy()

    def logn(x):
        return np.log(x / x.sum(axis=-1, keepdims=True))
    return dict(A=logn(rng.random((N, N)) + 0.05), pi=logn(rng.random(N) + 0.05), w=logn(rng.random((N, M)) + 0.05), mu=mu, var=var)


@pytest.mark.parametrize("clip", ["data.raw", "synthetic"])
def test_recognize_equals_oracle_chain(tmp_path, synth, oracle, clip):
    """Recognizer.recognize(pcm) == oracle.decode_diag(oracle.mfcc(pcm)[2], model), state for state --
    on the reference's own speech clip (test/data.raw, read as utils.read_pcm does) and on a synthetic one; int16 PCM
    and the same samples as float64."""
    from lnasr.gmmhmm import GMMHMM
    from lnasr.recognizer import Recognizer
    from conftest import GOLDEN
    import os
    if clip == "data.raw":
        pcm = np.fromfile(os.path.join(GOLDEN, "data.raw"), dtype="<i2")
    else:
        pcm = np.clip(synth.synth_signal(16000, seed=21) * 5, -32768, 32767).astype(np.int16)
    with np.errstate(divide="ignore"):
        feats = oracle.mfcc(pcm.astype(np.float64))[2]
    N, M, D = 8, 3, 39
    m = _model_on_features(feats, N, M, seed=5)
    si = np.zeros((N, M, D, D))
    si[:, :, np.arange(D), np.arange(D)] = m["var"]
    want = oracle.decode_diag(feats, m["A"], m["pi"], m["w"], m["mu"], m["var"])
    assert len(set(want.tolist())) >= 4                       # the path really moves between states
    path = str(tmp_path / "am.npz")
    GMMHMM(N, M, D, m["A"], m["pi"], m["w"], m["mu"], si).save(path)
    rec = Recognizer({"am": path, "lm": None})
    for samples in (pcm, pcm.astype(np.float64)):
        got = rec.recognize(samples)
        assert got.dtype == np.uint32 and got.shape == want.shape
        assert np.array_equal(got, want), "%s: %d of %d states differ" % (clip, int((got != want).sum()), len(want))
    # the two halves on their own, and the state the reference's decode() leaves behind
    _, _, f = rec.am.mfcc.calc_mfcc(pcm)
    assert_close(f, feats, 1e-9, 1e-9)
    assert np.array_equal(rec.am.gmmhmm.decode(f), want)
    ob, obm = oracle.gmm_map_b_diag(feats, m["w"], m["mu"], m["var"])
    assert_close(rec.am.gmmhmm.b, ob, 1e-9, 1e-7), assert_close(rec.am.gmmhmm.bm, obm, 1e-9, 1e-7)


def test_mfcc_ragged_edges(oracle):
    """Utterance lengths around every boundary of the four-frames-per-wave kernel: frames that start beyond the
    samples (n < L - S), a last frame cut anywhere, frame counts that are not multiples of four, odd sample offsets
    in the int16 batch (pairs of samples are loaded as one unaligned dword), groups of four frames that straddle
    utterances.  Batch == single utterance bit for bit; both against the oracle."""
    from lnasr.mfcc import MFCC
    rng = np.random.default_rng(5)
    lens = [10, 79, 401, 560, 561, 562, 721, 1000, 1001, 3333, 7777, 512, 513, 672, 673, 2 * 160 + 241, 4001]
    sigs16 = [rng.integers(-3000, 3000, n).astype(np.int16) for n in lens]
    sigs16[3][:] = 0                                               # an all-zero utterance: eps and -inf paths
    for sigs in (sigs16, [s.astype(np.float64) * 1.000244140625 for s in sigs16]):
        feat, foff = MFCC().calc_mfcc_batch(sigs)
        for u, x in enumerate(sigs):
            p, c, fu = MFCC().calc_mfcc(x)
            assert np.array_equal(feat[foff[u]:foff[u + 1]], fu, equal_nan=True), "batch vs single, utterance %d (n=%d)" % (u, len(x))
            with np.errstate(divide="ignore", invalid="ignore"):
                op, oc, of = oracle.mfcc(np.asarray(x, dtype=np.float64))
            assert_close(p, op, 1e-9, 1e-6, msg="power n=%d" % len(x))
            assert_close(c, oc, 1e-9, 1e-9, msg="cepstrum n=%d" % len(x))
            assert_close(fu, of, 1e-9, 1e-9, msg="features n=%d" % len(x))


@pytest.mark.parametrize("kw", [
    dict(frame_T=32e-3, frame_stride=8e-3, mfbank_num=26, cepstrum_num=12),      # L = 512 = fft_N, S = 128
    dict(frame_T=20e-3, frame_stride=5e-3, mfbank_num=48, cepstrum_num=20),      # 48 filters, 21 feature columns
    dict(frame_T=25e-3, frame_stride=10e-3, mfbank_num=40, cepstrum_num=39, hpf_alpha=0.9),   # n_cep = n_mel - 1
    dict(fs=8000, frame_T=50e-3, frame_stride=12.5e-3, mfbank_num=23, cepstrum_num=10),       # 8 kHz, L = 400, S = 100
])
def test_mfcc_fft512_other_parameters(kw, oracle):
    """The four-frames-per-wave fft-512 kernel away from the reference's defaults: frame length equal to the FFT
    size, other hops, filter counts up to the table limit (48), more than 16 feature columns (the unsplit DCT),
    another pre-emphasis; single utterance (power and cepstrum returned) and a ragged int16 batch, against the oracle."""
    from lnasr.mfcc import MFCC
    rng = np.random.default_rng(11)
    m = MFCC(**kw)
    ocfg = oracle.MfccCfg(**kw)
    assert m.fft_N == 512
    sigs = [rng.integers(-3000, 3000, n).astype(np.int16) for n in (5000, 1777, 12345)]
    feat, foff = m.calc_mfcc_batch(sigs)
    for u, x in enumerate(sigs):
        p, c, fu = m.calc_mfcc(x)
        with np.errstate(divide="ignore", invalid="ignore"):
            op, oc, of = oracle.mfcc(x.astype(np.float64), ocfg)
        assert_close(p, op, 1e-9, 1e-6, msg="power %d" % u)
        assert_close(c, oc, 1e-9, 1e-9, msg="cepstrum %d" % u)
        assert_close(fu, of, 1e-9, 1e-9, msg="features %d" % u)
        assert np.array_equal(feat[foff[u]:foff[u + 1]], fu, equal_nan=True)
