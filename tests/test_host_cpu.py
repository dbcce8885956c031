"""Host-side logic of the mirror classes that needs no GPU: the reference's parameter initialisation
(hmm.py:233-250, gmmhmm.py:166-188) against the reference-generated goldens, and small helpers."""
import numpy as np

from conftest import PKG_ROOT  # noqa: F401  (puts the package on sys.path)


def test_hmm_reset_reproduces_reference_draws(golden):
    from lnasr.hmm import HMM
    g = golden("hmm_train_4x5")
    np.random.seed(0)
    h = HMM(4, 5)
    h.reset("random")
    assert np.array_equal(h.A, g["A0"]) and np.array_equal(h.B, g["B0"]) and np.array_equal(h.pi, g["pi0"])
    h.reset("bogus")                                   # silent no-op, as in the reference
    assert np.array_equal(h.A, g["A0"])
    h.reset("uniform")
    assert np.array_equal(h.A, np.log(np.full((4, 4), 0.25))) and np.array_equal(h.B, np.log(np.full((4, 5), 0.2)))
    assert np.array_equal(h.pi, np.log(np.full(4, 0.25)))
    h32 = HMM(3, 2, precision=np.float32)
    h32.reset("uniform")
    assert h32.A.dtype == np.float32 and h32.B.dtype == np.float32 and h32.pi.dtype == np.float32
    h32.reset("random")                                # the reference's random draws stay float64 (observed)
    assert h32.A.dtype == np.float64 and np.allclose(np.exp(h32.A).sum(axis=1), 1.0)


def test_gmmhmm_reset_reproduces_reference_draws(golden):
    from lnasr.gmmhmm import GMMHMM
    g = golden("gmmhmm_small")
    np.random.seed(0)
    _ = np.array(0.6 * np.random.random_sample((40, 2)) - 0.3)      # the generator drew the observations first
    gh = GMMHMM(5, 3, 2)
    gh.reset("random")
    for k in ("A", "pi", "w", "mu", "si"):
        assert np.array_equal(getattr(gh, k), g[k]), k
    before = gh.A
    gh.reset("nothing")
    assert gh.A is before
    g32 = GMMHMM(4, 2, 3, precision=np.float32)
    g32.reset("random")
    assert g32.A.dtype == np.float64 and g32.w.dtype == np.float64          # stay float64, as in the reference
    assert g32.mu.dtype == np.float32 and g32.si.dtype == np.float32 and g32.si.shape == (4, 2, 3, 3)
    assert np.array_equal(g32.si[1, 1], np.eye(3, dtype=np.float32)) and np.abs(g32.mu).max() <= 0.3
    g32.reset("uniform")
    assert g32.A.dtype == np.float32 and not g32.mu.any() and np.array_equal(g32.si[3, 0], np.eye(3))
    assert np.allclose(np.exp(g32.w).sum(axis=1), 1.0)


def test_iteration_line_format(capsys):
    from lnasr.hmm import _report_iteration
    d = _report_iteration(7, -1234.5678, -1200.25, True)
    assert capsys.readouterr().out == "Iter:   7, L(lambda|O) = -1.234568e+03, L(lambda_new|O) = -1.200250e+03, eps = 34.317800\n"
    assert abs(d - 34.3178) < 1e-12
    _report_iteration(0, 1.0, 2.0, False)
    assert capsys.readouterr().out == ""
