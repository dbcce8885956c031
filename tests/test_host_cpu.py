"""Host-side logic of the mirror classes that needs no GPU: the reference's parameter initialisation
(hmm.py:233-250, gmmhmm.py:166-188) against the reference-generated goldens, and small helpers."""
import sys

import numpy as np
import pytest

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


# ---- model persistence (hmm.py:252-266, gmmhmm.py:190-208) ------------------------------------------------
class _FakeH5File:
    """A dict-backed stand-in for h5py.File (h5py is not installed in this image; the reference's save/load use exactly
    `File(name, mode)`, `create_dataset(name, data=...)`, `f[name]`, `.shape`, `close()`).  'w' writes a file that starts
    with the HDF5 signature so that format sniffing sees what a real file would show."""
    store = {}
    log = []

    def __init__(self, filename, mode="r"):
        self.filename, self.mode = str(filename), mode
        if mode == "w":
            _FakeH5File.store[self.filename] = {}
            with open(self.filename, "wb") as fp:
                fp.write(b"\x89HDF\r\n\x1a\n" + b"\0" * 24)
        self.d = _FakeH5File.store[self.filename]

    def create_dataset(self, name, data=None):
        assert self.mode == "w"
        _FakeH5File.log.append((self.filename, name))
        self.d[name] = np.array(data)

    def __getitem__(self, name):
        return self.d[name]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


@pytest.fixture
def fake_h5py(monkeypatch):
    import types
    mod = types.ModuleType("h5py")
    mod.File = _FakeH5File
    _FakeH5File.store.clear()
    _FakeH5File.log.clear()
    monkeypatch.setitem(sys.modules, "h5py", mod)
    return _FakeH5File


def _params(rng, n, m, d=None):
    if d is None:
        return dict(A=rng.standard_normal((n, n)), B=rng.standard_normal((n, m)), pi=rng.standard_normal(n))
    return dict(A=rng.standard_normal((n, n)), pi=rng.standard_normal(n), w=rng.standard_normal((n, m)),
                mu=rng.standard_normal((n, m, d)), si=rng.standard_normal((n, m, d, d)))


def test_hdf5_branch_writes_the_references_datasets(tmp_path, fake_h5py):
    """With h5py importable, save() emits exactly the datasets of hmm.py:252-258 ('A','B','pi') and gmmhmm.py:190-198
    ('A','pi','w','mu','si'), in the reference's order, holding the parameters bit for bit; load() reads them back and
    infers n, m (from B.shape, hmm.py:260-266) and n, m, d (from mu.shape, gmmhmm.py:200-208)."""
    from lnasr.hmm import HMM
    from lnasr.gmmhmm import GMMHMM
    rng = np.random.default_rng(0)
    p = _params(rng, 4, 7)
    f1 = str(tmp_path / "hmm.h5")
    HMM(4, 7, p["A"], p["B"], p["pi"]).save(f1)
    assert [n for f, n in fake_h5py.log if f == f1] == ["A", "B", "pi"]
    for k in ("A", "B", "pi"):
        assert np.array_equal(fake_h5py.store[f1][k], p[k]) and fake_h5py.store[f1][k].dtype == np.float64
    h = HMM()                                       # n = m = 1 until load says otherwise
    h.load(f1)
    assert (h.n, h.m) == (4, 7)
    assert all(isinstance(getattr(h, k), np.ndarray) and np.array_equal(getattr(h, k), p[k]) for k in ("A", "B", "pi"))

    q = _params(rng, 3, 2, 5)
    f2 = str(tmp_path / "gmmhmm.h5")
    GMMHMM(3, 2, 5, q["A"], q["pi"], q["w"], q["mu"], q["si"]).save(f2)
    assert [n for f, n in fake_h5py.log if f == f2] == ["A", "pi", "w", "mu", "si"]
    for k, v in q.items():
        assert np.array_equal(fake_h5py.store[f2][k], v)
    g = GMMHMM(n=2, m=2, d=39)                      # as AcousticModel builds it (recognizer.py:23-26)
    g.load(f2)
    assert (g.n, g.m, g.d) == (3, 2, 5)
    assert all(isinstance(getattr(g, k), np.ndarray) and np.array_equal(getattr(g, k), v) for k, v in q.items())
    # float32 models keep their dtype through the file
    q32 = {k: v.astype(np.float32) for k, v in q.items()}
    f3 = str(tmp_path / "gmmhmm32.h5")
    GMMHMM(3, 2, 5, q32["A"], q32["pi"], q32["w"], q32["mu"], q32["si"], precision=np.float32).save(f3)
    g.load(f3)
    assert all(getattr(g, k).dtype == np.float32 and np.array_equal(getattr(g, k), v) for k, v in q32.items())


def test_npz_round_trip_is_bit_exact(tmp_path):
    """Without h5py (this image) or with a .npz name: every array back bit for bit, shapes re-inferred, -inf and NaN
    entries included (log-domain parameters hold -inf for impossible transitions)."""
    from lnasr.hmm import HMM
    from lnasr.gmmhmm import GMMHMM
    rng = np.random.default_rng(1)
    p = _params(rng, 5, 3)
    p["A"][0, 1] = -np.inf
    p["B"][2, 2] = np.nan
    f1 = str(tmp_path / "hmm.npz")
    HMM(5, 3, p["A"], p["B"], p["pi"]).save(f1)
    h = HMM()
    h.load(f1)
    assert (h.n, h.m) == (5, 3)
    for k in ("A", "B", "pi"):
        assert getattr(h, k).tobytes() == p[k].tobytes() and getattr(h, k).dtype == p[k].dtype
    with np.load(f1) as z:
        assert sorted(z.files) == ["A", "B", "pi"]
    q = _params(rng, 2, 4, 6)
    q["w"][1, 3] = -np.inf
    try:
        import h5py  # noqa: F401
        f2 = str(tmp_path / "model.npz")
    except ImportError:
        f2 = str(tmp_path / "model_without_suffix")   # no h5py -> npz container whatever the name
    GMMHMM(2, 4, 6, q["A"], q["pi"], q["w"], q["mu"], q["si"]).save(f2)
    g = GMMHMM()
    g.load(f2)
    assert (g.n, g.m, g.d) == (2, 4, 6)
    for k, v in q.items():
        assert getattr(g, k).tobytes() == v.tobytes()
    with np.load(f2) as z:
        assert sorted(z.files) == ["A", "mu", "pi", "si", "w"]
