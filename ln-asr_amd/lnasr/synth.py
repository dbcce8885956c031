"""Deterministic synthetic models / observations.

Self-contained (numpy only, no package-relative imports) so that the golden
generator, the tests and bench.py can all load it by file path and obtain the
same tensors from the same seeds.

  obs   ~ N(0, 1)                 [U, T, D] float64, seed 1000 + shard
  mu    ~ U(-0.3, 0.3)            [N, M, D]
  var   ~ 0.5 + U(0, 1)           [N, M, D]  (diagonal of Sigma)
  A, pi, w = log(row-normalised U(0, 1))
"""
import numpy as np


def synth_model(N, M, D, seed=1, full_cov=False):
    """Return a dict of GMM-HMM parameters in the reference's conventions
    (natural-log A/pi/w, `si` as [N, M, D, D]) plus `var` [N, M, D]."""
    rng = np.random.default_rng(seed)
    a = 1.0 - rng.random((N, N))
    p = 1.0 - rng.random(N)
    w = 1.0 - rng.random((N, M))
    mu = 0.6 * rng.random((N, M, D)) - 0.3
    var = 0.5 + rng.random((N, M, D))
    si = np.zeros((N, M, D, D))
    idx = np.arange(D)
    si[:, :, idx, idx] = var
    if full_cov:
        # SPD, well conditioned: diag(var) + small symmetric low-rank part
        L = 0.15 * (rng.random((N, M, D, 3)) - 0.5)
        si = si + L @ np.swapaxes(L, -1, -2)
    return {
        "A": np.log(a / a.sum(axis=1, keepdims=True)),
        "pi": np.log(p / p.sum()),
        "w": np.log(w / w.sum(axis=1, keepdims=True)),
        "mu": mu,
        "var": var,
        "si": si,
    }


def synth_obs(U, T, D, seed=1000):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((U, T, D))


def synth_discrete(N, K, T, seed=3):
    """Discrete HMM (log A, log B, log pi) and an int32 symbol sequence."""
    rng = np.random.default_rng(seed)
    a = 1.0 - rng.random((N, N))
    b = 1.0 - rng.random((N, K))
    p = 1.0 - rng.random(N)
    obs = rng.integers(0, K, size=T).astype(np.int32)
    return {
        "A": np.log(a / a.sum(axis=1, keepdims=True)),
        "B": np.log(b / b.sum(axis=1, keepdims=True)),
        "pi": np.log(p / p.sum()),
        "obs": obs,
    }


def synth_signal(n, seed=7, fs=16000):
    """Noise + chirp test signal (config 2 shape: n=160240 -> 1000 frames)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    chirp = np.sin(2 * np.pi * (200.0 + 1500.0 * t) * t)
    return 1000.0 * (0.5 * rng.standard_normal(n) + chirp)
