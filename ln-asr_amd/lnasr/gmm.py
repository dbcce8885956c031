"""Gaussian densities with the reference's six free functions (lnasr/gmm.py:16-100).

`_log` variants return natural-log densities; means / covariances are linear.  Covariances
are full D x D matrices and need not be symmetric (LU semantics, test/gmm-test.py:42-43).
"""
import numpy as np

from . import _ffi


def gaussian_distribution(x: np.ndarray, mu, sigma2) -> np.ndarray:
    """Univariate normal pdf (gmm.py:16-26)."""
    x = np.asarray(x, dtype=np.float64)
    return _ffi.default_ctx().gauss1d(x.ravel(), mu, sigma2, False).reshape(x.shape)


def gaussian_distribution_log(x: np.ndarray, mu, sigma2) -> np.ndarray:
    """Univariate normal log-pdf (gmm.py:28-30)."""
    x = np.asarray(x, dtype=np.float64)
    return _ffi.default_ctx().gauss1d(x.ravel(), mu, sigma2, True).reshape(x.shape)


def gaussian_multi_distribution(x: np.ndarray, mu: np.ndarray, Sigma: np.ndarray) -> np.ndarray:
    """Multivariate normal pdf, x [L,D] -> [L] (gmm.py:32-57)."""
    return _ffi.default_ctx().gauss_multi(x, mu, Sigma, False)


def gaussian_multi_distribution_log(x: np.ndarray, mu: np.ndarray, Sigma: np.ndarray) -> np.ndarray:
    """Multivariate normal log-pdf (gmm.py:59-68)."""
    return _ffi.default_ctx().gauss_multi(x, mu, Sigma, True)


def gaussian_mixture_distribution(c: np.ndarray, x: np.ndarray, mu: np.ndarray, Sigma: np.ndarray):
    """Mixture pdf with linear weights c [M]; returned as [1, L] like the reference (gmm.py:70-88)."""
    return _ffi.default_ctx().gauss_mixture(c, x, mu, Sigma, False).reshape(1, -1)


def gaussian_mixture_distribution_log(c: np.ndarray, x: np.ndarray, mu: np.ndarray, Sigma: np.ndarray):
    """Mixture log-pdf; c holds LOG weights (gmm.py:90-100)."""
    return _ffi.default_ctx().gauss_mixture(c, x, mu, Sigma, True)
