#!/usr/bin/env python3
"""CPU emulation of the integer digit-plane bank's arithmetic (csrc/gmm_i8.hip), digit for digit: what the matrix cores
sum exactly is summed here in int64, and the fold is the kernel's (pairs joined in int32, three fused multiply-adds).

Two scalings of the operands:
  row      one exponent per GEMM row k, from the largest |W[k, c]| over ALL components (rounds 5's scheme): a sharp
           component raises the rows' exponents, every frame's exponent follows, the other components lose the bits
  comp     rows scaled by the model's own data scale per dimension, every component by its own exponent h_c
           (round 6): the absolute error of a component's quadratic is bounded by ITS largest coefficient times the
           frame's largest normalised feature

    python tools/i8_emulate.py             # the table DESIGN.md quotes: one whole sharp component, ratios 1 .. 1e6
"""
import sys

import numpy as np

KS = 7


def scale_exponent(m):
    """e with m 2^-e <= 0.498 (gmm_i8.hip: scale_exponent), m > 0; arrays"""
    m = np.asarray(m, dtype=np.float64)
    e = np.frexp(m)[1] + 1                       # ilogb(m) + 2 = (frexp exponent - 1) + 2
    e = np.where(np.ldexp(m, -e) > 0.498, e + 1, e)
    return np.where(m > 0, e, 0).astype(np.int64)


def digits(v):
    """|v| <= 0.498 -> 7 balanced base-256 digits (int64 [.., 7]) with v = sum s_i 256^-(i+1) + r, |r| <= 2^-57"""
    t = np.rint(np.ldexp(v, 56)).astype(np.int64)       # |v| 2^56 < 2^55: exact in int64 (v has 53 significant bits)
    out = np.zeros(v.shape + (KS,), dtype=np.int64)
    for i in range(KS - 1, 0, -1):
        d = ((t + 128) & 255) - 128
        out[..., i] = d
        t = (t - d) >> 8
    out[..., 0] = t
    return out


def coefficients(mu_c, var):
    """W [K = 2 D, C]: (-1/2 / var | mu / var) of the centred means"""
    iv = 1.0 / var
    return np.concatenate([(-0.5 * iv).T, (mu_c * iv).T], axis=0)


def features(x_c):
    """phi [F, K]: (x^2 | x) of the centred observations"""
    return np.concatenate([x_c * x_c, x_c], axis=1)


def row_exponents(W, scheme, mu_c=None, var=None):
    K = W.shape[0]
    if scheme == "row":
        return scale_exponent(np.abs(W).max(axis=1))
    # the data's own scale per dimension under the model: s_d^2 = geometric mean over the components of var + mu_c^2 (one
    # absurd component moves it by its C-th root only); quadratic rows 1 / s^2, linear rows 1 / s
    l2 = np.log2(var + mu_c * mu_c).mean(axis=0)
    gq = -np.rint(l2).astype(np.int64)
    gl = -np.rint(0.5 * l2).astype(np.int64)
    return np.concatenate([gq, gl])


def model_range(mu_c, var, z=4.0):
    """rho = max_c A_c X_c: A_c the component's largest normalised coefficient, X_c the largest normalised feature of a
    frame within z standard deviations of the component's own mean -- 2^-53 rho is the size of the bank's absolute error
    on a component's log-density at the frames that component describes (gmm_i8.hip: i8_range_kernel)"""
    W = coefficients(mu_c, var)
    g = row_exponents(W, "comp", mu_c, var)
    D = mu_c.shape[1]
    A = np.abs(np.ldexp(W, -g[:, None])).max(axis=0)
    reach = np.abs(mu_c) + z * np.sqrt(var)
    X = np.maximum(np.ldexp(reach * reach, g[None, :D]), np.ldexp(reach, g[None, D:])).max(axis=1)
    return float((A * X).max())


def quadratic_i8(x_c, mu_c, var, scheme="comp", planes=KS):
    """Q [F, C] = sum_k W[k, c] phi_k(x_f) by the sliced integer product, and the exponents used"""
    W = coefficients(mu_c, var)
    phi = features(x_c)
    g = row_exponents(W, scheme, mu_c, var)
    Ws = np.ldexp(W, -g[:, None])
    if scheme == "comp":
        h = scale_exponent(np.abs(Ws).max(axis=0))
    else:
        h = np.zeros(W.shape[1], dtype=np.int64)
    Ws = np.ldexp(Ws, -h[None, :])
    assert np.abs(Ws).max() <= 0.498
    ps = np.ldexp(phi, g[None, :])
    ef = scale_exponent(np.abs(ps).max(axis=1))
    ps = np.ldexp(ps, -ef[:, None])
    dw = digits(Ws)                                      # [K, C, 7]
    dp = digits(ps)                                      # [F, K, 7]
    acc = np.zeros((planes, phi.shape[0], W.shape[1]), dtype=np.int64)
    for i in range(planes):
        for j in range(planes - i):
            acc[i + j] += dp[:, :, i] @ dw[:, :, j]      # exact: < 2^24 per diagonal
    assert np.abs(acc).max() < 2 ** 24
    # the kernel's fold: unit = the weight of diagonal 1; pairs joined in int32, then fp64 fused multiply-adds
    r = None
    for n in range((planes + 1) // 2):
        pn = acc[2 * n] * 256 + acc[2 * n + 1] if 2 * n + 1 < planes else acc[2 * n]
        pn = pn.astype(np.float64)                       # |pn| < 2^32: exact
        wgt = [1.0, 2.0 ** -16, 2.0 ** -32, 2.0 ** -40][n]
        r = pn if r is None else pn * wgt + r            # product exact (power of two): one rounding, as the fma
    # value = r 256^-3 2^(e_f + h_c): the kernel's frame scale 2^(e_f - 24), times 2^h_c
    return np.ldexp(r, (ef[:, None] - 24) + h[None, :]), dict(g=g, h=h, ef=ef)


def quadratic_f64_gemm(x_c, mu_c, var):
    """the fp64 matrix-core route's arithmetic class: the same expanded product, one rounding per term (np.dot in fp64)"""
    return features(x_c) @ coefficients(mu_c, var)


def quadratic_exact(x_c, mu_c, var):
    """long-double evaluation of the same expanded sum (the value both routes approximate)"""
    W = coefficients(mu_c.astype(np.longdouble), var.astype(np.longdouble))
    return features(x_c.astype(np.longdouble)) @ W


def sharp_component_case(ratio, seed=0, C=32, D=39, F=2000):
    """one whole component's variances divided by `ratio`, its mean on the model centre; frames drawn from the others"""
    rng = np.random.default_rng(seed)
    mu = rng.uniform(-0.3, 0.3, (C, D))
    var = 0.5 + rng.uniform(0.0, 1.0, (C, D))
    centre = mu.mean(axis=0)
    mu[0] = centre
    var[0] /= ratio
    src = rng.integers(1, C, F)
    x = mu[src] + np.sqrt(var[src]) * rng.standard_normal((F, D))
    return x - centre, mu - centre, var


def main():
    print("%8s  %-23s %-23s %-12s" % ("ratio", "row scales: broad / all", "comp scales: broad / all", "fp64 GEMM"))
    for ratio in (1.0, 1e2, 1e3, 1e4, 1e6):
        x, mu, var = sharp_component_case(ratio)
        ex = quadratic_exact(x, mu, var)
        out = []
        for scheme in ("row", "comp"):
            q, _ = quadratic_i8(x, mu, var, scheme)
            err = np.abs(q.astype(np.longdouble) - ex).astype(np.float64)
            rel = err / np.maximum(1.0, np.abs(ex).astype(np.float64))
            out.append("%.1e / %.1e (rel %.0e)" % (err[:, 1:].max(), err.max(), rel.max()))
        e64 = np.abs(quadratic_f64_gemm(x, mu, var).astype(np.longdouble) - ex).astype(np.float64)
        print("%8.0e  %-23s %-23s %.1e / %.1e" % (ratio, out[0], out[1], e64[:, 1:].max(), e64.max()))


if __name__ == "__main__":
    sys.exit(main())
