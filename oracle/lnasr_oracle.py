"""CPU oracle: a numpy restatement of the ln-asr MFCC / GMM / HMM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`ln-asr_amd/`) may
import this file.  Its only legal users are `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py`, where it is the checker / the thing
timed beside the GPU, never the thing shipped.

Parity status: PINNED.  Every function below is checked against golden vectors
produced by importing the unmodified reference in the build container
(`tests/golden/gen_golden.py` -> `tests/golden/*.npz`, `tests/test_oracle.py`):
Viterbi `v/bt/path` bit-identical, everything else <= 1e-10 relative in fp64.

All `file:line` citations are relative to the reference checkout
(`/root/reference/`).  Conventions follow the reference: A, B, pi, w are
natural-log probabilities; `A[i, j]` is i -> j; `b` is [N, T] (state-major);
`alpha / beta / v / bt` are [T, N]; `bm` is [N, M, T].

The restatement is written from the maths (doc/source/hmm.rst, gmmhmm.rst,
mfcc.rst) and the observable behaviour; loops over states are vectorised, the
Baum-Welch M-step streams instead of materialising the reference's
N x M x T x D x D tensor (gmmhmm.py:138-146).
"""
import numpy as np

LOG2PI = np.log(2.0 * np.pi)


# ----------------------------------------------------------------------------
# U1 / U2: log-sum-exp                                     (utils.py:140-204)
# ----------------------------------------------------------------------------
def _lse_impl(seq, axis, op):
    seq = np.asarray(seq)
    if axis is None:                                   # utils.py:157-158
        return _lse_impl(seq.ravel(), 0, op)
    if isinstance(axis, tuple):                        # utils.py:159-165: last axis first
        out = seq
        for ax in reversed(axis):
            out = _lse_impl(out, ax, op)
        return out
    n = seq.shape[axis]
    if n == 0:                                         # utils.py:167-168 (drops ALL zero dims)
        return np.full(tuple(d for d in seq.shape if d != 0), -np.inf)
    if n == 1:                                         # utils.py:169-170
        return seq.take(0, axis)
    # utils.py:171-179: a LEFT fold of logaddexp, no max shift.  ufunc.reduce on a
    # ufunc without reorderable pairwise summation is exactly that fold.
    with np.errstate(invalid="ignore"):
        return op.reduce(seq, axis=axis)


def lse(seq, axis=None):
    """log(sum(exp(seq))) along `axis`; fold order as utils.py:176-179."""
    return _lse_impl(seq, axis, np.logaddexp)


def lse2(seq, axis=None):
    """Base-2 variant (utils.py:181-204)."""
    return _lse_impl(seq, axis, np.logaddexp2)


# ----------------------------------------------------------------------------
# U3 / U4: framing and window                              (utils.py:106-138)
# ----------------------------------------------------------------------------
def split_frames(signal, L, S):
    """Frames [Nf, L]; Nf = ceil(|len-(L-S)|/S) (utils.py:119), zero padded tail."""
    signal = np.asarray(signal)
    n = len(signal)
    nf = int(np.ceil(np.abs(n - (L - S)) / S))
    need = nf * S + (L - S)
    if need > n:                                       # utils.py:122-123 (np.append -> float64)
        pad = np.concatenate([signal, np.zeros(need - n)])
    else:
        pad = signal
    idx = np.arange(L)[None, :] + (S * np.arange(nf))[:, None]
    return pad[idx]


def create_hamming(N):
    """0.54 - 0.46 cos(2 pi n / (N-1))  (utils.py:132-138)."""
    return 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(N) / (N - 1))


# ----------------------------------------------------------------------------
# M1-M5: MFCC                                                 (mfcc.py:16-175)
# ----------------------------------------------------------------------------
class MfccCfg:
    def __init__(self, fs=16000, frame_T=25e-3, frame_stride=10e-3, hpf_alpha=0.97,
                 fft_N=512, mfbank_num=40, cepstrum_num=12):
        self.fs = fs
        self.frame_L = int(fs * frame_T)               # mfcc.py:40
        self.frame_step = int(fs * frame_stride)       # mfcc.py:42
        self.hpf_alpha = hpf_alpha
        self.fft_N = fft_N
        self.fft_size = fft_N // 2 + 1                 # mfcc.py:45
        self.mfbank_num = mfbank_num
        self.cepstrum_num = cepstrum_num


def mel(hz):
    return 2595 * np.log(1 + np.asarray(hz) / 700)     # natural log (mfcc.py:72)


def mel_inv(m):
    return 700 * (np.exp(np.asarray(m) / 2595) - 1)    # mfcc.py:76


def high_pass(signal, alpha):
    """y[0]=x[0]; y[t]=x[t]-alpha*x[t-1]  (mfcc.py:50-64)."""
    signal = np.asarray(signal)
    out = np.empty(len(signal), dtype=np.result_type(signal.dtype, np.float64))
    out[0] = signal[0]
    out[1:] = signal[1:] - alpha * signal[:-1]
    return out


def mel_bin_edges(cfg):
    """floor((N/fs)*hz) of M+2 mel-equispaced points on [0, fs/2]  (mfcc.py:89-96)."""
    hz = mel_inv(np.linspace(mel(0), mel(cfg.fs / 2), cfg.mfbank_num + 2))
    return np.floor((cfg.fft_N / cfg.fs) * hz).astype(np.int64)


def filter_bank(cfg):
    """Triangles rising on [lo,mi), falling on [mi,hi), unnormalised (mfcc.py:98-104)."""
    e = mel_bin_edges(cfg)
    fb = np.zeros((cfg.mfbank_num, cfg.fft_size))
    for m in range(cfg.mfbank_num):
        lo, mi, hi = int(e[m]), int(e[m + 1]), int(e[m + 2])
        if mi > lo:
            fb[m, lo:mi] = (np.arange(lo, mi) - lo) / (mi - lo)
        if hi > mi:
            fb[m, mi:hi] = (hi - np.arange(mi, hi)) / (hi - mi)
    return fb


def dct2_ortho_matrix(n):
    """DCT-II, norm='ortho' as a matrix C with y = x @ C.T (scipy.fftpack.dct, mfcc.py:152)."""
    k = np.arange(n)[:, None]
    j = np.arange(n)[None, :]
    C = np.cos(np.pi * k * (2 * j + 1) / (2 * n)) * np.sqrt(2.0 / n)
    C[0] *= np.sqrt(0.5)
    return C


def mfcc(signal, cfg=None):
    """(frames_power [Nf,257], frames_cepstrum [Nf,40], features [Nf,39])  (mfcc.py:108-175).

    Reproduces the reference's quirks: the second mean subtraction is applied
    through a view, so the RETURNED cepstrum columns 1..12 are mean-normalised
    twice (mfcc.py:155-156); delta row 0 is features[1] (mfcc.py:162-172).
    """
    cfg = cfg or MfccCfg()
    x = high_pass(np.asarray(signal, dtype=np.float64), cfg.hpf_alpha)
    frames = split_frames(x, cfg.frame_L, cfg.frame_step) * create_hamming(cfg.frame_L)
    mag = np.abs(np.fft.rfft(frames, cfg.fft_N))
    power = (1.0 / cfg.fft_N) * (mag ** 2)
    cep = power @ filter_bank(cfg).T
    cep = np.where(cep == 0, np.finfo(float).eps, cep)
    with np.errstate(divide="ignore"):
        cep = 20 * np.log10(cep)
    cep = cep - (np.mean(cep, axis=0) + 1e-8)
    cep = cep @ dct2_ortho_matrix(cfg.mfbank_num).T
    c = cfg.cepstrum_num
    cep[:, 1:1 + c] -= (np.mean(cep[:, 1:1 + c], axis=0) + 1e-8)
    with np.errstate(divide="ignore"):
        energy = np.log(np.sum(power, axis=1))
    f13 = np.concatenate([cep[:, 1:1 + c], energy[:, None]], axis=1)
    d1 = np.concatenate([f13[1:2], f13[1:] - f13[:-1]], axis=0)
    d2 = np.concatenate([d1[1:2], d1[1:] - d1[:-1]], axis=0)
    return power, cep, np.concatenate([f13, d1, d2], axis=1)


# ----------------------------------------------------------------------------
# G1-G3: Gaussian densities                                    (gmm.py:16-100)
# ----------------------------------------------------------------------------
def gaussian(x, mu, sigma2):
    return 1.0 / np.sqrt(2.0 * np.pi * sigma2) * np.exp(-(x - mu) * (x - mu) / (2.0 * sigma2))


def gaussian_log(x, mu, sigma2):
    return -0.5 * (LOG2PI + np.log(sigma2) + (x - mu) * (x - mu) / sigma2)


def _quad(x, mu, Sigma):
    """(x-mu)^T Sigma^-1 (x-mu) per row with LU semantics: Sigma need not be symmetric
    (gmm.py:62-68; test/gmm-test.py:42-43)."""
    xmu = np.asarray(x) - np.asarray(mu)
    inv = np.linalg.inv(np.asarray(Sigma, dtype=np.float64))
    return np.einsum("ld,ld->l", xmu @ inv, xmu)


def gaussian_multi(x, mu, Sigma):
    D = np.asarray(x).shape[1]
    det = np.linalg.det(np.asarray(Sigma, dtype=np.float64))
    r = 1.0 / (((2.0 * np.pi) ** (D / 2.0)) * (det ** 0.5))
    return r * np.exp(-0.5 * _quad(x, mu, Sigma))


def gaussian_multi_log(x, mu, Sigma):
    """-0.5 (D ln 2pi + ln det Sigma + quad)  (gmm.py:59-68): log of the determinant itself."""
    D = np.asarray(x).shape[1]
    det = np.linalg.det(np.asarray(Sigma, dtype=np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        return -0.5 * (D * LOG2PI + np.log(det) + _quad(x, mu, Sigma))


def gaussian_mixture(c, x, mu, Sigma):
    """Linear mixture, returned as [1, L]  (gmm.py:70-88)."""
    val = np.stack([gaussian_multi(x, mu[k], Sigma[k]) for k in range(len(c))])
    return np.asarray(c).reshape(1, -1) @ val


def gaussian_mixture_log(c, x, mu, Sigma):
    """Log mixture with LOG weights c  (gmm.py:90-100)."""
    val = np.stack([gaussian_multi_log(x, mu[k], Sigma[k]) for k in range(len(c))])
    return lse(np.asarray(c).reshape(-1, 1) + val, axis=0)


# ----------------------------------------------------------------------------
# X2: GMM-HMM emission                                       (gmmhmm.py:46-67)
# ----------------------------------------------------------------------------
def gmm_map_b(obs, w, mu, si, precision=np.float64):
    """Returns (b [N,T], bm [N,M,T]).  `si` is [N,M,D,D]."""
    obs = np.asarray(obs)
    N, M = w.shape
    T = obs.shape[0]
    bm = np.empty((N, M, T), dtype=precision)
    for j in range(N):
        for m in range(M):
            bm[j, m] = gaussian_multi_log(obs, mu[j, m], si[j, m])
    b = np.empty((N, T), dtype=precision)
    for j in range(N):
        b[j] = lse(w[j].reshape(M, 1) + bm[j], axis=0)
    return b, bm


def _nonfinite_frames(obs):
    """Frames holding a NaN or an infinite feature.  The reference multiplies (x - mu) with the FULL inverse matrix
    (gmm.py:63-68): for a diagonal Sigma its off-diagonal zeros meet the infinite entry (inf * 0 = NaN), so every
    component density of such a frame is NaN, not the -inf a diagonal-only sum of squares would give (golden
    `gmmhmm_nonfinite`).  The diagonal shortcuts below restore exactly that.  With D = 1 there is no off-diagonal: an
    infinite feature then simply gives (inf)^2 / var = inf, a density of -inf, which the shortcuts' own arithmetic
    yields; only a NaN makes the frame NaN."""
    obs = np.asarray(obs, dtype=np.float64)
    if obs.shape[1] == 1:
        return np.isnan(obs).any(axis=1)
    return ~np.isfinite(obs).all(axis=1)


def gmm_map_b_diag(obs, w, mu, var, precision=np.float64):
    """Same quantity for diagonal Sigma given as var [N,M,D]; vectorised (the CPU
    baseline's 'best vectorised form', BASELINE.md section 3 line B)."""
    obs = np.asarray(obs, dtype=np.float64)
    N, M, D = mu.shape
    T = obs.shape[0]
    iv = 1.0 / var
    const = -0.5 * (D * LOG2PI + np.log(np.prod(var, axis=2)))        # [N,M]; det = prod of diagonal
    # quad[t,n,m] = sum_d (x-mu)^2 iv  = x^2.iv - 2 x.(mu iv) + mu^2.iv
    q = (obs * obs) @ iv.reshape(N * M, D).T
    q -= 2.0 * (obs @ (mu * iv).reshape(N * M, D).T)
    q += np.sum(mu * mu * iv, axis=2).reshape(1, N * M)
    # the reference stores bm into an array of `precision` FIRST and forms b from those stored values
    # (gmmhmm.py:64-67): with float32 the mixture sum sees the rounded component log-densities
    bm = (const.reshape(1, N * M) - 0.5 * q).reshape(T, N, M).astype(precision)
    bm[_nonfinite_frames(obs)] = np.nan
    if D == 1:                                   # the reference's direct form gives (inf)^2 / var = inf; the expansion above
        bm[np.isinf(obs[:, 0])] = -np.inf        # would cancel inf - inf
    z = w[None] + bm.astype(np.float64)
    mx = z.max(axis=2, keepdims=True)
    b = (mx[..., 0] + np.log(np.exp(z - mx).sum(axis=2)))
    return np.ascontiguousarray(b.T).astype(precision), np.ascontiguousarray(bm.transpose(1, 2, 0))


def gmm_map_b_diag_direct(obs, w, mu, var, precision=np.float64):
    """gmm_map_b_diag with the reference's DIRECT arithmetic, sum_d (x - mu)^2 / var (gmm.py:66-68 for a diagonal
    Sigma), instead of the expanded quadratic: the form to check against when means lie many standard deviations
    from the origin or from each other (golden `gmmhmm_offset_means`), where the expansion cancels large terms."""
    obs = np.asarray(obs, dtype=np.float64)
    N, M, D = mu.shape
    T = obs.shape[0]
    const = -0.5 * (D * LOG2PI + np.log(np.prod(var, axis=2)))
    bm = np.empty((N, M, T), dtype=precision)                 # rounded at the store, as in the reference
    for n in range(N):
        for m in range(M):
            dx = obs - mu[n, m]
            bm[n, m] = const[n, m] - 0.5 * ((dx / var[n, m]) * dx).sum(axis=1)
    bm[:, :, _nonfinite_frames(obs)] = np.nan
    b = np.empty((N, T), dtype=precision)
    for n in range(N):
        b[n] = lse(w[n].reshape(M, 1) + bm[n], axis=0)
    return b, bm


# ----------------------------------------------------------------------------
# H2-H9: HMM trellises                                        (hmm.py:47-283)
# ----------------------------------------------------------------------------
def map_b(B, obs, precision=np.float64):
    """b[:, t] = B[:, obs[t]]  (hmm.py:47-61)."""
    return np.asarray(B)[:, np.asarray(obs)].astype(precision)


def forward(A, pi, b, precision=np.float64):
    """alpha[0]=pi+b[:,0]; alpha[t,j]=lse_i(alpha[t-1,i]+A[i,j])+b[j,t]  (hmm.py:89-99)."""
    N, T = b.shape
    alpha = np.empty((T, N), dtype=precision)
    alpha[0] = pi + b[:, 0]
    for t in range(1, T):
        alpha[t] = lse(alpha[t - 1][:, None] + A, axis=0) + b[:, t]
    return alpha


def backward(A, b, precision=np.float64):
    """beta[T-1]=0; beta[t,i]=lse_j(beta[t+1,j]+A[i,j]+b[j,t+1])  (hmm.py:124-130)."""
    N, T = b.shape
    beta = np.empty((T, N), dtype=precision)
    beta[-1] = 0.0
    for t in range(T - 2, -1, -1):
        beta[t] = lse((beta[t + 1] + A) + b[:, t + 1], axis=1)
    return beta


def viterbi(A, pi, b, precision=np.float64):
    """(v [T,N], bt [T,N] uint32, path [T] uint32); first index wins ties  (hmm.py:155-171)."""
    N, T = b.shape
    v = np.empty((T, N), dtype=precision)
    bt = np.zeros((T, N), dtype=np.uint32)
    v[0] = pi + b[:, 0]
    for t in range(1, T):
        val = v[t - 1][:, None] + A                    # val[i, j]
        v[t] = val.max(axis=0) + b[:, t]
        bt[t] = val.argmax(axis=0)
    path = np.empty(T, dtype=np.uint32)
    path[-1] = np.argmax(v[-1])
    for t in range(T - 2, -1, -1):
        path[t] = bt[t + 1, path[t + 1]]
    return v, bt, path


def loglik(alpha):
    return lse(alpha[-1])                              # hmm.py:274


def baumwelch(A, b, alpha, beta, precision=np.float64):
    """xi [T,N,N], gamma [T,N]; frame T-1 is -inf by construction  (hmm.py:190-199)."""
    N, T = b.shape
    xi = np.full((T, N, N), -np.inf, dtype=precision)
    for t in range(T - 1):
        numer = alpha[t][:, None] + A + b[:, t + 1][None, :] + beta[t + 1][None, :]
        xi[t] = numer - lse(numer)
    gamma = lse(xi, axis=2)
    return xi, gamma


def estimate_discrete(obs, n_sym, xi, gamma, precision=np.float64):
    """A, B, pi re-estimates  (hmm.py:213-219); unseen symbol -> -inf column."""
    obs = np.asarray(obs)
    N = gamma.shape[1]
    denom = lse(gamma, axis=0)
    A = lse(xi, axis=0) - denom.reshape(N, 1)
    B = np.empty((N, n_sym), dtype=precision)
    for k in range(n_sym):
        sel = gamma[obs == k, :]
        B[:, k] = (lse(sel, axis=0) if sel.shape[0] else np.full(N, -np.inf)) - denom
    return {"A": A, "B": B, "pi": gamma[0]}


def estimate_gmm(obs, w, mu_old, bm, alpha, beta, xi, gamma, min_std=0.01, precision=np.float64):
    """A, pi, w, mu, si re-estimates  (gmmhmm.py:69-156), streamed over (n, m).

    xi_mix uses alpha+beta, so frame T-1 IS included (unlike gamma); Sigma uses the
    OLD mean (gmmhmm.py:138) and adds min_std * I (gmmhmm.py:147)."""
    obs = np.asarray(obs, dtype=np.float64)
    T, D = obs.shape
    N, M = w.shape
    A = lse(xi, axis=0) - lse(gamma, axis=0).reshape(N, 1)
    pi = gamma[0]
    ab = alpha + beta                                              # [T,N]
    z = w[None] + np.transpose(bm, (2, 0, 1))                      # [T,N,M]
    xi_mix = np.empty((T, N, M), dtype=precision)
    for t in range(T):
        x = ab[t].reshape(N, 1) + z[t]
        x = x - lse(ab[t])
        x = x - lse(z[t], axis=1).reshape(N, 1)
        xi_mix[t] = x
    w_new = lse(xi_mix, axis=0) - lse(xi_mix, axis=(0, 2)).reshape(N, 1)
    g = np.exp(xi_mix)                                             # [T,N,M]
    gsum = g.sum(axis=0)                                           # [N,M]
    mu = np.einsum("tnm,td->nmd", g, obs) / gsum[..., None]
    si = np.empty((N, M, D, D), dtype=np.result_type(precision, np.float64))
    for n in range(N):
        for m in range(M):
            dt = obs - mu_old[n, m]
            si[n, m] = (dt * g[:, n, m][:, None]).T @ dt / gsum[n, m]
    si += min_std * np.eye(D)
    return {"A": A, "pi": pi, "w": w_new, "mu": mu, "si": si}


# ----------------------------------------------------------------------------
# Multi-utterance sufficient statistics: linear-domain sums of the
# per-utterance posteriors above; with U=1 they reproduce estimate_gmm.
# ----------------------------------------------------------------------------
def bw_stats_diag(obs_batch, A, pi, w, mu, var):
    """Accumulate over utterances: returns dict of
       xi_sum [N,N], gamma_sum [N] (t < T-1), gamma0 [N], mix [N,M], mix_x [N,M,D],
       mix_xx [N,M,D] (around the OLD mean), loglik (sum), n_utt."""
    N, M, D = mu.shape
    st = dict(xi_sum=np.zeros((N, N)), gamma_sum=np.zeros(N), gamma0=np.zeros(N),
              mix=np.zeros((N, M)), mix_x=np.zeros((N, M, D)), mix_xx=np.zeros((N, M, D)),
              loglik=0.0, n_utt=0)
    for obs in obs_batch:
        b, bm = gmm_map_b_diag(obs, w, mu, var)
        alpha = forward(A, pi, b)
        beta = backward(A, b)
        xi, gamma = baumwelch(A, b, alpha, beta)
        T = obs.shape[0]
        ab = alpha + beta
        z = w[None] + np.transpose(bm, (2, 0, 1))
        xm = ab[:, :, None] + z - lse(ab, axis=1)[:, None, None] - lse(z, axis=2)[:, :, None]
        g = np.exp(xm)
        with np.errstate(under="ignore"):
            st["xi_sum"] += np.exp(xi[:T - 1]).sum(axis=0)
            st["gamma_sum"] += np.exp(gamma[:T - 1]).sum(axis=0)
            st["gamma0"] += np.exp(gamma[0])
        st["mix"] += g.sum(axis=0)
        st["mix_x"] += np.einsum("tnm,td->nmd", g, obs)
        for n in range(N):
            for m in range(M):
                dt = obs - mu[n, m]
                st["mix_xx"][n, m] += (g[:, n, m][:, None] * dt * dt).sum(axis=0)
        st["loglik"] += float(loglik(alpha))
        st["n_utt"] += 1
    return st


def mstep_from_stats(st, min_std=0.01):
    """The M-step every rank runs on the all-reduced statistics (diagonal Sigma)."""
    with np.errstate(divide="ignore"):
        A = np.log(st["xi_sum"]) - np.log(st["gamma_sum"])[:, None]
        pi = np.log(st["gamma0"] / st["n_utt"])
        w = np.log(st["mix"]) - np.log(st["mix"].sum(axis=1))[:, None]
    mu = st["mix_x"] / st["mix"][..., None]
    var = st["mix_xx"] / st["mix"][..., None] + min_std
    return {"A": A, "pi": pi, "w": w, "mu": mu, "var": var}


# ----------------------------------------------------------------------------
# Whole-path helpers used by tests and by bench.py's cpu_baseline leg
# ----------------------------------------------------------------------------
def decode_diag(obs, A, pi, w, mu, var):
    """GMMHMM.decode (hmm.py:276-283) with the vectorised emission: path [T] uint32."""
    b, _ = gmm_map_b_diag(obs, w, mu, var)
    return viterbi(A, pi, b)[2]


def prep_diag_fused(w, mu, var):
    """Per-model constant for emit_b_fused: coefficient matrix [2D+1, N*M] of the expanded
    quadratic, rows = (-1/2 iv | mu iv | log w + const)."""
    N, M, D = mu.shape
    iv = 1.0 / var
    const = w - 0.5 * (D * LOG2PI + np.log(np.prod(var, axis=2)) + np.sum(mu * mu * iv, axis=2))
    return np.ascontiguousarray(np.concatenate(
        [(-0.5 * iv).reshape(N * M, D).T, (mu * iv).reshape(N * M, D).T, const.reshape(1, N * M)], axis=0))


def emit_b_fused(obs, W, N, M):
    """b [N,T] only (no bm): one GEMM [x^2 | x | 1] @ W followed by the M-way log-sum-exp --
    the fastest numpy form of gmmhmm.py:46-67 for diagonal Sigma (CPU baseline line B)."""
    T = obs.shape[0]
    F = np.concatenate([obs * obs, obs, np.ones((T, 1))], axis=1)
    z = (F @ W).reshape(T, N, M)
    mx = z.max(axis=2)
    z -= mx[:, :, None]
    np.exp(z, out=z)
    return np.ascontiguousarray((mx + np.log(z.sum(axis=2))).T)


def decode_diag_fused(obs, A, pi, W, N, M):
    return viterbi(A, pi, emit_b_fused(obs, W, N, M))[2]


def decode_refshaped(obs, A, pi, w, mu, si):
    """Reference-SHAPED loop structure (per-(j,m) density calls, per-(t,j) trellis
    updates: gmmhmm.py:64-67, hmm.py:162-166) -- BASELINE.md section 3 line A.
    Only used for timing; results equal decode_diag's."""
    N, M = w.shape
    T = obs.shape[0]
    b, _ = gmm_map_b(obs, w, mu, si)
    v = np.empty((T, N))
    bt = np.zeros((T, N), dtype=np.uint32)
    v[0] = pi + b[:, 0]
    for t in range(1, T):
        for j in range(N):
            val = v[t - 1] + A[:, j]
            v[t, j] = np.max(val) + b[j, t]
            bt[t, j] = np.argmax(val)
    path = np.empty(T, dtype=np.uint32)
    path[-1] = np.argmax(v[-1])
    for t in range(T - 2, -1, -1):
        path[t] = bt[t + 1, path[t + 1]]
    return path


def train_hmm_discrete(A, B, pi, obs, iters=1, eps=1e-4, log=None):
    """HMM.train (hmm.py:285-313) as a pure function; `log` collects the printed lines."""
    n_sym = B.shape[1]
    for k in range(iters):
        b = map_b(B, obs)
        alpha = forward(A, pi, b)
        beta = backward(A, b)
        xi, gamma = baumwelch(A, b, alpha, beta)
        with np.errstate(divide="ignore", invalid="ignore"):
            mdl = estimate_discrete(obs, n_sym, xi, gamma)
        old = lse(alpha[-1])
        A, B, pi = mdl["A"], mdl["B"], mdl["pi"]
        new = loglik(forward(A, pi, map_b(B, obs)))
        d = abs(old - new)
        if log is not None:
            log.append("Iter: {:3}, L(lambda|O) = {:.6e}, L(lambda_new|O) = {:.6e}, eps = {:.6f}"
                       .format(k, old, new, d))
        if d < eps:
            break
    return A, B, pi


def train_gmmhmm(A, pi, w, mu, si, obs, iters=1, eps=1e-4, log=None):
    """GMMHMM.train via hmm.py:285-313 + gmmhmm.py:69-164 as a pure function."""
    for k in range(iters):
        b, bm = gmm_map_b(obs, w, mu, si)
        alpha = forward(A, pi, b)
        beta = backward(A, b)
        xi, gamma = baumwelch(A, b, alpha, beta)
        mdl = estimate_gmm(obs, w, mu, bm, alpha, beta, xi, gamma)
        old = lse(alpha[-1])
        A, pi, w, mu, si = mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"]
        new = loglik(forward(A, pi, gmm_map_b(obs, w, mu, si)[0]))
        d = abs(old - new)
        if log is not None:
            log.append("Iter: {:3}, L(lambda|O) = {:.6e}, L(lambda_new|O) = {:.6e}, eps = {:.6f}"
                       .format(k, old, new, d))
        if d < eps:
            break
    return A, pi, w, mu, si
