// Log-sum-exp utility, Baum-Welch M-steps (hmm.py:202-225, gmmhmm.py:69-156) and the
// multi-utterance sufficient-statistics path.
#include <utility>

#include <algorithm>

#include "common.h"
#include "devmath.h"

namespace lnasr {

int emit_device(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm);
int device_lens(lnasr_ctx* ctx, const int32_t* lens, int U, int T, const int32_t** out);
int forward_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* alpha,
                   double* loglik);
int backward_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* beta);

// ------------------------------------------------------------------------------------------
// lse / lse2 (utils.py:140-204) over the middle axis of [outer, n, inner]
// ------------------------------------------------------------------------------------------
template <bool BASE2>
struct LseAcc {
    double m = LN_NEG_INF, s = 0.0;
    __device__ __forceinline__ void add(double z) {
        if (z == LN_NEG_INF) return;
        if (z > m) {
            s = s * (BASE2 ? exp2(m - z) : exp(m - z)) + 1.0;
            m = z;
        } else if (z <= m) {
            s += BASE2 ? exp2(z - m) : exp(z - m);
        } else if (m == m) {
            m = z;                      // a NaN term poisons the sum, as np.logaddexp does (see Lse, devmath.h)
        }
    }
    __device__ __forceinline__ void merge(double om, double os) {
        if (om == LN_NEG_INF) return;
        if (om > m) {
            s = s * (BASE2 ? exp2(m - om) : exp(m - om)) + os;
            m = om;
        } else if (om <= m) {
            s += os * (BASE2 ? exp2(om - m) : exp(om - m));
        } else if (m == m) {
            m = om;
        }
    }
    __device__ __forceinline__ double value() const {
        return (m == LN_NEG_INF) ? m : m + (BASE2 ? log2(s) : log(s));
    }
};

template <bool BASE2>
__global__ void lse_strided_kernel(const double* __restrict__ x, size_t outer, size_t n, size_t inner,
                                   double* __restrict__ out) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= outer * inner) return;
    size_t o = idx / inner, i = idx % inner;
    const double* p = x + o * n * inner + i;
    LseAcc<BASE2> acc;
    for (size_t k = 0; k < n; ++k) acc.add(p[k * inner]);
    out[idx] = acc.value();
}

// inner == 1: one wave per row
template <bool BASE2>
__global__ void lse_rows_kernel(const double* __restrict__ x, size_t outer, size_t n, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= outer) return;
    const double* p = x + row * n;
    LseAcc<BASE2> acc;
    for (size_t k = lane; k < n; k += 64) acc.add(p[k]);
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
        double om = shfl_xor_f64(acc.m, msk), os = shfl_xor_f64(acc.s, msk);
        acc.merge(om, os);
    }
    if (lane == 0) out[row] = acc.value();
}

// ------------------------------------------------------------------------------------------
// M-step pieces
// ------------------------------------------------------------------------------------------
// out[c] = lse_t X[t, c], t < T  (thread per column)
template <typename TS>
__global__ void col_lse_kernel(const TS* __restrict__ X, int T, size_t C, double* __restrict__ out) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    Lse acc;
    for (int t = 0; t < T; ++t) acc.add(ld(X, (size_t)t * C + c));
    out[c] = acc.value();
}

// A[i,j] = numer[i,j] - denom[i]; pi = gamma[0]                     (hmm.py:213-214,219)
template <typename TS>
__global__ void finish_A_pi_kernel(const double* __restrict__ numer, const double* __restrict__ denom,
                                   const TS* __restrict__ gamma0, int N, TS* __restrict__ A, TS* __restrict__ pi) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < N * N) st(A, idx, numer[idx] - denom[idx / N]);
    if (idx < N) pi[idx] = gamma0[idx];
}

__device__ __forceinline__ void atomic_max_f64(double* addr, double val) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *a;
    while (true) {
        double cur = __longlong_as_double((long long)old);
        if (!(val > cur)) return;
        unsigned long long prev = atomicCAS(a, old, (unsigned long long)__double_as_longlong(val));
        if (prev == old) return;
        old = prev;
    }
}

__global__ void fill_f64_kernel(double* p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// B[i,k] = lse_{t: obs[t]==k} gamma[t,i] - denom[i]                  (hmm.py:215-218)
template <typename TS>
__global__ void emit_B_max_kernel(const TS* __restrict__ gamma, const int32_t* __restrict__ obs, int T, int N, int K,
                                  double* __restrict__ mx) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * N) return;
    int t = idx / N, i = idx % N;
    int k = obs[t];
    if (k < 0 || k >= K) return;
    double g = ld(gamma, idx);
    if (g != LN_NEG_INF) atomic_max_f64(&mx[(size_t)i * K + k], g);
}
template <typename TS>
__global__ void emit_B_sum_kernel(const TS* __restrict__ gamma, const int32_t* __restrict__ obs, int T, int N, int K,
                                  const double* __restrict__ mx, double* __restrict__ sm) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * N) return;
    int t = idx / N, i = idx % N;
    int k = obs[t];
    if (k < 0 || k >= K) return;
    double g = ld(gamma, idx);
    if (g != LN_NEG_INF) atomicAdd(&sm[(size_t)i * K + k], exp(g - mx[(size_t)i * K + k]));
}
template <typename TS>
__global__ void emit_B_finish_kernel(const double* __restrict__ mx, const double* __restrict__ sm,
                                     const double* __restrict__ denom, int N, int K, TS* __restrict__ B) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * K) return;
    double m = mx[idx];
    double v = (m == LN_NEG_INF) ? m : m + log(sm[idx]);
    st(B, idx, v - denom[idx / K]);
}

// c1[t] = lse_j(alpha[t,j] + beta[t,j])                              (gmmhmm.py:99)  wave per frame
template <typename TS>
__global__ void ab_lse_kernel(const TS* __restrict__ alpha, const TS* __restrict__ beta, long F, int N,
                              double* __restrict__ c1) {
    const int lane = threadIdx.x & 63;
    long f = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (f >= F) return;
    Lse acc;
    for (int j = lane; j < N; j += 64) acc.add(ld(alpha, (size_t)f * N + j) + ld(beta, (size_t)f * N + j));
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
        double om = shfl_xor_f64(acc.m, msk), os = shfl_xor_f64(acc.s, msk);
        acc.merge(om, os);
    }
    if (lane == 0) c1[f] = acc.value();
}

// xi_mix[t,j,m] = (alpha+beta)[t,j] + w[j,m] + bm[t,j,m] - c1[t] - lse_m(w[j,:] + bm[t,j,:])   (gmmhmm.py:98-100)
template <typename TS>
__global__ void xi_mix_kernel(const TS* __restrict__ alpha, const TS* __restrict__ beta, const TS* __restrict__ bm,
                              const double* __restrict__ logw, const double* __restrict__ c1, long F, int N, int M,
                              TS* __restrict__ xm) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F * N) return;
    long f = idx / N;
    int j = (int)(idx % N);
    const double ab = ld(alpha, idx) + ld(beta, idx);
    Lse acc;
    for (int m = 0; m < M; ++m) acc.add(logw[(size_t)j * M + m] + ld(bm, (size_t)idx * M + m));
    const double c2 = acc.value();
    for (int m = 0; m < M; ++m) {
        double v = (ab + logw[(size_t)j * M + m]) + ld(bm, (size_t)idx * M + m);
        v = (v - c1[f]) - c2;
        st(xm, (size_t)idx * M + m, v);
    }
}

// w[j,m] = wn[j,m] - lse_m wn[j,:]                                   (gmmhmm.py:103)
template <typename TS>
__global__ void finish_w_kernel(const double* __restrict__ wn, int N, int M, TS* __restrict__ w) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    Lse acc;
    for (int m = 0; m < M; ++m) acc.add(wn[(size_t)j * M + m]);
    const double d = acc.value();
    for (int m = 0; m < M; ++m) st(w, (size_t)j * M + m, wn[(size_t)j * M + m] - d);
}

// Full-covariance mean / Sigma re-estimate for one component per block     (gmmhmm.py:106-147)
//   mu   = sum_t g_t o_t / sum_t g_t,  g_t = exp(xi_mix[t,c])
//   Sigma= sum_t g_t (o_t - mu_old)(o_t - mu_old)^T / sum_t g_t + min_std I
constexpr int kPairAcc = 8;
template <typename TS>
__global__ void mstep_full_kernel(const double* __restrict__ obs, const TS* __restrict__ xm,
                                  const double* __restrict__ mu_old, int T, int NM, int D, double min_std,
                                  double* __restrict__ mu_out, double* __restrict__ si_out) {
    extern __shared__ double sm[];
    constexpr int CH = 64;
    double* g = sm;                    // [CH]
    double* dt = sm + CH;              // [CH][D]  o - mu_old
    double* red = sm + CH + CH * D;    // [32]
    const int c = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double* mo = mu_old + (size_t)c * D;
    const int DD = D * D;
    const int npass = (DD + nt * kPairAcc - 1) / (nt * kPairAcc);
    double gsum = 0.0;
    for (int pass = 0; pass < npass; ++pass) {
        double acc[kPairAcc];
        double macc = 0.0;             // mean accumulator for d = tid (pass 0 only)
#pragma unroll
        for (int r = 0; r < kPairAcc; ++r) acc[r] = 0.0;
        double gl = 0.0;
        for (int t0 = 0; t0 < T; t0 += CH) {
            const int nch = (T - t0) < CH ? (T - t0) : CH;
            __syncthreads();
            for (int e = tid; e < nch; e += nt) g[e] = exp(ld(xm, (size_t)(t0 + e) * NM + c));
            for (int e = tid; e < nch * D; e += nt) {
                int f = e / D, d = e % D;
                dt[e] = obs[(size_t)(t0 + f) * D + d] - mo[d];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kPairAcc; ++r) {
                const int p = pass * nt * kPairAcc + r * nt + tid;
                if (p < DD) {
                    const int d = p / D, e = p % D;
                    double a = acc[r];
                    for (int f = 0; f < nch; ++f) a += g[f] * (dt[f * D + d] * dt[f * D + e]);
                    acc[r] = a;
                }
            }
            if (pass == 0) {
                if (tid < D)
                    for (int f = 0; f < nch; ++f) macc += g[f] * obs[(size_t)(t0 + f) * D + tid];
                if (tid == 0)
                    for (int f = 0; f < nch; ++f) gl += g[f];
            }
        }
        if (pass == 0) {
            __syncthreads();
            if (tid == 0) red[0] = gl;
            __syncthreads();
            gsum = red[0];
            if (tid < D) mu_out[(size_t)c * D + tid] = macc / gsum;
        }
#pragma unroll
        for (int r = 0; r < kPairAcc; ++r) {
            const int p = pass * nt * kPairAcc + r * nt + tid;
            if (p < DD) {
                const int d = p / D, e = p % D;
                si_out[(size_t)c * DD + p] = acc[r] / gsum + (d == e ? min_std : 0.0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// multi-utterance statistics (diagonal model), linear domain
// ------------------------------------------------------------------------------------------
struct StatsLayout {
    size_t xi, gsum, g0, mix, mix_x, mix_xx, ll, nutt, nfr, total;
};
static StatsLayout stats_layout(int N, int M, int D, bool diag) {
    StatsLayout L;
    size_t o = 0;
    L.xi = o; o += (size_t)N * N;
    L.gsum = o; o += N;
    L.g0 = o; o += N;
    L.mix = o; o += (size_t)N * M;
    L.mix_x = o; o += (size_t)N * M * D;
    L.mix_xx = o; o += diag ? (size_t)N * M * D : (size_t)N * M * D * D;
    L.ll = o; o += 1;
    L.nutt = o; o += 1;
    L.nfr = o; o += 1;
    L.total = o;
    return L;
}

// xi_sum[i,j] += sum_{t < L-1} exp(alpha[t,i] + A[i,j] + b[t+1,j] + beta[t+1,j] - loglik)
// block = 16x16 tile of (i,j) for one utterance
__global__ void acc_xi_kernel(const double* __restrict__ alpha, const double* __restrict__ beta,
                              const double* __restrict__ b, const double* __restrict__ A,
                              const double* __restrict__ loglik, const int32_t* __restrict__ lens, int T, int N,
                              double* __restrict__ xi_sum) {
    __shared__ double sa[16], sc[16];
    const int u = blockIdx.z;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int i = i0 + ti, j = j0 + tj;
    const int L = lens[u];
    const double ll = loglik[u];
    const double a = (i < N && j < N) ? A[(size_t)i * N + j] : LN_NEG_INF;
    const double* al = alpha + (size_t)u * T * N;
    const double* be = beta + (size_t)u * T * N;
    const double* bb = b + (size_t)u * T * N;
    double acc = 0.0;
    for (int t = 0; t + 1 < L; ++t) {
        __syncthreads();
        if (threadIdx.x < 16) {
            int ii = i0 + threadIdx.x;
            sa[threadIdx.x] = ii < N ? al[(size_t)t * N + ii] : LN_NEG_INF;
        } else if (threadIdx.x < 32) {
            int jj = j0 + threadIdx.x - 16;
            sc[threadIdx.x - 16] = jj < N ? bb[(size_t)(t + 1) * N + jj] + be[(size_t)(t + 1) * N + jj] : LN_NEG_INF;
        }
        __syncthreads();
        const double z = ((sa[ti] + a) + sc[tj]) - ll;
        if (z != LN_NEG_INF && z == z) acc += exp(z);
    }
    if (i < N && j < N && acc != 0.0) atomicAdd(&xi_sum[(size_t)i * N + j], acc);
}

// gamma0[i] += exp(alpha[0,i] + beta[0,i] - loglik) for utterances with at least two frames
// (gamma = lse_j xi is -inf on the last frame: hmm.py:190,199); loglik_sum, n_utt, n_frames
__global__ void acc_gamma0_kernel(const double* __restrict__ alpha, const double* __restrict__ beta,
                                  const double* __restrict__ loglik, const int32_t* __restrict__ lens, int U, int T,
                                  int N, double* __restrict__ g0, double* __restrict__ ll_sum,
                                  double* __restrict__ nutt, double* __restrict__ nfr) {
    const int u = blockIdx.x;
    const int L = lens[u];
    if (L >= 2)
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            double z = alpha[(size_t)u * T * N + i] + beta[(size_t)u * T * N + i] - loglik[u];
            if (z != LN_NEG_INF) atomicAdd(&g0[i], exp(z));
        }
    if (threadIdx.x == 0) {
        atomicAdd(ll_sum, loglik[u]);
        atomicAdd(nutt, 1.0);
        atomicAdd(nfr, (double)L);
    }
}

// mixture posteriors and first/second moments for one state over a chunk of frames:
//   g[f,m] = exp(alpha+beta - c1[f] + w[j,m] + bm[f,j,m] - b[f,j])
//   mix[j,m] += g; mix_x[j,m,d] += g o_d; mix_xx[j,m,d] += g (o_d - mu_old)^2
constexpr int kMixCH = 128;
__global__ void acc_mix_diag_kernel(const double* __restrict__ obs, const double* __restrict__ alpha,
                                    const double* __restrict__ beta, const double* __restrict__ b,
                                    const double* __restrict__ bm, const double* __restrict__ c1,
                                    const double* __restrict__ logw, const double* __restrict__ mu,
                                    const int32_t* __restrict__ lens, int T, int N, int M, int D,
                                    double* __restrict__ mix, double* __restrict__ mix_x,
                                    double* __restrict__ mix_xx) {
    extern __shared__ double sm[];
    double* g = sm;                          // [CH][M]
    double* ob = sm + kMixCH * M;            // [CH][D]
    const int j = blockIdx.x;
    const int u = blockIdx.z;
    const int t0 = blockIdx.y * kMixCH;
    const int L = lens[u];
    if (t0 >= L) return;
    const int nch = (L - t0) < kMixCH ? (L - t0) : kMixCH;
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t fbase = (size_t)u * T + t0;
    for (int e = tid; e < nch * M; e += nt) {
        const int f = e / M, m = e % M;
        const size_t fr = fbase + f;
        const size_t fj = fr * N + j;
        double z = (alpha[fj] + beta[fj] - c1[fr]) + (logw[(size_t)j * M + m] + bm[fj * M + m] - b[fj]);
        g[e] = (z == LN_NEG_INF || z != z) ? 0.0 : exp(z);
    }
    for (int e = tid; e < nch * D; e += nt) ob[e] = obs[fbase * D + e];
    __syncthreads();
    const int MD = M * D;
    for (int p = tid; p < MD; p += nt) {
        const int m = p / D, d = p % D;
        const double mo = mu[((size_t)j * M + m) * D + d];
        double sx = 0.0, sxx = 0.0;
        for (int f = 0; f < nch; ++f) {
            const double gv = g[f * M + m], o = ob[f * D + d];
            sx += gv * o;
            sxx += gv * ((o - mo) * (o - mo));
        }
        atomicAdd(&mix_x[((size_t)j * M + m) * D + d], sx);
        atomicAdd(&mix_xx[((size_t)j * M + m) * D + d], sxx);
    }
    for (int m = tid; m < M; m += nt) {
        double s = 0.0;
        for (int f = 0; f < nch; ++f) s += g[f * M + m];
        atomicAdd(&mix[(size_t)j * M + m], s);
    }
}

// full covariance: one block per (component c, utterance u).  Posteriors g[t] of the whole utterance sit in
// LDS, observations stream through in 64-frame chunks, every thread keeps up to kPairAcc (d,e) pairs of the
// D x D second-moment matrix (around the OLD mean, gmmhmm.py:138) in registers; one atomic per pair per block.
__global__ void acc_full_kernel(const double* __restrict__ obs, const double* __restrict__ alpha,
                                const double* __restrict__ beta, const double* __restrict__ b,
                                const double* __restrict__ bm, const double* __restrict__ c1,
                                const double* __restrict__ logw, const double* __restrict__ mu,
                                const int32_t* __restrict__ lens, int T, int N, int M, int D,
                                double* __restrict__ mix, double* __restrict__ mix_x, double* __restrict__ mix_xx) {
    extern __shared__ double sm[];
    constexpr int CH = 64;
    double* g = sm;                    // [T]
    double* dt = sm + T;               // [CH][D]
    const int c = blockIdx.x, u = blockIdx.y;
    const int j = c / M, m = c % M;
    const int L = lens[u];
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t fbase = (size_t)u * T;
    const double* mo = mu + (size_t)c * D;
    for (int t = tid; t < L; t += nt) {
        const size_t fr = fbase + t, fj = fr * N + j;
        const double z = (alpha[fj] + beta[fj] - c1[fr]) + (logw[c] + bm[fj * M + m] - b[fj]);
        g[t] = (z == LN_NEG_INF || z != z) ? 0.0 : exp(z);
    }
    __syncthreads();
    const int DD = D * D;
    const int npass = (DD + nt * kPairAcc - 1) / (nt * kPairAcc);
    for (int pass = 0; pass < npass; ++pass) {
        double acc[kPairAcc];
        double macc = 0.0, gacc = 0.0;
#pragma unroll
        for (int r = 0; r < kPairAcc; ++r) acc[r] = 0.0;
        for (int t0 = 0; t0 < L; t0 += CH) {
            const int nch = (L - t0) < CH ? (L - t0) : CH;
            __syncthreads();
            for (int e = tid; e < nch * D; e += nt) dt[e] = obs[(fbase + t0) * D + e] - mo[e % D];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kPairAcc; ++r) {
                const int p = pass * nt * kPairAcc + r * nt + tid;
                if (p < DD) {
                    const int d = p / D, e = p % D;
                    double a = acc[r];
                    for (int f = 0; f < nch; ++f) a += g[t0 + f] * (dt[f * D + d] * dt[f * D + e]);
                    acc[r] = a;
                }
            }
            if (pass == 0) {
                if (tid < D)
                    for (int f = 0; f < nch; ++f) macc += g[t0 + f] * obs[(fbase + t0 + f) * D + tid];
                if (tid == D)
                    for (int f = 0; f < nch; ++f) gacc += g[t0 + f];
            }
        }
        if (pass == 0) {
            if (tid < D) atomicAdd(&mix_x[(size_t)c * D + tid], macc);
            if (tid == D) atomicAdd(&mix[c], gacc);
        }
#pragma unroll
        for (int r = 0; r < kPairAcc; ++r) {
            const int p = pass * nt * kPairAcc + r * nt + tid;
            if (p < DD) atomicAdd(&mix_xx[(size_t)c * DD + p], acc[r]);
        }
    }
}

__global__ void rowsum_kernel(const double* __restrict__ X, int N, double* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double s = 0.0;
    for (int j = 0; j < N; ++j) s += X[(size_t)i * N + j];
    out[i] = s;
}

// M-step on (all-reduced) statistics, diagonal model
__global__ void mstep_stats_kernel(const double* __restrict__ st, StatsLayout L, int N, int M, int D, int diag,
                                   double min_std,
                                   double* __restrict__ A, double* __restrict__ pi, double* __restrict__ w,
                                   double* __restrict__ mu, double* __restrict__ var) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)N * N) A[idx] = log(st[L.xi + idx]) - log(st[L.gsum + idx / N]);
    if (idx < (size_t)N) pi[idx] = log(st[L.g0 + idx] / st[L.nutt]);
    if (idx < (size_t)N * M) {
        size_t j = idx / M;
        double s = 0.0;
        for (int m = 0; m < M; ++m) s += st[L.mix + j * M + m];
        w[idx] = log(st[L.mix + idx]) - log(s);
    }
    if (idx < (size_t)N * M * D) {
        double g = st[L.mix + idx / D];
        mu[idx] = st[L.mix_x + idx] / g;
        if (diag) var[idx] = st[L.mix_xx + idx] / g + min_std;
    }
    if (!diag && idx < (size_t)N * M * D * D) {
        const size_t c = idx / ((size_t)D * D), de = idx % ((size_t)D * D);
        var[idx] = st[L.mix_xx + idx] / st[L.mix + c] + ((de / D == de % D) ? min_std : 0.0);   // gmmhmm.py:147
    }
}

}  // namespace lnasr

using namespace lnasr;

extern "C" {

int lnasr_lse(lnasr_ctx* ctx, const double* x, size_t outer, size_t n, size_t inner, int base2, double* out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    const size_t no = outer * inner;
    if (no == 0) return 0;
    LN_REQUIRE(out, "null out");
    const void* d_x;
    void* d_o;
    LN_TRY(stage_in(ctx, x, outer * n * inner * 8, ctx->ws_in, &d_x));
    LN_TRY(stage_out(ctx, out, no * 8, ctx->ws_out, &d_o));
    if (inner == 1 && n >= 128) {
        const unsigned grid = (unsigned)((outer + 3) / 4);
        if (base2)
            LN_LAUNCH(ctx, "lse2_rows", lse_rows_kernel<true>, dim3(grid), dim3(256), 0, (const double*)d_x, outer, n,
                      (double*)d_o);
        else
            LN_LAUNCH(ctx, "lse_rows", lse_rows_kernel<false>, dim3(grid), dim3(256), 0, (const double*)d_x, outer, n,
                      (double*)d_o);
    } else {
        const unsigned grid = (unsigned)((no + 255) / 256);
        if (base2)
            LN_LAUNCH(ctx, "lse2_strided", lse_strided_kernel<true>, dim3(grid), dim3(256), 0, (const double*)d_x,
                      outer, n, inner, (double*)d_o);
        else
            LN_LAUNCH(ctx, "lse_strided", lse_strided_kernel<false>, dim3(grid), dim3(256), 0, (const double*)d_x,
                      outer, n, inner, (double*)d_o);
    }
    return finish_out(ctx, out, d_o, no * 8);
}

}  // extern "C"

namespace lnasr {

template <typename TS>
static int estimate_discrete_impl(lnasr_ctx* ctx, const int32_t* obs, int T, int N, int K, const void* xi,
                                  const void* gamma, void* A_out, void* B_out, void* pi_out) {
    const size_t es = sizeof(TS);
    const void *d_obs, *d_xi, *d_ga;
    LN_TRY(stage_in(ctx, obs, (size_t)T * 4, ctx->ws_in, &d_obs));
    LN_TRY(stage_in(ctx, xi, (size_t)T * N * N * es, ctx->ws_in2, &d_xi));
    LN_TRY(stage_in(ctx, gamma, (size_t)T * N * es, ctx->ws_in3, &d_ga));
    // scratch: denom[N] numer[N*N] mx[N*K] sm[N*K]
    const size_t nk = (size_t)N * K;
    LN_TRY(ctx->ws_misc.reserve(((size_t)N + (size_t)N * N + 2 * nk) * 8));
    double* denom = ctx->ws_misc.as<double>();
    double* numer = denom + N;
    double* mx = numer + (size_t)N * N;
    double* sm = mx + nk;
    void *dA, *dB, *dpi;
    LN_TRY(stage_out(ctx, A_out, (size_t)N * N * es, ctx->ws_out, &dA));
    LN_TRY(stage_out(ctx, B_out, nk * es, ctx->ws_out2, &dB));
    LN_TRY(stage_out(ctx, pi_out, (size_t)N * es, ctx->ws_out3, &dpi));
    LN_LAUNCH(ctx, "col_lse", col_lse_kernel<TS>, dim3((N + 63) / 64), dim3(64), 0, (const TS*)d_ga, T, (size_t)N,
              denom);
    LN_LAUNCH(ctx, "col_lse", col_lse_kernel<TS>, dim3((N * N + 63) / 64), dim3(64), 0, (const TS*)d_xi, T,
              (size_t)N * N, numer);
    LN_LAUNCH(ctx, "finish_A_pi", finish_A_pi_kernel<TS>, dim3((N * N + 255) / 256), dim3(256), 0, numer, denom,
              (const TS*)d_ga, N, (TS*)dA, (TS*)dpi);
    LN_LAUNCH(ctx, "fill_f64", fill_f64_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, mx, nk, LN_NEG_INF);
    LN_HIP(hipMemsetAsync(sm, 0, nk * 8, ctx->stream));
    const unsigned gtn = (unsigned)(((size_t)T * N + 255) / 256);
    LN_LAUNCH(ctx, "emit_B_max", emit_B_max_kernel<TS>, dim3(gtn), dim3(256), 0, (const TS*)d_ga,
              (const int32_t*)d_obs, T, N, K, mx);
    LN_LAUNCH(ctx, "emit_B_sum", emit_B_sum_kernel<TS>, dim3(gtn), dim3(256), 0, (const TS*)d_ga,
              (const int32_t*)d_obs, T, N, K, mx, sm);
    LN_LAUNCH(ctx, "emit_B_finish", emit_B_finish_kernel<TS>, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, mx, sm,
              denom, N, K, (TS*)dB);
    LN_TRY(finish_out(ctx, A_out, dA, (size_t)N * N * es));
    LN_TRY(finish_out(ctx, B_out, dB, nk * es));
    return finish_out(ctx, pi_out, dpi, (size_t)N * es);
}

template <typename TS>
static int estimate_gmm_impl(lnasr_ctx* ctx, const double* obs, int T, const void* bm, const void* alpha,
                             const void* beta, const void* xi, const void* gamma, double min_std, void* A_out,
                             void* pi_out, void* w_out, double* mu_out, double* si_out) {
    Model& md = ctx->model;
    const int N = md.N, M = md.M, D = md.D;
    const size_t es = sizeof(TS), NM = (size_t)N * M;
    const void *d_obs, *d_bm, *d_al, *d_be, *d_xi, *d_ga;
    LN_TRY(stage_in(ctx, obs, (size_t)T * D * 8, ctx->ws_in, &d_obs));
    LN_TRY(stage_in(ctx, xi, (size_t)T * N * N * es, ctx->ws_in2, &d_xi));
    LN_TRY(stage_in(ctx, gamma, (size_t)T * N * es, ctx->ws_in3, &d_ga));
    LN_TRY(stage_in(ctx, bm, (size_t)T * NM * es, ctx->ws_in4, &d_bm));
    LN_TRY(ctx->ws_alpha.reserve((size_t)T * N * es));
    LN_TRY(ctx->ws_beta.reserve((size_t)T * N * es));
    if (is_device_ptr(alpha)) d_al = alpha; else {
        LN_HIP(hipMemcpyAsync(ctx->ws_alpha.p, alpha, (size_t)T * N * es, hipMemcpyHostToDevice, ctx->stream));
        d_al = ctx->ws_alpha.p;
    }
    if (is_device_ptr(beta)) d_be = beta; else {
        LN_HIP(hipMemcpyAsync(ctx->ws_beta.p, beta, (size_t)T * N * es, hipMemcpyHostToDevice, ctx->stream));
        d_be = ctx->ws_beta.p;
    }
    // scratch: denom[N] numer[N*N] c1[T] wn[NM]; xi_mix [T,N,M] in ws_bm
    LN_TRY(ctx->ws_misc.reserve(((size_t)N + (size_t)N * N + T + NM) * 8));
    double* denom = ctx->ws_misc.as<double>();
    double* numer = denom + N;
    double* c1 = numer + (size_t)N * N;
    double* wn = c1 + T;
    LN_TRY(ctx->ws_bm.reserve((size_t)T * NM * es));
    TS* xm = ctx->ws_bm.as<TS>();
    void *dA, *dpi, *dw, *dmu, *dsi;
    LN_TRY(stage_out(ctx, A_out, (size_t)N * N * es, ctx->ws_out, &dA));
    LN_TRY(stage_out(ctx, pi_out, (size_t)N * es, ctx->ws_out2, &dpi));
    LN_TRY(stage_out(ctx, w_out, NM * es, ctx->ws_out3, &dw));
    LN_TRY(stage_out(ctx, mu_out, NM * D * 8, ctx->ws_misc2, &dmu));
    LN_TRY(stage_out(ctx, si_out, NM * D * D * 8, ctx->ws_stats, &dsi));
    LN_LAUNCH(ctx, "col_lse", col_lse_kernel<TS>, dim3((N + 63) / 64), dim3(64), 0, (const TS*)d_ga, T, (size_t)N,
              denom);
    LN_LAUNCH(ctx, "col_lse", col_lse_kernel<TS>, dim3((N * N + 63) / 64), dim3(64), 0, (const TS*)d_xi, T,
              (size_t)N * N, numer);
    LN_LAUNCH(ctx, "finish_A_pi", finish_A_pi_kernel<TS>, dim3((N * N + 255) / 256), dim3(256), 0, numer, denom,
              (const TS*)d_ga, N, (TS*)dA, (TS*)dpi);
    LN_LAUNCH(ctx, "ab_lse", ab_lse_kernel<TS>, dim3((T + 3) / 4), dim3(256), 0, (const TS*)d_al, (const TS*)d_be,
              (long)T, N, c1);
    LN_LAUNCH(ctx, "xi_mix", xi_mix_kernel<TS>, dim3((unsigned)(((size_t)T * N + 127) / 128)), dim3(128), 0,
              (const TS*)d_al, (const TS*)d_be, (const TS*)d_bm, md.logw.as<double>(), c1, (long)T, N, M, xm);
    LN_LAUNCH(ctx, "col_lse", col_lse_kernel<TS>, dim3((unsigned)((NM + 63) / 64)), dim3(64), 0, (const TS*)xm, T, NM,
              wn);
    LN_LAUNCH(ctx, "finish_w", finish_w_kernel<TS>, dim3((N + 63) / 64), dim3(64), 0, wn, N, M, (TS*)dw);
    const size_t shmem = (size_t)(64 + 64 * D + 32) * 8;
    LN_LAUNCH(ctx, "mstep_full", mstep_full_kernel<TS>, dim3((unsigned)NM), dim3(256), shmem, (const double*)d_obs,
              (const TS*)xm, md.mu.as<double>(), T, (int)NM, D, min_std, (double*)dmu, (double*)dsi);
    LN_TRY(finish_out(ctx, A_out, dA, (size_t)N * N * es));
    LN_TRY(finish_out(ctx, pi_out, dpi, (size_t)N * es));
    LN_TRY(finish_out(ctx, w_out, dw, NM * es));
    LN_TRY(finish_out(ctx, mu_out, dmu, NM * D * 8));
    return finish_out(ctx, si_out, dsi, NM * D * D * 8);
}

}  // namespace lnasr

extern "C" {

int lnasr_estimate_discrete(lnasr_ctx* ctx, const int32_t* obs, int T, int N, int K, int dtype, const void* xi,
                            const void* gamma, void* logA_out, void* logB_out, void* logpi_out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_REQUIRE(T >= 1 && N >= 1 && K >= 1, "bad shape");
    LN_REQUIRE(obs && xi && gamma && logA_out && logB_out && logpi_out, "null argument");
    LN_REQUIRE(dtype == LNASR_F64 || dtype == LNASR_F32, "bad dtype %d", dtype);
    if (dtype == LNASR_F32)
        return estimate_discrete_impl<float>(ctx, obs, T, N, K, xi, gamma, logA_out, logB_out, logpi_out);
    return estimate_discrete_impl<double>(ctx, obs, T, N, K, xi, gamma, logA_out, logB_out, logpi_out);
}

int lnasr_estimate_gmm(lnasr_ctx* ctx, const double* obs, int T, int dtype, const void* bm, const void* alpha,
                       const void* beta, const void* xi, const void* gamma, double min_std, void* logA_out,
                       void* logpi_out, void* logw_out, double* mu_out, double* si_out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (!ctx->model.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    LN_REQUIRE(T >= 1, "bad T");
    LN_REQUIRE(obs && bm && alpha && beta && xi && gamma && logA_out && logpi_out && logw_out && mu_out && si_out,
               "null argument");
    LN_REQUIRE(dtype == LNASR_F64 || dtype == LNASR_F32, "bad dtype %d", dtype);
    if (dtype == LNASR_F32)
        return estimate_gmm_impl<float>(ctx, obs, T, bm, alpha, beta, xi, gamma, min_std, logA_out, logpi_out,
                                        logw_out, mu_out, si_out);
    return estimate_gmm_impl<double>(ctx, obs, T, bm, alpha, beta, xi, gamma, min_std, logA_out, logpi_out, logw_out,
                                     mu_out, si_out);
}

int lnasr_bw_stats_len(lnasr_ctx* ctx, size_t* n) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && n, "null argument");
    Model& md = ctx->model;
    if (!md.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    *n = stats_layout(md.N, md.M, md.D, md.diag).total;
    return 0;
}

int lnasr_bw_accumulate(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, double* stats) {
    return lnasr_bw_accumulate_p(ctx, obs, U, T, lens, LNASR_F64, stats);
}

int lnasr_bw_accumulate_p(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, int dtype, double* stats) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && stats, "null argument");
    LN_REQUIRE(dtype == LNASR_F64 || dtype == LNASR_F32, "bad dtype %d", dtype);
    Model& md = ctx->model;
    if (!md.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    LN_REQUIRE(U >= 1 && T >= 1 && obs, "bad batch");
    const int N = md.N, M = md.M, D = md.D;
    const StatsLayout L = stats_layout(N, M, D, md.diag);
    const int32_t* d_lens;
    LN_TRY(device_lens(ctx, lens, U, T, &d_lens));
    const double* d_obs = nullptr;
    void* d_st;
    LN_TRY(stage_out(ctx, stats, L.total * 8, ctx->ws_stats, &d_st));
    double* st = (double*)d_st;
    LN_HIP(hipMemsetAsync(st, 0, L.total * 8, ctx->stream));
    const size_t cells = (size_t)U * T * N;
    // precision = float32: densities and moments on the fp32 matrix cores where the model allows it (f32stats.hip)
    const bool f32 = dtype == LNASR_F32 && f32_stats_ok(md);
    LN_TRY(ctx->ws_b.reserve(cells * 8));
    LN_TRY(ctx->ws_bm.reserve((cells + (size_t)kMixPadFrames * N) * M * (f32 ? 4 : 8)));
    LN_TRY(ctx->ws_alpha.reserve(cells * 8));
    LN_TRY(ctx->ws_beta.reserve(cells * 8));
    LN_TRY(ctx->ws_misc.reserve(((size_t)U + (size_t)U * T) * 8));
    double* ll = ctx->ws_misc.as<double>();
    double* c1 = ll + U;
    double* b = ctx->ws_b.as<double>();
    double* bm = ctx->ws_bm.as<double>();
    double* al = ctx->ws_alpha.as<double>();
    double* be = ctx->ws_beta.as<double>();
    LN_REQUIRE(U <= 65535, "U=%d too large for one statistics launch", U);
    const bool mfma_stats = !sw().no_mfma_stats;
    bool feat_tested = false;       // the emission kernel looked for NaN / infinite features itself (nanfix.hip)
    // (Cutting the batch into halves so that the trellises of one half run beside the emission / the statistics of the
    // other was built twice: round 2 with equal-priority streams (12.8 ms against 11.7), round 3 with highest-priority
    // trellis streams and grids sized for the CUs that are left (13.0 ms, float32 pass 9.0 against 7.7): the trellis
    // blocks then do get their CUs, but their frame loop waits for one fresh HBM row of b per frame, and beside the
    // 8.4 GB write / read streams of the emission / moments kernels that wait exceeds a frame -- 2.8 -> 4.1 ms per pass;
    // prefetching deeper costs registers the slice-0 role does not have.  DESIGN 4.5.)
    if (f32) {
        const void* dv;
        LN_TRY(stage_in(ctx, obs, (size_t)U * T * D * 8, ctx->ws_in, &dv));
        d_obs = (const double*)dv;
        LN_TRY(emit_device_f32(ctx, d_obs, (long)U * T, b, ctx->ws_bm.as<float>()));
    } else {
        // host observations: upload overlapped with the emission
        LN_TRY(emit_batch(ctx, obs, U, T, b, bm, &d_obs, &feat_tested));
    }
    // forward and backward only share their input: with few utterances per GPU (config 5: 64 blocks on 256 CUs)
    // the backward pass runs on a side stream beside the forward pass
    LN_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    LN_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    {
        StreamScope on(ctx, ctx->stream2);
        LN_TRY(backward_device(ctx, b, U, T, d_lens, LNASR_F64, be));
    }
    LN_HIP(hipEventRecord(ctx->ev_join, ctx->stream2));
    LN_TRY(forward_device(ctx, b, U, T, d_lens, LNASR_F64, al, ll));
    LN_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    if (mfma_stats) {
        LN_TRY(xi_sum_device(ctx, al, be, b, ll, d_lens, U, T, st + L.xi));
    } else {
        const int tiles = (N + 15) / 16;
        LN_LAUNCH(ctx, "acc_xi", acc_xi_kernel, dim3(tiles, tiles, U), dim3(256), 0, al, be, b, md.logA.as<double>(), ll,
                  d_lens, T, N, st + L.xi);
    }
    LN_LAUNCH(ctx, "rowsum", rowsum_kernel, dim3((N + 63) / 64), dim3(64), 0, st + L.xi, N, st + L.gsum);
    LN_LAUNCH(ctx, "acc_gamma0", acc_gamma0_kernel, dim3(U), dim3(64), 0, al, be, ll, d_lens, U, T, N, st + L.g0,
              st + L.ll, st + L.nutt, st + L.nfr);
    // frames past an utterance's length hold stale alpha/beta; ab_lse covers them but acc_mix skips them
    LN_LAUNCH(ctx, "ab_lse", ab_lse_kernel<double>, dim3((unsigned)(((size_t)U * T + 3) / 4)), dim3(256), 0, al, be,
              (long)U * T, N, c1);
    if (f32) {
        LN_TRY(mix_stats_device_f32(ctx, (const double*)d_obs, al, be, b, ctx->ws_bm.as<float>(), c1, d_lens, U, T,
                                    st + L.mix, st + L.mix_x, st + L.mix_xx));
    } else if (mix_stats_mfma_ok(md)) {
        LN_TRY(mix_stats_device(ctx, (const double*)d_obs, al, be, b, bm, c1, d_lens, U, T, st + L.mix, st + L.mix_x,
                                st + L.mix_xx));
    } else if (md.diag) {
        const size_t shmem = (size_t)kMixCH * (M + D) * 8;
        LN_REQUIRE(shmem <= 160 * 1024, "M=%d, D=%d too large for the statistics kernel", M, D);
        LN_LAUNCH(ctx, "acc_mix_diag", acc_mix_diag_kernel, dim3(N, (T + kMixCH - 1) / kMixCH, U), dim3(256), shmem,
                  (const double*)d_obs, al, be, b, bm, c1, md.logw.as<double>(), md.mu.as<double>(), d_lens, T, N, M,
                  D, st + L.mix, st + L.mix_x, st + L.mix_xx);
    } else if (full_stats_mfma_ok(md)) {
        LN_TRY(full_stats_device(ctx, (const double*)d_obs, al, be, b, bm, c1, d_lens, U, T, st + L.mix, st + L.mix_x,
                                 st + L.mix_xx));
    } else {
        const size_t shmem = ((size_t)T + 64 * (size_t)D) * 8;
        LN_REQUIRE(shmem <= 160 * 1024, "T=%d too long for the full-covariance statistics kernel", T);
        LN_LAUNCH(ctx, "acc_full", acc_full_kernel, dim3(N * M, U), dim3(256), shmem, (const double*)d_obs, al, be, b,
                  bm, c1, md.logw.as<double>(), md.mu.as<double>(), d_lens, T, N, M, D, st + L.mix, st + L.mix_x,
                  st + L.mix_xx);
    }
    // a NaN or an infinite feature anywhere in the batch: every posterior of its utterance is NaN in the reference, and so
    // is every sum it enters (one launch that returns at once unless the emission kernel raised the flag)
    LN_TRY(stats_nan_repair(ctx, (const double*)d_obs, U, T, d_lens, st, (long)L.ll + 1, !feat_tested));   // the counts stay
    return finish_out(ctx, stats, d_st, L.total * 8);
}

int lnasr_bw_mstep(lnasr_ctx* ctx, const double* stats, double min_std, double* logA, double* logpi, double* logw,
                   double* mu, double* sigma) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && stats && logA && logpi && logw && mu && sigma, "null argument");
    Model& md = ctx->model;
    if (!md.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    const int N = md.N, M = md.M, D = md.D;
    const StatsLayout L = stats_layout(N, M, D, md.diag);
    const void* d_st;
    LN_TRY(stage_in(ctx, stats, L.total * 8, ctx->ws_in, &d_st));
    const size_t NM = (size_t)N * M;
    const size_t nsig = md.diag ? NM * D : NM * D * D;
    LN_TRY(ctx->ws_misc.reserve(((size_t)N * N + N + NM + NM * D + nsig) * 8));
    double* dA = ctx->ws_misc.as<double>();
    double* dpi = dA + (size_t)N * N;
    double* dw = dpi + N;
    double* dmu = dw + NM;
    double* dvar = dmu + NM * D;
    size_t big = (size_t)N * N;
    if (nsig > big) big = nsig;
    LN_LAUNCH(ctx, "mstep_stats", mstep_stats_kernel, dim3((unsigned)((big + 255) / 256)), dim3(256), 0,
              (const double*)d_st, L, N, M, D, md.diag ? 1 : 0, min_std, dA, dpi, dw, dmu, dvar);
    struct { double* user; double* dev; size_t n; } outs[] = {
        {logA, dA, (size_t)N * N}, {logpi, dpi, (size_t)N}, {logw, dw, NM}, {mu, dmu, NM * D}, {sigma, dvar, nsig}};
    for (auto& o : outs) {
        hipMemcpyKind k = is_device_ptr(o.user) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        LN_HIP(hipMemcpyAsync(o.user, o.dev, o.n * 8, k, ctx->stream));
    }
    LN_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
