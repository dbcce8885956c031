// Baum-Welch sufficient statistics on the fp64 matrix cores (reference: hmm.py:186-199 for xi / gamma,
// gmmhmm.py:95-140 for the mixture posteriors and moments; summed over a batch of utterances).
//
// Both big sums are contractions over FRAMES, i.e. GEMMs with K = U*T:
//
//   xi_sum[i,j]   = sum_f exp(alpha[f,i] + A[i,j] + b[f+1,j] + beta[f+1,j] - loglik)
//                 = expA[i,j] * sum_f P[f,i] * Q[f,j],      P = exp(alpha - c_f),  Q = exp(b+beta (f+1) + c_f - loglik)
//   S[(j,m), col] = sum_f g[f,j,m] * feat[f,col],           feat = [x - c, (x - c)^2, 1]   (c: one centre per dimension)
//
// so N^2 (resp. N*M*D) exponentials per frame become 2N (resp. N*M), and the multiply-adds run as
// v_mfma_f64_16x16x4 at the fp64 peak instead of as scalar exp() chains.  The per-frame shift c_f keeps P in
// [0,1]; a frame on which that single shift cannot represent every term that matters (a state hopeless from
// the past but essential given the future: dynamic range beyond e^690) is detected from gamma and handed,
// whole, to an exact log-domain kernel, so no term of the reference's sum is ever silently dropped.
#include "common.h"
#include "devmath.h"

namespace lnasr {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr double kLinLo = -690.0;     // exponents below this are not representable well enough in the linear domain
constexpr double kGammaKeep = -60.0;  // log-posterior above which a dropped state / column would matter (e^-60 = 9e-27)

__device__ __forceinline__ bool finite_f64(double x) { return x == x && x != LN_NEG_INF && x != -LN_NEG_INF; }

// ------------------------------------------------------------------------------------------
// xi factors: one wave per frame f = u*T + t.  P[f,:] and Q[f,:] (Q already shifted to pair with frame f)
// are zero for t >= len-1, for padding rows, and for flagged frames (which go to xi_exact_kernel).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xi_factors_kernel(const double* __restrict__ alpha, const double* __restrict__ beta,
                                                         const double* __restrict__ b, const double* __restrict__ loglik,
                                                         const int32_t* __restrict__ lens, long F, long FP, int T, int N,
                                                         int NP, double* __restrict__ P, double* __restrict__ Q,
                                                         int* __restrict__ flag_count, int* __restrict__ flag_list) {
    const int lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= FP) return;
    double* Pf = P + (size_t)f * NP;
    double* Qf = Q + (size_t)f * NP;
    bool live = f < F;
    double ll = 0.0;
    if (live) {
        const int u = (int)(f / T), t = (int)(f % T);
        ll = loglik[u];
        live = t + 1 < lens[u] && finite_f64(ll);
    }
    if (!live) {
        for (int j = lane; j < NP; j += 64) Pf[j] = 0.0, Qf[j] = 0.0;
        return;
    }
    const double* a0 = alpha + (size_t)f * N;
    const double* a1 = a0 + N;
    const double* be0 = beta + (size_t)f * N;
    const double* be1 = be0 + N;
    const double* b1 = b + (size_t)(f + 1) * N;
    double c = LN_NEG_INF;
    for (int j = lane; j < N; j += 64) c = fmax(c, a0[j]);
    c = wave_max(c);
    bool flag = false;
    if (!finite_f64(c)) {
        for (int j = lane; j < NP; j += 64) Pf[j] = 0.0, Qf[j] = 0.0;
        return;
    }
    for (int j = lane; j < NP; j += 64) {
        double pv = 0.0, qv = 0.0;
        if (j < N) {
            const double aj = a0[j];
            const double pa = aj - c;
            if (pa >= kLinLo) pv = exp(pa);
            else if (aj + be0[j] - ll >= kGammaKeep) flag = true;           // underflowed but it matters
            const double zq = (b1[j] + be1[j]) + (c - ll);
            if (zq > -kLinLo) {
                if (a1[j] + be1[j] - ll >= kGammaKeep) flag = true;         // would overflow and it matters
            } else if (zq >= kLinLo) {
                qv = exp(zq);
            }
        }
        Pf[j] = pv;
        Qf[j] = qv;
    }
    if (__any(flag)) {
        for (int j = lane; j < NP; j += 64) Pf[j] = 0.0, Qf[j] = 0.0;
        if (lane == 0) flag_list[atomicAdd(flag_count, 1)] = (int)f;
    }
}

// flagged frames, exactly as the reference adds them (log domain, one exp per (i,j) term)
__global__ void xi_exact_kernel(const double* __restrict__ alpha, const double* __restrict__ beta,
                                const double* __restrict__ b, const double* __restrict__ logA,
                                const double* __restrict__ loglik, int T, int N, const int* __restrict__ flag_count,
                                const int* __restrict__ flag_list, double* __restrict__ xi_sum) {
    const int nflag = *flag_count;
    for (int e = blockIdx.x; e < nflag; e += gridDim.x) {
        const long f = flag_list[e];
        const double ll = loglik[f / T];
        const double* a0 = alpha + (size_t)f * N;
        const double* be1 = beta + (size_t)(f + 1) * N;
        const double* b1 = b + (size_t)(f + 1) * N;
        for (int p = threadIdx.x; p < N * N; p += blockDim.x) {
            const int i = p / N, j = p % N;
            const double z = ((a0[i] + logA[p]) + (b1[j] + be1[j])) - ll;
            if (z != LN_NEG_INF && z == z) atomicAdd(&xi_sum[p], exp(z));
        }
    }
}

// ------------------------------------------------------------------------------------------
// xi_sum += expA o (P^T Q): 128 x 128 output per block (2 x 2 waves, 4 x 4 MFMA tiles each), split over K.
// Operands go global -> registers directly (each is used by 4 MFMAs of the wave), three steps in flight.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xi_gemm_kernel(const double* __restrict__ P, const double* __restrict__ Q, long FP,
                                                      long FPB, int NP, int N, const double* __restrict__ expA,
                                                      double* __restrict__ xi_sum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int i0 = blockIdx.y * 128 + (w >> 1) * 64, j0 = blockIdx.x * 128 + (w & 1) * 64;
    const long f0 = (long)blockIdx.z * FPB;
    long f1 = f0 + FPB;
    if (f1 > FP) f1 = FP;
    if (f0 >= f1) return;
    int ib[4], jb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ib[a] = i0 + 16 * a < NP ? i0 + 16 * a : NP - 16;      // tiles past the edge recompute the last one
        jb[a] = j0 + 16 * a < NP ? j0 + 16 * a : NP - 16;
    }
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = double4_t{0.0, 0.0, 0.0, 0.0};
    constexpr int PD = 3;                                        // ring of operand sets
    double pa[PD][4], qb[PD][4];
    const long last = f1 - 4;
    auto load = [&](int slot, long f) {
        const long fc = f < last ? f : last;
        const double* pr = P + (size_t)(fc + kk) * NP + l15;
        const double* qr = Q + (size_t)(fc + kk) * NP + l15;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            pa[slot][a] = pr[ib[a]];
            qb[slot][a] = qr[jb[a]];
        }
    };
    load(0, f0);
    load(1, f0 + 4);
    long f = f0;
    while (f < f1) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            if (f < f1) {
                load((s + 2) % PD, f + 8);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[s][a], qb[s][c], acc[a][c], 0, 0, 0);
                f += 4;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (i0 + 16 * a >= NP) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (j0 + 16 * c >= NP) continue;
            const int j = j0 + 16 * c + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 16 * a + kk + 4 * r;
                if (i < N && j < N) {
                    const double v = acc[a][c][r] * expA[(size_t)i * N + j];
                    if (v != 0.0) atomicAdd(&xi_sum[(size_t)i * N + j], v);
                }
            }
        }
    }
}

int xi_sum_device(lnasr_ctx* ctx, const double* alpha, const double* beta, const double* b, const double* loglik,
                  const int32_t* d_lens, int U, int T, double* xi_sum) {
    LN_DEVPTR(alpha); LN_DEVPTR(beta); LN_DEVPTR(b); LN_DEVPTR(loglik); LN_DEVPTR(d_lens); LN_DEVPTR(xi_sum);
    Model& md = ctx->model;
    const int N = md.N, NP = (N + 15) / 16 * 16;
    const long F = (long)U * T, FP = (F + 3) / 4 * 4;
    LN_REQUIRE(F < (1L << 31), "U*T = %ld frames is too many for one statistics call", F);
    LN_TRY(ctx->ws_p.reserve((size_t)FP * NP * 8));
    LN_TRY(ctx->ws_q.reserve((size_t)FP * NP * 8));
    LN_TRY(ctx->ws_misc2.reserve((size_t)(F + 2) * sizeof(int)));
    int* flag_count = ctx->ws_misc2.as<int>();
    int* flag_list = flag_count + 2;
    LN_HIP(hipMemsetAsync(flag_count, 0, 2 * sizeof(int), ctx->stream));
    double* P = ctx->ws_p.as<double>();
    double* Q = ctx->ws_q.as<double>();
    LN_LAUNCH(ctx, "xi_factors", xi_factors_kernel, dim3((unsigned)(FP / 4)), dim3(256), 0, alpha, beta, b, loglik, d_lens,
              F, FP, T, N, NP, P, Q, flag_count, flag_list);
    const int nb = (NP + 127) / 128;
    long KS = (2L * ctx->prop.multiProcessorCount + nb * nb - 1) / (nb * nb);
    if (KS > FP / 4) KS = FP / 4;
    if (KS < 1) KS = 1;
    if (KS > 65535) KS = 65535;
    long FPB = ((FP + KS - 1) / KS + 3) / 4 * 4;
    KS = (FP + FPB - 1) / FPB;
    LN_LAUNCH(ctx, "xi_gemm", xi_gemm_kernel, dim3(nb, nb, (unsigned)KS), dim3(256), 0, (const double*)P, (const double*)Q,
              FP, FPB, NP, N, md.expA.as<double>(), xi_sum);
    LN_LAUNCH(ctx, "xi_exact", xi_exact_kernel, dim3(256), dim3(256), 0, alpha, beta, b, md.logA.as<double>(), loglik, T, N,
              (const int*)flag_count, (const int*)flag_list, xi_sum);
    return 0;
}

// ------------------------------------------------------------------------------------------
// Mixture posteriors and moments (diagonal model).  Block = kMixWaves waves = consecutive states over a contiguous
// range of frames; the wave of state j turns bm[f,j,:] into posteriors g (the MFMA A operand, 16 mixtures x 4
// frames per instruction) and multiplies them with the block's shared feature rows [x-c, (x-c)^2, 1] from LDS.
// Raw sums go to R[(j, m), col]; mix_finish_kernel turns them into the reference's three accumulators.
// ------------------------------------------------------------------------------------------
constexpr int kMixFrames = 32;       // frames per staged chunk (8 MFMA k-steps): the full-covariance moments kernel
constexpr int kMixDiagFrames = 32;   // ... and the diagonal one (mix_mfma_kernel).  Measured at config 5 (round 3, SQ counters in
                                     // the matrix pipe is busy 64 % of the kernel, other VALU 21 %): 64 frames per
                                     // chunk -- half the barriers -- 3.80 ms against 3.65; one generation of blocks instead of
                                     // two (only one 206-register block fits a CU at a time anyway) 3.67
constexpr int kMixRing = 8;          // bm prefetch ring (k-steps; must divide kMixFrames / 4)
// The ring's running pointer is not clamped: the last block reads up to (kMixFrames - 1) frames of chunk rounding plus
// 4 (kMixRing - 1) + 3 frames of prefetch past the batch.  Callers allocate bm with kMixPadFrames frames of slack
// (lnasr_bw_accumulate_p); this ties the three constants together.
static_assert((kMixDiagFrames - 1) + 4 * (kMixRing - 1) + 3 < kMixPadFrames, "bm prefetch ring outruns kMixPadFrames");
static_assert((kMixFrames - 1) + 4 * (kMixRing - 1) + 3 < kMixPadFrames, "bm prefetch ring outruns kMixPadFrames");
static_assert((kMixDiagFrames / 4) % kMixRing == 0, "kMixRing must divide kMixDiagFrames / 4");
constexpr int kMixWaves = 8;         // waves (= states) per block; 4-wave blocks at 3 waves/SIMD were measured: slower (5.7 vs 4.1 ms);
                                     // round 3, after independent blocks had paid in fwdbwd_cols2: 4-wave blocks again, two or three
                                     // per CU (218 / 140 registers): 3.98 / 3.96 ms against 3.65 -- this kernel is 85 % issue, and
                                     // half-size blocks stage every feature row twice
constexpr int kMixNT = 64 * kMixWaves;

__global__ void mix_centre_kernel(const double* __restrict__ mu, int NM, int D, double* __restrict__ centre) {
    __shared__ double scratch[8];
    const int d = blockIdx.x;
    double s = 0.0;
    for (int c = threadIdx.x; c < NM; c += blockDim.x) s += mu[(size_t)c * D + d];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) centre[d] = s / (double)NM;
}

template <int MT, int CT>
__global__ __launch_bounds__(64 * kMixWaves) void mix_mfma_kernel(const double* __restrict__ obs, const double* __restrict__ alpha,
                                                       const double* __restrict__ beta, const double* __restrict__ b,
                                                       const double* __restrict__ bm, const double* __restrict__ c1,
                                                       const double* __restrict__ logw, const double* __restrict__ centre,
                                                       const int32_t* __restrict__ lens, long F, long FPB, int T, int N,
                                                       int M, int D, int G, int MP, double* __restrict__ R) {
    constexpr int CH = kMixDiagFrames, NFP = 16 * CT, FST = NFP + ((CT & 1) ? 0 : 16);
    constexpr int OPT = (CH * 39 + kMixNT - 1) / kMixNT;             // obs doubles staged per thread (D <= 39)
    __shared__ double feat[2][CH * FST];
    __shared__ double sab[2][CH * kMixWaves], sbb[2][CH * kMixWaves];
    __shared__ double etab[kExpTab];
    __shared__ double cen[40];
    LN_BLOCK_T0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    // wave <-> (state j, group of 32 mixtures mg): M > 32 runs as G = ceil(M / 32) groups per state
    const int jv = blockIdx.x * kMixWaves + w;
    const int j = jv / G, mo = 32 * (jv % G);
    const bool active = j < N;
    const int jc = active ? j : N - 1;
    const long fb0 = (long)blockIdx.y * FPB;
    long fb1 = fb0 + FPB;
    if (fb1 > F) fb1 = F;
    if (fb0 >= fb1) return;
    exp_table_init(etab, tid, kMixNT);
    if (tid < D) cen[tid] = centre[tid];
    // constant feature columns: the ones column at 2D, zeros after it
    for (int e = tid; e < 2 * CH * (FST - 2 * D); e += kMixNT) {
        const int bufi = e / (CH * (FST - 2 * D)), r = e % (CH * (FST - 2 * D));
        const int fr = r / (FST - 2 * D), col = 2 * D + r % (FST - 2 * D);
        feat[bufi][fr * FST + col] = (col == 2 * D) ? 1.0 : 0.0;
    }
    double lw[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mo + 16 * mt + l15;
        lw[mt] = m < M ? logw[(size_t)jc * M + m] : LN_NEG_INF;
    }

    // ---- staging: global -> registers -> LDS, one chunk ahead ----
    double oreg[OPT], s_ab = 0.0, s_bb = 0.0;
    auto stage_load = [&](long f0) {
        const long nfr = (fb1 - f0) < CH ? (fb1 - f0) : CH;
#pragma unroll
        for (int r = 0; r < OPT; ++r) {
            const int e = tid + r * kMixNT;
            oreg[r] = e < nfr * D ? obs[(size_t)f0 * D + e] : 0.0;
        }
        if (tid < CH * kMixWaves) {
            const int fr = tid / kMixWaves, js = (blockIdx.x * kMixWaves + (tid % kMixWaves)) / G;
            const long f = f0 + fr;
            s_ab = LN_NEG_INF;
            s_bb = 0.0;
            if (f < fb1 && js < N) {
                const int u = (int)(f / T), t = (int)(f % T);
                if (t < lens[u]) {
                    const size_t fj = (size_t)f * N + js;
                    s_ab = alpha[fj] + beta[fj] - c1[f];
                    s_bb = b[fj];
                }
            }
        }
    };
    auto stage_store = [&](int bufi) {
#pragma unroll
        for (int r = 0; r < OPT; ++r) {
            const int e = tid + r * kMixNT;
            if (e < CH * D) {
                const int fr = e / D, d = e % D;
                const double x = oreg[r] - cen[d], xx = x * x;
                // a frame behind its utterance's end weighs 0, and 0 * inf is NaN on the matrix cores: whatever the caller
                // left in the padding (NaN, inf, or anything whose square overflows) enters the sums as 0
                const bool fin = __builtin_isfinite(xx);
                feat[bufi][fr * FST + d] = fin ? x : 0.0;
                feat[bufi][fr * FST + D + d] = fin ? xx : 0.0;
            }
        }
        if (tid < CH * kMixWaves) {
            sab[bufi][tid] = s_ab;
            sbb[bufi][tid] = s_bb;
        }
    };

    double4_t acc[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = double4_t{0.0, 0.0, 0.0, 0.0};

    // bm prefetch ring: k-step s of the block covers frames fb0 + 4 s + kk.  A running pointer per mixture tile (one
    // 64-bit add per load); the ring runs up to 4 kMixRing frames past the block's range: bm is allocated with that
    // much slack (kMixPadFrames), and what is read there is never used (those frames carry a log posterior of -inf).
    double ring[kMixRing][MT];
    const size_t bstride = (size_t)N * M;
    const double* bptr[MT];                // lanes past the last mixture read mixture `mo` instead (their log w is -inf):
#pragma unroll                             // an unconditional load, not a branch around one
    for (int mt = 0; mt < MT; ++mt)
        bptr[mt] = bm + ((size_t)(fb0 + kk) * N + jc) * M + mo + ((mo + 16 * mt + l15 < M) ? 16 * mt + l15 : 0);
    auto ring_load = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ring[slot][mt] = *bptr[mt];
            bptr[mt] += 4 * bstride;
        }
    };
    __syncthreads();                       // cen, etab, constant columns
    stage_load(fb0);
    stage_store(0);
#pragma unroll
    for (int s = 0; s < kMixRing - 1; ++s) ring_load(s);
    __syncthreads();

    int cur = 0;
    for (long f0 = fb0; f0 < fb1; f0 += CH, cur ^= 1) {
        const bool has_next = f0 + CH < fb1;
        if (has_next) stage_load(f0 + CH);
        if (active) {
#pragma unroll
            for (int ks = 0; ks < CH / 4; ++ks) {
                constexpr int RM = kMixRing;
                ring_load((ks + RM - 1) % RM);
                const int fk = 4 * ks + kk;
                const double ab = sab[cur][fk * kMixWaves + w], bb = sbb[cur][fk * kMixWaves + w];
                double g[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const double z = ab + ((lw[mt] + ring[ks % RM][mt]) - bb);
                    const double ev = exp_nonpos(z, etab);
                    g[mt] = (z >= -700.0) ? ev : 0.0;             // exact zeros stay zero; NaN -> 0
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const double fv = feat[cur][fk * FST + 16 * ct + l15];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[mt], fv, acc[mt][ct], 0, 0, 0);
                }
            }
        }
        if (has_next) stage_store(cur ^ 1);
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * mt + kk + 4 * r;
                    const double v = acc[mt][ct][r];
                    if (v != 0.0) atomicAdd(&R[((size_t)j * MP + mo + m) * NFP + 16 * ct + l15], v);
                }
    }
    LN_BLOCK_T1;
}
LN_BLOCK_TIMES_EXPORT(lnasr_dbg_block_times_bwstats)

// mix[j,m] = Sg;  mix_x = S1 + c Sg;  mix_xx = sum g (o - mu_old)^2 = S2 - 2 (mu - c) S1 + (mu - c)^2 Sg
__global__ void mix_finish_kernel(const double* __restrict__ R, const double* __restrict__ mu,
                                  const double* __restrict__ centre, int N, int M, int D, int MP, int NFP,
                                  double* __restrict__ mix, double* __restrict__ mix_x, double* __restrict__ mix_xx) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * M * D) return;
    const int d = (int)(idx % D);
    const size_t c = idx / D;
    const int m = (int)(c % M), j = (int)(c / M);
    const double* row = R + ((size_t)j * MP + m) * NFP;
    const double sg = row[2 * D], s1 = row[d], s2 = row[D + d];
    const double cd = centre[d], dl = mu[idx] - cd;
    if (d == 0) mix[c] = sg;
    mix_x[idx] = s1 + cd * sg;
    const double v = (s2 - 2.0 * dl * s1) + dl * dl * sg;
    mix_xx[idx] = v > 0.0 ? v : 0.0;
}

// ------------------------------------------------------------------------------------------
// Full covariance: the second moment of component (j,m) around its OLD mean (gmmhmm.py:138) is, with y = x - c
// and delta = mu_old - c,   sum g (y - delta)(y - delta)^T = P - delta S1^T - S1 delta^T + delta delta^T Sg,
// P[d,e] = sum g y_d y_e,  S1 = sum g y,  Sg = sum g  -- the K = D(D+1)/2 + D + 1 quadratic features of the
// full-covariance emission kernel (same pair table, 80-column chunks).  Block = 8 states x one column chunk x
// a contiguous frame range; the posteriors are recomputed per chunk (an exp per (frame, mixture): cheap next
// to the 10 MFMAs it feeds), the chunk's feature rows y_a y_b are built once per 32 frames in LDS.
// ------------------------------------------------------------------------------------------
constexpr int kFullWaves = 8, kFullNT = 64 * kFullWaves;   // the full-covariance kernel's LDS (69 KB) allows two blocks per CU

template <int MT>
__global__ __launch_bounds__(64 * kFullWaves) void full_mfma_kernel(const double* __restrict__ obs, const double* __restrict__ alpha,
                                                        const double* __restrict__ beta, const double* __restrict__ b,
                                                        const double* __restrict__ bm, const double* __restrict__ c1,
                                                        const double* __restrict__ logw, const double* __restrict__ centre,
                                                        const int* __restrict__ pairtab, const int32_t* __restrict__ lens,
                                                        long F, long FPB, int T, int N, int M, int D, int NCH, int G,
                                                        int MP, double* __restrict__ R) {
    constexpr int CH = kMixFrames, CT = 5, NFP = 16 * CT, FST = NFP, YST = 42;
    constexpr int OPT = (CH * 39 + kFullNT - 1) / kFullNT;
    __shared__ double feat[2][CH * FST];
    __shared__ double yrow[2][CH * YST];             // y[0..D-1], then 1.0 at D and 0.0 at D+1
    __shared__ double sab[2][CH * kFullWaves], sbb[2][CH * kFullWaves];
    __shared__ double etab[kExpTab];
    __shared__ double cen[40];
    __shared__ int pa[NFP], pb[NFP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int jv = blockIdx.x * kFullWaves + w, cc = blockIdx.y;
    const int j = jv / G, mo = 32 * (jv % G);      // (state, group of 32 mixtures)
    const bool active = j < N;
    const int jc = active ? j : N - 1;
    const long fb0 = (long)blockIdx.z * FPB;
    long fb1 = fb0 + FPB;
    if (fb1 > F) fb1 = F;
    if (fb0 >= fb1) return;
    exp_table_init(etab, tid, kFullNT);
    if (tid < D) cen[tid] = centre[tid];
    if (tid < NFP) {
        const int code = pairtab[cc * NFP + tid], d = code & 0xFF, e = (code >> 8) & 0xFF;
        pa[tid] = d < D ? d : (d == 0xFF ? D : D + 1);
        pb[tid] = e < D ? e : (e == 0xFF ? D : D + 1);
    }
    for (int e = tid; e < 2 * CH * 2; e += kFullNT) {
        const int bufi = e / (CH * 2), r = e % (CH * 2);
        yrow[bufi][(r >> 1) * YST + D + (r & 1)] = (r & 1) ? 0.0 : 1.0;
    }
    double lw[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mo + 16 * mt + l15;
        lw[mt] = m < M ? logw[(size_t)jc * M + m] : LN_NEG_INF;
    }
    double oreg[OPT], s_ab = 0.0, s_bb = 0.0;
    auto stage_load = [&](long f0) {
        const long nfr = (fb1 - f0) < CH ? (fb1 - f0) : CH;
#pragma unroll
        for (int r = 0; r < OPT; ++r) {
            const int e = tid + r * kFullNT;
            oreg[r] = e < nfr * D ? obs[(size_t)f0 * D + e] : 0.0;
        }
        if (tid < CH * kFullWaves) {
            const int fr = tid / kFullWaves, js = (blockIdx.x * kFullWaves + (tid % kFullWaves)) / G;
            const long f = f0 + fr;
            s_ab = LN_NEG_INF;
            s_bb = 0.0;
            if (f < fb1 && js < N) {
                const int u = (int)(f / T), t = (int)(f % T);
                if (t < lens[u]) {
                    const size_t fj = (size_t)f * N + js;
                    s_ab = alpha[fj] + beta[fj] - c1[f];
                    s_bb = b[fj];
                }
            }
        }
    };
    auto stage_store = [&](int bufi) {
#pragma unroll
        for (int r = 0; r < OPT; ++r) {
            const int e = tid + r * kFullNT;
            if (e < CH * D) {
                const int fr = e / D, d = e % D;
                const double y = oreg[r] - cen[d];
                yrow[bufi][fr * YST + d] = __builtin_fabs(y) < 1e150 ? y : 0.0;   // padding frames: see mix_mfma_kernel
                                                                                 // (products of two entries must stay finite)
            }
        }
        if (tid < CH * kFullWaves) {
            sab[bufi][tid] = s_ab;
            sbb[bufi][tid] = s_bb;
        }
    };
    auto build_feat = [&](int bufi) {
        for (int e = tid; e < CH * NFP; e += kFullNT) {
            const int fr = e / NFP, col = e % NFP;
            feat[bufi][fr * FST + col] = yrow[bufi][fr * YST + pa[col]] * yrow[bufi][fr * YST + pb[col]];
        }
    };

    double4_t acc[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = double4_t{0.0, 0.0, 0.0, 0.0};
    double ring[kMixRing][MT];
    const double* bmj = bm + (size_t)jc * M + mo;
    const size_t bstride = (size_t)N * M;
    int boff[MT];                          // lanes past the last mixture read mixture `mo` instead (their log w is -inf):
#pragma unroll                             // an unconditional load, not a branch around one
    for (int mt = 0; mt < MT; ++mt) boff[mt] = (mo + 16 * mt + l15 < M) ? 16 * mt + l15 : 0;
    auto ring_load = [&](int slot, long f) {
        const long fc = f < F ? f : F - 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ring[slot][mt] = bmj[(size_t)fc * bstride + boff[mt]];
    };
    __syncthreads();                       // cen, etab, pair table, constant columns
    stage_load(fb0);
    stage_store(0);
#pragma unroll
    for (int s = 0; s < kMixRing - 1; ++s) ring_load(s, fb0 + 4 * s + kk);
    __syncthreads();
    build_feat(0);
    __syncthreads();

    int cur = 0;
    for (long f0 = fb0; f0 < fb1; f0 += CH, cur ^= 1) {
        const bool has_next = f0 + CH < fb1;
        if (has_next) stage_load(f0 + CH);
        if (active) {
#pragma unroll
            for (int ks = 0; ks < CH / 4; ++ks) {
                constexpr int RM = kMixRing;
                ring_load((ks + RM - 1) % RM, f0 + 4 * (ks + RM - 1) + kk);
                const int fk = 4 * ks + kk;
                const double ab = sab[cur][fk * kFullWaves + w], bb = sbb[cur][fk * kFullWaves + w];
                double g[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const double z = ab + ((lw[mt] + ring[ks % RM][mt]) - bb);
                    const double ev = exp_nonpos(z, etab);
                    g[mt] = (z >= -700.0) ? ev : 0.0;
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const double fv = feat[cur][fk * FST + 16 * ct + l15];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[mt], fv, acc[mt][ct], 0, 0, 0);
                }
            }
        }
        if (has_next) stage_store(cur ^ 1);
        __syncthreads();
        if (has_next) build_feat(cur ^ 1);
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * mt + kk + 4 * r;
                    const double v = acc[mt][ct][r];
                    if (v != 0.0) atomicAdd(&R[(((size_t)j * MP + mo + m) * NCH + cc) * NFP + 16 * ct + l15], v);
                }
    }
}

// raw sums -> mix, mix_x and the D x D second moment around the old mean
__global__ void full_finish_kernel(const double* __restrict__ R, const double* __restrict__ mu,
                                   const double* __restrict__ centre, int N, int M, int D, int MP, int KF,
                                   double* __restrict__ mix, double* __restrict__ mix_x, double* __restrict__ mix_xx) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t DD = (size_t)D * D;
    if (idx >= (size_t)N * M * DD) return;
    const int e = (int)(idx % D), d = (int)((idx / D) % D);
    const size_t c = idx / DD;
    const int m = (int)(c % M), j = (int)(c / M);
    const double* row = R + ((size_t)j * MP + m) * KF;
    const int npair = D * (D + 1) / 2;
    const int lo = d < e ? d : e, hi = d < e ? e : d;
    const double P = row[lo * D - lo * (lo - 1) / 2 + (hi - lo)];
    const double sd = row[npair + d], se = row[npair + e], sg = row[npair + D];
    const double dd = mu[c * D + d] - centre[d], de = mu[c * D + e] - centre[e];
    mix_xx[idx] = ((P - dd * se) - sd * de) + dd * de * sg;
    if (e == 0) {
        mix_x[c * D + d] = sd + centre[d] * sg;
        if (d == 0) mix[c] = sg;
    }
}

bool full_stats_mfma_ok(const Model& md) {
    return !md.diag && md.wfull_Q > 0 && md.D <= 39 && !sw().no_mfma_stats;
}

int full_stats_device(lnasr_ctx* ctx, const double* obs, const double* alpha, const double* beta, const double* b,
                      const double* bm, const double* c1, const int32_t* d_lens, int U, int T, double* mix, double* mix_x,
                      double* mix_xx) {
    LN_DEVPTR(obs); LN_DEVPTR(alpha); LN_DEVPTR(beta); LN_DEVPTR(b); LN_DEVPTR(bm); LN_DEVPTR(c1); LN_DEVPTR(d_lens); LN_DEVPTR(mix); LN_DEVPTR(mix_x); LN_DEVPTR(mix_xx);
    Model& md = ctx->model;
    const int N = md.N, M = md.M, D = md.D, NCH = md.wfull_NCH;
    const int MG = (M + 31) / 32, MT = MG > 1 ? 2 : (M + 15) / 16, MP = MG > 1 ? 32 * MG : 16 * MT, KF = NCH * 80;
    const long F = (long)U * T;
    const size_t nraw = (size_t)N * MP * KF;
    LN_TRY(ctx->ws_raw.reserve((nraw + 64) * 8));
    double* R = ctx->ws_raw.as<double>();
    double* centre = R + nraw;
    LN_HIP(hipMemsetAsync(R, 0, nraw * 8, ctx->stream));
    LN_LAUNCH(ctx, "mix_centre", mix_centre_kernel, dim3(D), dim3(256), 0, md.mu.as<double>(), N * M, D, centre);
    const int SG = (N * MG + kFullWaves - 1) / kFullWaves;
    const long nchunk = (F + kMixFrames - 1) / kMixFrames;
    long G = (2L * ctx->prop.multiProcessorCount + (long)SG * NCH - 1) / ((long)SG * NCH);
    if (G > nchunk) G = nchunk;
    if (G < 1) G = 1;
    if (G > 65535) G = 65535;
    const long FPB = ((nchunk + G - 1) / G) * kMixFrames;
    G = (F + FPB - 1) / FPB;
    const dim3 grid(SG, NCH, (unsigned)G);
    const double* lw = md.logw.as<double>();
    if (MT == 1)
        LN_LAUNCH(ctx, "full_mfma", full_mfma_kernel<1>, grid, dim3(kFullNT), 0, obs, alpha, beta, b, bm, c1, lw,
                  (const double*)centre, (const int*)md.pairtab.as<int>(), d_lens, F, FPB, T, N, M, D, NCH, MG, MP, R);
    else
        LN_LAUNCH(ctx, "full_mfma", full_mfma_kernel<2>, grid, dim3(kFullNT), 0, obs, alpha, beta, b, bm, c1, lw,
                  (const double*)centre, (const int*)md.pairtab.as<int>(), d_lens, F, FPB, T, N, M, D, NCH, MG, MP, R);
    const size_t nel = (size_t)N * M * D * D;
    LN_LAUNCH(ctx, "full_finish", full_finish_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, (const double*)R,
              md.mu.as<double>(), (const double*)centre, N, M, D, MP, KF, mix, mix_x, mix_xx);
    return 0;
}

// launch wrappers for f32stats.hip (kernels are launched from the translation unit that defines them)
int mix_centre_launch(lnasr_ctx* ctx, double* centre) {
    Model& md = ctx->model;
    LN_LAUNCH(ctx, "mix_centre", mix_centre_kernel, dim3(md.D), dim3(256), 0, md.mu.as<double>(), md.N * md.M, md.D, centre);
    return 0;
}
int mix_finish_launch(lnasr_ctx* ctx, const double* R, const double* centre, int MP, int NFP, double* mix, double* mix_x,
                      double* mix_xx) {
    Model& md = ctx->model;
    const size_t nel = (size_t)md.N * md.M * md.D;
    LN_LAUNCH(ctx, "mix_finish", mix_finish_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, R, md.mu.as<double>(),
              centre, md.N, md.M, md.D, MP, NFP, mix, mix_x, mix_xx);
    return 0;
}

bool mix_stats_mfma_ok(const Model& md) {
    return md.diag && md.D <= 39 && !sw().no_mfma_stats;
}

template <int MT>
static int mix_launch(lnasr_ctx* ctx, int CT, dim3 grid, const double* obs, const double* alpha, const double* beta,
                      const double* b, const double* bm, const double* c1, const double* centre, const int32_t* d_lens,
                      long F, long FPB, int T, int G, int MP, double* R) {
    Model& md = ctx->model;
    const double* lw = md.logw.as<double>();
#define LN_MIX_CASE(ct)                                                                                              \
    case ct:                                                                                                         \
        LN_LAUNCH(ctx, "mix_mfma", (mix_mfma_kernel<MT, ct>), grid, dim3(kMixNT), 0, obs, alpha, beta, b, bm, c1, lw, centre, \
                  d_lens, F, FPB, T, md.N, md.M, md.D, G, MP, R);                                                    \
        break;
    switch (CT) {
        LN_MIX_CASE(1)
        LN_MIX_CASE(2)
        LN_MIX_CASE(3)
        LN_MIX_CASE(4)
        LN_MIX_CASE(5)
        default: return fail(LNASR_E_INVALID, "mix statistics: unsupported D=%d", md.D);
    }
#undef LN_MIX_CASE
    return 0;
}

int mix_stats_device(lnasr_ctx* ctx, const double* obs, const double* alpha, const double* beta, const double* b,
                     const double* bm, const double* c1, const int32_t* d_lens, int U, int T, double* mix, double* mix_x,
                     double* mix_xx, int phase) {
    LN_DEVPTR(obs); LN_DEVPTR(alpha); LN_DEVPTR(beta); LN_DEVPTR(b); LN_DEVPTR(bm); LN_DEVPTR(c1); LN_DEVPTR(d_lens); LN_DEVPTR(mix); LN_DEVPTR(mix_x); LN_DEVPTR(mix_xx);
    Model& md = ctx->model;
    const int N = md.N, M = md.M, D = md.D;
    // M > 32: MG groups of 32 mixtures per state, one wave each
    const int MG = (M + 31) / 32, MT = MG > 1 ? 2 : (M + 15) / 16, CT = (2 * D + 1 + 15) / 16;
    const int MP = MG > 1 ? 32 * MG : 16 * MT, NFP = 16 * CT;
    const long F = (long)U * T;
    const size_t nraw = (size_t)N * MP * NFP;
    LN_TRY(ctx->ws_raw.reserve((nraw + 64) * 8));
    double* R = ctx->ws_raw.as<double>();
    double* centre = R + nraw;
    if (phase & kMixBegin) {
        LN_HIP(hipMemsetAsync(R, 0, nraw * 8, ctx->stream));
        LN_LAUNCH(ctx, "mix_centre", mix_centre_kernel, dim3(D), dim3(256), 0, md.mu.as<double>(), N * M, D, centre);
    }
    if (phase & kMixAdd) {
        const int SG = (N * MG + kMixWaves - 1) / kMixWaves;
        const long nchunk = (F + kMixDiagFrames - 1) / kMixDiagFrames;
        long G = (2L * ctx->prop.multiProcessorCount + SG - 1) / SG;
        if (G > nchunk) G = nchunk;
        if (G < 1) G = 1;
        if (G > 65535) G = 65535;
        const long FPB = ((nchunk + G - 1) / G) * kMixDiagFrames;
        G = (F + FPB - 1) / FPB;
        const dim3 grid(SG, (unsigned)G);
        if (MT == 1)
            LN_TRY(mix_launch<1>(ctx, CT, grid, obs, alpha, beta, b, bm, c1, centre, d_lens, F, FPB, T, MG, MP, R));
        else
            LN_TRY(mix_launch<2>(ctx, CT, grid, obs, alpha, beta, b, bm, c1, centre, d_lens, F, FPB, T, MG, MP, R));
    }
    if (phase & kMixEnd) {
        const size_t nel = (size_t)N * M * D;
        LN_LAUNCH(ctx, "mix_finish", mix_finish_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, (const double*)R,
                  md.mu.as<double>(), (const double*)centre, N, M, D, MP, NFP, mix, mix_x, mix_xx);
    }
    return 0;
}

}  // namespace lnasr
