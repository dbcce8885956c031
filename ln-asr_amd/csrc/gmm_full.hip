// Full-covariance GMM bank on the fp64 matrix cores (gmm.py:59-68 inside gmmhmm.py:46-67).
//
// The reference's own M-step (gmmhmm.py:138-147) returns FULL covariance matrices, so every model
// that went through GMMHMM.train() is scored here.  With P = Sigma^-1 (LU semantics, not
// necessarily symmetric):
//
//   -1/2 (x-mu)^T P (x-mu) = sum_{d<=e} x_d x_e S[d,e] + sum_d x_d L[d] + K
//       S[d,d] = -1/2 P[d,d],  S[d,e] = -1/2 (P[d,e] + P[e,d]) (d < e),
//       L = 1/2 (P + P^T) mu,  K = -1/2 mu^T P mu  (+ log w - 1/2 (D ln 2pi + ln det Sigma))
//
// i.e. a GEMM  [frames x Kf] . [Kf x N*M]  with Kf = D(D+1)/2 + D + 1 quadratic features (820 at D = 39),
// followed by the same in-lane M-way log-sum-exp as the diagonal kernel.  The coefficient matrix
// (14 MB at the metric shape) cannot stay in registers: it is streamed from L2 in 80-feature chunks,
// and every fragment load is reused for FT = 4 frame tiles (64 frames) per wave.  blockIdx is decoded
// so that the blocks of one XCD work on the same state slab and its coefficients stay in that L2.
#include "common.h"
#include "devmath.h"

namespace lnasr {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kFC = 80;          // features per chunk (20 MFMA k-steps)
constexpr int kFKS = 20;

// feature k -> (d, e): pairs d <= e row by row, then the D linear terms (e = 0xFF), then the constant
// (d = e = 0xFF); anything beyond is padding (d = 0xFE).
__global__ void pair_table_kernel(int D, int Kpad, int* __restrict__ tab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kpad) return;
    const int npair = D * (D + 1) / 2;
    int d = 0xFE, e = 0xFE;
    if (k < npair) {
        int rem = k;
        d = 0;
        while (rem >= D - d) {
            rem -= D - d;
            ++d;
        }
        e = d + rem;
    } else if (k < npair + D) {
        d = k - npair;
        e = 0xFF;
    } else if (k == npair + D) {
        d = 0xFF;
        e = 0xFF;
    }
    tab[k] = d | (e << 8);
}

// wfull[(((sg*NCH + ch)*Q + q)*KS + ks)*64 + lane]: coefficient of component (state 4 sg + (row&3),
// mixture 4 q + (row>>2)), row = lane & 15, feature k = 80 ch + 4 ks + (lane >> 4)
__global__ void pack_wfull_kernel(const double* __restrict__ mu, const double* __restrict__ prec,
                                  const double* __restrict__ gconst, const double* __restrict__ logw,
                                  const int* __restrict__ pairtab, double* __restrict__ wfull, int N, int M, int D,
                                  int Q, int NCH, int fold_w) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(N / 4) * NCH * Q * kFKS * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t r1 = idx >> 6;
    const int ks = (int)(r1 % kFKS);
    r1 /= kFKS;
    const int q = (int)(r1 % Q);
    r1 /= Q;
    const int ch = (int)(r1 % NCH);
    const int sg = (int)(r1 / NCH);
    const int row = lane & 15, k = kFC * ch + 4 * ks + (lane >> 4);
    const int state = 4 * sg + (row & 3), mix = 4 * q + (row >> 2);
    const size_t c = (size_t)state * M + mix;
    const double* P = prec + c * D * D;
    const double* m = mu + c * D;
    const int code = pairtab[k], d = code & 0xFF, e = (code >> 8) & 0xFF;
    double val = 0.0;
    if (d < 0xFE && e < 0xFE) {
        val = d == e ? -0.5 * P[(size_t)d * D + d] : -0.5 * (P[(size_t)d * D + e] + P[(size_t)e * D + d]);
    } else if (d < 0xFE && e == 0xFF) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s += (P[(size_t)d * D + j] + P[(size_t)j * D + d]) * m[j];
        val = 0.5 * s;
    } else if (d == 0xFF) {
        double s = 0.0;
        for (int i = 0; i < D; ++i) {
            double t = 0.0;
            for (int j = 0; j < D; ++j) t += P[(size_t)i * D + j] * m[j];
            s += m[i] * t;
        }
        val = (fold_w ? logw[c] : 0.0) + gconst[c] - 0.5 * s;
    }
    wfull[idx] = val;
}

// Round 5: the same coefficients, one BLOCK per MFMA tile row group -- the 16 components (state group sg, mixture tile q) whose
// values interleave in every 512-byte fragment of wfull.  pack_wfull_kernel (one thread per output value) gathered its inputs:
// a wave read 16 different 12 KB precision matrices at two transposed positions each, and did so again for every k-step and
// for both variants (0.37 ms per launch at the metric shape, 0.74 ms per upload).  Here each component's matrix is read ONCE,
// coalesced, into the LDS, its K coefficients are formed there (same expressions, same summation order: the same bits),
// parked in a [K][16] tile, and the block writes the tile out as whole fragments -- both variants (with / without log w
// folded into the constant) from the one tile.
__global__ __launch_bounds__(256) void pack_wfull_tile_kernel(const double* __restrict__ mu, const double* __restrict__ prec,
                                                              const double* __restrict__ gconst,
                                                              const double* __restrict__ logw, const int* __restrict__ pairtab,
                                                              double* __restrict__ wfull, double* __restrict__ wfull_nw,
                                                              int N, int M, int D, int Q, int NCH) {
    extern __shared__ double lds[];
    const int K80 = NCH * kFC;
    double* coef = lds;                            // [K80][16]
    double* Pl = coef + (size_t)K80 * 16;          // [D][D]: the component in hand
    double* ml = Pl + D * D;                       // [D]
    double* tl = ml + D;                           // [D]: sum_j P_ij m_j
    __shared__ double c_lw[16], c_g[16], c_hs[16]; // constant term: (log w + gconst) - 0.5 m^T P m, kept in its parts
    const int tid = threadIdx.x;
    const int sg = blockIdx.x / Q, q = blockIdx.x % Q;
    const int kconst = D * (D + 1) / 2 + D;
    for (int row = 0; row < 16; ++row) {
        const int state = 4 * sg + (row & 3), mix = 4 * q + (row >> 2);
        const size_t c = (size_t)state * M + mix;
        __syncthreads();                           // the previous component's readers are done with Pl / ml / tl
        for (int e = tid; e < D * D; e += 256) Pl[e] = prec[c * D * D + e];
        if (tid < D) ml[tid] = mu[c * D + tid];
        __syncthreads();
        for (int i = tid; i < D; i += 256) {
            double t = 0.0;
            for (int j = 0; j < D; ++j) t += Pl[i * D + j] * ml[j];
            tl[i] = t;
        }
        __syncthreads();
        for (int k = tid; k < K80; k += 256) {
            const int code = pairtab[k], d = code & 0xFF, e = (code >> 8) & 0xFF;
            double val = 0.0;
            if (d < 0xFE && e < 0xFE) {
                val = d == e ? -0.5 * Pl[d * D + d] : -0.5 * (Pl[d * D + e] + Pl[e * D + d]);
            } else if (d < 0xFE && e == 0xFF) {
                double s = 0.0;
                for (int j = 0; j < D; ++j) s += (Pl[d * D + j] + Pl[j * D + d]) * ml[j];
                val = 0.5 * s;
            } else if (d == 0xFF) {
                double s = 0.0;
                for (int i = 0; i < D; ++i) s += ml[i] * tl[i];
                c_lw[row] = logw[c];
                c_g[row] = gconst[c];
                c_hs[row] = 0.5 * s;
            }
            coef[k * 16 + row] = val;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NCH * kFKS * 64; idx += 256) {
        const int lane = idx & 63, r1 = idx >> 6, ks = r1 % kFKS, ch = r1 / kFKS;
        const int k = kFC * ch + 4 * ks + (lane >> 4), row = lane & 15;
        const size_t out = ((((size_t)sg * NCH + ch) * Q + q) * kFKS + ks) * 64 + lane;
        double v = coef[k * 16 + row], vn = v;
        if (k == kconst) {
            v = (c_lw[row] + c_g[row]) - c_hs[row];
            vn = (0.0 + c_g[row]) - c_hs[row];
        }
        wfull[out] = v;
        wfull_nw[out] = vn;
    }
}

template <int Q, bool WRITE_BM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void emit_full_mfma(
    const double* __restrict__ obs, const double* __restrict__ wfull, const int* __restrict__ pairtab,
    const double* __restrict__ logw, const double* __restrict__ centre, double* __restrict__ b, double* __restrict__ bm, long F, int N, int D, int NG,
    int GY, int NCH, long nsuper, int super_per_block, long nchunks) {
    constexpr int NWV = 8, NT = 512, SPB = 32, FT = 4, FS = 17, OS = SPB + 1;
    extern __shared__ double smem[];
    double* xs = smem;                                   // [D][64] observations of the super-tile
    double* Fl = xs + (size_t)D * 64;                    // [2][FT][kFC * FS] feature chunks (double buffered)
    double* ob = Fl + 2 * FT * kFC * FS;                 // [FT][16 * OS] staged outputs
    double* etab = ob + FT * 16 * OS;                    // [kExpTab]
    double* ltab = etab + kExpTab;                       // [2 * kLogTab]
    int* ptab = reinterpret_cast<int*>(ltab + 2 * kLogTab);   // [NCH * kFC]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    exp_table_init(etab, tid, NT);
    log_table_init(ltab, tid, NT);
    for (int e = tid; e < NCH * kFC; e += NT) ptab[e] = pairtab[e];

    // XCD-aware decode: every XCD serves one state slab (its coefficients stay in that XCD's L2)
    const long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long i = bid >> 3;
    int y;
    long chunk;
    if (GY <= 8 && 8 % GY == 0) {
        y = xcd % GY;
        chunk = i * (8 / GY) + xcd / GY;
    } else {
        y = (int)(i % GY);
        chunk = (i / GY) * 8 + xcd;
    }
    if (chunk >= nchunks) return;
    const long sup0 = chunk * super_per_block;
    long sup1 = sup0 + super_per_block;
    if (sup1 > nsuper) sup1 = nsuper;
    const int sg = NWV * y + w;
    const bool active = sg < NG;
    const int sgc = active ? sg : 0;
    const int M = 4 * Q;
    const int g = lane >> 4, fr = lane & 15;
    const int my_state = 4 * sgc + g;
    double lw[WRITE_BM ? Q * 4 : 1];
    if (WRITE_BM) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) lw[q * 4 + r] = logw[(size_t)my_state * M + 4 * q + r];
    }
    const double* wbase = wfull + (size_t)sgc * NCH * Q * kFKS * 64 + lane;
    const long lim = F * (long)D;
    __syncthreads();

    for (long sup = sup0; sup < sup1; ++sup) {
        // observations of the 64 frames -> LDS, d-major
        const long fbase = sup * 64;
        for (int e = tid; e < 64 * D; e += NT) {
            const int f = e / D, d = e - f * D;
            long ge = fbase * D + e;
            ge = ge < lim ? ge : lim - 1;
            xs[d * 64 + f] = obs[ge] - centre[d];        // centred: the coefficients are those of (x - centre)
        }
        __syncthreads();
        // feature chunk builder: wave w fills rows kk = w, w+8, ... of chunk `ch` into buffer `buf`
        auto build = [&](int ch, int buf) {
            double* Fb = Fl + (size_t)buf * FT * kFC * FS;
#pragma unroll
            for (int r = 0; r < kFC / NWV; ++r) {
                const int kk = w + NWV * r;
                const int code = ptab[ch * kFC + kk], d = code & 0xFF, e = (code >> 8) & 0xFF;
                double val;
                if (d < 0xFE && e < 0xFE) val = xs[d * 64 + lane] * xs[e * 64 + lane];
                else if (d < 0xFE) val = xs[d * 64 + lane];
                else val = d == 0xFF ? 1.0 : 0.0;
                Fb[(lane >> 4) * (kFC * FS) + kk * FS + (lane & 15)] = val;
            }
        };
        double4_t acc[FT][Q];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[ft][q] = double4_t{0.0, 0.0, 0.0, 0.0};

        build(0, 0);
        __syncthreads();
        for (int ch = 0; ch < NCH; ++ch) {
            const int buf = ch & 1;
            if (ch + 1 < NCH) build(ch + 1, buf ^ 1);
            const double* Fb = Fl + (size_t)buf * FT * kFC * FS;
            const double* wc = wbase + (size_t)ch * Q * kFKS * 64;
            double wv[Q], wn[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) wv[q] = wc[(size_t)(q * kFKS) * 64];
#pragma unroll
            for (int ks = 0; ks < kFKS; ++ks) {
                if (ks + 1 < kFKS) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) wn[q] = wc[(size_t)(q * kFKS + ks + 1) * 64];
                }
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) {
                    const double bf = Fb[ft * (kFC * FS) + (4 * ks + g) * FS + fr];
#pragma unroll
                    for (int q = 0; q < Q; ++q)
                        acc[ft][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q], bf, acc[ft][q], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) wv[q] = wn[q];
            }
            __syncthreads();
        }
        // epilogue: in-lane M-way log-sum-exp per frame tile
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            if (WRITE_BM) {
                const long frame = fbase + ft * 16 + fr;
                if (active && frame < F) {
                    double* dst = bm + ((size_t)frame * N + my_state) * M;
#pragma unroll
                    for (int q = 0; q < Q; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[4 * q + r] = acc[ft][q][r];
                }
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ft][q][r] += lw[q * 4 + r];
            }
            double mx = acc[ft][0][0];
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (q + r > 0) mx = vmax_f64(mx, acc[ft][q][r]);
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += exp_nonpos(acc[ft][q][r] - mx, etab);
            ob[ft * 16 * OS + fr * OS + 4 * w + g] = (mx == LN_NEG_INF || mx != mx) ? mx : mx + log_sum(s, ltab);
        }
        __syncthreads();
        for (int e = tid; e < FT * 16 * SPB; e += NT) {
            const int ft = e / (16 * SPB), rem = e % (16 * SPB);
            const int f = rem / SPB, st = rem % SPB;
            const long frame = fbase + ft * 16 + f;
            const int state = SPB * y + st;
            if (frame < F && state < N) b[frame * N + state] = ob[ft * 16 * OS + f * OS + st];
        }
        // the next super-tile's xs / feature writes are fenced by the barrier at the top of the loop
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Direct form on the matrix cores (round 4): -1/2 (x - mu)^T P (x - mu) exactly as the reference groups it (gmm.py:66-68),
// for models the expanded quadratic above is refused for (kappa beyond the gate: means many standard deviations apart, or the
// nearly singular Sigma an M-step on too few frames leaves).  Per component and tile of 16 frames
//     Z = Y P    (Y = X - mu: 16 x Dp, P: Dp x Dp, Dp = 16 JT)  as JT * 4 JT  fp64 MFMAs,   q[f] = sum_e Z[f, e] Y[f, e]
// with P's fragments in registers (loaded once per component and wave, used for 64 frames), X in the LDS in the two layouts the
// operand (frame = lane & 15, d = 4 ks + lane >> 4) and the result (frame = 4 r + (lane >> 4), e = 16 jt + lane & 15) need.
// A wave owns one state and folds its M mixtures into a running log-sum-exp per frame; a block = 4 states x 64 frames.
// The generic VALU kernel this replaces took 3.7 ms for 1000 frames x 2048 components (62 ms before its grid fix).
// ------------------------------------------------------------------------------------------
template <int JT, bool WRITE_BM>
__global__ __launch_bounds__(256) void emit_full_direct_mfma(const double* __restrict__ obs, const double* __restrict__ mu,
                                                            const double* __restrict__ prec,
                                                            const double* __restrict__ gconst,
                                                            const double* __restrict__ logw, double* __restrict__ b,
                                                            double* __restrict__ bm, long F, int N, int M, int D) {
    constexpr int Dp = 16 * JT, KS = 4 * JT, FT = 4, XS = Dp + 1, MC = 16;
    __shared__ double xs[64 * XS];                           // X of the block's 64 frames, zero beyond D and beyond F
    __shared__ double vls[4][MC * 64];                       // per wave: log w + density of up to MC mixtures x 64 frames
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l = lane & 15;
    const long fbase = (long)blockIdx.x * 64;
    for (int e = tid; e < 64 * Dp; e += 256) {
        const int f = e / Dp, d = e - f * Dp;
        const long fr = fbase + f;
        xs[f * XS + d] = (d < D && fr < F) ? obs[fr * D + d] : 0.0;
    }
    __syncthreads();
    const int n = 4 * blockIdx.y + w;
    if (n >= N) return;                                      // (after the only barrier; the waves never synchronise again)
    double* vl = vls[w];
    Lse tot;                                                 // this lane's frame (fbase + lane), over the mixture groups
    for (int m0 = 0; m0 < M; m0 += MC) {
        const int mc = M - m0 < MC ? M - m0 : MC;
        for (int mm = 0; mm < mc; ++mm) {
            const size_t c = (size_t)n * M + m0 + mm;
            const double* P = prec + c * D * D;
            const double* mcn = mu + c * D;
            double Bf[KS][JT], mu_a[KS], mu_d[JT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d = 4 * ks + g;
                mu_a[ks] = d < D ? mcn[d] : 0.0;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int e = 16 * jt + l;
                    Bf[ks][jt] = (d < D && e < D) ? P[(size_t)d * D + e] : 0.0;
                }
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) mu_d[jt] = 16 * jt + l < D ? mcn[16 * jt + l] : 0.0;
            const double gc = gconst[c], lw = logw[c];
#pragma unroll 1                                                 // one tile at a time: unrolled, the four tiles' operands and
            for (int ft = 0; ft < FT; ++ft) {                    // accumulators were all live at once (450 registers at JT = 3)
                double4_t z[JT];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) z[jt] = double4_t{0.0, 0.0, 0.0, 0.0};
                const double* xa = xs + (16 * ft + l) * XS + g;  // operand layout: frame l of the tile, d = 4 ks + g
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const double y = xa[4 * ks] - mu_a[ks];
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
                        z[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, Bf[ks][jt], z[jt], 0, 0, 0);
                }
                // accumulator r of lane group g is row 4 r + g of the tile:  q[frame 4 r + g] = sum_e Z[f, e] (x[f, e] - mu[e]),
                // this lane holding e = 16 jt + l
                double mine = 0.0;                               // lane l < 4 keeps frame 4 l + g
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double* xd = xs + (16 * ft + 4 * r + g) * XS + l;
                    double t = 0.0;
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) t = fma(z[jt][r], xd[16 * jt] - mu_d[jt], t);
                    t += dpp_f64<0xB1>(t);                       // over the 16 lanes of the row
                    t += dpp_f64<0x4E>(t);
                    t += dpp_f64<0x141>(t);
                    t += dpp_f64<0x140>(t);
                    mine = l == r ? t : mine;
                }
                if (l < 4) {
                    const double val = gc - 0.5 * mine;
                    const int fl = 16 * ft + 4 * l + g;
                    vl[mm * 64 + fl] = lw + val;
                    if (WRITE_BM && fbase + fl < F) bm[((size_t)(fbase + fl) * N + n) * M + m0 + mm] = val;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // log-sum-exp over this group's mixtures, one frame per lane (a NaN term poisons the sum, as np.logaddexp does)
        for (int mm = 0; mm < mc; ++mm) tot.add(vl[mm * 64 + lane]);
        __builtin_amdgcn_wave_barrier();
    }
    if (fbase + lane < F) b[(size_t)(fbase + lane) * N + n] = tot.value();
}

// returns 1 if the launch was issued, 0 if this model is not covered (D > 64), < 0 on error
int emit_full_direct_device(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm) {
    LN_DEVPTR(obs); LN_DEVPTR(b); LN_DEVPTR(bm);
    Model& md = ctx->model;
    const int N = md.N, M = md.M, D = md.D;
    if (D > 64 || sw().no_mfma) return 0;
    const int JT = (D + 15) / 16;
    const dim3 grid((unsigned)((F + 63) / 64), (unsigned)((N + 3) / 4)), block(256);
    LN_REQUIRE(grid.y <= 65535, "N=%d too large for the direct-form kernel", N);
#define LN_FD(jt)                                                                                                     \
    do {                                                                                                              \
        if (bm)                                                                                                       \
            LN_LAUNCH(ctx, "emit_full_direct_mfma_bm", (emit_full_direct_mfma<jt, true>), grid, block, 0, obs,         \
                      md.mu.as<double>(), md.prec.as<double>(), md.gconst.as<double>(), md.logw.as<double>(), b, bm, F, N, M, D); \
        else                                                                                                          \
            LN_LAUNCH(ctx, "emit_full_direct_mfma", (emit_full_direct_mfma<jt, false>), grid, block, 0, obs,           \
                      md.mu.as<double>(), md.prec.as<double>(), md.gconst.as<double>(), md.logw.as<double>(), b, bm, F, N, M, D); \
    } while (0)
    if (JT == 1) LN_FD(1);
    else if (JT == 2) LN_FD(2);
    else if (JT == 3) LN_FD(3);
    else LN_FD(4);
#undef LN_FD
    return 1;
}

// padded copy of a full-covariance model: [N,M] -> [Np,Mp]; padding components get weight -inf, zero precision
__global__ void pad_full_model_kernel(const double* __restrict__ mu, const double* __restrict__ prec,
                                      const double* __restrict__ gconst, const double* __restrict__ logw, int N, int M,
                                      int D, int Np, int Mp, double* __restrict__ mu_p, double* __restrict__ prec_p,
                                      double* __restrict__ gconst_p, double* __restrict__ logw_p) {
    const size_t c = blockIdx.x;                       // padded component
    if (c >= (size_t)Np * Mp) return;
    const int n = (int)(c / Mp), m = (int)(c % Mp);
    const bool real = n < N && m < M;
    const size_t s = (size_t)n * M + m;
    if (threadIdx.x == 0) {
        gconst_p[c] = real ? gconst[s] : 0.0;
        logw_p[c] = real ? logw[s] : LN_NEG_INF;
    }
    for (int d = threadIdx.x; d < D; d += blockDim.x) mu_p[c * D + d] = real ? mu[s * D + d] : 0.0;
    for (int e = threadIdx.x; e < D * D; e += blockDim.x) prec_p[c * D * D + e] = real ? prec[s * D * D + e] : 0.0;
}

int pack_full_model(lnasr_ctx* ctx) {
    Model& md = ctx->model;
    const int N = md.N, M = md.M, D = md.D;
    md.wfull_Q = 0;
    // same shape rules as the diagonal bank (gmm.hip): pad N to a multiple of 4 and M to 4, 8, 16 or 16 k
    const int Mp = M <= 4 ? 4 : (M <= 8 ? 8 : (M + 15) / 16 * 16);
    const int Np = (N + 3) / 4 * 4;
    const int virt = Mp > 16 ? Mp / 16 : 1;
    const int Nv = Np * virt, Mv = Mp / virt;
    md.pad_N = md.pad_M = 0;
    if (D > 64) return 0;
    if ((Np != N || Mp != M) && sw().no_pad) return 0;
    const int Q = Mv / 4;
    if (!(Q == 1 || Q == 2 || Q == 4)) return 0;
    const int K = D * (D + 1) / 2 + D + 1;
    const int NCH = (K + kFC - 1) / kFC;
    const size_t lds = ((size_t)D * 64 + 2 * 4 * kFC * 17 + 4 * 16 * 33 + kExpTab + 2 * kLogTab) * 8 + (size_t)NCH * kFC * 4;
    if (lds > 160 * 1024) return 0;
    const double *mu_s = md.mu_c.as<double>(), *pr_s = md.prec.as<double>(), *gc_s = md.gconst.as<double>(),
                 *lw_s = md.logw.as<double>();
    if (Np != N || Mp != M) {
        const size_t NMp = (size_t)Np * Mp;
        LN_TRY(md.pad_mu.reserve(NMp * D * 8));
        LN_TRY(md.pad_prec.reserve(NMp * D * D * 8));
        LN_TRY(md.pad_gconst.reserve(NMp * 8));
        LN_TRY(md.pad_logw.reserve(NMp * 8));
        LN_LAUNCH(ctx, "pad_full_model", pad_full_model_kernel, dim3((unsigned)NMp), dim3(128), 0, mu_s, pr_s, gc_s, lw_s, N,
                  M, D, Np, Mp, md.pad_mu.as<double>(), md.pad_prec.as<double>(), md.pad_gconst.as<double>(),
                  md.pad_logw.as<double>());
        mu_s = md.pad_mu.as<double>();
        pr_s = md.pad_prec.as<double>();
        gc_s = md.pad_gconst.as<double>();
        lw_s = md.pad_logw.as<double>();
        md.pad_N = Np;
        md.pad_M = Mp;
    }
    LN_TRY(md.pairtab.reserve((size_t)NCH * kFC * 4));
    LN_LAUNCH(ctx, "pair_table", pair_table_kernel, dim3((NCH * kFC + 255) / 256), dim3(256), 0, D, NCH * kFC,
              md.pairtab.as<int>());
    const size_t total = (size_t)(Nv / 4) * NCH * Q * kFKS * 64;
    LN_TRY(md.wfull.reserve(total * 8));
    LN_TRY(md.wfull_nw.reserve(total * 8));
    const size_t tile_lds = ((size_t)NCH * kFC * 16 + (size_t)D * D + 2 * D) * 8;
    if (tile_lds <= 150 * 1024 && !sw().no_tilepack) {
        LN_LAUNCH(ctx, "pack_wfull_tile", pack_wfull_tile_kernel, dim3((unsigned)((Nv / 4) * Q)), dim3(256), tile_lds, mu_s, pr_s,
                  gc_s, lw_s, md.pairtab.as<int>(), md.wfull.as<double>(), md.wfull_nw.as<double>(), Nv, Mv, D, Q, NCH);
    } else {
        for (int fold = 1; fold >= 0; --fold)
            LN_LAUNCH(ctx, "pack_wfull", pack_wfull_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mu_s, pr_s,
                      gc_s, lw_s, md.pairtab.as<int>(), (fold ? md.wfull : md.wfull_nw).as<double>(), Nv, Mv, D, Q, NCH, fold);
    }
    md.wfull_Q = Q;
    md.wfull_NCH = NCH;
    md.wfull_virt = virt;
    // sixteen mixtures per (virtual) state: the same bank on the integer matrix cores for large batches (gmm_full_i8.hip)
    md.fi8_nk = 0;
    md.fi8_rho = 0.0;
    if (Q == 4 && !sw().no_i8emit) LN_TRY(pack_full_i8_model(ctx, mu_s, pr_s, gc_s, lw_s, Np, Mp, Nv));
    return 0;
}

template <int Q>
static int launch_full(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm) {
    Model& md = ctx->model;
    const int N = (md.pad_N ? md.pad_N : md.N) * md.wfull_virt, D = md.D, NCH = md.wfull_NCH;   // (virtual, padded) states
    const double* lw = md.pad_N ? md.pad_logw.as<double>() : md.logw.as<double>();
    const int GY = (N + 31) / 32;
    const long nsuper = (F + 63) / 64;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    long want_chunks = (long)cus / GY;
    if (want_chunks < 8) want_chunks = 8;
    want_chunks = (want_chunks / 8) * 8;
    long spb = (nsuper + want_chunks - 1) / want_chunks;
    if (spb < 1) spb = 1;
    const long nchunks = (nsuper + spb - 1) / spb;
    const long grid = (GY <= 8 && 8 % GY == 0) ? ((nchunks * GY + 7) / 8) * 8 : ((nchunks + 7) / 8) * 8 * GY;
    const size_t shm = ((size_t)D * 64 + 2 * 4 * kFC * 17 + 4 * 16 * 33 + kExpTab + 2 * kLogTab) * 8 + (size_t)NCH * kFC * 4;
    if (bm)
        LN_LAUNCH(ctx, "emit_full_mfma_bm", (emit_full_mfma<Q, true>), dim3((unsigned)grid), dim3(512), shm, obs,
                  md.wfull_nw.as<double>(), md.pairtab.as<int>(), lw, (const double*)md.centre.as<double>(), b, bm, F, N, D, N / 4, GY, NCH,
                  nsuper, (int)spb, nchunks);
    else
        LN_LAUNCH(ctx, "emit_full_mfma", (emit_full_mfma<Q, false>), dim3((unsigned)grid), dim3(512), shm, obs,
                  md.wfull.as<double>(), md.pairtab.as<int>(), lw, (const double*)md.centre.as<double>(), b, bm, F, N, D, N / 4, GY, NCH,
                  nsuper, (int)spb, nchunks);
    return 0;
}


// returns 1 if the launch was issued, 0 if this model / dtype is not covered, < 0 on error
int emit_full_mfma_device(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm) {
    LN_DEVPTR(obs); LN_DEVPTR(b); LN_DEVPTR(bm);
    Model& md = ctx->model;
    if (md.wfull_Q == 0 || sw().no_mfma) return 0;
    const int N = md.N, M = md.M, virt = md.wfull_virt;
    const int Np = md.pad_N ? md.pad_N : N, Mp = md.pad_M ? md.pad_M : M;
    const bool padded = md.pad_N != 0;
    double* bdst = b;
    double* bmdst = bm;
    if (virt > 1 || padded) {
        LN_TRY(ctx->ws_misc2.reserve((size_t)F * Np * virt * 8));
        bdst = ctx->ws_misc2.as<double>();
    }
    if (bm && padded) {
        LN_TRY(ctx->ws_pad.reserve((size_t)F * Np * Mp * 8));
        bmdst = ctx->ws_pad.as<double>();
    }
    int rc;
    if (!bm && md.wfull_Q == 4 && md.fi8_Nv == Np * virt && full_i8_route(md, F)) rc = emit_full_i8_device(ctx, obs, F, bdst);
    else if (md.wfull_Q == 4) rc = launch_full<4>(ctx, obs, F, bdst, bmdst);
    else if (md.wfull_Q == 2) rc = launch_full<2>(ctx, obs, F, bdst, bmdst);
    else rc = launch_full<1>(ctx, obs, F, bdst, bmdst);
    if (rc) return rc;
    if (virt > 1 || padded) {
        const size_t cells = (size_t)F * N;
        LN_LAUNCH(ctx, "merge_virtual_states", merge_virtual_kernel, dim3((unsigned)((cells + 255) / 256)),
                  dim3(256), 0, bdst, cells, N, Np, virt, b);
    }
    if (bm && padded) {
        const size_t cells = (size_t)F * N * M;
        LN_LAUNCH(ctx, "compact_bm", compact_bm_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0,
                  (const double*)bmdst, cells, N, M, Np, Mp, bm);
    }
    return 1;
}

}  // namespace lnasr
