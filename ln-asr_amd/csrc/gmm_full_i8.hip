// The FULL-covariance GMM bank with its GEMM on the INTEGER matrix cores (gmm.py:59-68 inside gmmhmm.py:46-67 over a batch).
//
// gmm_full.hip evaluates  -1/2 (x-mu)^T P (x-mu) = sum_{d<=e} x_d x_e S[d,e] + sum_d x_d L[d] + K  as a K = 820 GEMM on the fp64
// matrix cores (0.67 of their peak).  MI355X runs v_mfma_i32_16x16x64_i8 at 64 x that rate, so the same product is formed
// here as in gmm_i8.hip: a 56-bit FIXED-POINT product of seven base-256 digit planes per operand, the 28 plane pairs with
// i + j <= 6, one int32 accumulator per diagonal, folded in fp64 -- 28 x 13 integer matrix instructions per 16 x 16 tile for
// K = 819 features (D = 39) against 205 fp64 ones at four times the cycles.  Scales as in gmm_i8.hip: g per GEMM row from the
// model's own data scale (pair row (d, e): ~ s_d s_e, linear row: ~ s_d; s_d^2 = geometric mean over the components of
// Sigma_dd + (mu_d - centre_d)^2), h_c per component, e_f per frame; error ~ 3 * 2^-53 * sqrt(K / 80) * A_c X_f on a component's
// quadratic.  rho = max_c A_c X_c is measured at upload (full_comp_scale_kernel) and gated at kFullI8RhoMax (beyond it, and
// for small batches, emit_full_mfma scores the model); lnasr_model_range reports it.
//
// Unlike the diagonal bank (W stationary in registers: K = 80) neither operand fits anywhere but HBM / L2: W planes are
// 5.8 KB per component (11.9 MB at the metric shape), feature planes 5.8 KB per frame.  The kernel is a real GEMM:
//   * block = 8 waves, two per SIMD: four CONSUMERS (112 accumulators + 72 fragment registers each) and four LOADERS; tile = 64
//     components (two state pairs) x 64 frames; consumer (cu, fg) owns state pair cu and frame tiles 2 fg, 2 fg + 1;
//   * both operands are stored in MFMA fragment order, [tile][k-step][plane][lane][16 B], so a k-step's share of a tile is
//     seven lane-linear 1 KB pieces that go HBM / L2 -> LDS by global_load_lds_dwordx4 (double buffered: the pieces of
//     k-step s + 1 are in flight under the 112 matrix instructions of step s) and LDS -> registers by one ds_read_b128 each;
//   * blockIdx is decoded so that an XCD owns a fixed group of component slabs (its W planes, 1.5 MB, stay in its L2) and
//     the 32 blocks resident on it share 8 frame blocks' feature planes; W planes arrive one k-step ahead, feature planes
//     (HBM / MALL for the first block that asks) two steps ahead in a third buffer: counted vmcnt, raw s_barrier;
//   * a block walks several consecutive frame blocks of its slab (about four blocks per CU in all): the tables are built once,
//     and while the consumers run a tile's epilogue the loaders are already two steps into the next tile (-5 % at 131,072 frames);
//   * epilogue once per tile: seven conversions and six FMAs per value (the pair join of gmm_i8.hip would overflow at
//     K = 832), 2^(h_c + e_f), the constant, the 16-way log-sum-exp of gmm_i8.hip, b straight to global memory.
// Measured (N M = 2048, D = 39; tools/full_i8_try.py, full_i8_clock.py): 0.65-0.70 ms per 16,000 frames against 0.99 for
// emit_full_mfma, 4.78 against 7.05 per 131,072; in-kernel clock 1.73 GHz -- the integer matrix unit is power-limited on these
// boxes at about half its nominal rate, which is what separates this from the 0.43 ms its instruction counts promise.
// b only (bm: the fp64 kernel).  Kernels: full_coef / full_row_scale / full_comp_scale / full_pack_planes (once per model),
// full_slice_features (per batch, in chunks of kFullChunkFrames frames: 5.8 KB of planes per frame), emit_full_i8.
#include <cstdio>

#include "common.h"
#include "devmath.h"

namespace lnasr {
namespace {

typedef int int4_t __attribute__((ext_vector_type(4)));

constexpr int kS = 7;                                  // digit planes per operand
constexpr double kFullI8Reach = 4.0;

__device__ __forceinline__ int scale_exponent(double m) {
    int e = ilogb(m) + 2;
    if (ldexp(m, -e) > 0.498) ++e;
    return e;
}

// feature k -> (d, e): pairs d <= e row by row (k < D (D + 1) / 2), then the D linear terms (e = -1); beyond: d = -1 (zero row)
__device__ __forceinline__ void pair_of(int k, int D, int& d, int& e) {
    const int npair = D * (D + 1) / 2;
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
        e = -1;
    } else {
        d = -1;
        e = -1;
    }
}

// |v| <= 0.498 -> the seven balanced base-256 digits of v, packed: byte 6 = plane 0 ... byte 0 = plane 6 (as slice_features_kernel)
__device__ __forceinline__ unsigned long long digits_of(double v) {
    const double a = v * 4294967296.0;                 // v 2^32
    const double hd = __builtin_rint(a);
    const double ld = __builtin_rint((a - hd) * 16777216.0);
    const long long t = ((long long)(int)hd << 24) + (long long)(int)ld;
    return (unsigned long long)(t + 0x0080808080808080LL) ^ 0x0080808080808080ULL;
}

// sixteen packed values -> the 16 bytes of plane i (byte position 6 - i of every value)
__device__ __forceinline__ int4_t plane_of(const unsigned long long u[16], int i) {
    const int sh = 8 * (6 - i);
    unsigned w[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        w[m] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) w[m] |= (unsigned)((u[4 * m + b] >> sh) & 255ull) << (8 * b);
    }
    return int4_t{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
}

// Wc[c * Kp + k]: coefficient of feature k for component c (0 for k past the features); mu = centred means, P = Sigma^-1
__global__ void full_coef_kernel(const double* __restrict__ mu, const double* __restrict__ prec, int C, int D, int Kp,
                                 double* __restrict__ Wc) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)C * Kp) return;
    const size_t c = idx / Kp;
    const int k = (int)(idx % Kp);
    int d, e;
    pair_of(k, D, d, e);
    const double* P = prec + c * D * D;
    double val = 0.0;
    if (d >= 0 && e >= 0) {
        val = d == e ? -0.5 * P[(size_t)d * D + d] : -0.5 * (P[(size_t)d * D + e] + P[(size_t)e * D + d]);
    } else if (d >= 0) {
        const double* m = mu + c * D;
        double s = 0.0;
        for (int j = 0; j < D; ++j) s += (P[(size_t)d * D + j] + P[(size_t)j * D + d]) * m[j];
        val = 0.5 * s;
    }
    Wc[idx] = val;
}

// gl[d] = -round(1/2 mean_c log2(Sigma_c[d][d] + mu_c[d]^2)) over the REAL components ([N M] arrays); pow2gl[d] = 2^gl[d]
__global__ void full_row_scale_kernel(const double* __restrict__ mu, const double* __restrict__ sigma, int NM, int D,
                                      int* __restrict__ gl, double* __restrict__ pow2gl) {
    const int d = blockIdx.x;
    __shared__ double red[256];
    __shared__ int cnt[256];
    double sum = 0.0;
    int n = 0;
    for (int c = threadIdx.x; c < NM; c += 256) {
        const double var = sigma[((size_t)c * D + d) * D + d], m = mu[(size_t)c * D + d];
        const double s2 = var + m * m;
        if (s2 > 0.0 && __builtin_isfinite(s2)) {
            sum += log2(s2);
            ++n;
        }
    }
    red[threadIdx.x] = sum;
    cnt[threadIdx.x] = n;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[threadIdx.x] += red[threadIdx.x + s];
            cnt[threadIdx.x] += cnt[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double l = cnt[0] > 0 ? red[0] / cnt[0] : 0.0;
        if (!(l > -900.0)) l = -900.0;
        if (l > 900.0) l = 900.0;
        const int g = -(int)__builtin_rint(0.5 * l);
        gl[d] = g;
        pow2gl[d] = ldexp(1.0, g);
    }
}

__device__ __forceinline__ int row_exponent(int k, int D, const int* __restrict__ gl) {
    int d, e;
    pair_of(k, D, d, e);
    return d < 0 ? 0 : (e < 0 ? gl[d] : gl[d] + gl[e]);
}

// One wave per (padded) component c = n Mp + m:  h[c], the constant (log w + gconst - 1/2 mu^T P mu), and rho (atomic max).
__global__ __launch_bounds__(256) void full_comp_scale_kernel(const double* __restrict__ Wc, const double* __restrict__ mu_p,
                                                              const double* __restrict__ prec_p, const double* __restrict__ gconst_p,
                                                              const double* __restrict__ logw_p, const double* __restrict__ mu_r,
                                                              const double* __restrict__ sigma_r, const int* __restrict__ gl, int N,
                                                              int M, int Np, int Mp, int D, int Kfeat, int Kp,
                                                              int* __restrict__ h, double* __restrict__ constc,
                                                              unsigned long long* __restrict__ rho_bits) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= Np * Mp) return;
    const int n = c / Mp, m = c % Mp;
    const bool real = n < N && m < M;
    double A = 0.0;
    for (int k = lane; k < Kfeat; k += 64) A = fmax(A, fabs(ldexp(Wc[(size_t)c * Kp + k], -row_exponent(k, D, gl))));
    double r = 0.0;
    if (real) {
        const size_t cr = (size_t)n * M + m;
        for (int d = lane; d < D; d += 64) {
            const double var = sigma_r[(cr * D + d) * D + d];
            r = fmax(r, ldexp(fabs(mu_r[cr * D + d]) + kFullI8Reach * sqrt(var > 0.0 ? var : 0.0), gl[d]));
        }
    }
    double q = 0.0;                                         // mu^T P mu
    {
        const double* P = prec_p + (size_t)c * D * D;
        const double* mm = mu_p + (size_t)c * D;
        for (int i = lane; i < D * D; i += 64) q += mm[i / D] * P[i] * mm[i % D];
    }
    for (int o = 32; o > 0; o >>= 1) {
        A = fmax(A, __shfl_xor(A, o));
        r = fmax(r, __shfl_xor(r, o));
        q += __shfl_xor(q, o);
    }
    if (lane == 0) {
        h[c] = (A > 0.0 && __builtin_isfinite(A)) ? scale_exponent(A) : 0;
        constc[c] = logw_p[c] + gconst_p[c] - 0.5 * q;
        double rho = real ? A * fmax(r * r, r) : 0.0;
        if (!__builtin_isfinite(rho)) rho = __builtin_huge_val();
        atomicMax(rho_bits, (unsigned long long)__double_as_longlong(rho));
    }
}

// W digit planes in A-operand order: wq[((ct * nk + ks) * 7 + plane) * 64 + lane], comp tile ct = 2 sp + q; row r = lane & 15
// <-> state 2 sp + ((r >> 2) & 1), mixture 8 (r >> 3) + 4 q + (r & 3) (the row mapping of gmm_i8.hip); k = 64 ks + 16 (lane >> 4) + byte
__global__ void full_pack_planes_kernel(const double* __restrict__ Wc, const int* __restrict__ gl, const int* __restrict__ h,
                                        int4_t* __restrict__ wq, int Nv, int D, int Kp, int nk) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)Nv * nk * 64;             // (Nv / 2 pairs) x 2 tiles x nk x 64 lanes
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t r1 = idx >> 6;
    const int ks = (int)(r1 % nk);
    const size_t ct = r1 / nk;
    const int q = (int)(ct & 1);
    const size_t sp = ct >> 1;
    const int r = lane & 15, hh = lane >> 4;
    const size_t c = (2 * sp + ((r >> 2) & 1)) * 16 + 8 * (r >> 3) + 4 * q + (r & 3);
    const int hc = h[c];
    unsigned long long u[16];
#pragma unroll
    for (int by = 0; by < 16; ++by) {
        const int k = 64 * ks + 16 * hh + by;
        u[by] = digits_of(ldexp(Wc[c * Kp + k], -row_exponent(k, D, gl) - hc));
    }
#pragma unroll
    for (int i = 0; i < kS; ++i) wq[((ct * nk + ks) * kS + i) * 64 + lane] = plane_of(u, i);
}

// Feature planes in B-operand order: fq[((ft * nk + ks) * 7 + plane) * 64 + lane], lane = 16 hh + col, frame 16 ft + col,
// k = 64 ks + 16 hh + byte; fsc[frame] = 2^(e_f - 24) (NaN for a frame with a non-finite feature: the pass over such frames,
// nanfix.hip, re-scores them anyway).  Block = one frame tile: thread (col, chunk) owns 16 features of one frame.
__global__ __launch_bounds__(1024) void full_slice_features_kernel(const double* __restrict__ obs, const double* __restrict__ centre,
                                                                   const double* __restrict__ pow2gl, int4_t* __restrict__ fq,
                                                                   double* __restrict__ fsc, long F, int D, int nk) {
    extern __shared__ double sh[];
    double* xs = sh;                                       // [16][D] centred, scaled by 2^gl: x_d / s_d
    double* smax = sh + 16 * 40;                           // [nk * 4][16]
    unsigned char* tab = reinterpret_cast<unsigned char*>(smax + 64 * 16);   // [nk * 64][2]: (d, e) of feature k; 255 = none
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const long ft = blockIdx.x, frame0 = ft * 16;
    for (int i = tid; i < 16 * D; i += nthreads) {
        const long f = frame0 + i / D;
        const int d = i % D;
        xs[i] = f < F ? (obs[f * D + d] - centre[d]) * pow2gl[d] : 0.0;
    }
    for (int k = tid; k < nk * 64; k += nthreads) {
        int d, e;
        pair_of(k, D, d, e);
        tab[2 * k] = (unsigned char)(d < 0 ? 255 : d);
        tab[2 * k + 1] = (unsigned char)(e < 0 ? 255 : e);
    }
    __syncthreads();
    const int col = tid & 15, chunk = tid >> 4;            // chunk < nk * 4
    const double* xf = xs + col * D;
    double val[16];
    double mx = 0.0;
#pragma unroll
    for (int by = 0; by < 16; ++by) {
        const int k = 16 * chunk + by;
        const int d = tab[2 * k], e = tab[2 * k + 1];
        double v = 0.0;
        if (d != 255) v = e != 255 ? xf[d] * xf[e] : xf[d];   // (x_d / s_d)(x_e / s_e) = x_d x_e 2^(gl_d + gl_e)
        val[by] = v;
        mx = fmax(mx, fabs(v));
    }
    smax[chunk * 16 + col] = mx;                           // (a NaN value: fmax drops it; the frame is caught below)
    __syncthreads();
    bool bad = false;
    for (int d = 0; d < D; ++d) bad |= !__builtin_isfinite(xf[d]);
    mx = 0.0;
    for (int ch = 0; ch < nk * 4; ++ch) mx = fmax(mx, smax[ch * 16 + col]);
    if (!__builtin_isfinite(mx)) bad = true;               // a product overflowed
    const int ef = (mx > 0.0 && !bad) ? scale_exponent(mx) : 0;
    const double down = bad ? 0.0 : ldexp(1.0, -ef);
    unsigned long long u[16];
#pragma unroll
    for (int by = 0; by < 16; ++by) u[by] = digits_of(bad ? 0.0 : val[by] * down);
    const int ks = chunk >> 2, hh = chunk & 3;
#pragma unroll
    for (int i = 0; i < kS; ++i) fq[(((size_t)ft * nk + ks) * kS + i) * 64 + 16 * hh + col] = plane_of(u, i);
    if (chunk == 0 && frame0 + col < F) fsc[frame0 + col] = bad ? __builtin_nan("") : ldexp(1.0, ef - 24);
}

// The GEMM.  Grid: bid -> xcd = bid & 7, inner = bid >> 3; slab (4 states = 64 components) = xcd + 8 (inner % spx), frame
// block (64 frames) = inner / spx, spx = slabs per XCD.  Block = 8 waves, two per SIMD: waves 0-3 CONSUME (wave (cu, fg):
// state pair cu, frame tiles 2 fg, 2 fg + 1 -- 112 matrix instructions per k-step), waves 4-7 LOAD (the 56 one-KB pieces of the
// next k-step by LDS-DMA, 14 each): an LDS-DMA instruction holds its wave's issue slot for ~60-100 cycles, which in a
// consumer's own stream starved the matrix pipe (first version, four waves doing both: 4900 cycles per k-step for 1792 of
// matrix work); beside a consumer on the same SIMD it is free.  The consumer streams the B planes one at a time (plane i
// meets A planes 0 .. 6 - i: the same 28 pairs), so its fragments are 56 + 16 registers and everything fits two waves per SIMD.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void emit_full_i8(
    const int4_t* __restrict__ fq, const double* __restrict__ fsc, const int4_t* __restrict__ wq,
    const double* __restrict__ constc, const int* __restrict__ hexp, double* __restrict__ b, long F, int Nv, int nk,
    long nftiles, int slabs, int spx, long fblocks, int tiles_pb) {
    __shared__ int4_t bufA[2][4 * kS * 64];                // [buffer][(cu, q)][plane][lane]: 28 KB each; W planes: from this XCD's L2
    __shared__ int4_t bufB[3][4 * kS * 64];                // [buffer][frame tile][plane][lane]; feature planes: from HBM / MALL, TWO steps ahead
    __shared__ double etab[kExpTabBig];
    __shared__ double ltab[2 * kLogTab];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    LN_BLOCK_T0;
    const long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long inner = bid >> 3;
    const int slab = xcd + 8 * (int)(inner % spx);
    const long fblock0 = (inner / spx) * tiles_pb;         // the block walks tiles_pb consecutive frame blocks of its slab: the
    if (slab >= slabs || fblock0 >= fblocks) return;       // tables are built once, the loaders run ahead into the next tile
    const int ntile = (int)(fblocks - fblock0 < tiles_pb ? fblocks - fblock0 : tiles_pb);    // while the consumers finish this one
    const int nsteps = ntile * nk;                          // k-steps of the whole walk: step gs = tile gs / nk, k-step gs % nk
    exp_table_big_init(etab, tid, 512);
    log_table_init(ltab, tid, 512);
    const bool loader = w >= 4;
    const int cu = w & 1, fg = (w >> 1) & 1;
    // The 28 + 28 one-KB pieces of a k-step: loader wave lw moves pieces lw, lw + 4, ... (seven of each operand).  The W
    // planes of step s + 1 come from this XCD's L2 and land within a step; the feature planes are shared by the XCD's four
    // slabs only, so the first block to ask pays HBM / MALL latency (~2 us: with one step of lead every k-step took 3860
    // cycles for 1792 of matrix work) -- they are requested TWO steps ahead into a third buffer, and the loader waits with a
    // COUNTED vmcnt (its seven newest requests may stay in flight across the barrier: a raw s_barrier, not __syncthreads,
    // whose fence would drain them).
    const int lw = w - 4;
    auto load_A = [&](int gs) {
        const int ks = gs % nk;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int p = lw + 4 * i;                       // wave-uniform: 0 .. 27
            const int t = p / kS, pl = p % kS;              // tile of the block (0 .. 3), plane
            const size_t ct = (size_t)(2 * slab + (t >> 1)) * 2 + (t & 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(wq + ((ct * nk + ks) * kS + pl) * 64 + lane),
                (__attribute__((address_space(3))) void*)(&bufA[gs & 1][(t * kS + pl) * 64]), 16, 0, 0);
        }
    };
    auto load_B = [&](int gs) {
        const int buf = gs % 3, ks = gs % nk;
        const long ft0 = (fblock0 + gs / nk) * 4;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int p = lw + 4 * i;
            const int t = p / kS, pl = p % kS;
            long ftile = ft0 + t;
            if (ftile >= nftiles) ftile = nftiles - 1;      // (past the batch: any valid tile; the results are not stored)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(fq + (((size_t)ftile * nk + ks) * kS + pl) * 64 + lane),
                (__attribute__((address_space(3))) void*)(&bufB[buf][(t * kS + pl) * 64]), 16, 0, 0);
        }
    };
    // The two roles walk the same sequence of barriers in code of their own (a wave-uniform branch: every wave of the block
    // executes nk barriers), so that neither role's live values are charged to the other's loop.
    if (loader) {
        load_A(0);
        load_B(0);
        if (nsteps > 1) load_B(1);
        for (int gs = 0; gs < nsteps; ++gs) {
            // everything but the newest feature-plane requests (step gs + 1's, issued last) has landed
            if (gs + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0x0F77);    // vmcnt(7)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
            __builtin_amdgcn_s_barrier();                   // step gs is in the LDS; the consumers have read step gs - 1
            if (gs + 1 < nsteps) load_A(gs + 1);
            if (gs + 2 < nsteps) load_B(gs + 2);
        }
        return;
    }
    int4_t acc[kS][2][2];
#pragma unroll
    for (int d = 0; d < kS; ++d)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[d][q][t] = int4_t{0, 0, 0, 0};
    // epilogue constants: this lane holds, for frame tile t, frame col = lane & 15 and state 2 sp + (g & 1), the mixtures
    // 8 (g >> 1) + 4 q + e
    const int g = lane >> 4, col = lane & 15;
    const int sp = 2 * slab + cu;
    const int state = 2 * sp + (g & 1), mb = 8 * (g >> 1);
    double cst[8];
    int hx[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        cst[m] = constc[(size_t)state * 16 + mb + m];
        hx[m] = hexp[(size_t)state * 16 + mb + m];
    }
    auto swap_max = [](double x) {
        auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        return vmax_f64(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
    };
    auto swap_sum = [](double x) {
        auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    };
    auto epilogue = [&](long ft0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long frame = (ft0 + 2 * fg + t) * 16 + col;
        const double scale = frame < F ? fsc[frame] : 0.0;
        double v[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // unit: the weight of diagonal 1 (256^-3); diagonal d weighs 256^(1 - d) of it
                double r = (double)acc[6][q][t][e];
                r = fma(r, 0x1p-8, (double)acc[5][q][t][e]);
                r = fma(r, 0x1p-8, (double)acc[4][q][t][e]);
                r = fma(r, 0x1p-8, (double)acc[3][q][t][e]);
                r = fma(r, 0x1p-8, (double)acc[2][q][t][e]);
                r = fma(r, 0x1p-8, (double)acc[1][q][t][e]);
                r = fma((double)acc[0][q][t][e], 256.0, r);
                v[q][e] = fma(r, ldexp(scale, hx[4 * q + e]), cst[4 * q + e]);
            }
        double mx = v[0][0];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q + e > 0) mx = vmax_f64(mx, v[q][e]);
        mx = swap_max(mx);
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += exp_nonpos_big2(v[q][e] - mx, etab);
        s = swap_sum(s);
        const double res = (mx == LN_NEG_INF) ? mx : mx + log_sum(s, ltab);
        if (g < 2 && frame < F && state < Nv) b[frame * Nv + state] = res;
    }
    };
    drain_vmem();
    for (int gs = 0; gs < nsteps; ++gs) {
        const int cur = gs & 1, curB = gs % 3;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // the LDS reads below stay below the barrier
        // Fragments are read in the order of use and the 28 plane pairs are walked by GROUPS g = max(i, j): group g needs the
        // planes g of both operands and nothing newer, so the matrix instructions start behind the first four reads and the
        // other 24 reads return under them.
        int4_t Af[2][kS], Bf[2][kS];
#pragma unroll
        for (int p = 0; p < kS; ++p) {
#pragma unroll
            for (int q = 0; q < 2; ++q) Af[q][p] = bufA[cur][((2 * cu + q) * kS + p) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) Bf[t][p] = bufB[curB][((2 * fg + t) * kS + p) * 64 + lane];
        }
#define LN_PAIR(j, i)                                                                                           \
    do {                                                                                                        \
        if ((j) + (i) < kS) {                                                                                   \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int t = 0; t < 2; ++t)         \
                acc[(j) + (i)][q][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[q][j], Bf[t][i], acc[(j) + (i)][q][t], 0, 0, 0); \
        }                                                                                                       \
    } while (0)
#pragma unroll
        for (int g = 0; g < kS; ++g) {
#pragma unroll
            for (int o = 0; o < g; ++o) {                   // (o, g) and (g, o): A plane o with B plane g, A plane g with B plane o
                LN_PAIR(o, g);
                LN_PAIR(g, o);
            }
            LN_PAIR(g, g);
        }
#undef LN_PAIR
        if ((gs + 1) % nk != 0) continue;                   // the tile's last k-step: its epilogue, then the next tile from zero
        epilogue((fblock0 + gs / nk) * 4);
#pragma unroll
        for (int d = 0; d < kS; ++d)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[d][q][t] = int4_t{0, 0, 0, 0};
    }
    LN_BLOCK_T1;
}
LN_BLOCK_TIMES_EXPORT(lnasr_dbg_block_times_full_i8)

}  // namespace

// Measured (tests/test_gpu_full_i8.py::test_full_range_gate, relative to max(1, |b|), against long double): rho = 101: 8e-16,
// 365: 2.8e-15, 9197 (cond(Sigma) = 2710): 2.7e-13 -- where the fp64 matrix-core GEMM has 1.6e-13: on ill-conditioned
// covariances the expanded quadratic loses in any arithmetic.  4096 leaves a factor of eight under 1e-12 max(1, |b|).
constexpr double kFullI8RhoMax = 4096.0;
constexpr long kFullChunkFrames = 1L << 17;               // feature planes of 131,072 frames at a time: 0.76 GB

bool full_i8_model_ok(const Model& md) {
    return md.fi8_nk > 0 && !sw().no_i8emit && md.fi8_rho <= (sw().i8_rho_max_set ? sw().i8_rho_max : kFullI8RhoMax);
}
bool full_i8_route(const Model& md, long F) {
    return full_i8_model_ok(md) && (sw().i8_min_frames >= 0 ? F >= sw().i8_min_frames : F >= 2048);
}

// once per full-covariance model (pack_full_model has padded the parameters): scales, range, digit planes.  mu_p / prec_p /
// gconst_p / logw_p: the (padded) [Np, Mp] arrays the fp64 packer used; Nv = Np * virt states of 16 mixtures
int pack_full_i8_model(lnasr_ctx* ctx, const double* mu_p, const double* prec_p, const double* gconst_p, const double* logw_p,
                       int Np, int Mp, int Nv) {
    Model& md = ctx->model;
    md.fi8_nk = 0;
    md.fi8_rho = 0.0;
    const int D = md.D, N = md.N, M = md.M;
    const int Kfeat = D * (D + 1) / 2 + D;
    const int nk = (Kfeat + 63) / 64, Kp = nk * 64;
    if (D > 40 || nk > 13 || Nv % 4 != 0) return 0;
    const size_t C = (size_t)Np * Mp;
    LN_TRY(ctx->ws_misc.reserve(C * Kp * 8));             // Wc: scratch, dead after the planes are packed
    double* Wc = ctx->ws_misc.as<double>();
    LN_TRY(md.fi8_gl.reserve(64 * 4));
    LN_TRY(md.fi8_pow2gl.reserve(64 * 8));
    LN_TRY(md.fi8_h.reserve(C * 4));
    LN_TRY(md.fi8_const.reserve(C * 8));
    LN_TRY(md.fi8_wq.reserve((size_t)Nv * nk * kS * 64 * 16));
    LN_TRY(md.flags.reserve(256));
    unsigned long long* rbits = reinterpret_cast<unsigned long long*>(md.flags.as<char>() + 24);
    LN_HIP(hipMemsetAsync(rbits, 0, 8, ctx->stream));
    LN_LAUNCH(ctx, "full_coef", full_coef_kernel, dim3((unsigned)((C * Kp + 255) / 256)), dim3(256), 0, mu_p, prec_p, (int)C, D, Kp, Wc);
    LN_LAUNCH(ctx, "full_row_scale", full_row_scale_kernel, dim3(D), dim3(256), 0, (const double*)md.mu_c.as<double>(),
              (const double*)md.sigma_raw.as<double>(), N * M, D, md.fi8_gl.as<int>(), md.fi8_pow2gl.as<double>());
    LN_LAUNCH(ctx, "full_comp_scale", full_comp_scale_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (const double*)Wc, mu_p,
              prec_p, gconst_p, logw_p, (const double*)md.mu_c.as<double>(), (const double*)md.sigma_raw.as<double>(),
              (const int*)md.fi8_gl.as<int>(), N, M, Np, Mp, D, Kfeat, Kp, md.fi8_h.as<int>(), md.fi8_const.as<double>(), rbits);
    const size_t total = (size_t)Nv * nk * 64;
    LN_LAUNCH(ctx, "full_pack_planes", full_pack_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
              (const double*)Wc, (const int*)md.fi8_gl.as<int>(), (const int*)md.fi8_h.as<int>(), md.fi8_wq.as<int4_t>(), Nv, D, Kp, nk);
    unsigned long long rb = 0;
    LN_HIP(hipMemcpyAsync(&rb, rbits, 8, hipMemcpyDeviceToHost, ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream));
    md.fi8_rho = __builtin_bit_cast(double, rb);
    md.fi8_nk = nk;
    md.fi8_Nv = Nv;
    return 0;
}

// b [F, Nv] for the (virtual, padded) states the planes were packed for
int emit_full_i8_device(lnasr_ctx* ctx, const double* obs, long F, double* b) {
    Model& md = ctx->model;
    const int D = md.D, nk = md.fi8_nk, Nv = md.fi8_Nv;
    const int slabs = Nv / 4, spx = (slabs + 7) / 8;
    const size_t shm = (size_t)(16 * 40 + 64 * 16) * 8 + (size_t)nk * 64 * 2;
    for (long f0 = 0; f0 < F; f0 += kFullChunkFrames) {
        const long Fc = F - f0 < kFullChunkFrames ? F - f0 : kFullChunkFrames;
        const long nftiles = (Fc + 15) / 16, fblocks = (nftiles + 3) / 4;
        LN_TRY(ctx->ws_fs.reserve((size_t)nftiles * nk * kS * 64 * 16 + (size_t)nftiles * 16 * 8));
        int4_t* fq = ctx->ws_fs.as<int4_t>();
        double* fsc = reinterpret_cast<double*>(fq + (size_t)nftiles * nk * kS * 64);
        LN_LAUNCH(ctx, "full_slice_features", full_slice_features_kernel, dim3((unsigned)nftiles), dim3(16 * nk * 4), shm,
                  obs + (size_t)f0 * D, (const double*)md.centre.as<double>(), (const double*)md.fi8_pow2gl.as<double>(), fq, fsc,
                  Fc, D, nk);
        // a block walks `tiles_pb` consecutive frame blocks of its slab: about four blocks per CU in all
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        long tiles_pb = (long)slabs * fblocks / (4L * cus);
        if (tiles_pb < 1) tiles_pb = 1;
        if (tiles_pb > 64) tiles_pb = 64;
        const long walks = (fblocks + tiles_pb - 1) / tiles_pb;
        const long grid = 8L * spx * walks;
        LN_LAUNCH(ctx, "emit_full_i8", emit_full_i8, dim3((unsigned)grid), dim3(512), 0, (const int4_t*)fq, (const double*)fsc,
                  (const int4_t*)md.fi8_wq.as<int4_t>(), (const double*)md.fi8_const.as<double>(),
                  (const int*)md.fi8_h.as<int>(), b + (size_t)f0 * Nv, Fc, Nv, nk, nftiles, slabs, spx, fblocks, (int)tiles_pb);
    }
    return 0;
}

}  // namespace lnasr
