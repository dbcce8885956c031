// The diagonal GMM bank with its GEMM on the INTEGER matrix cores (gmm.py:46-67 / gmmhmm.py:46-67 over a batch).
//
// MI355X halves the fp64 matrix rate of its predecessor (78.6 TFLOP/s = the fp64 vector rate, and an fp64 MFMA blocks every
// vector instruction of its SIMD, gmm.hip) and doubles the int8 rate (v_mfma_i32_16x16x64_i8: 2048 int-ops per cycle and
// SIMD, 64 x the fp64 rate).  The quadratic part of the component log-density,
//   Q[c, f] = sum_k W[k, c] phi_k(x_f),   phi = (x_d^2 | x_d),  W = (-1/2 / var | mu / var),   K = 2 D <= 80,
// is therefore evaluated in FIXED POINT on the integer matrix cores (an Ozaki-type splitting of both operands).  It is a
// fixed-point product with stated scales, not an exact one; what it keeps and what it drops:
//   * three families of binary scales bring the operands to |v| <= 0.498:
//       g_k  per GEMM row, from the model's own data scale in that dimension, s_d^2 = geometric mean over the components of
//            var + (mu - centre)^2: the quadratic row of dimension d is scaled by 2^-g ~ s_d^2, the linear one by ~ s_d, the
//            features by the inverse (x / s_d is O(1) for the frames the model describes, whatever the units of the data);
//       h_c  per COMPONENT, from its largest scaled coefficient (W[k, c] 2^-g_k 2^-h_c <= 0.498);
//       e_f  per FRAME, from its largest scaled feature (phi_k 2^g_k 2^-e_f <= 0.498);
//   * both operands are written in SEVEN balanced base-256 digits v = sum_i s_i 256^-(i+1), s_i in [-128, 127]: 56 bits
//     below the component's / the frame's own largest entry, what is dropped is <= 2^-57 of that;
//   * the digit planes are int8 matrices; the products of planes i and j with i + j = d all carry the weight 256^-(d+2) and
//     are summed by the matrix core in ONE int32 accumulator per d (|acc_d| <= 7 * 80 * 128^2 < 2^24: no overflow, no
//     rounding); planes with i + j > 6 are not formed: 28 plane products, K = 80 each.  What that drops: the six pairs of
//     diagonal 7 weigh 2^-72, their digit products reach 2^14 and there are 80 of each -- at most 2^-49 of the scale
//     2^(h_c + e_f), 2^-55 for digits of random sign; every later diagonal is 2^-8 of the one before;
//   * Q = 2^(h_c + e_f) sum_d acc_d 256^-(d+2): the diagonals are joined pairwise in int32 (acc_2n * 256 + acc_2n+1 < 2^31 for
//     2n <= 4), converted and summed in fp64, three fused multiply-adds.
// ERROR.  |dQ[c, f]| ~ 3 * 2^-53 * A_c X_f (measured, tools/i8_emulate.py and tests/test_gpu_i8.py; worst case ~ 2^-49 A_c X_f),
// A_c = the component's largest scaled coefficient, X_f = the frame's largest scaled feature: the error of a component is
// relative to ITS OWN largest possible term at that frame, as a floating-point GEMM's is -- a sharp component (small var)
// costs the broad ones nothing.  (Round 5 scaled row k by the largest |W[k, c]| over ALL components: one component with
// var / 10^3 then cost every other component ten bits, 1e-11 instead of 1e-14 on b.)  What the fixed point cannot give is
// range WITHIN a component: a component that is sharp in one dimension and broad in another keeps 56 bits below its
// sharpest coefficient only.  pack_i8_model therefore measures rho = max_c A_c X_c, X_c = the largest scaled feature of a
// frame within four standard deviations of the component's own mean (i8_comp_scale_kernel): 2^-53 rho is the size of the
// absolute error on a component's log-density at the frames that component describes.  Models with rho > kI8RhoMax
// (2048: <= 1e-12) are scored by the fp64 matrix-core bank (emit_diag_mfma, gmm.hip) instead -- like kappa > 4096 sends a
// model to the direct form.  lnasr_model_range reports rho and the decision.  On the metric's model (var in [0.5, 1.5],
// rho = 26) the integer bank is the more accurate of the two matrix-core routes against a long-double evaluation (9e-15
// against 4e-14 for the fp64 GEMM, which rounds after each of its 80 terms).
// The constant of the component (log w + gconst - 1/2 sum mu^2 / var) is added in fp64, then the M-way log-sum-exp of gmm.hip.
//
// Work per 16-component x 16-frame tile: 38 integer MFMAs (28 plane pairs x K = 64, plus the K = 64..79 remainders of up to
// four pairs of one diagonal packed into one instruction: 10) at 16.7 cycles against 20 fp64 MFMAs at 64.  The integer
// instruction does not hide vector work either (measured, tools/ubench/i8coexec, i8rate: beside a saturated integer matrix
// pipe an fp64 vector instruction costs 2.9 of its 4 cycles), so the kernel is bound by ISSUE: 76 matrix + ~245 vector
// instructions per wave and tile; what the design wins is the matrix share, 1270 cycles per wave-tile instead of 2560.
//
// Kernels: i8_row_scale_kernel / i8_comp_scale_kernel / pack_wslice_kernel (once per model), slice_features_kernel (once per
// batch: digit planes of the features, [tile][plane][k-chunk][frame][16 B], 568 B per frame), emit_diag_i8 (the bank).
// Instantiations of the bank: 16 mixtures per (virtual) state, 8 (M = 5 .. 8) or 32 (M = 32 as it is: no virtual states, no
// merge pass); with or without the K = 64 .. 79 remainder instructions (2 D <= 64 has none); b alone, b + bm (whole rows
// through an LDS image), or b + float q32 from the first four planes only (the float32 statistics pass's hand-off).
#include <cstdio>
#include <vector>

#include "common.h"
#include "devmath.h"

namespace lnasr {
namespace {

typedef int int4_t __attribute__((ext_vector_type(4)));

constexpr int kS = 7;                                 // digit planes per operand (balanced base-256 digits)
constexpr int kChunks = 5;                            // K = 80 in chunks of 16
constexpr int kPlaneI4 = kChunks * 16;                // int4 (16-byte) units per plane of a 16-frame tile: [chunk][frame]
constexpr int kTileI4 = kS * kPlaneI4 + 8;            // + 16 frame scales (doubles) = 568 int4 = 9,088 B
constexpr int kSlots = kS + 2;                        // stationary A quads per tile: the planes' K = 0..63, two packed remainders
constexpr int kZeroI4 = kTileI4;                      // a chunk of zeros behind the tile in the LDS image
constexpr int kLdsI4 = kTileI4 + 16;

// |v| <= 0.498 -> s[0..6], v = sum s_i 256^-(i+1) + r, |r| <= 2^-57; s_0 in [-127, 127], the others in [-128, 127]
__device__ __forceinline__ void split_digits(double v, int s[kS]) {
    const double a = v * 4294967296.0;                // 2^32
    const double hd = __builtin_rint(a);
    const double ld = __builtin_rint((a - hd) * 16777216.0);   // 2^24; a - hd is exact
    int hi = (int)hd, lo = (int)ld;
#pragma unroll
    for (int i = kS - 1; i >= 4; --i) {
        const int d = ((lo + 128) & 255) - 128;
        s[i] = d;
        lo = (lo - d) >> 8;
    }
    hi += lo;
#pragma unroll
    for (int i = 3; i >= 1; --i) {
        const int d = ((hi + 128) & 255) - 128;
        s[i] = d;
        hi = (hi - d) >> 8;
    }
    s[0] = hi;
}

// exponent e with m 2^-e <= 0.498 (the top digit stays within an int8), m > 0 finite
__device__ __forceinline__ int scale_exponent(double m) {
    int e = ilogb(m) + 2;
    if (ldexp(m, -e) > 0.498) ++e;
    return e;
}

__device__ __forceinline__ double coef_of(const double* __restrict__ mu, const double* __restrict__ ivar, size_t c, int k, int D) {
    if (k < D) return -0.5 * ivar[c * D + k];
    if (k < 2 * D) return mu[c * D + (k - D)] * ivar[c * D + (k - D)];
    return 0.0;
}

// Row scales from the model's own data scale.  Block d: l = mean over the real components (ivar > 0: padding components
// have none) of log2(var + mu_c^2) in dimension d; the quadratic row of d gets g = -round(l), the linear row -round(l / 2);
// pow2g[k] = 2^g[k].  Rows past 2 D (zero coefficients): g = 0.
__global__ void i8_row_scale_kernel(const double* __restrict__ mu, const double* __restrict__ ivar, int NM, int D,
                                    int* __restrict__ g, double* __restrict__ pow2g) {
    const int d = blockIdx.x;
    __shared__ double red[256];
    __shared__ int cnt[256];
    double sum = 0.0;
    int n = 0;
    for (int c = threadIdx.x; c < NM; c += 256) {
        const double iv = ivar[(size_t)c * D + d], m = mu[(size_t)c * D + d];
        if (iv > 0.0) {
            sum += log2(1.0 / iv + m * m);
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
        if (!(l > -900.0)) l = -900.0;                    // (also a NaN: the packers' results are then refused by rho)
        if (l > 900.0) l = 900.0;
        const int gq = -(int)__builtin_rint(l), gl = -(int)__builtin_rint(0.5 * l);
        g[d] = gq;
        pow2g[d] = ldexp(1.0, gq);
        g[D + d] = gl;
        pow2g[D + d] = ldexp(1.0, gl);
    }
    if (d == 0)
        for (int k = 2 * D + threadIdx.x; k < 80; k += 256) {
            g[k] = 0;
            pow2g[k] = 1.0;
        }
}

// Component scales and the model's range.  One wave per component c:
//   h[c] = exponent with |W[k, c]| 2^-g[k] 2^-h[c] <= 0.498 for every row k (0 for a padding component);
//   rho  = max_c A_c X_c (atomic max of the bit pattern; the values are >= 0), A_c = max_k |W[k, c]| 2^-g[k] and X_c = the
//          largest scaled feature of a frame within kI8Reach standard deviations of the component's mean:
//          max_d max((|mu_c| + 4 sd)^2 2^g[d], (|mu_c| + 4 sd) 2^g[D + d]).  A non-finite value counts as +infinity.
constexpr double kI8Reach = 4.0;
__global__ __launch_bounds__(256) void i8_comp_scale_kernel(const double* __restrict__ mu, const double* __restrict__ ivar,
                                                            const int* __restrict__ g, int NM, int D, int* __restrict__ h,
                                                            unsigned long long* __restrict__ rho_bits) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= NM) return;
    double A = 0.0, X = 0.0;
    bool real = false;
    for (int d = lane; d < D; d += 64) {
        const double iv = ivar[(size_t)c * D + d], m = mu[(size_t)c * D + d];
        if (!(iv > 0.0)) continue;
        real = true;
        A = fmax(A, fmax(ldexp(0.5 * iv, -g[d]), ldexp(fabs(m) * iv, -g[D + d])));
        const double reach = fabs(m) + kI8Reach / sqrt(iv);
        X = fmax(X, fmax(ldexp(reach * reach, g[d]), ldexp(reach, g[D + d])));
    }
    for (int o = 32; o > 0; o >>= 1) {
        A = fmax(A, __shfl_xor(A, o));
        X = fmax(X, __shfl_xor(X, o));
        real |= __shfl_xor((int)real, o) != 0;
    }
    if (lane == 0) {
        h[c] = (real && A > 0.0 && __builtin_isfinite(A)) ? scale_exponent(A) : 0;
        double r = real ? A * X : 0.0;
        if (!__builtin_isfinite(r)) r = __builtin_huge_val();
        atomicMax(rho_bits, (unsigned long long)__double_as_longlong(r));
    }
}

// Digit planes of the coefficients as stationary MFMA A operands, per state pair sp, tile q (0, 1):
//   wq[(((sp*2 + q)*9 + slot)*64 + lane)]  (16 bytes)   row r = lane & 15 <-> state 2 sp + ((r >> 2) & 1), mixture 8 (r >> 3) + 4 q + (r & 3)
//   slot j < 7 : plane j, k = 16 (lane >> 4) + byte                   (the K = 0..63 step of a plane pair)
//   slot 7 + p : plane (lane >> 4) + 4 p, k = 64 + byte               (the packed remainder steps; plane 7 does not exist: zeros)
// constc[state*16 + m] = log w + gconst - 1/2 sum mu^2 / var;  constc_nw: the same without log w
__global__ void pack_wslice_kernel(const double* __restrict__ mu, const double* __restrict__ ivar,
                                   const double* __restrict__ gconst, const double* __restrict__ logw,
                                   const int* __restrict__ g, const int* __restrict__ hexp, int4_t* __restrict__ wq,
                                   double* __restrict__ constc, double* __restrict__ constc_nw, int N, int D, int MX) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(MX == 16 ? N / 2 : (MX == 8 ? N / 4 : N)) * 2 * kSlots * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t r1 = idx >> 6;
    const int slot = (int)(r1 % kSlots);
    r1 /= kSlots;
    const int q = (int)(r1 & 1);
    const int sp = (int)(r1 >> 1);
    const int r = lane & 15, h = lane >> 4;
    // 8 mixtures per state (MX = 8): a wave's two tiles are 4 states x 8 mixtures, row r <-> state 4 sp + (r >> 2), mixture
    // 4 q + (r & 3) -- a lane then holds all eight mixtures of its state
    // 32 mixtures per state (MX = 32): the two tiles are ONE state, row r <-> mixture 8 (r >> 2) + 4 q + (r & 3)
    const int state = MX == 16 ? 2 * sp + ((r >> 2) & 1) : (MX == 8 ? 4 * sp + (r >> 2) : sp);
    const int mix = MX == 16 ? 8 * (r >> 3) + 4 * q + (r & 3) : (MX == 8 ? 4 * q + (r & 3) : 8 * (r >> 2) + 4 * q + (r & 3));
    const size_t c = (size_t)state * MX + mix;
    const int plane = slot < kS ? slot : h + 4 * (slot - kS);
    const int k0 = slot < kS ? 16 * h : 64;
    unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int by = 0; by < 16; ++by) {
        const int k = k0 + by;
        int s[kS];
        split_digits(ldexp(coef_of(mu, ivar, c, k, D), -g[k] - hexp[c]), s);
        int dg = 0;
#pragma unroll
        for (int i = 0; i < kS; ++i) dg = plane == i ? s[i] : dg;      // (plane 7 of the last remainder slot: zero)
        w[by >> 2] |= ((unsigned)dg & 255u) << (8 * (by & 3));
    }
    wq[idx] = int4_t{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    if (slot == 0 && h == 0) {
        double sq = 0.0;
        for (int d = 0; d < D; ++d) {
            const double m = mu[c * D + d];
            sq += m * m * ivar[c * D + d];
        }
        constc[c] = logw[c] + gconst[c] - 0.5 * sq;
        constc_nw[c] = gconst[c] - 0.5 * sq;             // without the weight: what bm holds (gmmhmm.py:57-60)
    }
}

// Digit planes of the features.  Block = 320 threads x 64 frames = 4 tiles: thread t owns (tile t / 80, k-chunk (t % 80) / 16,
// frame t % 16) -- the 80 threads of a tile write 1280 contiguous bytes of every plane (a wave: runs of 256 B .. 1 KB).  The row
// scales 2^g_k wait in the LDS: read per lane from global memory they were a third of the kernel (0.26 -> 0.20 ms per 1,024,000
// frames, round 6; the store pattern -- one wave per k-chunk, four pieces of 256 B 9 KB apart per instruction, against this one --
// measured the same: 0.2017 / 0.2001 ms).
//   Fs[tile*kTileI4 + (plane*5 + chunk)*16 + frame]  (16 bytes: k = 16 chunk + byte),   then 16 doubles: 2^(e_f - 24)
// (NaN for a frame with a NaN / infinite feature: every density of that frame comes out NaN, as the reference's does).
__global__ __launch_bounds__(320) void slice_features_kernel(const double* __restrict__ obs, const double* __restrict__ centre,
                                                             const double* __restrict__ pow2g, int4_t* __restrict__ Fs,
                                                             long F, int D, unsigned* __restrict__ nan_flag,
                                                             unsigned nan_epoch) {
    const int tid = threadIdx.x;
    const int ch = (tid % 80) >> 4, fl = (tid / 80) * 16 + (tid & 15);
    const long frame0 = (long)blockIdx.x * 64;
    const long frame = frame0 + fl;
    const bool live = frame < F;
    // the block's 64 frames are one contiguous run of obs: coalesced into the LDS, centred on the way
    __shared__ double xs[64 * 40];
    __shared__ double smax[kChunks][64];
    __shared__ int sbad[kChunks][64];
    __shared__ double p2g[80];
    if (tid < 80) p2g[tid] = pow2g[tid];
    {
        const long base = frame0 * D, lim = F * (long)D;
        for (int e = tid; e < 64 * D; e += 320) {
            const long ge = base + e;
            xs[e] = ge < lim ? obs[ge] - centre[e % D] : 0.0;
        }
    }
    __syncthreads();
    const double* xf = xs + fl * D;
    double val[16];
    double mx = 0.0;
    bool bad = false;
#pragma unroll
    for (int by = 0; by < 16; ++by) {
        const int k = 16 * ch + by;
        double v = 0.0;
        if (k < 2 * D) {
            const double x = xf[k < D ? k : k - D];
            bad |= !__builtin_isfinite(x);
            v = (k < D ? x * x : x) * p2g[k];
        }
        val[by] = live ? v : 0.0;
        mx = fmax(mx, fabs(val[by]));
    }
    smax[ch][fl] = mx;
    sbad[ch][fl] = (bad && live) ? 1 : 0;
    __syncthreads();
    int badi = 0;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        mx = fmax(mx, smax[c][fl]);
        badi |= sbad[c][fl];
    }
    if (!__builtin_isfinite(mx)) badi = 1;            // x^2 overflowed
    const int ef = (mx > 0.0 && !badi) ? scale_exponent(mx) : 0;
    const long tile = frame >> 4;
    const int f16 = (int)(frame & 15);
    int4_t* dst = Fs + tile * kTileI4;
    if (tile <= ((F - 1) >> 4)) {                         // frames past F inside the last tile: zero planes
        // all seven digits of a value at once: t = v 2^56 as an integer; adding 0x80 at every byte position carries exactly
        // as the digit-by-digit extraction does, and (byte ^ 0x80) is the balanced digit.  Byte 6 = plane 0 ... byte 0 = plane 6.
        unsigned lo32[16], hi32[16];
        const double down = badi ? 0.0 : ldexp(1.0, 32 - ef);
#pragma unroll
        for (int by = 0; by < 16; ++by) {
            const double a = val[by] * down;              // v 2^32, |v| <= 0.498
            const double hd = __builtin_rint(a);
            const double ld = __builtin_rint((a - hd) * 16777216.0);
            const long long t = ((long long)(int)hd << 24) + (long long)(int)ld;
            const unsigned long long u = (unsigned long long)(t + 0x0080808080808080LL) ^ 0x0080808080808080ULL;
            lo32[by] = (unsigned)u;
            hi32[by] = (unsigned)(u >> 32);
        }
#pragma unroll
        for (int i = 0; i < kS; ++i) {
            const int bi = 6 - i;                         // byte of u that holds plane i
            const unsigned sel = bi < 4 ? (unsigned)(bi | ((4 + bi) << 8) | 0x0c0c0000u)
                                        : (unsigned)((bi - 4) | ((bi) << 8) | 0x0c0c0000u);
            unsigned w[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned* src = bi < 4 ? lo32 : hi32;
                const unsigned t01 = __builtin_amdgcn_perm(src[4 * m + 1], src[4 * m], sel);
                const unsigned t23 = __builtin_amdgcn_perm(src[4 * m + 3], src[4 * m + 2], sel);
                w[m] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
            }
            dst[(i * kChunks + ch) * 16 + f16] = int4_t{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        }
        if (ch == 0) {
            double* sc = reinterpret_cast<double*>(dst + kS * kPlaneI4);
            sc[f16] = badi ? __builtin_nan("") : ldexp(1.0, ef - 24);
        }
    }
    if (badi && live && ch == 0) *nan_flag = nan_epoch;
}

// The bank.  Block = 8 waves, two per SIMD; wave w owns the state pair sp = 8 y + w of slab y (16 states per block): two
// 16-component tiles, W planes in 72 registers.  Row r of tile q <-> state 2 sp + ((r >> 2) & 1), mixture 8 (r >> 3) + 4 q + (r & 3):
// lane group g = lane >> 4 then holds, for frame lane & 15, the mixtures 8 (g >> 1) + {0..7} of state 2 sp + (g & 1) -- the M-way
// log-sum-exp is eight values in the lane and one exchange with lane ^ 32 (v_permlane32_swap).  The block walks a range of
// 16-frame tiles whose digit planes it copies into the LDS (double buffered); a wave reads the seven K = 0..63 fragments of a
// tile ONCE (with the ten remainder fragments: 17 x 16 bytes per lane) and keeps them for all 28 plane pairs.  Diagonals go in
// ascending order (d = 0 needs plane 0 only: its matrix instructions start behind the first LDS read), in the groups (0, 1),
// (2, 3), (4, 5), 6; a group's two accumulators are joined in an int32 (|acc_d| 256 + |acc_d+1| < 2^31 for d <= 4) and folded
// into the running fp64 value while the next group is in the matrix core.
// REM = false (2 D <= 64): no remainder instructions, 28 per tile.
// Two phases per tile, the two waves of a SIMD in OPPOSITE ones (waves 0-3: matrix phase of tile t, then its log-sum-exp;
// waves 4-7: log-sum-exp of tile t - 1, then the matrix phase of tile t): the vector work of one wave issues beside the matrix
// instructions of the other.  ABL = 6 is the diagnostic build's instantiation with s_memtime stamps around the phases.
// WRITE_BM: also bm [F, N, 16], the component log-densities WITHOUT the weights (constc is then the weightless constant; the
// weights of the block's 16 states wait in the LDS and are added on the way into the log-sum-exp).  The block's share of a
// frame's bm row is 2 KB contiguous (16 states x 16 mixtures): the waves park their values in an LDS image of the tile
// ([frame][state][mixture], double buffered) and the log-sum-exp waves of the NEXT phase write it out as whole rows, 16
// bytes per lane -- tools/ubench/wpattern: 6.1 TB/s, against 3.8 for the 32-byte pieces a lane could store directly.
constexpr int kBmRow = 16 * 16 + 4;                   // doubles per frame of the image (padded: two-way bank conflicts at most)
template <int ABL, bool REM = true, bool WRITE_BM = false, int MIX = 16, int NPL = kS, bool Q32 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void emit_diag_i8(
    const int4_t* __restrict__ Fs, const int4_t* __restrict__ wq, const double* __restrict__ constc,
    const int* __restrict__ hexp, const double* __restrict__ logw, double* __restrict__ b, double* __restrict__ bm, long F,
    int N, int GY, long ntiles,
    int tiles_per_block, long nchunks, unsigned long long* __restrict__ dbg, float* __restrict__ q32 = nullptr) {
    // Q32 (the float32 statistics pass, f32stats.hip): only the first NPL = 4 digit planes of either operand are multiplied (ten
    // plane pairs: 32 bits below the row's / column's scale, ~1e-6 absolute on a log-density where that pass promises 8e-6), the
    // weighted component log-densities go out as float q32 [F, N, MIX] and b is the float log-sum-exp of those rounded values,
    // exactly the arithmetic emit_diag_mfma_f32 has after ITS matrix product.
    static_assert(!Q32 || (!WRITE_BM && MIX != 8 && NPL == 4), "float32 hand-off: b + q32, 16 / 32 mixtures, four planes");
    static_assert(Q32 || NPL == kS, "the fp64 results need every plane");
    __shared__ int4_t Fl[2][kLdsI4];
    // MIX mixtures per state: 16 (a wave = 2 states, a state's mixtures in lanes l and l ^ 32), 8 (4 states, all eight in the
    // lane), 32 (1 state, its mixtures in the four lanes l, l ^ 16, l ^ 32, l ^ 48: M = 32 without a merge pass)
    constexpr bool M8 = MIX == 8;
    static_assert(MIX == 16 || MIX == 8 || MIX == 32, "mixtures per state");
    static_assert(!(M8 && WRITE_BM), "the 8-mixture instantiation writes b only");
    constexpr int SW = MIX == 16 ? 2 : (MIX == 8 ? 4 : 1);  // states per wave
    constexpr int SPB = 8 * SW, OST = SPB + 1;            // states per block, padded row of the staged outputs
    __shared__ double ob[2][16 * 33];
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, ts = 0;
    const unsigned long long k0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define LN_STAMP(i)                                                \
    if (ABL == 6) {                                                \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        ph[i] += now - ts;                                         \
        ts = now;                                                  \
    }
    __shared__ double etab[kExpTabBig];
    __shared__ double ltab[2 * kLogTab];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    exp_table_big_init(etab, tid, 512);
    log_table_init(ltab, tid, 512);
    if (tid < 32) Fl[tid >> 4][kZeroI4 + (tid & 15)] = int4_t{0, 0, 0, 0};
    __shared__ double lws[WRITE_BM ? 256 : 1];
    __shared__ double csts[WRITE_BM ? 256 : 1];
    __shared__ int hs[WRITE_BM ? 256 : 1];
    __shared__ __attribute__((aligned(32))) double bms[WRITE_BM ? 2 * 16 * kBmRow : 1];   // written / read as 32- and 16-byte vectors
    constexpr int kQRow = 16 * 16 + 8;                    // floats per frame of the q32 image (the block's 1 KB of a frame + pad)
    __shared__ __attribute__((aligned(16))) float qms[Q32 ? 2 * 16 * kQRow : 1];          // ... as 16-byte vectors
    const long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long bi = bid >> 3;
    const int y = (int)(bi % GY);
    const long chunk = (bi / GY) * 8 + xcd;
    if (chunk >= nchunks) return;
    const long tile0 = chunk * tiles_per_block;
    long tile1 = tile0 + tiles_per_block;
    if (tile1 > ntiles) tile1 = ntiles;
    // (N a multiple of 2, not of 16: the waves of the last slab that have no state pair score pair 0 again and stage values
    // nobody flushes)
    const int sp = SW * (8 * y + w) < N ? 8 * y + w : 0;               // the wave's unit: state pair / four states / state
    const int g = lane >> 4, col = lane & 15;
    const int lst = MIX == 16 ? (g & 1) : (MIX == 8 ? g : 0);          // the lane's state within the wave's unit
    const int mb = MIX == 16 ? 8 * (g >> 1) : (MIX == 8 ? 0 : 8 * g);  // its first mixture (eight consecutive ones)
    const int sl = SW * w + lst;                                       // its state within the block's slab
    if (WRITE_BM && tid < 256) {                          // [state of the slab][mixture]; states past N: never used
        const long st = (long)SPB * y + tid / MIX;
        lws[tid] = logw[(st < N ? st : N - 1) * MIX + tid % MIX];
    }

    int4_t Wm[2][kS], Wl[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int j = 0; j < kS; ++j) Wm[q][j] = wq[(((size_t)sp * 2 + q) * kSlots + j) * 64 + lane];
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (REM) Wl[q][p] = wq[(((size_t)sp * 2 + q) * kSlots + kS + p) * 64 + lane];
    }
    // the components' constants and their exponents h_c (the planes hold W 2^-g 2^-h: the product is put back by 2^(e_f + h_c))
    // (WRITE_BM: both wait in the LDS beside the weights -- 24 registers the instantiation does not have)
    double cst[WRITE_BM ? 1 : 8];
    int hx[WRITE_BM ? 1 : 8];
    const double* cstp = cst;
    const int* hxp = hx;
    if (WRITE_BM) {
        if (tid < 256) {
            const long st = (long)SPB * y + tid / MIX;
            csts[tid] = constc[(st < N ? st : N - 1) * MIX + tid % MIX];
            hs[tid] = hexp[(st < N ? st : N - 1) * MIX + tid % MIX];
        }
        cstp = csts + sl * MIX + mb;
        hxp = hs + sl * MIX + mb;
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            cst[m] = constc[(size_t)(SW * sp + lst) * MIX + mb + m];
            hx[m] = hexp[(size_t)(SW * sp + lst) * MIX + mb + m];
        }
    }

    // (NPL < kS: only the first NPL planes and the scales behind plane kS - 1 are copied; they keep their places in the image)
    constexpr int kUsedI4 = NPL * kPlaneI4 + 8;
    auto tile_index = [](int e) { return e < NPL * kPlaneI4 ? e : e + (kS - NPL) * kPlaneI4; };
#ifndef LNASR_I8_NO_GLDS
    // The next tile's planes go from global memory straight into the LDS image (global_load_lds_dwordx4: 1 KB per wave and
    // instruction, LDS address = a wave-uniform base + 16 lane -- the image is lane-linear, NPL * kPlaneI4 is a multiple of
    // 64), no staging registers and no ds_write pass.  Issued at the top of a tile into the buffer the barrier that ended the
    // previous tile has freed; the barrier in the middle of the tile waits for it (a pending LDS-DMA is a pending LDS write
    // on the VM counter: __syncthreads drains it), a phase of ~1500 cycles later.
    auto load_tile = [&](long tile, int buf) {
        const int4_t* src = Fs + tile * kTileI4;
#pragma unroll
        for (int r = 0; r < (kUsedI4 + 511) / 512; ++r) {
            const int e0 = 64 * w + 512 * r, e = e0 + lane;        // e0: wave-uniform
            if (e < kUsedI4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tile_index(e)),
                                                 (__attribute__((address_space(3))) void*)(&Fl[buf][tile_index(e0)]), 16, 0, 0);
        }
    };
    auto store_tile = [](int) {};
#else
    int4_t pre[2];
    auto load_tile = [&](long tile, int) {
        const int4_t* src = Fs + tile * kTileI4;
#pragma unroll
        for (int r = 0; r < (kUsedI4 + 511) / 512; ++r) {
            const int e = tid + 512 * r;
            pre[r] = src[tile_index(e < kUsedI4 ? e : kUsedI4 - 1)];
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < (kUsedI4 + 511) / 512; ++r) {
            const int e = tid + 512 * r;
            if (e < kUsedI4) Fl[buf][tile_index(e)] = pre[r];
        }
    };
#endif
    auto flush = [&](int buf, long tile) {               // 16 frames x SPB states (16: waves 0-3; 8 mixtures: every thread)
        if (tid < 16 * SPB) {
            const int f = tid / SPB, st = tid % SPB;
            const long frame = tile * 16 + f;
            if (frame < F && SPB * y + st < N) b[frame * N + SPB * y + st] = ob[buf][f * OST + st];
        }
    };

    load_tile(tile0, 0);
    store_tile(0);
    drain_vmem();
    __syncthreads();

    const int hb = g * 16 + col;
    // max / sum of the two halves of a state's mixtures (lanes l and l ^ 32), left in both lanes
    auto pair_max = [](double x) {
        auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        return vmax_f64(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
    };
    auto pair_sum = [](double x) {
        auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    };
    // (32 mixtures: the halves of lanes l and l ^ 16 as well)
    auto quad_max = [](double x) {
        auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        return vmax_f64(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
    };
    auto quad_sum = [](double x) {
        auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    };
    // log-sum-exp of the tile whose values `v` carries (this lane: frame `col`, state 2 sp + (g & 1), mixtures
    // 8 (g >> 1) + 4 q + e; the other eight are in lane ^ 32) -> the staging row of that tile
    double v[2][4];
    float vf[2][4];                                       // Q32: the rounded values, carried instead of v
    auto lse_to = [&](double* obuf) {
        if (Q32) {
            auto swap32 = [](float x) {
                auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
            };
            auto add32 = [](float x) {
                auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                return __uint_as_float(t[0]) + __uint_as_float(t[1]);
            };
            auto swap16 = [](float x) {
                auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
            };
            auto add16 = [](float x) {
                auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                return __uint_as_float(t[0]) + __uint_as_float(t[1]);
            };
            float mxf = vf[0][0];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q + e > 0) mxf = fmaxf(mxf, vf[q][e]);
            mxf = swap32(mxf);
            if (MIX == 32) mxf = swap16(mxf);
            float sf = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) sf += __expf(vf[q][e] - mxf);
            sf = add32(sf);
            if (MIX == 32) sf = add16(sf);
            // mx = -inf: the differences are NaN, s is NaN; the state's value is -inf (as in emit_diag_mfma_f32)
            const double resf = (mxf == -__builtin_huge_valf()) ? (double)mxf : (double)mxf + (double)__logf(sf);
            if ((MIX == 16 && g < 2) || (MIX == 32 && g == 0)) obuf[col * OST + sl] = resf;
            return;
        }
        if (WRITE_BM) {
            const double* lw = lws + sl * MIX + mb;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q][e] += lw[4 * q + e];
        }
        double mx = v[0][0];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q + e > 0) mx = vmax_f64(mx, v[q][e]);
        if (!M8) mx = pair_max(mx);
        if (MIX == 32) mx = quad_max(mx);
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += exp_nonpos_big2(v[q][e] - mx, etab);
        if (!M8) s = pair_sum(s);
        if (MIX == 32) s = quad_sum(s);
        const double res = (mx == LN_NEG_INF) ? mx : mx + log_sum(s, ltab);
        if (M8 || (MIX == 16 && g < 2) || (MIX == 32 && g == 0)) obuf[col * OST + sl] = res;
    };

    // the matrix phase of one tile: planes from the LDS to registers, 76 matrix instructions with the folds between them -> v
    auto matrix_phase = [&](int buf, long tile) {
        const int4_t* Fc = &Fl[buf][0];
        const double scale = reinterpret_cast<const double*>(Fc + kS * kPlaneI4)[col];
        // (reads in the order of use: diagonal d needs planes 0..d, so the matrix instructions start behind the first read)
        int4_t fm[NPL], fl[NPL][2];
#pragma unroll
        for (int d = 0; d < NPL; ++d) {
            fm[d] = Fc[d * kPlaneI4 + hb];
#pragma unroll
            for (int p = 0; p < 2; ++p) {                 // K = 64..79: lane group g carries planes (d - g - 4p, g + 4p)
                if (4 * p > d || !REM || (WRITE_BM && d >= 4)) continue;   // (WRITE_BM: 24 registers short -- the last six
                const int i = d - g - 4 * p;                               // remainder fragments are read next to their use)
                fl[d][p] = Fc[i >= 0 ? i * kPlaneI4 + 64 + col : kZeroI4 + col];
            }
        }
        // (The fold in 64-bit INTEGER arithmetic -- T = p0 2^32 + p1 2^16 + p2 + round(acc_6 / 256), two conversions and one
        // fused multiply-add: 4 fp64 + 11 integer instructions per value instead of 8 + 3 -- was built on the strength of
        // tools/ubench/i8coexec showing shift-adds free beside the integer matrix pipe: 4.94 ms against 4.73 on one box, three
        // alternations.  In this instruction mix they are not free.)
        double r[2][4];                                   // running fp64 value of the diagonals folded so far
#pragma unroll
        for (int n = 0; n < (NPL + 1) / 2; ++n) {
            int4_t acc[2][2];
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                const int d = 2 * n + dd;
                if (d >= NPL) continue;
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[dd][q] = int4_t{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i <= d; ++i)              // K = 0..63 of planes (i, d - i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[dd][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wm[q][d - i], fm[i], acc[dd][q], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (4 * p > d || !REM) continue;
                    if (WRITE_BM && d >= 4) {
                        const int i = d - g - 4 * p;
                        fl[d][p] = Fc[i >= 0 ? i * kPlaneI4 + 64 + col : kZeroI4 + col];
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[dd][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wl[q][p], fl[d][p], acc[dd][q], 0, 0, 0);
                }
            }
            // unit: the weight of diagonal 1; the pair (2n, 2n+1) sits 2^-16n below it, diagonal 6 2^-40
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double pn = 2 * n + 1 >= NPL ? (double)acc[0][q][e] : (double)(acc[0][q][e] * 256 + acc[1][q][e]);
                    r[q][e] = n == 0 ? pn : fma(pn, n == 1 ? 0x1p-16 : (n == 2 ? 0x1p-32 : 0x1p-40), r[q][e]);
                }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = fma(r[q][e], ldexp(scale, hxp[4 * q + e]), cstp[4 * q + e]);
        if (Q32) {                                        // eight consecutive mixtures of (frame col, the lane's state): 32 bytes
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[q][e] = (float)v[q][e];
            typedef float float4v __attribute__((ext_vector_type(4)));   // parked like bm: whole rows go out in the next phase
            float4v* dst = reinterpret_cast<float4v*>(qms + (size_t)buf * 16 * kQRow + col * kQRow + sl * MIX + mb);
            dst[0] = float4v{vf[0][0], vf[0][1], vf[0][2], vf[0][3]};
            dst[1] = float4v{vf[1][0], vf[1][1], vf[1][2], vf[1][3]};
        }
        if (WRITE_BM) {                                   // this lane: eight consecutive mixtures of (frame col, its state)
            typedef double double4v __attribute__((ext_vector_type(4)));
            double4v* dst = reinterpret_cast<double4v*>(bms + (size_t)buf * 16 * kBmRow + col * kBmRow + sl * MIX + mb);
            dst[0] = double4v{v[0][0], v[0][1], v[0][2], v[0][3]};
            dst[1] = double4v{v[1][0], v[1][1], v[1][2], v[1][3]};
        }
    };
    // the parked tile -> bm: 256 threads, 16 bytes each, eight times: whole 2 KB rows
    auto flush_bm = [&](int buf, long tile, int t256) {
        typedef double double2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = k * 256 + t256;
            const int f = idx >> 7, off = idx & 127;      // frame of the tile, 16-byte piece of its 2 KB
            const long frame = tile * 16 + f;
            const int st = SPB * y + off * 2 / MIX;
            if (frame < F && st < N)
                *reinterpret_cast<double2v*>(bm + ((size_t)frame * N + SPB * y) * MIX + off * 2) =
                    *reinterpret_cast<const double2v*>(bms + (size_t)buf * 16 * kBmRow + f * kBmRow + off * 2);
        }
    };

    // the parked q32 tile -> global: 256 threads, 16 bytes each, four times: whole 1 KB rows
    auto flush_q = [&](int buf, long tile, int t256) {
        typedef float float4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = k * 256 + t256;
            const int f = idx >> 6, off = idx & 63;       // frame of the tile, 16-byte piece of its 1 KB
            const long frame = tile * 16 + f;
            const int st = SPB * y + off * 4 / MIX;
            if (frame < F && st < N)
                *reinterpret_cast<float4v*>(q32 + ((size_t)frame * N + SPB * y) * MIX + off * 4) =
                    *reinterpret_cast<const float4v*>(qms + (size_t)buf * 16 * kQRow + f * kQRow + off * 4);
        }
    };

    // Two phases per tile, the two waves of a SIMD in OPPOSITE ones: waves 0-3 run the matrix phase of tile t while waves 4-7
    // run the log-sum-exp of tile t - 1, then the roles swap -- the vector work of one wave issues beside the matrix
    // instructions of the other instead of both waves wanting the same unit at the same time.
    const int half = w >> 2;
    for (long tile = tile0; tile < tile1; ++tile) {
        const int cur = (int)((tile - tile0) & 1);
        const bool has_next = tile + 1 < tile1;
        if (ABL == 6) ts = __builtin_amdgcn_s_memtime();
        if (has_next) load_tile(tile + 1, cur ^ 1);
        if (half == 0) matrix_phase(cur, tile);
        else if (tile > tile0) {
            lse_to(&ob[cur ^ 1][0]);
            if (WRITE_BM) flush_bm(cur ^ 1, tile - 1, tid - 256);      // complete since the last barrier
            if (Q32) flush_q(cur ^ 1, tile - 1, tid - 256);
        }
        LN_STAMP(0);
        __syncthreads();
        LN_STAMP(1);
        if (half == 0) {
            lse_to(&ob[cur][0]);
            if (!M8 && tile > tile0) flush(cur ^ 1, tile - 1);   // both halves have staged tile t - 1 by now
        } else {
            matrix_phase(cur, tile);
        }
        if (M8 && tile > tile0) flush(cur ^ 1, tile - 1);  // (32 states: every thread has a value to write)
        if (has_next) store_tile(cur ^ 1);
        LN_STAMP(3);
        __syncthreads();
        LN_STAMP(4);
    }
    if (tile1 > tile0) {
        const int last = (int)((tile1 - 1 - tile0) & 1);
        if (half == 1) lse_to(&ob[last][0]);
        __syncthreads();
        flush(last, tile1 - 1);
        if (WRITE_BM && half == 1) flush_bm(last, tile1 - 1, tid - 256);
        if (Q32 && half == 1) flush_q(last, tile1 - 1, tid - 256);
    }
    if (ABL == 6 && lane == 0) {
        for (int i = 0; i < 5; ++i) dbg[((size_t)blockIdx.x * 8 + w) * 7 + i] = ph[i];
        dbg[((size_t)blockIdx.x * 8 + w) * 7 + 5] = __builtin_amdgcn_s_memtime() - k0;
        dbg[((size_t)blockIdx.x * 8 + w) * 7 + 6] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#undef LN_STAMP
}

}  // namespace

bool i8_emit_shape_ok(int Nv, int Mv, int D) {
    // a wave scores 32 components: two 16-mixture states, four 8-mixture states or one 32-mixture state
    const bool units = (Mv == 16 && Nv % 2 == 0) || (Mv == 8 && Nv % 4 == 0) || Mv == 32;
    return units && D >= 1 && 2 * D <= 80;
}

// The integer bank scores this model's batches of F frames: planes packed for the shape the fp64 bank was packed for, the
// model's range within what seven planes hold (rho, above), a few thousand frames to amortise 36 KB of digit planes per
// wave and four slabs of 16 states or more (128 states x 1000 frames: 0.018 against 0.016 ms for one utterance, 0.033 /
// 0.036 for four, 0.36 / 0.41 for 64; 512,000 frames of 32 states: 0.87 / 0.83, of 48: 1.17 / 1.57, of 64: 1.46 / 1.59)
bool i8_model_ok(const Model& md) {
    return md.i8_NG > 0 && md.i8_NG == md.wpack_NG && !sw().no_i8emit && md.i8_rho <= sw().i8_rho_max;
}
bool i8_route(const Model& md, long F) {
    return i8_model_ok(md) && (sw().i8_min_frames >= 0 ? F >= sw().i8_min_frames : (F >= 4096 && md.i8_NG >= 12));
}

// once per model (after prep_diag / centre_and_kappa): scales, the model's range, digit planes of the coefficients
int pack_i8_model(lnasr_ctx* ctx, const double* mu_s, const double* iv_s, const double* gc_s, const double* lw_s, int Nv,
                  int Mv) {
    ctx->model.i8_logw = lw_s;
    ctx->model.i8_mix = Mv;
    Model& md = ctx->model;
    const int D = md.D;
    const size_t NMv = (size_t)Nv * Mv;
    LN_TRY(md.i8_g.reserve(80 * 4));
    LN_TRY(md.i8_pow2g.reserve(80 * 8));
    const size_t units = Mv == 16 ? Nv / 2 : (Mv == 8 ? Nv / 4 : Nv);   // waves' worth of components: 32 each
    LN_TRY(md.i8_wq.reserve(units * 2 * kSlots * 64 * 16));
    LN_TRY(md.i8_const.reserve(NMv * 8));
    LN_TRY(md.i8_const_nw.reserve(NMv * 8));
    LN_TRY(md.i8_h.reserve(NMv * 4));
    LN_TRY(md.flags.reserve(256));
    unsigned long long* rbits = reinterpret_cast<unsigned long long*>(md.flags.as<char>() + 16);
    LN_HIP(hipMemsetAsync(rbits, 0, 8, ctx->stream));
    LN_LAUNCH(ctx, "i8_row_scale", i8_row_scale_kernel, dim3(D), dim3(256), 0, mu_s, iv_s, (int)NMv, D, md.i8_g.as<int>(),
              md.i8_pow2g.as<double>());
    LN_LAUNCH(ctx, "i8_comp_scale", i8_comp_scale_kernel, dim3((unsigned)((NMv + 3) / 4)), dim3(256), 0, mu_s, iv_s,
              (const int*)md.i8_g.as<int>(), (int)NMv, D, md.i8_h.as<int>(), rbits);
    const size_t total = units * 2 * kSlots * 64;
    LN_LAUNCH(ctx, "pack_wslice", pack_wslice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mu_s, iv_s, gc_s,
              lw_s, (const int*)md.i8_g.as<int>(), (const int*)md.i8_h.as<int>(), md.i8_wq.as<int4_t>(),
              md.i8_const.as<double>(), md.i8_const_nw.as<double>(), Nv, D, Mv);
    unsigned long long rb = 0;
    LN_HIP(hipMemcpyAsync(&rb, rbits, 8, hipMemcpyDeviceToHost, ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream));
    md.i8_rho = __builtin_bit_cast(double, rb);
    md.i8_NG = (Mv == 32 ? 2 * Nv : Nv) / 4;             // in units of four 16-mixture (virtual) states, like wpack_NG
    return 0;
}

// b [F, Nk] (and bm [F, Nk, mix] unless null); Nk = the (virtual, padded) states the planes were packed for -- with 32
// mixtures per state (i8_mix = 32) the model's real states: b and bm then need no merge / no virtual layout
int emit_diag_i8_device(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm, float* q32) {
    Model& md = ctx->model;
    const int mix = md.i8_mix, D = md.D;
    const int Nk = mix == 32 ? md.i8_NG * 2 : md.i8_NG * 4;
    const long ntiles = (F + 15) / 16;
    LN_TRY(ctx->ws_fs.reserve((size_t)ntiles * kTileI4 * 16));
    int4_t* Fs = ctx->ws_fs.as<int4_t>();
    LN_LAUNCH(ctx, "slice_features", slice_features_kernel, dim3((unsigned)((ntiles * 16 + 63) / 64)), dim3(320), 0, obs,
              (const double*)md.centre.as<double>(), (const double*)md.i8_pow2g.as<double>(), Fs, F, D,
              ctx->nan_flag.as<unsigned>(), ctx->nan_epoch);
    const int spb = mix == 16 ? 16 : (mix == 8 ? 32 : 8);            // states per block
    const int GY = (Nk + spb - 1) / spb;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    long want_chunks = (long)cus / GY;                   // one 8-wave block per CU: two waves per SIMD
    if (want_chunks < 8) want_chunks = 8;
    want_chunks = (want_chunks / 8) * 8;
    long tpb = (ntiles + want_chunks - 1) / want_chunks;
    if (tpb < 1) tpb = 1;
    const long nchunks = (ntiles + tpb - 1) / tpb;
    const long grid = ((nchunks + 7) / 8) * 8 * GY;
#define LN_I8K(abl, rem, wbm, mixv, name)                                                                         \
    LN_LAUNCH(ctx, name, (emit_diag_i8<abl, rem, wbm, mixv, kS, false>), dim3((unsigned)grid), dim3(512), 0,         \
              (const int4_t*)Fs, (const int4_t*)md.i8_wq.as<int4_t>(),                                               \
              (const double*)(wbm ? md.i8_const_nw.as<double>() : md.i8_const.as<double>()),                         \
              (const int*)md.i8_h.as<int>(), md.i8_logw, b, bm, F, Nk, GY, ntiles, (int)tpb, nchunks, dbg, (float*)nullptr)
#define LN_I8Q(rem, mixv, name)                                                                                   \
    LN_LAUNCH(ctx, name, (emit_diag_i8<0, rem, false, mixv, 4, true>), dim3((unsigned)grid), dim3(512), 0,           \
              (const int4_t*)Fs, (const int4_t*)md.i8_wq.as<int4_t>(), (const double*)md.i8_const.as<double>(),      \
              (const int*)md.i8_h.as<int>(), md.i8_logw, b, (double*)nullptr, F, Nk, GY, ntiles, (int)tpb, nchunks, dbg, q32)
    unsigned long long* dbg = nullptr;
#ifdef LNASR_I8_STAMPS                                    // diagnostic build: in-kernel phase stamps (s_memtime), printed per launch
    if (getenv("LNASR_I8_STAMPS") && !bm && mix == 16) {
        const size_t n = (size_t)grid * 8 * 7;
        LN_HIP(hipMalloc((void**)&dbg, n * 8));           // (diagnostic build: not a workspace of the context, freed below)
        LN_HIP(hipMemsetAsync(dbg, 0, n * 8, ctx->stream));
        LN_I8K(6, true, false, 16, "emit_i8_stamps");
        std::vector<unsigned long long> h(n);
        LN_HIP(hipMemcpyAsync(h.data(), dbg, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        LN_HIP(hipStreamSynchronize(ctx->stream));
        LN_HIP(hipFree(dbg));
        double sum[7] = {0, 0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < n; ++i) sum[i % 7] += (double)h[i];
        const double per = 1.0 / ((double)grid * 8 * (double)tpb);
        fprintf(stderr, "i8 stamps, cycles per wave-tile: phase one %.0f  barrier %.0f  (%.0f)  phase two %.0f  barrier %.0f\n",
                sum[0] * per, sum[1] * per, sum[2] * per, sum[3] * per, sum[4] * per);
        fprintf(stderr, "   wave lifetime %.0f cycles = %.3f ms: clock %.2f GHz\n", sum[5] / ((double)grid * 8),
                sum[6] / ((double)grid * 8) * 1e-5, sum[5] / sum[6] * 0.1);
        return 0;
    }
#endif
    const bool k64 = 2 * D <= 64;                         // K <= 64: the 28 plane pairs and nothing else
    if (q32) {                                            // the float32 statistics pass's hand-off: b + q32, four planes
        LN_REQUIRE(!bm && mix != 8, "the float32 hand-off takes 16 or 32 mixtures per state and no bm");
        if (mix == 32 && k64) LN_I8Q(false, 32, "emit_diag_i8_q32_m32_k64");
        else if (mix == 32) LN_I8Q(true, 32, "emit_diag_i8_q32_m32");
        else if (k64) LN_I8Q(false, 16, "emit_diag_i8_q32_k64");
        else LN_I8Q(true, 16, "emit_diag_i8_q32");
        return 0;
    }
    if (mix == 8) {
        LN_REQUIRE(!bm, "the 8-mixture integer bank writes b only");
        if (k64) LN_I8K(0, false, false, 8, "emit_diag_i8_m8_k64");
        else LN_I8K(0, true, false, 8, "emit_diag_i8_m8");
    } else if (mix == 32) {
        if (bm && k64) LN_I8K(0, false, true, 32, "emit_diag_i8_m32_k64_bm");
        else if (bm) LN_I8K(0, true, true, 32, "emit_diag_i8_m32_bm");
        else if (k64) LN_I8K(0, false, false, 32, "emit_diag_i8_m32_k64");
        else LN_I8K(0, true, false, 32, "emit_diag_i8_m32");
    } else {
        if (bm && k64) LN_I8K(0, false, true, 16, "emit_diag_i8_k64_bm");
        else if (bm) LN_I8K(0, true, true, 16, "emit_diag_i8_bm");
        else if (k64) LN_I8K(0, false, false, 16, "emit_diag_i8_k64");
        else LN_I8K(0, true, false, 16, "emit_diag_i8");
    }
#undef LN_I8K
#undef LN_I8Q
    return 0;
}

}  // namespace lnasr
