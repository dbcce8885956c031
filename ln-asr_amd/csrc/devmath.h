// Device-side helpers: wave64 reductions, log-sum-exp pieces, storage conversion.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace lnasr {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;   // ln(2 pi)

#define LN_NEG_INF (-__builtin_huge_val())

// Two blocks that share a CU are not served alike: the SIMD arbiter goes by priority, then AGE, so the older block runs
// nearly unimpeded, the younger gets the leftover issue slots, finishes long after it and is alone on the CU (unable to
// fill it) for the difference (tools/block_times.py).  Kernels whose grid is two resident blocks per CU call this once
// per loop turn: the two halves of the grid (blocks b and b + nblocks / 2 go to the same CU: the first half fills every
// CU once) raise their priority in alternate 164 us periods of the 100 MHz clock, each gets the arbiter's favour half of
// the time, and they finish together.  If the pairing is not what is assumed nothing is lost: two blocks of one class
// tie on priority and fall back on age.
__device__ __forceinline__ void alternate_priority(unsigned block_class) {
    if ((((unsigned)__builtin_amdgcn_s_memrealtime() >> 14) ^ block_class) & 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// Developer diagnostic (-DLNASR_BLOCK_TIMES, tools/block_times.py; never in the shipped build): first and last 100 MHz
// tick of every block of an instrumented kernel, per translation unit.  Shows start-up skew, tails and imbalance that
// the kernel's duration alone hides (it found the MFCC kernel's 20 % tail: DESIGN 4.6).
#ifdef LNASR_BLOCK_TIMES
static __device__ unsigned long long g_block_times[2 * 16384];
static __device__ unsigned long long g_block_cycles[16384];      // shader-clock cycles of the block's life (s_memtime):
#define LN_BLOCK_T0                                                      /* cycles / ticks * 100 MHz = its clock */ \
    const unsigned long long bt0__ = __builtin_amdgcn_s_memrealtime(), bc0__ = __builtin_amdgcn_s_memtime()
#define LN_BLOCK_T1                                                                  \
    do {                                                                             \
        const unsigned bid__ = blockIdx.x + gridDim.x * blockIdx.y;                  \
        if (threadIdx.x == 0 && bid__ < 16384) {                                     \
            g_block_times[2 * bid__] = bt0__;                                        \
            g_block_times[2 * bid__ + 1] = __builtin_amdgcn_s_memrealtime();         \
            g_block_cycles[bid__] = __builtin_amdgcn_s_memtime() - bc0__;            \
        }                                                                            \
    } while (0)
#define LN_BLOCK_TIMES_EXPORT(fn)                                                    \
    extern "C" int fn(unsigned long long* out) {                                     \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                         \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_times), sizeof(g_block_times)) == hipSuccess ? 0 : -1; \
    }                                                                                \
    extern "C" int fn##_cycles(unsigned long long* out) {                            \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                         \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_cycles), sizeof(g_block_cycles)) == hipSuccess ? 0 : -1; \
    }
#else
#define LN_BLOCK_T0
#define LN_BLOCK_T1
#define LN_BLOCK_TIMES_EXPORT(fn)
#endif

template <typename TS>
__device__ __forceinline__ double ld(const TS* p, size_t i) { return (double)p[i]; }
template <typename TS>
__device__ __forceinline__ void st(TS* p, size_t i, double v) { p[i] = (TS)v; }
// value as it reads back after being stored in TS
template <typename TS>
__device__ __forceinline__ double rnd(double v) { return (double)(TS)v; }

// Drain the vector-memory counter.  Put after loop-invariant operands have been loaded into registers and before
// a loop that prefetches: otherwise the compiler's wait-count model carries the preamble's loads into the loop
// as "possibly outstanding" and guards every use of those registers with s_waitcnt vmcnt(n), n counting down to
// 0 -- which, the counter being in-order, stalls on the loop's OWN freshly issued prefetch.
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0) only

// one-instruction max: the library fmax first canonicalises operands the compiler cannot prove quiet
// (MFMA results, LDS / global loads), which doubles the instruction count of max chains
__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, shfl_xor_f64(v, m));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

// wave64 max of a double through DPP (no LDS crossbar): quad_perm, half-mirror and mirror steps
// leave every lane of a 16-lane row with the row's max; the four row results are then read as
// scalars.  ctrl codes: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    // every control used here (quad_perm, row_half_mirror, row_mirror, row_ror) writes every lane from a valid lane, so
    // there is no `old` value to preserve: v_mov_b32_dpp straight from the source, one instruction per word (with
    // update_dpp(old = x, x) the compiler copies x first: two)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
// max over the wave, left in EVERY lane: four DPP steps inside the 16-lane rows, then the rows meet through
// v_permlane16_swap / v_permlane32_swap (gfx950) -- 22 instructions where DPP with a copied `old` operand plus four
// row leaders read back as scalars took 47, on the dependent chain of the backtrace and of the trellis finalisation.
// max is exact, so the result is the same bits whatever the order.
__device__ __forceinline__ double wave_max_dpp(double x) {
    x = vmax_f64(x, dpp_f64<0xB1>(x));
    x = vmax_f64(x, dpp_f64<0x4E>(x));
    x = vmax_f64(x, dpp_f64<0x141>(x));
    x = vmax_f64(x, dpp_f64<0x140>(x));
    {
        auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        x = vmax_f64(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
    }
    {
        auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        x = vmax_f64(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
    }
    return x;
}
// same walk with additions (association: quads, rows of 16, then the four rows)
__device__ __forceinline__ double wave_sum_dpp(double x) {
    x += dpp_f64<0xB1>(x);
    x += dpp_f64<0x4E>(x);
    x += dpp_f64<0x141>(x);
    x += dpp_f64<0x140>(x);
    const double r0 = readlane_f64(x, 0), r1 = readlane_f64(x, 16), r2 = readlane_f64(x, 32),
                 r3 = readlane_f64(x, 48);
    return (r0 + r1) + (r2 + r3);
}

// max over the 16 lanes of a DPP row of a FLOAT, left in every lane of the row (the compiler folds each move into a
// v_max_f32_dpp).  Used for the LEVEL of an emission row -- max_j b[t, j] to float accuracy -- which the scaled linear-domain
// recursions add to their lagged shift (fwdbwd.hip): a level is good to a few nats, it does not need fp64.
__device__ __forceinline__ float row_max_f32(float x) {
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)));
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)));
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true)));
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true)));
    return x;
}

// fp64 DPP exists on gfx950 for v_fmac_f64 / v_mov_b64 with row_newbcast only: lane NB of every
// 16-lane row is broadcast to the whole row
template <int NB>
__device__ __forceinline__ void fmac_bcast(double& acc, double v, double a) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(a), "n"(NB));
}
template <int NB>
__device__ __forceinline__ double mov_bcast(double v) {
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(NB));
    return r;
}
template <int I, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// rare path (sum left the safe range): log-sum-exp of prev[i] + logA[i][j] in the log domain; kept out
// of line so that its registers are not charged to the frame loop
static __device__ __noinline__ double exact_column_lse(const double* prev, const double* Acol, int N) {
    double mx = LN_NEG_INF;
    for (int i = 0; i < N; ++i) mx = fmax(mx, prev[i] + Acol[(size_t)i * N]);
    if (mx == LN_NEG_INF || mx != mx) return mx;
    double s2 = 0.0;
    for (int i = 0; i < N; ++i) s2 += exp(prev[i] + Acol[(size_t)i * N] - mx);
    return mx + log(s2);
}

// (value, index) arg-max with numpy.argmax semantics: first index among equal maxima.
__device__ __forceinline__ void argmax_combine(double& v, int& i, double ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}
// Callers start from (-inf, kNoIndex).  A NaN candidate never wins a comparison: a row of NaNs leaves kNoIndex standing,
// which comes back as 0 -- numpy.argmax's answer for an all-NaN row, and never an address outside the row.
constexpr int kNoIndex = 0x7fffffff;
__device__ __forceinline__ void wave_argmax(double& v, int& i) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double ov = shfl_xor_f64(v, m);
        int oi = __shfl_xor(i, m, 64);
        argmax_combine(v, i, ov, oi);
    }
    i = i == kNoIndex ? 0 : i;
}

// Block-wide reductions through a small LDS scratch (>= 2 * nwaves doubles). All threads must call.
__device__ __forceinline__ double block_max(double v, double* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    if (nw == 1) return v;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = scratch[0];
    for (int k = 1; k < nw; ++k) r = fmax(r, scratch[k]);
    return r;
}
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (nw == 1) return v;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = scratch[0];
    for (int k = 1; k < nw; ++k) r += scratch[k];
    return r;
}

// Online log-sum-exp accumulator (max-shifted; an all -inf input yields -inf like the
// reference's logaddexp fold, utils.py:176-179).
struct Lse {
    double m = LN_NEG_INF, s = 0.0;
    // A NaN term poisons the sum, as np.logaddexp does (utils.py:140-179): it fails both comparisons below and becomes
    // the running maximum, after which nothing compares either way and value() is NaN + log(s).
    __device__ __forceinline__ void add(double z) {
        if (z == LN_NEG_INF) return;
        if (z > m) {
            s = s * exp(m - z) + 1.0;   // m = -inf: s is 0, exp(-inf) = 0
            m = z;
        } else if (z <= m) {
            s += exp(z - m);
        } else if (m == m) {
            m = z;                      // z is NaN
        }
    }
    __device__ __forceinline__ void merge(double om, double os) {
        if (om == LN_NEG_INF) return;
        if (om > m) {
            s = s * exp(m - om) + os;
            m = om;
        } else if (om <= m) {
            s += os * exp(om - m);
        } else if (m == m) {
            m = om;
        }
    }
    __device__ __forceinline__ double value() const { return (m == LN_NEG_INF) ? m : m + log(s); }
};

// exp(x) for x <= 0 (log-sum-exp terms after the max shift), ~1 ulp, 10 fp64 + 4 int instructions:
//   x = k ln2/256 + r, |r| <= ln2/512   =>   exp(x) = 2^(k>>8) * 2^((k&255)/256) * (1 + r + r^2/2 + r^3/6 + r^4/24)
// (remainder r^5/120 < 4e-17).  k comes out of the low mantissa bits of x*256/ln2 + 1.5*2^52; the
// power of two is applied by adding to the exponent field of the table entry, which is safe
// because x is clamped to >= -700 (e^-700 is 1e-304: no contribution to a sum whose largest term is 1).
constexpr int kExpTab = 256;
__device__ __forceinline__ double exp_nonpos(double x, const double* tab) {
    constexpr double kInvStep = 369.32993046757463;          // 256 / ln 2
    constexpr double kStepHi = 2.7076061740622863e-03;        // ln 2 / 256 (hi)
    constexpr double kStepLo = 9.058776616587108e-20;         // ln 2 / 256 - hi
    constexpr double kMagic = 6755399441055744.0;             // 1.5 * 2^52
    x = vmax_f64(x, -700.0);
    const double kp = fma(x, kInvStep, kMagic);
    const int k = __double2loint(kp);
    const double kd = kp - kMagic;
    double r = fma(-kd, kStepHi, x);
    r = fma(-kd, kStepLo, r);
    double q = fma(r, 1.0 / 24.0, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    const double p = fma(q, r * r, r);
    const double t0 = tab[k & (kExpTab - 1)];
    const int hi = __double2hiint(t0) + ((k >> 8) << 20);
    const double t = __hiloint2double(hi, __double2loint(t0));
    return fma(t, p, t);
}

// Same scheme with a 2048-entry table (16 KB of LDS): |r| <= ln2/4096, one polynomial degree less
// (remainder r^4/24 < 3.5e-17): 9 fp64 instructions.  Used where the LDS is there for the taking.
constexpr int kExpTabBig = 2048;
__device__ __forceinline__ double exp_nonpos_big(double x, const double* tab) {
    constexpr double kInvStep = 369.32993046757463 * 8.0;        // 2048 / ln 2 (exact scaling)
    constexpr double kStepHi = 2.7076061740622863e-03 / 8.0;      // ln 2 / 2048 (hi)
    constexpr double kMagic = 6755399441055744.0;
    x = vmax_f64(x, -700.0);
    const double kp = fma(x, kInvStep, kMagic);
    const unsigned k = (unsigned)__double2loint(kp);
    const double kd = kp - kMagic;
    // one-step reduction: the low part of ln2/2048 (1.1e-20) times |kd| <= 2.1e6 is at most 2.4e-14 and
    // proportional to |x|, i.e. a relative error of |x| * 3.3e-17 on a term of size e^x -- below 1.3e-17
    // absolute for every x <= 0, a twentieth of an ulp of the sum (>= 1) these terms are added into
    const double r = fma(-kd, kStepHi, x);
    const double q = fma(r, 1.0 / 6.0, 0.5);
    const double p = fma(q, r * r, r);
    const double t0 = tab[k & (kExpTabBig - 1)];
    // 2^(k >> 11) goes into the exponent field: (k with its low 11 bits cleared) << 9 lands on bit 20
    const unsigned hi = (unsigned)__double2hiint(t0) + ((k & ~(unsigned)(kExpTabBig - 1)) << 9);
    const double t = __hiloint2double((int)hi, __double2loint(t0));
    return fma(t, p, t);
}

// exp_nonpos_big with the exponent applied by ONE integer instruction: the table holds 2^(j/2048) with (j << 9)
// already subtracted from its high word (exp_table_big_init), so that adding (k << 9) -- k = 2048 e + j -- leaves
// exactly e << 20 on the exponent field: v_lshl_add_u32 instead of and + shift + add.
__device__ __forceinline__ void exp_table_big_init(double* etab, int tid, int nt) {
    for (int e = tid; e < kExpTabBig; e += nt) {
        const double t = exp2((double)e * (1.0 / kExpTabBig));
        etab[e] = __hiloint2double(__double2hiint(t) - (e << 9), __double2loint(t));
    }
}
__device__ __forceinline__ double exp_nonpos_big2(double x, const double* tab) {
    constexpr double kInvStep = 369.32993046757463 * 8.0;        // 2048 / ln 2
    constexpr double kStepHi = 2.7076061740622863e-03 / 8.0;      // ln 2 / 2048
    constexpr double kMagic = 6755399441055744.0;                 // 1.5 * 2^52
    x = vmax_f64(x, -700.0);
    const double kp = fma(x, kInvStep, kMagic);
    const unsigned k = (unsigned)__double2loint(kp);
    const double kd = kp - kMagic;
    const double r = fma(-kd, kStepHi, x);                        // one-step reduction: see exp_nonpos_big
    const double q = fma(r, 1.0 / 6.0, 0.5);
    const double p = fma(q, r * r, r);
    const double t0 = tab[k & (kExpTabBig - 1)];
    const unsigned hi = (unsigned)__double2hiint(t0) + (k << 9);      // one v_lshl_add_u32
    const double t = __hiloint2double((int)hi, __double2loint(t0));
    return fma(t, p, t);
}

// log(s) for the sum of a max-shifted log-sum-exp: 1 <= s <= 2^20.  s = 2^e m, m in [1,2);
// c_j = 1 + (j + 1/2)/128 from the top seven mantissa bits; r = m/c_j - 1 (|r| <= 1/256, formed exactly
// by an fma with the tabulated reciprocal); log s = e ln2 + log c_j + (r - r^2/2 + ... - r^6/6)
// (remainder r^7/7 < 2e-18).  ltab[2j] = fl(1/c_j), ltab[2j+1] = -log(ltab[2j]).  ~10 fp64 + 6 integer
// instructions against ~35 for the library log.
constexpr int kLogTab = 128;
__device__ __forceinline__ void log_table_init(double* ltab, int tid, int nt) {
    for (int e = tid; e < kLogTab; e += nt) {
        const double rc = 1.0 / (1.0 + ((double)e + 0.5) * (1.0 / kLogTab));
        ltab[2 * e] = rc;
        ltab[2 * e + 1] = -log(rc);
    }
}
__device__ __forceinline__ double log_sum(double s, const double* ltab) {
    constexpr double kLn2 = 0.69314718055994530942;
    const int hi = __double2hiint(s);
    const int e = (hi >> 20) - 1023;
    const int j = (hi >> 13) & (kLogTab - 1);
    const double m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, __double2loint(s));
    const double r = fma(m, ltab[2 * j], -1.0);
    double q = fma(r, -1.0 / 6.0, 0.2);
    q = fma(q, r, -0.25);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -0.5);
    q = fma(q, r, 1.0);
    return fma((double)e, kLn2, fma(q, r, ltab[2 * j + 1]));
}

// exp(x) for any x: same scheme as exp_nonpos below 690, +inf above (callers treat an infinite term as
// "this sum must be redone in the log domain", exactly what the library exp's overflow would trigger)
__device__ __forceinline__ double exp_clamped(double x, const double* tab) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(690.0));
    const double e = exp_nonpos(r, tab);
    return x > 690.0 ? __builtin_huge_val() : e;
}
__device__ __forceinline__ void exp_table_init(double* etab, int tid, int nt) {
    for (int e = tid; e < kExpTab; e += nt) etab[e] = exp2((double)e * (1.0 / kExpTab));
}

// b[cell] = lse_h bv[cell*k + h]: M = 16*k mixtures are scored as k virtual 16-mixture states, then merged
static __global__ void merge_virtual_kernel(const double* __restrict__ bv, size_t cells, int N, int Np, int k,
                                            double* __restrict__ b) {
    // bv [F, Np, k] (Np >= N: states padded to a multiple of four) -> b [F, N]
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    const size_t f = idx / N;
    const int n = (int)(idx % N);
    const double* src = bv + (f * Np + n) * k;
    if (k == 1) {
        b[idx] = src[0];
        return;
    }
    Lse acc;
    for (int h = 0; h < k; ++h) acc.add(src[h]);
    b[idx] = acc.value();
}

// bm of the padded model [F, Np, Mp] -> [F, N, M]
static __global__ void compact_bm_kernel(const double* __restrict__ src, size_t cells, int N, int M, int Np, int Mp,
                                  double* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    const int m = (int)(idx % M);
    const size_t fn = idx / M;
    const int n = (int)(fn % N);
    const size_t f = fn / N;
    dst[idx] = src[((f * Np + n) * Mp) + m];
}

}  // namespace lnasr
