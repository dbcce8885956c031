// Model upload and GMM emission scoring (gmm.py:16-100, gmmhmm.py:46-67).
//
// Kernels
//   prep_diag_kernel      1/var, Gaussian constants                       (once per model)
//   lu_invert_kernel      Sigma^-1 and log|det| with partial pivoting     (once per model)
//   pack_wfrag_kernel     coefficient matrix in MFMA A-fragment order     (once per model)
//   emit_diag_mfma        b = lse_m(w + N(o; mu, diag var)) as an fp64 MFMA GEMM
//                         [x^2, x, 1] (frames x K) . [-1/2 iv, mu iv, const] (K x N*M)
//                         with the M-way log-sum-exp fused in-lane          (hot kernel)
//   emit_diag_generic     any N/M/D, fp64 VALU, direct (x-mu)^2/var form
//   emit_full_generic     full covariance: (x-mu)^T P (x-mu)
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "devmath.h"

namespace lnasr {

typedef double double4_t __attribute__((ext_vector_type(4)));

int pack_full_model(lnasr_ctx* ctx);                                                   // gmm_full.hip
int emit_full_mfma_device(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm);
int emit_full_direct_device(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm);   // gmm_full.hip

// ------------------------------------------------------------------------------------------
// model preparation
// ------------------------------------------------------------------------------------------
__global__ void transpose_exp_kernel(const double* __restrict__ A, double* __restrict__ AT,
                                     double* __restrict__ E, double* __restrict__ ET, int N) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    int i = idx / N, j = idx % N;
    double a = A[idx];
    AT[(size_t)j * N + i] = a;
    const double e = exp(a);
    E[idx] = e;
    ET[(size_t)j * N + i] = e;
}

// flag[0] = 1 if any off-diagonal entry of any Sigma is not exactly zero.  256 blocks walk the components; a block that finds one
// reports it with ONE store (thread 0, after a block-wide OR, and only if nobody has reported yet) and every block stops at the
// first report.  What this kernel costs is the traffic on the flag's ONE L2 line, not the 25 MB it may read: an atomicOr per
// non-zero entry was three million atomics (1.1 ms of every upload of a genuinely full model); one per wave still let every
// resident wave of a one-thread-per-entry grid queue its atomic (0.12 ms); a store per wave from 2048 blocks, 0.30 ms (round 5,
// measured on the way here).  A diagonal model -- the case that reads everything -- streams its matrices once.
__global__ void detect_offdiag_kernel(const double* __restrict__ S, int NM, int D, int* flag) {
    const int DD = D * D;
    for (int c = blockIdx.x; c < NM; c += gridDim.x) {
        // ONE thread reads the word and the block decides together: every wave reading it for itself could see another
        // block's report between two waves' loads, and part of the block would walk into the barrier below without the rest
        if (__syncthreads_or(threadIdx.x == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return;
        const double* Sc = S + (size_t)c * DD;
        bool hit = false;
        for (int e = threadIdx.x; e < DD; e += blockDim.x) {
            const int r = e / D, col = e - r * D;
            hit |= r != col && !(Sc[e] == 0.0);
        }
        if (__syncthreads_or(hit)) {
            if (threadIdx.x == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

__global__ void extract_diag_kernel(const double* __restrict__ S, double* __restrict__ var, size_t NMD, int D) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NMD) return;
    size_t c = idx / D;
    int d = (int)(idx % D);
    var[idx] = S[c * D * D + (size_t)d * D + d];
}

// numpy.linalg.det returns sign * exp(sum log|u_ii|); the reference then takes log(det)
// (gmm.py:62,65).  Keeping the exp/log round trip keeps its range behaviour (det overflow ->
// +inf, underflow -> -inf, negative det -> NaN).
__device__ __forceinline__ double log_of_det(double logabs, int sign) {
    double det = (double)sign * exp(logabs);
    return log(det);
}

// one thread per component
__global__ void prep_diag_kernel(const double* __restrict__ var, double* __restrict__ ivar,
                                 double* __restrict__ gconst, int NM, int D, int* __restrict__ degenerate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= NM) return;
    double la = 0.0;
    int sign = 1;
    bool bad = false;
    for (int d = 0; d < D; ++d) {
        double v = var[(size_t)c * D + d];
        const double iv = 1.0 / v;
        ivar[(size_t)c * D + d] = iv;
        la += log(fabs(v));
        if (v < 0.0) sign = -sign;
        if (v == 0.0) sign = 0;
        bad |= !(iv == iv) || iv == __builtin_huge_val() || iv == -__builtin_huge_val();
    }
    const double gc = -0.5 * ((double)D * kLog2Pi + log_of_det(la, sign));
    gconst[c] = gc;
    bad |= !(gc == gc) || gc == __builtin_huge_val() || gc == -__builtin_huge_val();
    // a singular / non-finite variance: the expanded-quadratic (matrix-core) form would turn it into NaNs that the
    // log-sum-exp ignores; such models stay on the direct-form kernels, which give the inf / nan numpy gives
    if (bad) atomicOr(degenerate, 1);
}

// centre[d] = mean over the components of mu[c][d] (0 if that mean is not finite: such a model is refused the
// expanded form below anyway).  One block per dimension.
__global__ void centre_kernel(const double* __restrict__ mu, int NM, int D, double* __restrict__ centre) {
    __shared__ double scratch[8];
    const int d = blockIdx.x;
    double s = 0.0;
    for (int c = threadIdx.x; c < NM; c += blockDim.x) s += mu[(size_t)c * D + d];
    s = block_sum(s, scratch);
    s /= (double)NM;
    if (threadIdx.x == 0) centre[d] = (s - s == 0.0) ? s : 0.0;
}

// mu_c = mu - centre, and kappa = max_c (mu_c)^T P_c (mu_c): the size of the terms the expanded quadratic cancels.
// One WAVE per component (rows of P read coalesced; a thread per component walked its own 12 KB matrix with a stride of
// D doubles: 1.3 ms per pass at the metric shape); prec == nullptr: diagonal model (ivar).  A non-finite value counts as +inf.
__global__ void centre_mu_kernel(const double* __restrict__ mu, const double* __restrict__ centre,
                                 const double* __restrict__ ivar, const double* __restrict__ prec, int NM, int D,
                                 double* __restrict__ mu_c, unsigned long long* __restrict__ kappa_bits) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= NM) return;
    const double* m = mu + (size_t)c * D;
    double* mc = mu_c + (size_t)c * D;
    double k = 0.0;                                    // lane-partial; summed over the wave at the end (all terms >= 0)
    for (int j0 = 0; j0 < D; j0 += 64) {
        const int j = j0 + lane;
        const double mj = j < D ? m[j] - centre[j] : 0.0;
        if (j < D) mc[j] = mj;
        if (prec) {
            const double* P = prec + (size_t)c * D * D;
            for (int i = 0; i < D; ++i) {              // k += |mc_i| * sum_j |P_ij mc_j|
                const double mi = m[i] - centre[i];
                if (j < D) k += fabs(mi) * fabs(P[(size_t)i * D + j] * mj);
            }
        } else if (j < D) {
            k += mj * mj * fabs(ivar[(size_t)c * D + j]);
        }
    }
    k = wave_sum(k);
    if (!(k - k == 0.0)) k = __builtin_huge_val();
    if (lane == 0) atomicMax(kappa_bits, (unsigned long long)__double_as_longlong(k));   // k >= 0: the bit patterns order like the values
}

// One WAVE (a 64-thread block) per component: Gauss-Jordan elimination with partial pivoting on
// [Sigma | I] held in LDS.  LU semantics: Sigma need not be symmetric (test/gmm-test.py:42-43).
// Round 4: the pivot search is a wave arg-max over the rows (it was thread 0 walking the column: 39 dependent LDS reads per
// step), the row factor a[r][k] / pivot is formed ONCE per row by lane r (it was formed per element: 3,000 fp64 divisions
// per step), and the elimination walks rows with the lane's two columns fixed (no integer division per element).  Same
// pivots, same factors, same subtractions as before: the inverse is bit for bit the one the old kernel produced.
// 1.78 -> 1.19 ms for the 2048 components of the metric shape (what is left is the column walk through the LDS).
__global__ void lu_invert_kernel(const double* __restrict__ S, double* __restrict__ P,
                                 double* __restrict__ gconst, int D, const int* __restrict__ only) {
    extern __shared__ double lds[];
    const int W = 2 * D;
    double* a = lds;                          // [D][2D]
    double* fac = lds + (size_t)D * W;        // [D] row factors of the current step
    const int c = blockIdx.x, lane = threadIdx.x;            // blockDim.x == 64
    if (only && !only[c]) return;             // second pass behind lu_invert_reg_kernel: the components it declined
    const double* Sc = S + (size_t)c * D * D;
    for (int e = lane; e < D * W; e += 64) {
        int r = e / W, col = e % W;
        a[e] = col < D ? Sc[(size_t)r * D + col] : (col - D == r ? 1.0 : 0.0);
    }
    __syncthreads();
    double logabs = 0.0;
    int sign = 1;
    for (int k = 0; k < D; ++k) {
        // pivot: first row r >= k with the largest |a[r][k]| (rows beyond 64 in strides; D <= 88)
        double bv = -1.0;
        int best = kNoIndex;
        for (int r = k + lane; r < D; r += 64) {
            const double v = fabs(a[r * W + k]);
            if (v > bv) { bv = v; best = r; }                // NaN never wins, as in the serial search
        }
        wave_argmax(bv, best);
        const int pr = (best == 0 && !(bv >= 0.0)) ? k : best;   // nothing comparable in the column (all NaN): keep row k
        const double pv = a[pr * W + k];
        __syncthreads();
        if (pr != k) {
            for (int col = lane; col < W; col += 64) {
                double t = a[k * W + col];
                a[k * W + col] = a[pr * W + col];
                a[pr * W + col] = t;
            }
            sign = -sign;
        }
        logabs += log(fabs(pv));
        if (pv < 0.0) sign = -sign;
        if (pv == 0.0) sign = 0;
        __syncthreads();
        for (int r = lane; r < D; r += 64) fac[r] = a[r * W + k] / pv;
        __syncthreads();
        // eliminate column k from every other row (row k itself is scaled afterwards)
        for (int col = lane; col < W; col += 64) {
            if (col != k) {
                const double akc = a[k * W + col];
                for (int r = 0; r < D; ++r)
                    if (r != k) a[r * W + col] -= fac[r] * akc;
            }
        }
        __syncthreads();
        for (int r = lane; r < D; r += 64)
            if (r != k) a[r * W + k] = 0.0;
        for (int col = lane; col < W; col += 64) a[k * W + col] /= pv;
        __syncthreads();
    }
    for (int e = lane; e < D * D; e += 64) {
        int r = e / D, col = e % D;
        P[(size_t)c * D * D + e] = a[r * W + D + col];
    }
    if (lane == 0) gconst[c] = -0.5 * ((double)D * kLog2Pi + log_of_det(logabs, sign));
}

// Round 5: the same elimination with the matrix in REGISTERS (D <= 40): one wave per component, lane r holds row r of the
// augmented system [Sigma | I] -- but only the DP columns that are not yet (left part) or no longer (right part) unit
// vectors, in a window that slides one column to the left per step: x[0] is always the pivot column, step k consumes it and
// appends the right-part column that the pivot row's unit entry turns into.  The step loop is a real loop, the column loop is
// unrolled over fixed registers.  Rows are never moved: a lane remembers which logical row it holds (`pos`), the pivot search
// runs over the lanes whose logical row is >= k with the old kernel's rule (largest |a|, lowest logical row among equals), and
// the row / column permutation is undone when the result is written.  The pivot row reaches the other lanes through the LDS
// (one broadcast read per column).  Per step ~DP fused multiply-adds and ~2 DP LDS / move instructions per lane against five
// block barriers and a column walk through the LDS: 0.60 ms -> see DESIGN 4.10 for the 2048 components of the metric shape.
// Same pivots; factors are formed with the pivot's reciprocal (one division per step instead of one per row).
// A component with a zero / non-finite pivot or a non-finite entry is left to lu_invert_kernel (`bad[c]` = 1), whose way of
// spreading the infinities and NaNs is the pinned one.
template <int DP>
__global__ __launch_bounds__(64) void lu_invert_reg_kernel(const double* __restrict__ S, double* __restrict__ P,
                                                           double* __restrict__ gconst, int D, int* __restrict__ bad) {
    __shared__ double st[DP * DP];            // Sigma on the way in, the inverse on the way out (coalesced both ways)
    __shared__ double bc[DP];                 // the pivot row of the step
    __shared__ double lg[DP];                 // log |pivot| per step
    __shared__ int colmap[DP];                // slot j of the final window <-> original column colmap[j]
    const int c = blockIdx.x, lane = threadIdx.x;
    const double* Sc = S + (size_t)c * D * D;
    bool nonfinite = false;
    for (int e = lane; e < D * D; e += 64) {
        const double v = Sc[e];
        nonfinite |= !(v - v == 0.0);
        st[e] = v;
    }
    if (__any(nonfinite)) {
        if (lane == 0) bad[c] = 1;
        return;
    }
    __syncthreads();
    double x[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) {
        double v = (lane == j) ? 1.0 : 0.0;                       // padding rows / columns: identity
        if (lane < D && j < D) v = st[lane * D + j];
        x[j] = lane < DP ? v : 0.0;
    }
    int pos = lane;                           // logical row held by this lane
    double mypv = 1.0;                        // lane k keeps the pivot of step k
    int nswap = 0;
    for (int k = 0; k < DP; ++k) {
        double av = (lane < DP && pos >= k) ? fabs(x[0]) : -1.0;
        av = av >= 0.0 ? av : -1.0;                               // a NaN never wins
        const double mx = wave_max_dpp(av);
        if (!(mx > 0.0) || mx == __builtin_huge_val()) {          // singular / overflowed: the LDS kernel's business
            if (lane == 0) bad[c] = 1;
            return;
        }
        unsigned long long cand = __ballot(av == mx);
        int p = __ffsll((long long)cand) - 1;
        if (cand & (cand - 1)) {                                  // equal magnitudes: the lowest logical row
            int best = 1 << 30;
            while (cand) {
                const int l = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const int pl = __builtin_amdgcn_readlane(pos, l);
                if (pl < best) { best = pl; p = l; }
            }
        }
        const int q = __builtin_amdgcn_readlane(pos, p);
        if (pos == k) pos = q;                                    // the swap of rows k and q, on the labels only
        if (lane == p) pos = k;
        nswap += q != k;
        const double pv = readlane_f64(x[0], p);
        const double rp = 1.0 / pv;
        if (lane == k) mypv = pv;
        const bool isp = lane == p;
        if (isp) {
#pragma unroll
            for (int j = 1; j < DP; ++j) {
                bc[j] = x[j];
                x[j] = 0.0;                                       // so that the update below leaves row_p * rp here
            }
        }
        __syncthreads();
        const double fac = isp ? -rp : x[0] * rp;
#pragma unroll
        for (int j = 1; j < DP; ++j) x[j - 1] = fma(-fac, bc[j], x[j]);
        x[DP - 1] = -fac;                                         // the entering right-part column: unit entry in the pivot row
        __syncthreads();
    }
    // lane l holds logical row pos; slot j belongs to the original column of the lane that was pivot at step j
    if (lane < DP) colmap[pos] = lane;
    if (lane < DP) lg[lane] = log(fabs(mypv));
    __syncthreads();
    if (lane < DP && pos < D) {
#pragma unroll
        for (int j = 0; j < DP; ++j) {
            const int col = colmap[j];
            if (col < D) st[pos * D + col] = x[j];
        }
    }
    const unsigned long long negs = __ballot(lane < D && mypv < 0.0);
    __syncthreads();
    for (int e = lane; e < D * D; e += 64) P[(size_t)c * D * D + e] = st[e];
    if (lane == 0) {
        double logabs = 0.0;
        for (int k = 0; k < D; ++k) logabs += lg[k];
        const int sign = ((__popcll(negs) + nswap) & 1) ? -1 : 1;
        gconst[c] = -0.5 * ((double)D * kLog2Pi + log_of_det(logabs, sign));
    }
}

// Coefficients of the expanded quadratic, laid out as MFMA A-operand fragments:
//   wpack[((sg*Q + q)*KS + ks)*64 + lane]  = Wt[comp(sg, q, row = lane&15)][k = 4*ks + (lane>>4)]
//   row r of tile q  <->  state 4*sg + (r&3), mixture 4*q + (r>>2)
// so that after the MFMA, lane group (lane>>4) = g holds ALL mixtures of state 4*sg+g for
// frame (lane&15) in its own registers and the M-way log-sum-exp needs no cross-lane traffic.
//   k <  D      : -0.5 / var[d=k]
//   D <= k < 2D : mu / var
//   k == 2D     : log w + gconst - 0.5 * sum mu^2 / var
//   k >  2D     : 0 (padding)
__global__ void pack_wfrag_kernel(const double* __restrict__ mu, const double* __restrict__ ivar,
                                  const double* __restrict__ gconst, const double* __restrict__ logw,
                                  double* __restrict__ wpack, int N, int M, int D, int Q, int KS, int fold_w) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)(N / 4) * Q * KS * 64;
    if (idx >= total) return;
    int lane = (int)(idx & 63);
    size_t r1 = idx >> 6;
    int ks = (int)(r1 % KS);
    r1 /= KS;
    int q = (int)(r1 % Q);
    int sg = (int)(r1 / Q);
    int row = lane & 15, k = 4 * ks + (lane >> 4);
    int state = 4 * sg + (row & 3), mix = 4 * q + (row >> 2);
    size_t c = (size_t)state * M + mix;
    double val = 0.0;
    if (k < D) {
        val = -0.5 * ivar[c * D + k];
    } else if (k < 2 * D) {
        int d = k - D;
        val = mu[c * D + d] * ivar[c * D + d];
    } else if (k == 2 * D) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            double m = mu[c * D + d];
            s += m * m * ivar[c * D + d];
        }
        val = (fold_w ? logw[c] : 0.0) + gconst[c] - 0.5 * s;
    }
    wpack[idx] = val;
}

// Measured on MI355X (tools/ubench): v_mfma_f64_16x16x4 issues every 64.5 cycles per SIMD
// (78 TFLOP/s chip-wide = the fp64 vector rate) and does NOT overlap with VALU work of the same
// or of a co-resident wave -- the fp64 matrix op occupies the vector datapath.  The kernel's
// time is therefore MFMA cycles + epilogue VALU cycles, and the epilogue is written to use as
// few vector instructions as possible.

// exp_nonpos / log_sum (devmath.h): table-driven exp and log, 10 fp64 instructions each.

// ------------------------------------------------------------------------------------------
// hot kernel: diagonal-covariance GMM bank on the fp64 matrix cores
// ------------------------------------------------------------------------------------------
// Block = NWV waves; wave w owns state group sg = NWV*y + w (4 states, Q*4 = M mixtures each)
// and keeps its coefficient fragments in registers for the whole launch (Q*KS doubles per
// lane).  The block walks a chunk of 16-frame tiles, TPS tiles per barrier interval: the
// tiles' feature matrices F[k][frame] (x^2 | x | 1) are built once in LDS (double buffered)
// and read as the MFMA B operand by all waves.  blockIdx is decoded XCD-aware: the GY blocks
// that share a frame chunk sit on one XCD (b and b+8 share an XCD), so the obs tile is fetched
// from HBM once and served to the other slabs from that XCD's L2.
template <int Q, int KS, int NWV, int TPS, bool WRITE_BM, int ABL = 0>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(2, 2))) void emit_diag_mfma(
    const double* __restrict__ obs, const double* __restrict__ wpack, const double* __restrict__ logw,
    const double* __restrict__ centre, double* __restrict__ b, double* __restrict__ bm, long F, int N, int D, int NG,
    int GY, long ntiles, int tiles_per_block, long nchunks, unsigned* __restrict__ nan_flag, unsigned nan_epoch) {
    constexpr int KP = 4 * KS;
    constexpr int NT = 64 * NWV;
    constexpr int SPB = 4 * NWV;                // states per block
    constexpr int FS = 17;                      // padded row stride of F: conflict-free staging writes
    __shared__ double Fl[2][TPS][KP * FS];
    constexpr int OS = SPB + 1;                 // padded row stride of the staged outputs
    __shared__ double ob[2][TPS][16 * OS];
    __shared__ double etab[kExpTabBig];
    __shared__ double ltab[2 * kLogTab];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    LN_BLOCK_T0;
    exp_table_big_init(etab, tid, NT);
    log_table_init(ltab, tid, NT);
    const long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long i = bid >> 3;
    const int y = (int)(i % GY);
    const long chunk = (i / GY) * 8 + xcd;
    if (chunk >= nchunks) return;
    const long tile0 = chunk * tiles_per_block;
    long tile1 = tile0 + tiles_per_block;
    if (tile1 > ntiles) tile1 = ntiles;
    const int sg = NWV * y + w;
    const bool active = sg < NG;
    const int sgc = active ? sg : 0;
    const int M = 4 * Q;

    // stationary A fragments
    double wr[Q][KS];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wr[q][ks] = wpack[(((size_t)sgc * Q + q) * KS + ks) * 64 + lane];

    // log weights of the lane's state (bm path: the coefficients carry none).  In the LDS, not in registers: with
    // Q = 4, KS = 20 the 32 VGPRs they took made the kernel spill, and a scratch reload waits on vmcnt -- which counts
    // in issue order, so every reload inside the MFMA chain also waited for all the bm stores before it (the write
    // stream ran in series with the arithmetic: 4.57 ms at config 5)
    __shared__ double lws[WRITE_BM ? SPB * 4 * Q : 1];
    const int g = lane >> 4, fr = lane & 15;
    const int my_state = 4 * sgc + g;
    const double* lw = lws + (WRITE_BM ? (4 * w + g) * 4 * Q : 0);
    if (WRITE_BM && fr < 4 * Q) lws[(4 * w + g) * 4 * Q + fr] = logw[(size_t)my_state * M + fr];

    // constant rows of F (k >= 2D): 1 for k == 2D, 0 above
    {
        const int rows = KP - 2 * D;
        for (int e = tid; e < 2 * TPS * rows * 16; e += NT) {
            const int bt = e / (rows * 16), r = e % (rows * 16);
            const int k = 2 * D + r / 16;
            Fl[bt / TPS][bt % TPS][k * FS + (r % 16)] = (k == 2 * D) ? 1.0 : 0.0;
        }
    }

    const int EL = TPS * 16 * D;                        // obs elements per stage
    constexpr int MAXLD = (TPS * 16 * 64 + NT - 1) / NT;   // supports D <= 64
    const long lim = F * (long)D;
    double pre[MAXLD];
    double cen[MAXLD];                                  // centre of this thread's dimension (a stage starts on a frame
    int foff[MAXLD];                                    // boundary, so element e of every stage is dimension e mod D)
#pragma unroll                                          // foff: LDS offset (doubles) of the x^2 slot; -1: none
    for (int r = 0; r < MAXLD; ++r) {
        const int e = tid + r * NT;
        const int f = e / D, d = e - f * D;             // f = frame within the stage
        foff[r] = e < EL ? (f >> 4) * (KP * FS) + d * FS + (f & 15) : -1;
        cen[r] = centre[d];
    }
    auto load_stage = [&](long tile) {
        const long base = tile * 16 * (long)D;
#pragma unroll
        for (int r = 0; r < MAXLD; ++r) {
            long ge = base + tid + r * NT;
            ge = ge < lim ? ge : lim - 1;               // clamped, always a valid address
            pre[r] = obs[ge];
        }
    };
    // lanes that staged a NaN or an infinity (a compare into the scalar unit per element): the reference turns such a
    // frame's densities into NaN, this arithmetic does not always (nanfix.hip) -- raise the flag and let that pass decide
    unsigned long long nonfinite = 0;
    auto store_stage = [&](int buf) {
        double* Fd = &Fl[buf][0][0];
        const int xo = D * FS;
#pragma unroll
        for (int r = 0; r < MAXLD; ++r) {
            nonfinite |= __ballot(!__builtin_isfinite(pre[r]));
            if (foff[r] >= 0) {
                // centred: the coefficients are those of (x - centre).  (Turning a non-finite feature into NaN right here
                // -- every density of the frame is NaN by itself then, no pass afterwards -- costs two selects per staged
                // element: measured +0.02..0.03 ms on this kernel, three alternations on one box.  The compare above is
                // free; decode does not need the rows, so the 128-state bank leaves them to nan_rows_device.)
                const double x = pre[r] - cen[r];
                Fd[foff[r]] = x * x;
                Fd[foff[r] + xo] = x;
            }
        }
    };
    auto flush_stage = [&](int buf, long tile) {         // TPS*16 frames x SPB states, SPB*8-byte rows
#pragma unroll
        for (int r = 0; r < (TPS * 16 * SPB + NT - 1) / NT; ++r) {
            const int e = tid + r * NT;
            if (e < TPS * 16 * SPB) {
                const int tt = e / (16 * SPB), rem = e % (16 * SPB);
                const int f = rem / SPB, st = rem % SPB;
                const long frame = (tile + tt) * 16 + f;
                const int state = SPB * y + st;
                if (tile + tt < tile1 && frame < F && state < N) b[frame * N + state] = ob[buf][tt][f * OS + st];
            }
        }
    };

    load_stage(tile0);
    store_stage(0);
    drain_vmem();                                        // wr[][] resident: no vmcnt guards inside the loop
    __syncthreads();

    int cur = 0;
    for (long tile = tile0; tile < tile1; tile += TPS, cur ^= 1) {
        const bool has_next = tile + TPS < tile1;
        if (has_next) load_stage(tile + TPS);
        if (tile > tile0) flush_stage(cur ^ 1, tile - TPS);

#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            double4_t acc[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
            const double* Fc = &Fl[cur][tt][0];
            // B operands two k-steps ahead of the MFMAs that use them: read right before its use, every LDS read's
            // latency sat between two MFMA groups (both waves of a SIMD walk the tile in the same phase)
            double bfq[2];
#pragma unroll
            for (int ks = 0; ks < 2 && ks < KS; ++ks) bfq[ks] = Fc[(4 * ks + g) * FS + fr];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const double bf = bfq[ks & 1];
                if (ks + 2 < KS) bfq[ks & 1] = Fc[(4 * (ks + 2) + g) * FS + fr];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    if (ABL == 2) acc[q][ks & 3] += bf * wr[q][ks];      // ablation: no matrix op
                    else acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q][ks], bf, acc[q], 0, 0, 0);
                }
            }
            if (ABL == 1) {                                              // ablation: no log-sum-exp
                double sum = 0.0;
#pragma unroll
                for (int q = 0; q < Q; ++q) sum += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
                ob[cur][tt][fr * OS + 4 * w + g] = sum;
                continue;
            }

            if (WRITE_BM) {     // coefficients carry no log w here: bm is the raw component log-density
                const long frame = (tile + tt) * 16 + fr;
                if (active && tile + tt < tile1 && frame < F) {
                    // a lane owns the 4 Q mixtures of its (frame, state): 32-byte vector stores (M is a multiple of 4 on
                    // this path and bm is 256-byte aligned) instead of one 8-byte request per value
                    double4_t* dst = reinterpret_cast<double4_t*>(bm + ((size_t)frame * N + my_state) * M);
#pragma unroll
                    for (int q = 0; q < Q; ++q) dst[q] = acc[q];
                }
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[q][r] += lw[q * 4 + r];
            }
            // in-lane M-way log-sum-exp: this lane holds c[q][r] = mixture 4q+r of state my_state
            double mx = acc[0][0];
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (q + r > 0) mx = vmax_f64(mx, acc[q][r]);
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s += exp_nonpos_big2(acc[q][r] - mx, etab);
            }
            // mx = -inf: every term is NaN -> clamped -> s = 0 -> log gives -inf; mx = +inf/NaN propagate
            ob[cur][tt][fr * OS + 4 * w + g] = (mx == LN_NEG_INF) ? mx : mx + log_sum(s, ltab);

        }

        // With bm stores in flight the prefetched features are consumed as late as possible: vmcnt counts in issue
        // order, so a wait for those loads hoisted into the next MFMA chain (where the scheduler likes to put it) also
        // waits for the bm stores issued before them.  (Consuming them right after the MFMA chain, ahead of this
        // tile's stores, was measured too: slower -- the previous tile's stores are still in flight there.)
        if (WRITE_BM) __builtin_amdgcn_sched_barrier(0);
        if (has_next) store_stage(cur ^ 1);
        __syncthreads();
    }
    // flush the last stage
    if (tile1 > tile0) {
        const long nst = (tile1 - tile0 + TPS - 1) / TPS;
        flush_stage((int)((nst - 1) & 1), tile0 + (nst - 1) * TPS);
    }
    if (nonfinite != 0 && lane == 0) *nan_flag = nan_epoch;
    LN_BLOCK_T1;
}
LN_BLOCK_TIMES_EXPORT(lnasr_dbg_block_times_gmm)

// ------------------------------------------------------------------------------------------
// single GMM (N = 1, M <= 16): HBM-bound row of the path (gmm.py:90-100 over a batch)
// ------------------------------------------------------------------------------------------
// 320 B of traffic per frame against 20 MFMAs: the kernel is laid out for bytes in flight, not for FLOPs.
// Every wave is independent (no block barriers) and walks 16-frame tiles with the next tile's observations
// already requested; the 16-component x 16-frame MFMA chain is followed by a reduction of the 16 mixtures
// across the wave's four 16-lane rows.
//
// No LDS staging.  With the features ordered as K = [x^2 block | x block], each block DP = 4*ceil((D+1)/4) wide
// and the constant 1 sitting in the spare slot D of the x^2 block, the MFMA B operand of k-step ks is, for lane
// (g = lane >> 4, frame fr = lane & 15), simply x[fr][4 j + g] (or its square, or 1): every lane loads its own
// DP/4 values of its own frame straight from global memory (32-byte pieces of 16 rows per instruction: the ten
// instructions of a tile hit the same 39 lines in L1), squares them in registers, and nothing goes through LDS.
//   wdirect[ks*64 + lane]: component m = lane & 15, slot k = 4 ks + (lane >> 4)
__global__ void pack_single_direct_kernel(const double* __restrict__ mu, const double* __restrict__ ivar,
                                          const double* __restrict__ gconst, double* __restrict__ wdirect, int M, int D,
                                          int KS) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KS * 64) return;
    const int DP = 2 * KS;
    const int lane = idx & 63, ks = idx >> 6;
    const int m = lane & 15, k = 4 * ks + (lane >> 4);
    const int d = k < DP ? k : k - DP;
    double val = 0.0;
    if (m < M) {
        if (k < DP) {
            if (d < D) {
                val = -0.5 * ivar[(size_t)m * D + d];
            } else if (d == D) {
                double sq = 0.0;
                for (int e = 0; e < D; ++e) sq += mu[(size_t)m * D + e] * mu[(size_t)m * D + e] * ivar[(size_t)m * D + e];
                val = gconst[m] - 0.5 * sq;
            }
        } else if (d < D) {
            val = mu[(size_t)m * D + d] * ivar[(size_t)m * D + d];
        }
    } else if (k < DP && d == D) {
        val = LN_NEG_INF;                      // padding component: contributes exp(-inf) = 0
    }
    wdirect[idx] = val;
}

template <int KS, bool WRITE_BM, bool CEN>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void emit_single_direct(const double* __restrict__ obs,
                                                          const double* __restrict__ wdirect,
                                                          const double* __restrict__ logw,
                                                          const double* __restrict__ centre, double* __restrict__ b,
                                                          double* __restrict__ bm, long F, int M, int D, long ntiles,
                                                          unsigned* __restrict__ nan_flag, unsigned nan_epoch) {
    constexpr int NL = KS / 2;                 // loads per lane and tile
    constexpr unsigned kTick = 2;              // tiles per draw (1: 0.079 ms, the draw then sits in front of every prefetch; 2: 0.074;
                                               // round 4, another box: 2: 0.078, 4: 0.082, 8: 0.081)
    __shared__ double etab[kExpTab];
    __shared__ double ltab[2 * kLogTab];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ unsigned ticket;                // next undrawn tile of the block's range (relative)
    for (int e = tid; e < kExpTab; e += 1024) etab[e] = exp2((double)e * (1.0 / kExpTab));
    log_table_init(ltab, tid, 1024);
    if (tid == 0) ticket = 16 * kTick;         // every wave's first chunk is its own
    __syncthreads();                           // tables ready; from here on waves never synchronise
    double wr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wr[ks] = wdirect[ks * 64 + lane];
    const int g = lane >> 4, fr = lane & 15;
    double lw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lw[r] = (g + 4 * r < M) ? logw[g + 4 * r] : 0.0;
    const long lim = F * (long)D;
    // slot 4 j + g of this lane: a real dimension, the constant (== D), or padding (> D).  The slots from D on lie in the
    // memory that FOLLOWS the frame; their coefficients are 0, but 0 * NaN is NaN on the matrix cores (a bad value in a later
    // frame would poison this one: seed 312 of the bad-feature sweep), so they are not read at all: their lanes aim the load
    // past the buffer's end, where the hardware returns 0.  Only the last NL - JT loads of a lane can hold such a slot (the
    // model upload picks the smallest KS that takes D: D >= 4 JT), so only those carry a per-lane offset.
    constexpr int JT = KS == 8 ? 0 : NL - 3;
    constexpr int kPastEnd = 0x7FFFFFF8;       // byte offset beyond any num_records
    bool is_one[NL - JT];
    int tail_off[NL - JT];
#pragma unroll
    for (int j = JT; j < NL; ++j) {
        is_one[j - JT] = (4 * j + g) == D;
        tail_off[j - JT] = (4 * j + g) < D ? (fr * D + g) * 8 + 32 * j : kPastEnd;
    }
    double cen[CEN ? NL : 1];                  // CEN (model re-centred, Model::centred): centre of the lane's dimensions;
#pragma unroll                                 // a model conditioned about the origin pays neither registers nor subtractions
    for (int j = 0; j < NL; ++j)
        if (CEN) cen[j] = (4 * j + g) < D ? centre[4 * j + g] : 0.0;
    double pre[NL];
    auto load_tile = [&](long tile) {
        const long e0 = tile * 16 * (long)D;
        long left = (lim - e0) * 8;
        left = left < 0 ? 0 : (left > 0x7FFFFFFFL ? 0x7FFFFFFFL : left);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(obs + e0), 0, (int)left, 0x00020000);
        const int vo = (fr * D + g) * 8;
#pragma unroll
        for (int j = 0; j < NL; ++j)
            pre[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, j < JT ? vo + 32 * j : tail_off[j < JT ? 0 : j - JT], 0, 0));
    };
    // A block (16 waves, four per SIMD: one block per CU) owns a contiguous range of tiles and its waves draw them from a
    // ticket counter in the LDS, kTick at a time.  A static stride gives every wave the same number of tiles, but the
    // four waves of a SIMD are not served alike -- the oldest issues whenever it can -- so the youngest ones set the
    // kernel's duration while the others have long finished (the MFCC frame kernel showed it: DESIGN 4.6).  (A
    // device-wide counter was tried first: 32,000 atomics on one address serialise in the L2 -- 0.47 ms against 0.08.)
    const long per_block = (ntiles + gridDim.x - 1) / gridDim.x;
    const long tbeg = (long)blockIdx.x * per_block, tend = tbeg + per_block < ntiles ? tbeg + per_block : ntiles;
    auto draw = [&]() -> long {                    // first tile of the next chunk (wave-uniform); >= tend: none left
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&ticket, kTick);
        return tbeg + (long)__builtin_amdgcn_readfirstlane(t);
    };
    drain_vmem();                              // wr[] resident: no vmcnt guards on it inside the loop
    unsigned long long nonfinite = 0;
    long tile = tbeg + (long)w * kTick;
    unsigned left = kTick;
    load_tile(tile < tend ? tile : ntiles);
    while (tile < tend) {
        long nxt = tile + 1;
        if (--left == 0) {
            nxt = draw();
            left = kTick;
        }
        if (nxt >= tend) nxt = ntiles;             // nothing left: the prefetch below then reads nothing
        double bx[NL], bq[NL];
        unsigned long long bad_lanes = 0;          // lanes of this tile that hold a NaN or an infinite feature
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            bad_lanes |= __ballot(!__builtin_isfinite(pre[j]));     // (slots past D read 0)
            bx[j] = CEN ? pre[j] - cen[CEN ? j : 0] : pre[j];
            bq[j] = (j >= JT && is_one[j < JT ? 0 : j - JT]) ? 1.0 : bx[j] * bx[j];
        }
        load_tile(nxt);
        double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < NL; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[j], bq[j], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NL; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[NL + j], bx[j], acc, 0, 0, 0);
        const long frame = tile * 16 + fr;
        bool bad_frame = false;
        if (bad_lanes != 0) {                      // wave-uniform, and never taken on clean data
            // a NaN OR an infinite feature: every density of its frame is NaN in the reference (gmm.py:63-68: inf * 0 against
            // the inverse's zero off-diagonals); this arithmetic would give -inf or NaN depending on the signs an infinity
            // meets, so the frame's outputs are overwritten -- here, not in a pass afterwards, which this 0.09 ms call would feel
            nonfinite |= bad_lanes;
            const unsigned long long fm = bad_lanes | (bad_lanes >> 16) | (bad_lanes >> 32) | (bad_lanes >> 48);
            bad_frame = (fm >> fr) & 1ull;         // the frame's four lanes sit 16 apart
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = bad_frame ? __builtin_nan("") : acc[r];
        }
        if (WRITE_BM && frame < F) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (g + 4 * r < M) bm[(size_t)frame * M + g + 4 * r] = acc[r];
        }
        double c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = acc[r] + lw[r];
        double mx = vmax_f64(vmax_f64(c[0], c[1]), vmax_f64(c[2], c[3]));
        mx = fmax(mx, shfl_xor_f64(mx, 16));
        mx = fmax(mx, shfl_xor_f64(mx, 32));
        double sacc = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc += exp_nonpos(c[r] - mx, etab);
        sacc += shfl_xor_f64(sacc, 16);
        sacc += shfl_xor_f64(sacc, 32);
        if (g == 0 && frame < F) b[frame] = bad_frame ? __builtin_nan("") : ((mx == LN_NEG_INF) ? mx : mx + log_sum(sacc, ltab));
        tile = nxt;
    }
    if (nonfinite != 0 && lane == 0) *nan_flag = nan_epoch;
}

// ------------------------------------------------------------------------------------------
// generic VALU kernels (any N, M, D; fp64 arithmetic, TS storage)
// ------------------------------------------------------------------------------------------
// block = 256 threads: 64 frames (lanes) x 4 waves; wave w walks states w, w+4, ...
template <typename TS>
__global__ void emit_diag_generic(const double* __restrict__ obs, const double* __restrict__ mu,
                                  const double* __restrict__ ivar, const double* __restrict__ gconst,
                                  const double* __restrict__ logw, TS* __restrict__ b, TS* __restrict__ bm,
                                  long F, int N, int M, int D, int fw) {
    // fw = 1: the 4 waves share one 64-frame tile and split the states (N >= 4);
    // fw = 4: every wave scores all states of its own 64-frame tile (single GMM / tiny N)
    extern __shared__ double xs_all[];        // [fw][D][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: parameter reads become s_loads
    const long fb = (long)blockIdx.x * 64 * fw;
    for (int e = tid; e < 64 * fw * D; e += blockDim.x) {
        int f = e / D, d = e % D;
        long fr = fb + f;
        xs_all[((f >> 6) * D + d) * 64 + (f & 63)] = fr < F ? obs[fr * D + d] : 0.0;
    }
    __syncthreads();
    const double* xs = xs_all + (size_t)(w % fw) * D * 64;
    const long frame = fb + (long)(w % fw) * 64 + lane;
    // blockIdx.y: a slab of states (the grid is frame tiles x state slabs -- with frame tiles alone a 1000-frame utterance
    // gave 16 blocks to 256 CUs: round 4, `train` at the metric shape spent 62 ms here)
    const int slab = (N + gridDim.y - 1) / gridDim.y;
    const int n_lo = blockIdx.y * slab, n_hi = n_lo + slab < N ? n_lo + slab : N;
    for (int n = n_lo + w / fw; n < n_hi; n += 4 / fw) {
        Lse acc;
        for (int m = 0; m < M; ++m) {
            const size_t c = (size_t)n * M + m;
            const double* mc = mu + c * D;
            const double* ic = ivar + c * D;
            double q = 0.0;
            for (int d = 0; d < D; ++d) {
                double xm = xs[d * 64 + lane] - mc[d];
                q += (xm * ic[d]) * xm;
            }
            double val = gconst[c] - 0.5 * q;
            if (bm && frame < F) st<TS>(bm, ((size_t)frame * N + n) * M + m, val);
            acc.add(logw[c] + rnd<TS>(val));
        }
        if (frame < F) st<TS>(b, (size_t)frame * N + n, acc.value());
    }
}

template <typename TS>
__global__ void emit_full_generic(const double* __restrict__ obs, const double* __restrict__ mu,
                                  const double* __restrict__ prec, const double* __restrict__ gconst,
                                  const double* __restrict__ logw, TS* __restrict__ b, TS* __restrict__ bm,
                                  long F, int N, int M, int D, int fw) {
    // fw = 1: the 4 waves share one 64-frame tile and split the states (N >= 4);
    // fw = 4: every wave scores all states of its own 64-frame tile (single GMM / tiny N)
    extern __shared__ double xs_all[];        // [fw][D][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long fb = (long)blockIdx.x * 64 * fw;
    for (int e = tid; e < 64 * fw * D; e += blockDim.x) {
        int f = e / D, d = e % D;
        long fr = fb + f;
        xs_all[((f >> 6) * D + d) * 64 + (f & 63)] = fr < F ? obs[fr * D + d] : 0.0;
    }
    __syncthreads();
    const double* xs = xs_all + (size_t)(w % fw) * D * 64;
    const long frame = fb + (long)(w % fw) * 64 + lane;
    // blockIdx.y: a slab of states (the grid is frame tiles x state slabs -- with frame tiles alone a 1000-frame utterance
    // gave 16 blocks to 256 CUs: round 4, `train` at the metric shape spent 62 ms here)
    const int slab = (N + gridDim.y - 1) / gridDim.y;
    const int n_lo = blockIdx.y * slab, n_hi = n_lo + slab < N ? n_lo + slab : N;
    for (int n = n_lo + w / fw; n < n_hi; n += 4 / fw) {
        Lse acc;
        for (int m = 0; m < M; ++m) {
            const size_t c = (size_t)n * M + m;
            const double* mc = mu + c * D;
            const double* pc = prec + c * D * D;
            double q = 0.0;
            for (int d = 0; d < D; ++d) {
                double t = 0.0;                                  // ((x-mu) P)[d]
                for (int e = 0; e < D; ++e) t += (xs[e * 64 + lane] - mc[e]) * pc[(size_t)e * D + d];
                q += t * (xs[d * 64 + lane] - mc[d]);
            }
            double val = gconst[c] - 0.5 * q;
            if (bm && frame < F) st<TS>(bm, ((size_t)frame * N + n) * M + m, val);
            acc.add(logw[c] + rnd<TS>(val));
        }
        if (frame < F) st<TS>(b, (size_t)frame * N + n, acc.value());
    }
}

// `precision=np.float32` emission in the reference's ROUNDING ORDER (gmmhmm.py:64-67): bm is stored into a float32
// array first, then b = lse_m(w + bm) is formed in float64 from those ROUNDED values and rounded on its way into the
// float32 `b`.  The fp64 kernels above produce bm in double; this pass rounds it, writes the float bm (if asked for)
// and re-forms b from the rounded values: b and bm come out bit-equal to the reference's float32 arrays wherever the
// fp64 density itself does not sit within ~1e-14 of a float32 rounding boundary (golden gmmhmm_f32_mid: everywhere).
// One thread per (frame, state); bm64 [F, N, M].
__global__ void round32_lse_kernel(const double* __restrict__ bm64, const double* __restrict__ logw, size_t cells, int N,
                                   int M, float* __restrict__ b32, float* __restrict__ bm32) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    const int n = (int)(idx % N);
    const double* src = bm64 + idx * M;
    const double* lw = logw + (size_t)n * M;
    Lse acc;
    for (int m = 0; m < M; ++m) {
        const float r = (float)src[m];
        if (bm32) bm32[idx * M + m] = r;
        acc.add(lw[m] + (double)r);
    }
    b32[idx] = (float)acc.value();
}

// free functions of gmm.py --------------------------------------------------------------
__global__ void gauss1d_kernel(const double* __restrict__ x, size_t L, double mu, double s2, int want_log,
                               double* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    double d = x[i] - mu;
    if (want_log)
        out[i] = -0.5 * (kLog2Pi + log(s2) + d * d / s2);                        // gmm.py:30
    else
        out[i] = 1.0 / sqrt(2.0 * M_PI * s2) * exp(-(d * d) / (2.0 * s2));       // gmm.py:26
}

// quad[m][l] for M components; linear pdf or log-pdf per component into comp[m*L + l]
__global__ void gauss_multi_kernel(const double* __restrict__ x, size_t L, int D, int M,
                                   const double* __restrict__ mu, const double* __restrict__ prec,
                                   const double* __restrict__ gconst, int want_log, double* __restrict__ comp) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const double* xl = x + l * D;
    for (int m = 0; m < M; ++m) {
        const double* mc = mu + (size_t)m * D;
        const double* pc = prec + (size_t)m * D * D;
        double q = 0.0;
        for (int d = 0; d < D; ++d) {
            double t = 0.0;
            for (int e = 0; e < D; ++e) t += (xl[e] - mc[e]) * pc[(size_t)e * D + d];
            q += t * (xl[d] - mc[d]);
        }
        if (want_log) {
            comp[(size_t)m * L + l] = gconst[m] - 0.5 * q;                         // gmm.py:65-68
        } else {
            // r = 1 / ((2 pi)^(D/2) sqrt(det)) = exp(gconst)                      // gmm.py:46
            comp[(size_t)m * L + l] = exp(gconst[m]) * exp(-0.5 * q);
        }
    }
}

__global__ void mixture_combine_kernel(const double* __restrict__ comp, const double* __restrict__ c, int M,
                                       size_t L, int want_log, double* __restrict__ out) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    if (want_log) {                                                                // gmm.py:100
        Lse acc;
        for (int m = 0; m < M; ++m) acc.add(c[m] + comp[(size_t)m * L + l]);
        out[l] = acc.value();
    } else {                                                                       // gmm.py:88
        double s = 0.0;
        for (int m = 0; m < M; ++m) s += c[m] * comp[(size_t)m * L + l];
        out[l] = s;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// padded copy of a diagonal model: [N,M] -> [Np,Mp]; padding components get weight -inf (and neutral parameters)
__global__ void pad_model_kernel(const double* __restrict__ mu, const double* __restrict__ ivar,
                                 const double* __restrict__ gconst, const double* __restrict__ logw, int N, int M, int D,
                                 int Np, int Mp, double* __restrict__ mu_p, double* __restrict__ ivar_p,
                                 double* __restrict__ gconst_p, double* __restrict__ logw_p) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Np * Mp) return;
    const int n = (int)(idx / Mp), m = (int)(idx % Mp);
    const bool real = n < N && m < M;
    const size_t c = (size_t)n * M + m;
    gconst_p[idx] = real ? gconst[c] : 0.0;
    logw_p[idx] = real ? logw[c] : LN_NEG_INF;
    for (int d = 0; d < D; ++d) {
        mu_p[idx * D + d] = real ? mu[c * D + d] : 0.0;
        ivar_p[idx * D + d] = real ? ivar[c * D + d] : 0.0;
    }
}

static bool mfma_shape_supported(int N, int M, int D, int* Q, int* KS) {
    if (N % 4 != 0 || M % 4 != 0) return false;
    int q = M / 4, ks = (2 * D + 1 + 3) / 4;
    if (!(q == 1 || q == 2 || q == 4)) return false;
    ks = ks <= 7 ? 7 : (ks <= 14 ? 14 : (ks <= 20 ? 20 : (ks <= 32 ? 32 : 0)));   // K rows past 2D+1: zero coefficients
    if (ks == 0) return false;                                    // D <= 63
    if (ks == 32 && q > 2) return false;                          // 40 <= D <= 63: at most 8 mixtures per tile (registers)
    *Q = q;
    *KS = ks;
    return true;
}

int upload_hmm(lnasr_ctx* ctx, int N, const double* logA, const double* logpi) {
    Model& md = ctx->model;
    LN_REQUIRE(N >= 1 && N <= 4096, "N=%d out of range [1,4096]", N);
    LN_REQUIRE(logA && logpi, "null model pointer");
    size_t nn = (size_t)N * N * 8;
    LN_TRY(md.logA.reserve(nn));
    LN_TRY(md.logAT.reserve(nn));
    LN_TRY(md.expA.reserve(nn));
    LN_TRY(md.expAT.reserve(nn));
    LN_TRY(md.logpi.reserve((size_t)N * 8));
    hipMemcpyKind kA = is_device_ptr(logA) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipMemcpyKind kP = is_device_ptr(logpi) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    LN_HIP(hipMemcpyAsync(md.logA.p, logA, nn, kA, ctx->stream));
    LN_HIP(hipMemcpyAsync(md.logpi.p, logpi, (size_t)N * 8, kP, ctx->stream));
    int total = N * N;
    LN_LAUNCH(ctx, "transpose_exp", transpose_exp_kernel, dim3((total + 255) / 256), dim3(256), 0,
              md.logA.as<double>(), md.logAT.as<double>(), md.expA.as<double>(), md.expAT.as<double>(), N);
    LN_HIP(hipStreamSynchronize(ctx->stream));   // host buffers may go away
    md.N = N;
    md.has_hmm = true;
    return 0;
}

static int copy_in(lnasr_ctx* ctx, DevBuf& dst, const void* src, size_t bytes) {
    LN_REQUIRE(src, "null model pointer");
    LN_TRY(dst.reserve(bytes));
    hipMemcpyKind k = is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    LN_HIP(hipMemcpyAsync(dst.p, src, bytes, k, ctx->stream));
    return 0;
}

// LNASR_KAPPA_MAX overrides the conditioning gate (tests: show what the expanded form does beyond it)
static double kappa_max() { return sw().kappa_max; }

// md.centre, md.mu_c, md.kappa / direct_form; prec == nullptr for a diagonal model.  A model that is well conditioned
// about the ORIGIN keeps centre = 0 (md.centred = false: kernels that would pay for the subtraction skip it); only a
// model that is not is re-centred on the mean of its means.
static int centre_and_kappa(lnasr_ctx* ctx, const double* prec) {
    Model& md = ctx->model;
    const int NM = md.N * md.M, D = md.D;
    LN_TRY(md.centre.reserve((size_t)D * 8));
    LN_TRY(md.mu_c.reserve((size_t)NM * D * 8));
    LN_TRY(md.flags.reserve(256));
    unsigned long long* kbits = reinterpret_cast<unsigned long long*>(md.flags.as<char>() + 8);
    for (int pass = 0; pass < 2; ++pass) {
        LN_HIP(hipMemsetAsync(kbits, 0, 8, ctx->stream));
        if (pass == 0)
            LN_HIP(hipMemsetAsync(md.centre.p, 0, (size_t)D * 8, ctx->stream));
        else
            LN_LAUNCH(ctx, "centre", centre_kernel, dim3(D), dim3(256), 0, md.mu.as<double>(), NM, D, md.centre.as<double>());
        LN_LAUNCH(ctx, "centre_mu", centre_mu_kernel, dim3((NM + 3) / 4), dim3(256), 0, md.mu.as<double>(),
                  (const double*)md.centre.as<double>(), prec ? nullptr : (const double*)md.ivar.as<double>(), prec, NM, D,
                  md.mu_c.as<double>(), kbits);
        unsigned long long kb = 0;
        LN_HIP(hipMemcpyAsync(&kb, kbits, 8, hipMemcpyDeviceToHost, ctx->stream));
        LN_HIP(hipStreamSynchronize(ctx->stream));
        md.kappa = __builtin_bit_cast(double, kb);
        md.direct_form = !(md.kappa <= kappa_max());
        md.centred = pass == 1;
        if (!md.direct_form) break;
    }
    return 0;
}

static int model_upload_impl(lnasr_ctx* ctx, int N, int M, int D, const double* logA, const double* logpi,
                             const double* logw, const double* mu, const double* sigma, int sigma_kind,
                             bool allow_diag_detect) {
    LN_REQUIRE(ctx, "null ctx");
    LN_REQUIRE(M >= 1 && D >= 1 && M <= 4096 && D <= 1024, "M=%d D=%d out of range", M, D);
    LN_REQUIRE(sigma_kind == 0 || sigma_kind == 1, "sigma_kind must be 0 (full) or 1 (diag)");
    LN_REQUIRE(sigma, "null sigma");
    Model& md = ctx->model;
    md.has_gmm = false;
    ++md.upload_serial;
    LN_TRY(upload_hmm(ctx, N, logA, logpi));
    const size_t NM = (size_t)N * M, NMD = NM * D;
    LN_TRY(copy_in(ctx, md.logw, logw, NM * 8));
    LN_TRY(copy_in(ctx, md.mu, mu, NMD * 8));
    LN_TRY(md.gconst.reserve(NM * 8));
    bool diag = sigma_kind == 1;
    if (sigma_kind == 1) {
        LN_TRY(copy_in(ctx, md.var, sigma, NMD * 8));
    } else {
        LN_REQUIRE(D <= 88, "full covariance supports D <= 88 (got %d)", D);
        LN_TRY(copy_in(ctx, md.sigma_raw, sigma, NMD * D * 8));
        if (allow_diag_detect) {
            LN_TRY(md.flags.reserve(256));
            LN_HIP(hipMemsetAsync(md.flags.p, 0, 4, ctx->stream));
            LN_LAUNCH(ctx, "detect_offdiag", detect_offdiag_kernel, dim3((unsigned)(NM < 256 ? NM : 256)), dim3(256), 0,
                      md.sigma_raw.as<double>(), (int)NM, D, md.flags.as<int>());
            int flag = 0;
            LN_HIP(hipMemcpyAsync(&flag, md.flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            LN_HIP(hipStreamSynchronize(ctx->stream));
            diag = flag == 0;
        }
        if (diag) {
            LN_TRY(md.var.reserve(NMD * 8));
            LN_LAUNCH(ctx, "extract_diag", extract_diag_kernel, dim3((unsigned)((NMD + 255) / 256)), dim3(256), 0,
                      md.sigma_raw.as<double>(), md.var.as<double>(), NMD, D);
        }
    }
    md.N = N;
    md.M = M;
    md.D = D;
    md.diag = diag;
    md.wpack_Q = md.wpack_KS = 0;
    md.i8_NG = 0;
    md.i8_rho = 0.0;
    md.fi8_nk = 0;
    md.fi8_rho = 0.0;
    md.pad_N = md.pad_M = 0;
    md.wfull_Q = 0;
    if (diag) {
        LN_TRY(md.ivar.reserve(NMD * 8));
        LN_TRY(md.flags.reserve(256));
        LN_HIP(hipMemsetAsync(md.flags.p, 0, 16, ctx->stream));
        LN_LAUNCH(ctx, "prep_diag", prep_diag_kernel, dim3((unsigned)((NM + 127) / 128)), dim3(128), 0,
                  md.var.as<double>(), md.ivar.as<double>(), md.gconst.as<double>(), (int)NM, D, md.flags.as<int>());
        unsigned long long fl[2] = {0, 0};
        LN_HIP(hipMemcpyAsync(fl, md.flags.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        LN_HIP(hipStreamSynchronize(ctx->stream));
        LN_TRY(centre_and_kappa(ctx, nullptr));
        // a singular / non-finite variance, or means too many standard deviations apart for the expanded quadratic:
        // the direct-form kernels score this model
        const int degenerate = (int)(fl[0] & 0xFFFFFFFFu) | (md.direct_form ? 1 : 0);
        int Q = 0, KS = 0;
        md.wdirect_KS = 0;
        if (!degenerate && N == 1 && M <= 16 && D <= 64) {
            int ksd = 2 * ((D + 1 + 3) / 4);                 // LDS-free layout: K = 2 * 4 ceil((D+1)/4)
            ksd = ksd <= 8 ? 8 : (ksd <= 14 ? 14 : (ksd <= 20 ? 20 : 0));   // wider blocks: zero coefficients
            if (ksd) {
                LN_TRY(md.wdirect.reserve((size_t)ksd * 64 * 8));
                LN_LAUNCH(ctx, "pack_single_direct", pack_single_direct_kernel, dim3((ksd * 64 + 255) / 256), dim3(256), 0,
                          md.mu_c.as<double>(), md.ivar.as<double>(), md.gconst.as<double>(), md.wdirect.as<double>(), M,
                          D, ksd);
                md.wdirect_KS = ksd;
            }
        }
        // The bank kernel takes N a multiple of 4 and 4, 8 or 16 mixtures per (virtual) state; M = 16 k mixtures are
        // scored as k virtual 16-mixture states per state ([N,M,D] is [N k,16,D] in memory).  Any other N / M is
        // padded to the next such shape on a copy of the parameters.
        // (40 <= D <= 63: 128 K rows leave registers for 8 mixtures per tile, so M > 8 runs as virtual 8-mixture states)
        const int mtile = (2 * D + 1 + 3) / 4 > 20 ? 8 : 16;
        const int Mp = M <= 4 ? 4 : (M <= 8 ? 8 : (M + mtile - 1) / mtile * mtile);
        const int Np = (N + 3) / 4 * 4;
        const int virt = Mp > mtile ? Mp / mtile : 1;
        const int Nv = Np * virt, Mv = Mp / virt;
        md.wpack_virt = 1;
        md.pad_N = md.pad_M = 0;
        if (!degenerate && mfma_shape_supported(Nv, Mv, D, &Q, &KS) && ((Np == N && Mp == M) || !sw().no_pad)) {
            // the packers see the CENTRED means: the kernels subtract md.centre from every observation they load
            const double *mu_s = md.mu_c.as<double>(), *iv_s = md.ivar.as<double>(), *gc_s = md.gconst.as<double>(),
                         *lw_s = md.logw.as<double>();
            if (Np != N || Mp != M) {
                const size_t NMp = (size_t)Np * Mp;
                LN_TRY(md.pad_mu.reserve(NMp * D * 8));
                LN_TRY(md.pad_ivar.reserve(NMp * D * 8));
                LN_TRY(md.pad_gconst.reserve(NMp * 8));
                LN_TRY(md.pad_logw.reserve(NMp * 8));
                LN_LAUNCH(ctx, "pad_model", pad_model_kernel, dim3((unsigned)((NMp + 127) / 128)), dim3(128), 0, mu_s, iv_s,
                          gc_s, lw_s, N, M, D, Np, Mp, md.pad_mu.as<double>(), md.pad_ivar.as<double>(),
                          md.pad_gconst.as<double>(), md.pad_logw.as<double>());
                mu_s = md.pad_mu.as<double>();
                iv_s = md.pad_ivar.as<double>();
                gc_s = md.pad_gconst.as<double>();
                lw_s = md.pad_logw.as<double>();
                md.pad_N = Np;
                md.pad_M = Mp;
            }
            size_t total = (size_t)(Nv / 4) * Q * KS * 64;
            LN_TRY(md.wpack.reserve(total * 8));
            LN_TRY(md.wpack_nw.reserve(total * 8));
            LN_LAUNCH(ctx, "pack_wfrag", pack_wfrag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mu_s, iv_s,
                      gc_s, lw_s, md.wpack.as<double>(), Nv, Mv, D, Q, KS, 1);
            LN_LAUNCH(ctx, "pack_wfrag", pack_wfrag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mu_s, iv_s,
                      gc_s, lw_s, md.wpack_nw.as<double>(), Nv, Mv, D, Q, KS, 0);
            md.wpack_Q = Q;
            md.wpack_KS = KS;
            md.wpack_NG = Nv / 4;
            md.wpack_virt = virt;
            md.i8_NG = 0;
            if ((Q == 4 || Q == 2) && (KS == 20 || KS == 14) && i8_emit_shape_ok(Nv, Mv, D) && !sw().no_i8emit) {
                // M = 32 on the model as it is: a wave scores one state's 32 mixtures and joins them itself -- no virtual
                // states, no merge pass (the arrays are the same memory either way: [N, 32, D] = [2 N, 16, D])
                if (virt == 2 && Mv == 16 && Np == N && Mp == M) LN_TRY(pack_i8_model(ctx, mu_s, iv_s, gc_s, lw_s, N, 32));
                else LN_TRY(pack_i8_model(ctx, mu_s, iv_s, gc_s, lw_s, Nv, Mv));
            }
        }
    } else {
        LN_TRY(md.prec.reserve(NMD * D * 8));
        size_t shmem = (size_t)(D * 2 * D + D) * 8;
        const int* only = nullptr;
        if (D <= 40 && !sw().no_reglu) {
            // matrices in registers; a component with a zero / non-finite pivot is flagged and redone by the LDS kernel
            LN_TRY(md.lu_bad.reserve(NM * 4));
            LN_HIP(hipMemsetAsync(md.lu_bad.p, 0, NM * 4, ctx->stream));
#define LN_LUREG(DPv)                                                                                             \
    LN_LAUNCH(ctx, "lu_invert_reg", lu_invert_reg_kernel<DPv>, dim3((unsigned)NM), dim3(64), 0,                     \
              md.sigma_raw.as<double>(), md.prec.as<double>(), md.gconst.as<double>(), D, md.lu_bad.as<int>())
            if (D <= 8) LN_LUREG(8);
            else if (D <= 16) LN_LUREG(16);
            else if (D <= 24) LN_LUREG(24);
            else if (D <= 32) LN_LUREG(32);
            else LN_LUREG(40);
#undef LN_LUREG
            only = md.lu_bad.as<int>();
        }
        LN_LAUNCH(ctx, "lu_invert", lu_invert_kernel, dim3((unsigned)NM), dim3(64), shmem, md.sigma_raw.as<double>(),
                  md.prec.as<double>(), md.gconst.as<double>(), D, only);
        LN_TRY(centre_and_kappa(ctx, md.prec.as<double>()));
        md.wfull_Q = 0;
        md.pad_N = md.pad_M = 0;
        if (!md.direct_form) LN_TRY(pack_full_model(ctx));
    }
    LN_HIP(hipStreamSynchronize(ctx->stream));
    md.has_gmm = true;
    return 0;
}

template <int Q, int KS, int NWV, int TPS>
static int launch_mfma_cfg(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm) {
    Model& md = ctx->model;
    const int N = md.wpack_NG * 4, D = md.D;           // (virtual, padded) states scored by this launch
    const double* lw = md.pad_N ? md.pad_logw.as<double>() : md.logw.as<double>();
    constexpr int SPB = 4 * NWV;
    const int GY = (N + SPB - 1) / SPB;
    const long ntiles = (F + 15) / 16;
    // one resident generation of blocks (2 waves per SIMD); every block walks many tiles so the
    // fragment load amortises
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    long want_chunks = ((long)cus * 8 / NWV) / GY;
    if (want_chunks < 8) want_chunks = 8;
    want_chunks = (want_chunks / 8) * 8;
    long tpb = (ntiles + want_chunks - 1) / want_chunks;
    tpb = ((tpb + TPS - 1) / TPS) * TPS;
    if (tpb < TPS) tpb = TPS;
    const long nchunks = (ntiles + tpb - 1) / tpb;
    const long grid = ((nchunks + 7) / 8) * 8 * GY;
    if (bm) {
        LN_LAUNCH(ctx, "emit_diag_mfma_bm", (emit_diag_mfma<Q, KS, NWV, TPS, true>), dim3((unsigned)grid),
                  dim3(64 * NWV), 0, obs, md.wpack_nw.as<double>(), lw, (const double*)md.centre.as<double>(), b, bm, F, N, D,
                  md.wpack_NG, GY, ntiles, (int)tpb, nchunks, ctx->nan_flag.as<unsigned>(), ctx->nan_epoch);
    } else {
        LN_LAUNCH(ctx, "emit_diag_mfma", (emit_diag_mfma<Q, KS, NWV, TPS, false>), dim3((unsigned)grid),
                  dim3(64 * NWV), 0, obs, md.wpack.as<double>(), lw, (const double*)md.centre.as<double>(), b, bm, F, N, D,
                  md.wpack_NG, GY, ntiles, (int)tpb, nchunks, ctx->nan_flag.as<unsigned>(), ctx->nan_epoch);
    }
    return 0;
}

template <int Q, int KS>
static int launch_mfma(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm) {
#ifdef LNASR_ABLATION_BUILD          // developer builds only: -DLNASR_ABLATION_BUILD + LNASR_EMIT=x|y
    const char v = sw().emit ? sw().emit : 'a';
    if (Q == 4 && KS == 20) {
        if (v == 'x' || v == 'y') {      // ablation kernels (wrong results by design; timing only)
            Model& md = ctx->model;
            const int GY = (md.N + 31) / 32;
            const long ntiles = (F + 15) / 16;
            long tpb = (ntiles + 63) / 64;
            const long nchunks = (ntiles + tpb - 1) / tpb;
            const long grid = ((nchunks + 7) / 8) * 8 * GY;
            if (v == 'x')
                LN_LAUNCH(ctx, "emit_ablate_no_lse", (emit_diag_mfma<4, 20, 8, 1, false, 1>), dim3((unsigned)grid), dim3(512), 0,
                          obs, md.wpack.as<double>(), md.logw.as<double>(), (const double*)md.centre.as<double>(), b, bm, F, md.N, md.D, md.wpack_NG, GY, ntiles, (int)tpb, nchunks, ctx->nan_flag.as<unsigned>(), ctx->nan_epoch);
            else
                LN_LAUNCH(ctx, "emit_ablate_no_mfma", (emit_diag_mfma<4, 20, 8, 1, false, 2>), dim3((unsigned)grid), dim3(512), 0,
                          obs, md.wpack.as<double>(), md.logw.as<double>(), (const double*)md.centre.as<double>(), b, bm, F, md.N, md.D, md.wpack_NG, GY, ntiles, (int)tpb, nchunks, ctx->nan_flag.as<unsigned>(), ctx->nan_epoch);
            return 0;
        }
    }
#endif
    // the integer matrix cores (gmm_i8.hip: i8_route has the rule)
    if (((Q == 4 && ctx->model.i8_mix == 16) || (Q == 2 && ctx->model.i8_mix == 8 && !bm)) && (KS == 20 || KS == 14) &&
        i8_route(ctx->model, F))
        return emit_diag_i8_device(ctx, obs, F, b, bm);
    // 8 waves per block (32 states), one 16-frame tile per barrier interval: fastest of the tilings tried
    // (4 or 8 waves x 1, 2, 4 tiles per interval) on MI355X
    return launch_mfma_cfg<Q, KS, 8, 1>(ctx, obs, F, b, bm);
}

// device-resident emission over F = U*T flattened frames; b [F,N], bm [F,N,M] or null
int emit_device(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm) {
    bool scanned = false, rows_done = false;
    LN_TRY(emit_device_raw(ctx, obs, F, dtype, b, bm, &scanned, &rows_done));
    // frames with a NaN or an infinite feature: NaN rows, as the reference's product with the full inverse gives
    if (scanned && rows_done && !nan_rows_rescored(ctx->model)) return 0;
    return nan_rows_device(ctx, obs, F, dtype, b, bm, !scanned);
}

int emit_device_raw(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm, bool* scanned, bool* rows_done) {
    LN_DEVPTR(obs); LN_DEVPTR(b); LN_DEVPTR(bm);
    Model& md = ctx->model;
    bool rows_dummy = false;
    if (!rows_done) rows_done = &rows_dummy;
    *scanned = false;
    *rows_done = false;
    if (!md.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    if (F <= 0) return 0;
    const int N = md.N, M = md.M, D = md.D;
    if (dtype == LNASR_F32 && f32_via_f64()) {
        // the tuned kernels are f64: score in double WITH the per-mixture densities (needed whether or not the caller
        // wants them), then round in the reference's order (round32_lse_kernel)
        const size_t cells = (size_t)F * N;
        LN_TRY(ctx->ws_w1.reserve(cells * 8));
        LN_TRY(ctx->ws_w3.reserve(cells * M * 8));
        LN_TRY(emit_device_raw(ctx, obs, F, LNASR_F64, ctx->ws_w1.p, ctx->ws_w3.p, scanned, nullptr));   // the rounding pass
                                                                                        // below does not carry NaN rows through
        LN_LAUNCH(ctx, "round32_lse", round32_lse_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0,
                  (const double*)ctx->ws_w3.as<double>(), (const double*)md.logw.as<double>(), cells, N, M, (float*)b,
                  (float*)bm);
        return 0;
    }
    if (md.diag && dtype == LNASR_F64 && md.wdirect_KS > 0 && !sw().no_mfma) {
        const long ntiles = (F + 15) / 16;
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        // every wave resident: one 16-wave block per CU (128 VGPRs, 4 KB of LDS); each block walks its own range of
        // tiles, its waves drawing them from a ticket counter in the LDS
        long grid = (ntiles + 31) / 32;
        if (grid > (long)cus) grid = (long)cus;
#define LN_DLAUNCH(ks, wbm, cen, name)                                                                            \
    LN_LAUNCH(ctx, name, (emit_single_direct<ks, wbm, cen>), dim3((unsigned)grid), dim3(1024), 0, obs,            \
              md.wdirect.as<double>(), md.logw.as<double>(), (const double*)md.centre.as<double>(), (double*)b,   \
              (double*)bm, F, M, D, ntiles, ctx->nan_flag.as<unsigned>(), ctx->nan_epoch)
#define LN_DCASE(ks)                                                                                              \
    if (md.wdirect_KS == ks) {                                                                                    \
        LN_REQUIRE(D >= (ks == 8 ? 0 : 4 * (ks / 2 - 3)), "D=%d below what the KS=%d single-GMM kernel assumes", D, ks); \
        *scanned = true;                                                                                          \
        *rows_done = true;                                                                                        \
        if (bm && md.centred) LN_DLAUNCH(ks, true, true, "emit_single_direct_bm");                                \
        else if (bm) LN_DLAUNCH(ks, true, false, "emit_single_direct_bm");                                        \
        else if (md.centred) LN_DLAUNCH(ks, false, true, "emit_single_direct");                                   \
        else LN_DLAUNCH(ks, false, false, "emit_single_direct");                                                  \
        return 0;                                                                                                 \
    }
        LN_DCASE(20) LN_DCASE(14) LN_DCASE(8)
#undef LN_DLAUNCH
#undef LN_DCASE
    }
    if (md.diag && dtype == LNASR_F64 && md.wpack_Q > 0 && !sw().no_mfma) {
        const int Q = md.wpack_Q, KS = md.wpack_KS;
        const int Np = md.pad_N ? md.pad_N : N, Mp = md.pad_M ? md.pad_M : M, virt = md.wpack_virt;
        const bool padded = md.pad_N != 0;
        // M = 32 on the integer matrix cores: b [F, N] and bm [F, N, 32] directly (gmm_i8.hip; same rule as in launch_mfma)
        if (md.i8_mix == 32 && !padded && i8_route(md, F)) {
            LN_TRY(emit_diag_i8_device(ctx, obs, F, (double*)b, (double*)bm));
            *scanned = true;
            return 0;
        }
        double* bdst = (double*)b;
        double* bmdst = (double*)bm;
        if (virt > 1 || padded) {
            LN_TRY(ctx->ws_misc2.reserve((size_t)F * Np * virt * 8));
            bdst = ctx->ws_misc2.as<double>();
        }
        if (bm && padded) {
            LN_TRY(ctx->ws_pad.reserve((size_t)F * Np * Mp * 8));
            bmdst = ctx->ws_pad.as<double>();
        }
        int rc = LNASR_E_UNSUPPORTED;
#define LN_CASE(q, ks) \
    if (Q == q && KS == ks) rc = launch_mfma<q, ks>(ctx, obs, F, bdst, bmdst);
        LN_CASE(4, 20) LN_CASE(2, 20) LN_CASE(1, 20) LN_CASE(4, 14) LN_CASE(2, 14) LN_CASE(1, 14)
        LN_CASE(4, 7) LN_CASE(2, 7) LN_CASE(1, 7) LN_CASE(2, 32) LN_CASE(1, 32)
#undef LN_CASE
        LN_TRY(rc);
        *scanned = true;

        // (Merging the two virtual states of an M = 32 model inside the emission kernels' flush stage instead -- no
        // second pass over b -- was built and measured at config 5 in round 3: the fp64 kernel went from 3.70 to 3.90 ms
        // (at 256 registers the extra code spills into the tile loop) and the fp32 one from 1.85 to 2.20 ms, for 0.27 /
        // 0.25 ms of merge kernel saved: not kept.)
        if (virt > 1 || padded) {
            const size_t cells = (size_t)F * N;
            LN_LAUNCH(ctx, "merge_virtual_states", merge_virtual_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256),
                      0, bdst, cells, N, Np, virt, (double*)b);
        }
        if (bm && padded) {
            const size_t cells = (size_t)F * N * M;
            LN_LAUNCH(ctx, "compact_bm", compact_bm_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0,
                      (const double*)bmdst, cells, N, M, Np, Mp, (double*)bm);
        }
        return 0;
    }
    if (!md.diag && dtype == LNASR_F64) {
        int rc = emit_full_mfma_device(ctx, obs, F, (double*)b, (double*)bm);
        if (rc != 0) return rc < 0 ? rc : 0;
        // refused (kappa beyond the gate, or a shape the fragment packing does not take): the reference's direct form, still on
        // the matrix cores
        rc = emit_full_direct_device(ctx, obs, F, (double*)b, (double*)bm);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    const int fw = N >= 4 ? 1 : 4;
    const unsigned gx = (unsigned)((F + 64 * fw - 1) / (64 * fw));
    // state slabs: enough blocks to fill the chip when the frame tiles alone do not, never fewer than one state per wave
    unsigned gy = 1;
    if (fw == 1) {
        const unsigned want = (4u * (unsigned)(ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256) + gx - 1) / gx;
        const unsigned most = (unsigned)(N / 4 > 0 ? N / 4 : 1);
        gy = want < 1 ? 1 : (want > most ? most : want);
    }
    const dim3 grid(gx, gy);
    const size_t shmem = (size_t)D * 64 * 8 * fw;
    LN_REQUIRE(shmem <= 160 * 1024, "D=%d too large for the generic emission kernel", D);
    if (md.diag) {
        if (dtype == LNASR_F32)
            LN_LAUNCH(ctx, "emit_diag_generic_f32", emit_diag_generic<float>, grid, dim3(256), shmem, obs,
                      md.mu.as<double>(), md.ivar.as<double>(), md.gconst.as<double>(), md.logw.as<double>(),
                      (float*)b, (float*)bm, F, N, M, D, fw);
        else
            LN_LAUNCH(ctx, "emit_diag_generic", emit_diag_generic<double>, grid, dim3(256), shmem, obs,
                      md.mu.as<double>(), md.ivar.as<double>(), md.gconst.as<double>(), md.logw.as<double>(),
                      (double*)b, (double*)bm, F, N, M, D, fw);
    } else {
        if (dtype == LNASR_F32)
            LN_LAUNCH(ctx, "emit_full_generic_f32", emit_full_generic<float>, grid, dim3(256), shmem, obs,
                      md.mu.as<double>(), md.prec.as<double>(), md.gconst.as<double>(), md.logw.as<double>(),
                      (float*)b, (float*)bm, F, N, M, D, fw);
        else
            LN_LAUNCH(ctx, "emit_full_generic", emit_full_generic<double>, grid, dim3(256), shmem, obs,
                      md.mu.as<double>(), md.prec.as<double>(), md.gconst.as<double>(), md.logw.as<double>(),
                      (double*)b, (double*)bm, F, N, M, D, fw);
    }
    return 0;
}

// Upload pipeline of emit_batch.  The host observations reach HBM in `nch` chunks on the side stream while the emission
// kernel already runs on the chunks that have arrived.  Three sources:
//   * page-locked host memory (lnasr_host_alloc / hipHostRegister): one hipMemcpyAsync per chunk, PCIe rate, no thread;
//   * pageable memory (plain numpy): `h2d_threads` host threads copy each chunk -- split between them -- into a slot of
//     the context's pinned staging ring, the thread that completes a chunk enqueues its DMA; a slot is reused once the
//     DMA that read it has finished (its event);
//   * LNASR_NO_PINNED_RING: hipMemcpyAsync straight from pageable memory in a helper thread (the runtime's own
//     staging: ~39-48 GB/s measured), the round-1 path.
namespace {

// (Chunk sizes growing geometrically from 16 utterances -- only the first chunk's transfer is exposed, so make it small --
// were measured in round 3: 8.34 ms end to end against 8.05 with sixteen equal chunks; an emission launch on 16 or 32
// utterances does not fill the chip for long enough to pay for its own ramp.  ONE small first chunk (a quarter of the
// others) and fifteen equal ones: 8.08-8.15 ms against 8.07 -- the exposed part is not the first transfer but the sixteen
// launches' own ramps.)
struct ChunkPlan {
    int U, nch;
    size_t utt_bytes;
    int lo(int c) const { return (int)((long)U * c / nch); }
    size_t off(int c) const { return (size_t)lo(c) * utt_bytes; }
    size_t len(int c) const { return (size_t)(lo(c + 1) - lo(c)) * utt_bytes; }
};

struct UploadState {
    std::mutex mu;
    std::condition_variable cv;
    int enqueued = 0;                  // chunks 0 .. enqueued-1 have their DMA and event on the side stream
    std::vector<char> done;            // per chunk: DMA enqueued (ring path completes chunks out of order)
    hipError_t err = hipSuccess;
    void mark(int c, hipError_t e) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (e != hipSuccess && err == hipSuccess) err = e;
            done[c] = 1;
            while (enqueued < (int)done.size() && done[enqueued]) ++enqueued;
            if (e != hipSuccess) enqueued = (int)done.size();      // release every waiter
        }
        cv.notify_all();
    }
    bool wait_for(int c) {             // false: an upload failed
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return enqueued > c; });
        return err == hipSuccess;
    }
};

}  // namespace

static int ensure_ring(lnasr_ctx* ctx, size_t slot_bytes) {
    if (ctx->ring_cap >= slot_bytes && ctx->ring[0]) return 0;
    LN_HIP(hipStreamSynchronize(ctx->stream2));
    for (int k = 0; k < lnasr_ctx::kRingSlots; ++k) {
        if (ctx->ring[k]) (void)hipHostFree(ctx->ring[k]);
        ctx->ring[k] = nullptr;
    }
    ctx->ring_cap = 0;
    for (int k = 0; k < lnasr_ctx::kRingSlots; ++k) {
        LN_HIP(hipHostMalloc(&ctx->ring[k], slot_bytes, hipHostMallocDefault));
        if (!ctx->ring_ev[k]) LN_HIP(hipEventCreateWithFlags(&ctx->ring_ev[k], hipEventDisableTiming));
    }
    ctx->ring_cap = slot_bytes;
    return 0;
}

int emit_batch(lnasr_ctx* ctx, const double* obs, int U, int T, double* b, double* bm, const double** d_obs_out,
               bool* scanned_or_null) {
    Model& md = ctx->model;
    const size_t utt = (size_t)T * md.D;                       // doubles per utterance
    constexpr int kMinChunk = 64, kMaxChunks = 16;
    int nch = U / kMinChunk;
    nch = nch > kMaxChunks ? kMaxChunks : nch;
    if (is_device_ptr(obs) || nch < 2 || sw().no_pipeline) {
        const void* d_obs;
        LN_TRY(stage_in(ctx, obs, (size_t)U * utt * 8, ctx->ws_in, &d_obs));
        if (d_obs_out) *d_obs_out = (const double*)d_obs;
        if (scanned_or_null)
            return emit_device_raw(ctx, (const double*)d_obs, (long)U * T, LNASR_F64, b, bm, scanned_or_null, nullptr);
        return emit_device(ctx, (const double*)d_obs, (long)U * T, LNASR_F64, b, bm);
    }
    LN_TRY(ctx->ws_in.reserve((size_t)U * utt * 8));
    char* d_obs = ctx->ws_in.as<char>();
    const char* h_obs = reinterpret_cast<const char*>(obs);
    if (d_obs_out) *d_obs_out = ctx->ws_in.as<double>();
    while ((int)ctx->copy_events.size() < nch) {
        hipEvent_t e;
        LN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->copy_events.push_back(e);
    }
    const ChunkPlan plan{U, nch, utt * 8};
    bool all_scanned = true, all_rows_done = true;
    // the side stream must not run ahead of work that still reads the staging buffer
    LN_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    LN_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    auto emit_chunk = [&](int c) -> int {
        const int u0 = plan.lo(c), u1 = plan.lo(c + 1);
        if (hipStreamWaitEvent(ctx->stream, ctx->copy_events[c], 0) != hipSuccess)
            return fail(LNASR_E_HIP, "hipStreamWaitEvent failed in the upload pipeline");
        bool scanned = false, rows_done = false;
        const int rc = emit_device_raw(ctx, ctx->ws_in.as<double>() + (size_t)u0 * utt, (long)(u1 - u0) * T, LNASR_F64,
                                       b + (size_t)u0 * T * md.N, bm ? bm + (size_t)u0 * T * md.N * md.M : nullptr, &scanned,
                                       &rows_done);
        all_scanned = all_scanned && scanned;
        all_rows_done = all_rows_done && rows_done;
        return rc;
    };
    // non-finite frames (nanfix.hip): once for the whole batch, behind the last chunk
    auto fix_rows = [&]() -> int {
        if (scanned_or_null) {
            *scanned_or_null = all_scanned;
            return 0;
        }
        if (all_scanned && all_rows_done && !nan_rows_rescored(md)) return 0;
        return nan_rows_device(ctx, ctx->ws_in.as<double>(), (long)U * T, LNASR_F64, b, bm, !all_scanned);
    };

    if (is_pinned_host_ptr(obs)) {
        // page-locked source: the DMA engine reads it directly
        for (int c = 0; c < nch; ++c) {
            LN_HIP(hipMemcpyAsync(d_obs + plan.off(c), h_obs + plan.off(c), plan.len(c), hipMemcpyHostToDevice, ctx->stream2));
            LN_HIP(hipEventRecord(ctx->copy_events[c], ctx->stream2));
        }
        for (int c = 0; c < nch; ++c) LN_TRY(emit_chunk(c));
        return fix_rows();
    }

    UploadState st;
    st.done.assign(nch, 0);
    std::vector<std::thread> workers;
    // everything the worker lambdas capture by reference lives in THIS scope, which also joins them
    constexpr int K = lnasr_ctx::kRingSlots;
    const int nth = sw().h2d_threads;
    std::vector<std::atomic<int>> pieces(nch);          // pieces[c]: threads that have finished their part of chunk c
    for (auto& p : pieces) p.store(0);
    if (sw().no_pinned_ring) {
        workers.emplace_back([&] {
            (void)hipSetDevice(ctx->device);
            for (int c = 0; c < nch; ++c) {
                hipError_t e = hipMemcpyAsync(d_obs + plan.off(c), h_obs + plan.off(c), plan.len(c), hipMemcpyHostToDevice,
                                              ctx->stream2);
                if (e == hipSuccess) e = hipEventRecord(ctx->copy_events[c], ctx->stream2);
                st.mark(c, e);
                if (e != hipSuccess) return;
            }
        });
    } else {
        size_t slot = 0;
        for (int c = 0; c < nch; ++c) slot = plan.len(c) > slot ? plan.len(c) : slot;
        LN_TRY(ensure_ring(ctx, slot));
        // The previous call returned once its DMAs were ENQUEUED: with device outputs and no host-side wait (decode into
        // a DeviceArray, statistics into device memory) its last copies may still be reading the ring.  A slot is
        // written again only after the copy that last read it has finished.
        for (int s = 0; s < K; ++s)
            if (ctx->ring_busy[s]) {
                LN_HIP(hipEventSynchronize(ctx->ring_ev[s]));
                ctx->ring_busy[s] = false;
            }
        for (int c = 0; c < nch; ++c) ctx->ring_busy[c % K] = true;     // every slot used below ends with a recorded event
        for (int t = 0; t < nth; ++t) {
            workers.emplace_back([&, t] {
                (void)hipSetDevice(ctx->device);
                for (int c = 0; c < nch; ++c) {
                    const int s = c % K;
                    if (c >= K) {
                        // the slot's previous DMA (chunk c - K) must have been enqueued, then finished
                        if (!st.wait_for(c - K)) return;
                        if (hipEventSynchronize(ctx->ring_ev[s]) != hipSuccess) {
                            st.mark(c, hipErrorUnknown);
                            return;
                        }
                    }
                    const size_t len = plan.len(c);
                    const size_t p0 = (len * t / nth) & ~(size_t)63, p1 = t + 1 == nth ? len : (len * (t + 1) / nth) & ~(size_t)63;
                    memcpy((char*)ctx->ring[s] + p0, h_obs + plan.off(c) + p0, p1 - p0);
                    if (pieces[c].fetch_add(1, std::memory_order_acq_rel) + 1 == nth) {
                        hipError_t e = hipMemcpyAsync(d_obs + plan.off(c), ctx->ring[s], len, hipMemcpyHostToDevice, ctx->stream2);
                        if (e == hipSuccess) e = hipEventRecord(ctx->copy_events[c], ctx->stream2);
                        if (e == hipSuccess) e = hipEventRecord(ctx->ring_ev[s], ctx->stream2);
                        st.mark(c, e);
                        if (e != hipSuccess) return;
                    }
                }
            });
        }
    }
    int rc = 0;
    for (int c = 0; c < nch && rc == 0; ++c) {
        if (!st.wait_for(c)) break;
        rc = emit_chunk(c);
    }
    for (auto& w : workers) w.join();
    if (st.err != hipSuccess) return fail(LNASR_E_HIP, "upload pipeline: %s", hipGetErrorString(st.err));
    return rc != 0 ? rc : fix_rows();
}

}  // namespace lnasr

using namespace lnasr;

extern "C" {

int lnasr_model_upload(lnasr_ctx* ctx, int N, int M, int D, const double* logA, const double* logpi,
                       const double* logw, const double* mu, const double* sigma, int sigma_kind) {
    LN_ENTER(ctx);
    return model_upload_impl(ctx, N, M, D, logA, logpi, logw, mu, sigma, sigma_kind, true);
}

int lnasr_model_upload_full(lnasr_ctx* ctx, int N, int M, int D, const double* logA, const double* logpi,
                            const double* logw, const double* mu, const double* sigma) {
    LN_ENTER(ctx);
    return model_upload_impl(ctx, N, M, D, logA, logpi, logw, mu, sigma, 0, false);
}

int lnasr_hmm_upload(lnasr_ctx* ctx, int N, const double* logA, const double* logpi) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    ctx->model.has_gmm = false;
    return upload_hmm(ctx, N, logA, logpi);
}

int lnasr_model_is_diag(lnasr_ctx* ctx, int* out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && out, "null argument");
    if (!ctx->model.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    *out = ctx->model.diag ? 1 : 0;
    return 0;
}

int lnasr_model_conditioning(lnasr_ctx* ctx, double* kappa, int* direct_form) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (!ctx->model.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    if (kappa) *kappa = ctx->model.kappa;
    if (direct_form) *direct_form = ctx->model.direct_form ? 1 : 0;
    return 0;
}

int lnasr_model_range(lnasr_ctx* ctx, double* rho, int* integer_bank) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (!ctx->model.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    const bool full = !ctx->model.diag;
    if (rho) *rho = full ? ctx->model.fi8_rho : ctx->model.i8_rho;
    if (integer_bank) *integer_bank = (full ? full_i8_model_ok(ctx->model) : i8_model_ok(ctx->model)) ? 1 : 0;
    return 0;
}

int lnasr_gmm_logprob(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, int dtype, void* b,
                      void* bm_or_null) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    (void)lens;   // emission is per frame; padded frames are scored too and ignored downstream
    LN_REQUIRE(U >= 0 && T >= 0, "negative batch shape");
    LN_REQUIRE(dtype == LNASR_F64 || dtype == LNASR_F32, "bad dtype %d", dtype);
    Model& md = ctx->model;
    if (!md.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    const long F = (long)U * T;
    if (F == 0) return 0;
    LN_REQUIRE(b, "null b");
    const size_t es = dtype_size(dtype);
    const void* d_obs;
    void *d_b, *d_bm = nullptr;
    LN_TRY(stage_in(ctx, obs, (size_t)F * md.D * 8, ctx->ws_in, &d_obs));
    LN_TRY(stage_out(ctx, b, (size_t)F * md.N * es, ctx->ws_out, &d_b));
    if (bm_or_null) LN_TRY(stage_out(ctx, bm_or_null, (size_t)F * md.N * md.M * es, ctx->ws_out2, &d_bm));
    LN_TRY(emit_device(ctx, (const double*)d_obs, F, dtype, d_b, d_bm));
    LN_TRY(finish_out(ctx, b, d_b, (size_t)F * md.N * es));
    if (bm_or_null) LN_TRY(finish_out(ctx, bm_or_null, d_bm, (size_t)F * md.N * md.M * es));
    return 0;
}

int lnasr_gauss1d(lnasr_ctx* ctx, const double* x, size_t L, double mu, double sigma2, int want_log, double* out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (L == 0) return 0;
    const void* d_x;
    void* d_o;
    LN_TRY(stage_in(ctx, x, L * 8, ctx->ws_in, &d_x));
    LN_TRY(stage_out(ctx, out, L * 8, ctx->ws_out, &d_o));
    LN_LAUNCH(ctx, "gauss1d", gauss1d_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, (const double*)d_x, L,
              mu, sigma2, want_log, (double*)d_o);
    return finish_out(ctx, out, d_o, L * 8);
}

static int gauss_components(lnasr_ctx* ctx, int M, const double* x, size_t L, int D, const double* mu,
                            const double* Sigma, int want_log, const double** d_comp_out) {
    LN_REQUIRE(D >= 1 && D <= 88, "D=%d out of range [1,88]", D);
    const void *d_x, *d_mu, *d_S;
    LN_TRY(stage_in(ctx, x, L * D * 8, ctx->ws_in, &d_x));
    LN_TRY(stage_in(ctx, mu, (size_t)M * D * 8, ctx->ws_in2, &d_mu));
    LN_TRY(stage_in(ctx, Sigma, (size_t)M * D * D * 8, ctx->ws_in3, &d_S));
    LN_TRY(ctx->ws_misc.reserve((size_t)M * D * D * 8 + (size_t)M * 8));
    double* d_P = ctx->ws_misc.as<double>();
    double* d_g = d_P + (size_t)M * D * D;
    LN_LAUNCH(ctx, "lu_invert", lu_invert_kernel, dim3((unsigned)M), dim3(64), (size_t)(D * 2 * D + D) * 8,
              (const double*)d_S, d_P, d_g, D, (const int*)nullptr);
    LN_TRY(ctx->ws_misc2.reserve((size_t)M * L * 8));
    LN_LAUNCH(ctx, "gauss_multi", gauss_multi_kernel, dim3((unsigned)((L + 127) / 128)), dim3(128), 0,
              (const double*)d_x, L, D, M, (const double*)d_mu, d_P, d_g, want_log, ctx->ws_misc2.as<double>());
    *d_comp_out = ctx->ws_misc2.as<double>();
    return 0;
}

int lnasr_gauss_multi(lnasr_ctx* ctx, const double* x, size_t L, int D, const double* mu, const double* Sigma,
                      int want_log, double* out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (L == 0) return 0;
    const double* d_comp;
    LN_TRY(gauss_components(ctx, 1, x, L, D, mu, Sigma, want_log, &d_comp));
    if (is_device_ptr(out)) {
        LN_HIP(hipMemcpyAsync(out, d_comp, L * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    return finish_out(ctx, out, d_comp, L * 8);
}

int lnasr_gauss_mixture(lnasr_ctx* ctx, const double* c, int M, const double* x, size_t L, int D, const double* mu,
                        const double* Sigma, int want_log, double* out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_REQUIRE(M >= 1, "M must be >= 1");
    if (L == 0) return 0;
    const double* d_comp;
    LN_TRY(gauss_components(ctx, M, x, L, D, mu, Sigma, want_log, &d_comp));
    const void* d_c;
    void* d_o;
    LN_TRY(stage_in(ctx, c, (size_t)M * 8, ctx->ws_in4, &d_c));
    LN_TRY(stage_out(ctx, out, L * 8, ctx->ws_out, &d_o));
    LN_LAUNCH(ctx, "mixture_combine", mixture_combine_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, d_comp,
              (const double*)d_c, M, L, want_log, (double*)d_o);
    return finish_out(ctx, out, d_o, L * 8);
}

}  // extern "C"
