// Trellises beyond the register file of a CU (N > 128 streaming kernels, N > 256 frame-parallel kernels):
// Viterbi, forward and backward.  Dispatch thresholds: trellis.h (measured with tools/bign.py).
#include "common.h"
#include "devmath.h"
#include "trellis.h"

namespace lnasr {

// ------------------------------------------------------------------------------------------
// N > 128: the transition matrix no longer fits the register file of a CU.  It is streamed from
// L2 once per frame and shared by UB utterances; thread (column jc, row slice rs) accumulates over
// its slice of rows, slices are combined through LDS and the UB*NP (utterance, column) results are
// finalised by all threads.  Same arithmetic as the register-resident kernels.
// ------------------------------------------------------------------------------------------
template <int UB>
__global__ __launch_bounds__(1024) void viterbi_stream(const double* __restrict__ b, const double* __restrict__ A,
                                                       const double* __restrict__ pi,
                                                       const int32_t* __restrict__ lens, double* __restrict__ v,
                                                       int U, int T, int N, int NP, int RS, int RPS) {
    extern __shared__ double sm[];
    double* Vs = sm;                           // [2][UB][NP]
    double* Ps = sm + 2 * UB * NP;             // [RS][UB][NP]
    const int tid = threadIdx.x, nt = blockDim.x;
    const int jc = tid % NP, rs = tid / NP;
    const int u0 = blockIdx.x * UB;
    int len[UB], maxlen = 0;
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        len[k] = (u0 + k < U) ? lens[u0 + k] : 0;
        maxlen = len[k] > maxlen ? len[k] : maxlen;
    }
    for (int q = tid; q < UB * NP; q += nt) {
        const int k = q / NP, j = q % NP;
        double val = LN_NEG_INF;
        if (j < N && len[k] > 0) {
            const size_t o = ((size_t)(u0 + k) * T) * N + j;
            val = pi[j] + b[o];
            v[o] = val;
        }
        Vs[k * NP + j] = val;
    }
    __syncthreads();
    const int i0 = rs * RPS, i1 = (i0 + RPS < N) ? i0 + RPS : N;
    const int jcl = jc < N ? jc : N - 1;
    for (int t = 1; t < maxlen; ++t) {
        const double* Vc = Vs + ((t - 1) & 1) * UB * NP;
        double* Vn = Vs + (t & 1) * UB * NP;
        if (rs < RS) {
            double acc[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) acc[k] = LN_NEG_INF;
#pragma unroll 8
            for (int i = i0; i < i1; ++i) {
                const double a = A[(size_t)i * N + jcl];
#pragma unroll
                for (int k = 0; k < UB; ++k) acc[k] = fmax(acc[k], Vc[k * NP + i] + a);
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) Ps[(rs * UB + k) * NP + jc] = acc[k];
        }
        __syncthreads();
        for (int q = tid; q < UB * NP; q += nt) {
            const int k = q / NP, j = q % NP;
            double val = Vc[k * NP + j];
            if (j < N && t < len[k]) {
                double m = Ps[k * NP + j];
                for (int r = 1; r < RS; ++r) m = fmax(m, Ps[(r * UB + k) * NP + j]);
                const size_t o = ((size_t)(u0 + k) * T + t) * N + j;
                val = m + b[o];
                v[o] = val;
            }
            Vn[k * NP + j] = val;
        }
        __syncthreads();
    }
}

// lvl[row] = max_j b[row, j]: the LEVEL of every emission row, which the scaled linear-domain recursions below add to their
// lagged shift (fwdbwd.hip, fwdbwd_cols2, has the story: the lagged maximum alone assumes emissions of a steady size, and on
// data a model does not fit every column took the exact fallback -- 98 ms for one 1000-frame utterance at N = 300).  One wave
// per row; for these sizes one more pass over b is noise.
__global__ void row_level_kernel(const double* __restrict__ b, long rows, int N, double* __restrict__ lvl) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    double m = LN_NEG_INF;
    for (int j = lane; j < N; j += 64) m = vmax_f64(m, b[(size_t)row * N + j]);
    m = wave_max_dpp(m);
    if (lane == 0) lvl[row] = m;
}

template <bool BACKWARD, int UB>
__global__ __launch_bounds__(1024) void fwdbwd_stream(const double* __restrict__ b, const double* __restrict__ Aexp,
                                                      const double* __restrict__ Alog, const double* __restrict__ pi,
                                                      const int32_t* __restrict__ lens, double* __restrict__ out,
                                                      double* __restrict__ loglik, int U, int T, int N, int NP, int RS,
                                                      int RPS, const double* __restrict__ lvl, unsigned* __restrict__ fb_count) {
    extern __shared__ double sm[];
    const int NB = NP / 64;                    // 64-column blocks
    double* Es = sm;                           // [2][UB][NP]
    double* Cs = Es + 2 * UB * NP;             // [2][UB][NP]
    double* Ps = Cs + 2 * UB * NP;             // [RS][UB][NP]
    double* Ms = Ps + RS * UB * NP;            // [2][UB][NB]
    double* Ss = Ms + 2 * UB * NB;             // [2][UB]
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
    const int jc = tid % NP, rs = tid / NP;
    const int u0 = blockIdx.x * UB;
    int len[UB], maxlen = 0;
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        len[k] = (u0 + k < U) ? lens[u0 + k] : 0;
        maxlen = len[k] > maxlen ? len[k] : maxlen;
    }
    auto frame_of = [&](int k, int step) { return BACKWARD ? len[k] - 1 - step : step; };
    // step 0 (jobs: q -> (k = q / NP, j = q % NP); a wave never straddles two utterances since NP % 64 == 0)
    for (int q = tid; q < UB * NP; q += nt) {
        const int k = q / NP, j = q % NP;
        double prev = LN_NEG_INF;
        if (j < N && len[k] > 0) {
            const int t0 = frame_of(k, 0);
            const size_t o = ((size_t)(u0 + k) * T + t0) * N + j;
            if (BACKWARD) {
                if (out) out[o] = 0.0;
                prev = 0.0 + b[o];
            } else {
                prev = pi[j] + b[o];
                if (out) out[o] = prev;
            }
        }
        Cs[k * NP + j] = prev;
        const double m = wave_max_dpp(prev);
        if (lane == 0) Ms[k * NB + (j >> 6)] = m;
    }
    __syncthreads();
    for (int q = tid; q < UB * NP; q += nt) {
        const int k = q / NP, j = q % NP;
        double S = Ms[k * NB];
        for (int c = 1; c < NB; ++c) S = fmax(S, Ms[k * NB + c]);
        const double prev = Cs[k * NP + j];
        Es[k * NP + j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp(prev - S);
        if (j == 0) Ss[k] = S;
    }
    __syncthreads();
    const int i0 = rs * RPS, i1 = (i0 + RPS < N) ? i0 + RPS : N;
    const int jcl = jc < N ? jc : N - 1;
    for (int step = 1; step < maxlen; ++step) {
        const int cur = (step - 1) & 1, nxt = step & 1;
        const double* Ec = Es + cur * UB * NP;
        const double* Cc = Cs + cur * UB * NP;
        if (rs < RS) {
            double acc[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) acc[k] = 0.0;
#pragma unroll 8
            for (int i = i0; i < i1; ++i) {
                const double a = Aexp[(size_t)i * N + jcl];
#pragma unroll
                for (int k = 0; k < UB; ++k) acc[k] = fma(Ec[k * NP + i], a, acc[k]);
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) Ps[(rs * UB + k) * NP + jc] = acc[k];
        }
        __syncthreads();
        for (int q = tid; q < UB * NP; q += nt) {
            const int k = q / NP, j = q % NP;
            double* En = Es + nxt * UB * NP;
            double* Cn = Cs + nxt * UB * NP;
            if (step < len[k]) {
                double sum = Ps[k * NP + j];
                for (int r = 1; r < RS; ++r) sum += Ps[(r * UB + k) * NP + j];
                const double Sc = Ss[cur * UB + k];
                double val;
                if (j >= N || (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF)) {   // (a padding column: nothing to redo)
                    val = Sc + log(sum);
                } else {
                    const int jl = j < N ? j : N - 1;
                    atomicAdd(fb_count, 1u);       // diagnostic (lnasr_debug_fallback_columns)
                    double mx = LN_NEG_INF;
                    for (int i = 0; i < N; ++i) mx = fmax(mx, Cc[k * NP + i] + Alog[(size_t)i * N + jl]);
                    if (mx == LN_NEG_INF || mx != mx) {
                        val = mx;
                    } else {
                        double s2 = 0.0;
                        for (int i = 0; i < N; ++i) s2 += exp(Cc[k * NP + i] + Alog[(size_t)i * N + jl] - mx);
                        val = mx + log(s2);
                    }
                }
                const int t = frame_of(k, step);
                double prev = LN_NEG_INF;
                if (j < N) {
                    const size_t o = ((size_t)(u0 + k) * T + t) * N + j;
                    const double bv = b[o];
                    if (BACKWARD) {
                        if (out) out[o] = val;
                        prev = val + bv;
                    } else {
                        prev = val + bv;
                        if (out) out[o] = prev;
                    }
                }
                double S = Ms[(cur * UB + k) * NB];
                for (int c = 1; c < NB; ++c) S = fmax(S, Ms[(cur * UB + k) * NB + c]);
                S += lvl[(size_t)(u0 + k) * T + t];    // the lagged maximum, carried to this frame's emission level
                Cn[k * NP + j] = prev;
                En[k * NP + j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp(prev - S);
                const double m = wave_max_dpp(prev);
                if (lane == 0) Ms[(nxt * UB + k) * NB + (j >> 6)] = m;
                if (j == 0) Ss[nxt * UB + k] = S;
            } else {
                Cn[k * NP + j] = Cc[k * NP + j];
                En[k * NP + j] = Ec[k * NP + j];
                if (lane == 0) Ms[(nxt * UB + k) * NB + (j >> 6)] = Ms[(cur * UB + k) * NB + (j >> 6)];
                if (j == 0) Ss[nxt * UB + k] = Ss[cur * UB + k];
            }
        }
        __syncthreads();
    }
    if (!BACKWARD && loglik && tid < 64) {
        const int fin = (maxlen > 0 ? maxlen - 1 : 0) & 1;
        for (int k = 0; k < UB; ++k) {
            if (u0 + k < U && len[k] > 0) {
                double mx = Ms[(fin * UB + k) * NB];
                for (int c = 1; c < NB; ++c) mx = fmax(mx, Ms[(fin * UB + k) * NB + c]);
                double sacc = 0.0;
                for (int j = lane; j < NP; j += 64) {
                    const double pv = Cs[(fin * UB + k) * NP + j];
                    sacc += (mx == LN_NEG_INF || pv == LN_NEG_INF) ? 0.0 : exp(pv - mx);
                }
                sacc = wave_sum(sacc);
                if (lane == 0) loglik[u0 + k] = (mx == LN_NEG_INF) ? mx : mx + log(sacc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// N > 256, frame-parallel: beyond 256 states A (2 MB at N = 512) fits no CU, and a block that walks the frames of
// its own utterances has to stream all of A from L2 on every frame -- with U/4 blocks doing that, the L2 is the
// bound (42 us per frame at N = 512).  Here ONE launch advances ALL utterances by one frame: the (utterance,
// column) outputs are tiled over the whole chip (128 columns x UT utterances x a slice of the rows per block), A
// is read once per utterance tile instead of once per four utterances, and row slices meet in a floating-point
// atomic max.  b[t,j] is added before the slices meet (rounding is monotonic: same bits as adding it after).
// T launches per call; v has to be pre-filled with -inf (fill_f64_kernel).
// ------------------------------------------------------------------------------------------
template <int UT, int CPT>
__global__ __launch_bounds__(256) void viterbi_frame_kernel(const double* __restrict__ b, const double* __restrict__ A,
                                                            const int32_t* __restrict__ lens, double* __restrict__ v,
                                                            int U, int T, int N, int t, int RPS, int atomic) {
    // CPT columns per lane (tile = 64 CPT columns): every v value read from LDS serves CPT transitions, so the wider
    // tile is the one for large batches, where the LDS broadcast reads are the bound
    constexpr int TW = 64 * CPT;
    extern __shared__ double fsm[];
    double* Vp = fsm;                                  // [UT][RPS]  v[t-1] of the slice rows
    double* comb = fsm + (size_t)UT * RPS;             // [UT][TW]   the four waves meet here (LDS atomic max)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = blockIdx.y * UT;
    const int r0 = blockIdx.z * RPS;
    const int r1 = r0 + RPS < N ? r0 + RPS : N;
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < UT; ++k) any_live |= u0 + k < U && t < lens[u0 + k];
    if (!any_live) return;                             // block-uniform
    for (int e = tid; e < UT * RPS; e += 256) {
        const int k = e / RPS, i = r0 + e % RPS;
        const bool ok = u0 + k < U && t < lens[u0 + k] && i < N;
        Vp[e] = ok ? v[((size_t)(u0 + k) * T + (t - 1)) * N + i] : LN_NEG_INF;
    }
    for (int e = tid; e < UT * TW; e += 256) comb[e] = LN_NEG_INF;
    __syncthreads();
    double acc[UT][CPT];
#pragma unroll
    for (int k = 0; k < UT; ++k)
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[k][c] = LN_NEG_INF;
    int jc[CPT];                                       // clamped: loads are unconditional
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int j = blockIdx.x * TW + lane + 64 * c;
        jc[c] = j < N ? j : N - 1;
    }
    // wave w walks a contiguous quarter of the slice two rows at a time: the two v values of an utterance come out
    // of LDS in one 128-bit broadcast read, A two steps ahead in registers
    const int RPW = RPS / 4;                            // RPS is a multiple of 8
    const int w0 = r0 + w * RPW;
    int w1 = w0 + RPW;
    w1 = w1 < r1 ? w1 : r1;
    constexpr int PF = 2;
    double a[PF][2][CPT];
    auto loadA = [&](int i, int slot) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int ic = i + h;
            ic = ic < N ? ic : N - 1;
#pragma unroll
            for (int c = 0; c < CPT; ++c) a[slot][h][c] = A[(size_t)ic * N + jc[c]];
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) loadA(w0 + 2 * p, p);
    for (int i = w0; i < w1; i += 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int ii = i + 2 * p;
            double x[2][CPT];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < CPT; ++c) x[h][c] = a[p][h][c];
            loadA(ii + 2 * PF, p);
            if (ii < w1) {                             // wave-uniform; a row past the slice end holds -inf in Vp
                const double* vp = Vp + (ii - r0);
#pragma unroll
                for (int k = 0; k < UT; ++k) {
                    const double va = vp[k * RPS], vb = vp[k * RPS + 1];
#pragma unroll
                    for (int c = 0; c < CPT; ++c) acc[k][c] = vmax_f64(acc[k][c], vmax_f64(va + x[0][c], vb + x[1][c]));
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < UT; ++k)
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            __hip_atomic_fetch_max(&comb[k * TW + 64 * c + lane], acc[k][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    for (int e = tid; e < UT * TW; e += 256) {
        const int k = e / TW, c = e % TW, j = blockIdx.x * TW + c;
        if (j < N && u0 + k < U && t < lens[u0 + k]) {
            const double m = comb[e];
            const size_t o = ((size_t)(u0 + k) * T + t) * N + j;
            const double val = m + b[o];
            if (atomic) atomicMax(&v[o], val);
            else v[o] = val;
        }
    }
}

__global__ void viterbi_frame0_kernel(const double* __restrict__ b, const double* __restrict__ pi,
                                      const int32_t* __restrict__ lens, double* __restrict__ v, int U, int T, int N) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)U * N) return;
    const int u = (int)(idx / N), j = (int)(idx % N);
    if (lens[u] > 0) v[(size_t)u * T * N + j] = pi[j] + b[(size_t)u * T * N + j];
}

__global__ void fill_value_kernel(double* __restrict__ p, size_t n, double value) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = value;
}

// Forward / backward for N > 256, one launch per step over the whole batch (same tiling as viterbi_frame_kernel,
// rows not sliced across blocks so that every sum is added in a fixed order).  State between launches lives in
// global memory: E[2][U][N] = exp(prev - shift), P[2][U][N] = prev (log domain, for the exact fallback) and a
// ring M[4][U] of per-utterance maxima of prev (filled with a floating-point atomic max: order independent).
// Step s of an utterance of length L is frame s (forward) or L-1-s (backward).  With h_s = M_{max(s-1,0)}:
//     val = h_{s-1} + log(sum_i E_{s-1}[i] expA[i,j]),  prev_s = val + b,  E_s = exp(prev_s - h_s),  M_s = max_j prev_s
// -- the lagged shift of the register-resident kernels, sums outside [1e-290, 1e290] redone exactly.
template <int UT, bool BACKWARD>
__global__ __launch_bounds__(256) void fwdbwd_frame_kernel(const double* __restrict__ b, const double* __restrict__ Aexp,
                                                           const double* __restrict__ Alog,
                                                           const int32_t* __restrict__ lens, double* __restrict__ out,
                                                           double* __restrict__ E, double* __restrict__ P,
                                                           double* __restrict__ Mring, int U, int T, int N, int s,
                                                           const double* __restrict__ lvl,
                                                           unsigned* __restrict__ fb_count) {
    extern __shared__ double fsm[];
    double* Ep = fsm;                                   // [UT][NP]  e values of the previous step
    const int NP = (N + 7) / 8 * 8;
    double* part = fsm + (size_t)UT * NP;               // [4][UT][128]
    __shared__ double etab[kExpTab];
    __shared__ double ltab[2 * kLogTab];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * 128 + lane, j1 = j0 + 64;
    const int u0 = blockIdx.y * UT;
    const int cur = (s - 1) & 1, nxt = s & 1;
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < UT; ++k) any_live |= u0 + k < U && s < lens[u0 + k];
    // the ring slot of step s+1 is cleared by the first column tile (nobody reads or accumulates it during this launch)
    if (blockIdx.x == 0 && tid < UT && u0 + tid < U) Mring[(size_t)((s + 1) & 3) * U + u0 + tid] = LN_NEG_INF;
    if (!any_live) return;
    exp_table_init(etab, tid, 256);
    log_table_init(ltab, tid, 256);
    for (int e = tid; e < UT * NP; e += 256) {
        const int k = e / NP, i = e % NP;
        const bool ok = u0 + k < U && s < lens[u0 + k] && i < N;
        Ep[e] = ok ? E[((size_t)cur * U + u0 + k) * N + i] : 0.0;
    }
    __syncthreads();
    double acc[UT][2];
#pragma unroll
    for (int k = 0; k < UT; ++k) acc[k][0] = acc[k][1] = 0.0;
    const int jc0 = j0 < N ? j0 : N - 1, jc1 = j1 < N ? j1 : N - 1;
    const int RPW = NP / 4;                              // a multiple of 2
    const int w0 = w * RPW;
    int w1 = w0 + RPW;
    w1 = w1 < N ? w1 : N;
    constexpr int PF = 2;
    double a0[PF][2], a1[PF][2];
    auto loadA = [&](int i, int slot) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int ic = i + h;
            ic = ic < N ? ic : N - 1;
            a0[slot][h] = Aexp[(size_t)ic * N + jc0];
            a1[slot][h] = Aexp[(size_t)ic * N + jc1];
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) loadA(w0 + 2 * p, p);
    for (int i = w0; i < w1; i += 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int ii = i + 2 * p;
            const double x00 = a0[p][0], x01 = a0[p][1], x10 = a1[p][0], x11 = a1[p][1];
            loadA(ii + 2 * PF, p);
            if (ii < w1) {                               // rows >= N carry e = 0 in Ep
                const double* ep = Ep + ii;
#pragma unroll
                for (int k = 0; k < UT; ++k) {
                    const double ea = ep[k * NP], eb = ep[k * NP + 1];
                    acc[k][0] = fma(eb, x01, fma(ea, x00, acc[k][0]));
                    acc[k][1] = fma(eb, x11, fma(ea, x10, acc[k][1]));
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < UT; ++k) {
        part[(w * UT + k) * 128 + lane] = acc[k][0];
        part[(w * UT + k) * 128 + 64 + lane] = acc[k][1];
    }
    __syncthreads();
    for (int e = tid; e < UT * 128; e += 256) {          // a wave works on 64 columns of ONE utterance per round
        const int k = e >> 7, c = e & 127, j = blockIdx.x * 128 + c, u = u0 + k;
        const int L = u < U ? lens[u] : 0;
        double prev = LN_NEG_INF;
        if (j < N && s < L) {
        const double sum = (part[(0 * UT + k) * 128 + c] + part[(1 * UT + k) * 128 + c]) +
                           (part[(2 * UT + k) * 128 + c] + part[(3 * UT + k) * 128 + c]);
        // shift of a row = the maximum of the log-domain row before it + the level of its own emission row (row 0: its own maximum)
        const int tcur = BACKWARD ? L - 1 - s : s, tprv = BACKWARD ? L - s : s - 1;
        const double hprev = s >= 2 ? Mring[(size_t)((s - 2) & 3) * U + u] + lvl[(size_t)u * T + tprv]
                                    : Mring[(size_t)0 * U + u];                       // shift of E_{s-1}
        const double hcur = Mring[(size_t)((s - 1) & 3) * U + u] + lvl[(size_t)u * T + tcur];   // shift for E_s
        double val;
        if (sum >= kSafeSum && sum <= 1e290 && hprev != LN_NEG_INF) {
            val = hprev + log_sum(sum, ltab);
        } else {
            val = exact_column_lse(P + ((size_t)cur * U + u) * N, Alog + j, N);
            atomicAdd(fb_count, 1u);               // diagnostic (lnasr_debug_fallback_columns)
        }
        const int t = BACKWARD ? L - 1 - s : s;
        const size_t o = ((size_t)u * T + t) * N + j;
        if (BACKWARD) {
            out[o] = val;
            prev = val + b[o];
        } else {
            prev = val + b[o];
            out[o] = prev;
        }
        P[((size_t)nxt * U + u) * N + j] = prev;
        E[((size_t)nxt * U + u) * N + j] = (hcur == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - hcur, etab);
        }
        // one atomic per wave, not per state: 512 atomics on one address per frame cost ~100 us
        const double m = wave_max_dpp(prev);
        if (lane == 0 && m != LN_NEG_INF) atomicMax(&Mring[(size_t)(s & 3) * U + u], m);
    }
}

// step 0: prev_0 into out / P[0], its per-utterance maximum into M[0] (atomic max), M[1] cleared
template <bool BACKWARD>
__global__ void fwdbwd_frame0a_kernel(const double* __restrict__ b, const double* __restrict__ pi,
                                      const int32_t* __restrict__ lens, double* __restrict__ out, double* __restrict__ P,
                                      double* __restrict__ Mring, int U, int T, int N) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)U * N) return;
    const int u = (int)(idx / N), j = (int)(idx % N);
    const int L = lens[u];
    double prev = LN_NEG_INF;
    if (L > 0) {
        const int t = BACKWARD ? L - 1 : 0;
        const size_t o = ((size_t)u * T + t) * N + j;
        if (BACKWARD) {
            out[o] = 0.0;
            prev = 0.0 + b[o];
        } else {
            prev = pi[j] + b[o];
            out[o] = prev;
        }
        atomicMax(&Mring[u], prev);
    }
    P[idx] = prev;
}
__global__ void fwdbwd_frame0b_kernel(const double* __restrict__ P, const double* __restrict__ Mring,
                                      double* __restrict__ E, int U, int N) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)U * N) return;
    const double h = Mring[idx / N], prev = P[idx];
    E[idx] = (h == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp(prev - h);
}
// loglik[u] = lse_j alpha[len-1, j]  (hmm.py:274): one wave per utterance
__global__ void final_loglik_kernel(const double* __restrict__ alpha, const int32_t* __restrict__ lens,
                                    double* __restrict__ loglik, int U, int T, int N) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (u >= U) return;
    const int L = lens[u];
    if (L <= 0) return;
    const double* row = alpha + ((size_t)u * T + (L - 1)) * N;
    double mx = LN_NEG_INF;
    for (int j = lane; j < N; j += 64) mx = fmax(mx, row[j]);
    mx = wave_max(mx);
    double sacc = 0.0;
    for (int j = lane; j < N; j += 64) sacc += (mx == LN_NEG_INF || row[j] == LN_NEG_INF) ? 0.0 : exp(row[j] - mx);
    sacc = wave_sum(sacc);
    if (lane == 0) loglik[u] = (mx == LN_NEG_INF) ? mx : mx + log(sacc);
}

// levels of every emission row of the batch (row_level_kernel); the two directions of a statistics pass run on two streams and
// each fills its own buffer
static int emission_levels(lnasr_ctx* ctx, const double* b, int U, int T, bool backward, const double** out) {
    DevBuf& ws = backward ? ctx->ws_lvl2 : ctx->ws_lvl;
    const long rows = (long)U * T;
    LN_TRY(ws.reserve((size_t)rows * 8));
    LN_LAUNCH(ctx, "emission_levels", row_level_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, b, rows, ctx->model.N,
              ws.as<double>());
    *out = ws.as<double>();
    return 0;
}

template <bool BACKWARD>
static int fwdbwd_frames(lnasr_ctx* ctx, const double* b, const double* Aexp, const double* Alog, int U, int T,
                         const int32_t* d_lens, double* out, double* loglik) {
    Model& md = ctx->model;
    const int N = md.N;
    constexpr int UT = 4;
    const int NP = (N + 7) / 8 * 8;
    const int ctiles = (N + 127) / 128, utiles = (U + UT - 1) / UT;
    LN_REQUIRE(utiles <= 65535, "U=%d too large for the frame-parallel forward/backward", U);
    const size_t shm = ((size_t)UT * NP + 4 * UT * 128) * 8;
    LN_REQUIRE(shm + 4096 <= 160 * 1024, "N=%d too large for the frame-parallel forward/backward", N);
    const size_t un = (size_t)U * N;
    DevBuf& ws = BACKWARD ? ctx->ws_q : ctx->ws_p;     // the two directions may run side by side (bw_accumulate)
    LN_TRY(ws.reserve((2 * un + 2 * un + 4 * (size_t)U) * 8));
    double* E = ws.as<double>();
    double* P = E + 2 * un;
    double* Mring = P + 2 * un;
    const double* lvl;
    LN_TRY(emission_levels(ctx, b, U, T, BACKWARD, &lvl));
    LN_LAUNCH(ctx, "fwdbwd_fill", fill_value_kernel, dim3((unsigned)((4 * (size_t)U + 255) / 256)), dim3(256), 0, Mring,
              (size_t)4 * U, LN_NEG_INF);
    LN_LAUNCH(ctx, BACKWARD ? "backward_frame0" : "forward_frame0", fwdbwd_frame0a_kernel<BACKWARD>,
              dim3((unsigned)((un + 255) / 256)), dim3(256), 0, b, md.logpi.as<double>(), d_lens, out, P, Mring, U, T, N);
    LN_LAUNCH(ctx, BACKWARD ? "backward_frame0" : "forward_frame0", fwdbwd_frame0b_kernel, dim3((unsigned)((un + 255) / 256)),
              dim3(256), 0, (const double*)P, (const double*)Mring, E, U, N);
    for (int s = 1; s < T; ++s)
        LN_LAUNCH(ctx, BACKWARD ? "backward_frame" : "forward_frame", (fwdbwd_frame_kernel<UT, BACKWARD>),
                  dim3(ctiles, utiles), dim3(256), shm, b, Aexp, Alog, d_lens, out, E, P, Mring, U, T, N, s, lvl, ctx->nan_flag.as<unsigned>() + 3);
    if (!BACKWARD && loglik)
        LN_LAUNCH(ctx, "forward_loglik", final_loglik_kernel, dim3((U + 3) / 4), dim3(256), 0, (const double*)out, d_lens,
                  loglik, U, T, N);
    return 0;
}


// ---- launchers (dispatch: viterbi.hip / fwdbwd.hip by the rules of trellis.h) ---------------------------------------
int viterbi_frames_device(lnasr_ctx* ctx, const double* b, int U, int T, const int32_t* d_lens, double* v) {
    // one launch per frame over the whole batch (see viterbi_frame_kernel)
    Model& md = ctx->model;
    const int N = md.N;
    const double* A = md.logA.as<double>();
    const double* pi = md.logpi.as<double>();
    constexpr int UT = 8;
    const int CPT = dispatch::viterbi_frame_cpt(U);          // wide tiles when the batch alone fills the chip
    const int TW = 64 * CPT;
    const int ctiles = (N + TW - 1) / TW, utiles = (U + UT - 1) / UT;
    LN_REQUIRE(utiles <= 65535, "U=%d too large for the frame-parallel Viterbi", U);
    int RS = (3 * ctx->prop.multiProcessorCount + ctiles * utiles - 1) / (ctiles * utiles);
    if (RS > N / 64) RS = N / 64;
    if (RS < 1) RS = 1;
    int RPS = ((N + RS - 1) / RS + 7) / 8 * 8;
    RS = (N + RPS - 1) / RPS;
    const size_t cells = (size_t)U * T * N;
    const size_t shm = ((size_t)UT * RPS + UT * TW) * 8;
    LN_REQUIRE(shm <= 160 * 1024, "N=%d too large for the frame-parallel Viterbi", N);
    if (RS > 1)
        LN_LAUNCH(ctx, "viterbi_fill", fill_value_kernel, dim3(2048), dim3(256), 0, v, cells, LN_NEG_INF);
    LN_LAUNCH(ctx, "viterbi_frame0", viterbi_frame0_kernel, dim3((unsigned)(((size_t)U * N + 255) / 256)), dim3(256), 0, b,
              pi, d_lens, v, U, T, N);
    for (int t = 1; t < T; ++t) {
        if (CPT == 4)
            LN_LAUNCH(ctx, "viterbi_frame", (viterbi_frame_kernel<UT, 4>), dim3(ctiles, utiles, RS), dim3(256), shm, b, A,
                      d_lens, v, U, T, N, t, RPS, RS > 1 ? 1 : 0);
        else
            LN_LAUNCH(ctx, "viterbi_frame", (viterbi_frame_kernel<UT, 2>), dim3(ctiles, utiles, RS), dim3(256), shm, b, A,
                      d_lens, v, U, T, N, t, RPS, RS > 1 ? 1 : 0);
    }
    return 0;
}

int viterbi_stream_device(lnasr_ctx* ctx, const double* b, int U, int T, const int32_t* d_lens, double* v) {
    Model& md = ctx->model;
    const int N = md.N;
    constexpr int UB = 4;
    const int NP = ((N + 63) / 64) * 64, RS = 1024 / NP, RPS = (N + RS - 1) / RS;
    const size_t shm = (size_t)(2 + RS) * UB * NP * 8;
    LN_LAUNCH(ctx, "viterbi_stream", (viterbi_stream<UB>), dim3((U + UB - 1) / UB), dim3(NP * RS), shm, b,
              md.logA.as<double>(), md.logpi.as<double>(), d_lens, v, U, T, N, NP, RS, RPS);
    return 0;
}

int fwdbwd_frames_device(lnasr_ctx* ctx, bool backward, const double* b, int U, int T, const int32_t* d_lens, double* out,
                         double* loglik) {
    Model& md = ctx->model;
    if (backward)
        return fwdbwd_frames<true>(ctx, b, md.expAT.as<double>(), md.logAT.as<double>(), U, T, d_lens, out, nullptr);
    if (!out) {                                         // log-likelihood only: alpha still has to exist somewhere
        LN_TRY(ctx->ws_alpha.reserve((size_t)U * T * md.N * 8));
        out = ctx->ws_alpha.as<double>();
    }
    return fwdbwd_frames<false>(ctx, b, md.expA.as<double>(), md.logA.as<double>(), U, T, d_lens, out, loglik);
}

template <bool BACKWARD>
static int fwdbwd_stream_launch(lnasr_ctx* ctx, const double* b, int U, int T, const int32_t* d_lens, double* out,
                                double* loglik) {
    Model& md = ctx->model;
    const int N = md.N;
    const double* Alin = BACKWARD ? md.expAT.as<double>() : md.expA.as<double>();
    const double* Alog = BACKWARD ? md.logAT.as<double>() : md.logA.as<double>();
    const double* pi = md.logpi.as<double>();
    const int NP = ((N + 63) / 64) * 64, RS = 1024 / NP, RPS = (N + RS - 1) / RS, NB = NP / 64;
    const int ub = dispatch::fwdbwd_stream_ub(N, U, ctx->prop.multiProcessorCount);
    const size_t shm = ((size_t)(4 + RS) * ub * NP + 2 * ub * NB + 2 * ub) * 8;
    double* ll = BACKWARD ? nullptr : loglik;
    const double* lvl;
    LN_TRY(emission_levels(ctx, b, U, T, BACKWARD, &lvl));
    if (ub == 1)
        LN_LAUNCH(ctx, BACKWARD ? "backward_stream_u1" : "forward_stream_u1", (fwdbwd_stream<BACKWARD, 1>), dim3(U),
                  dim3(NP * RS), shm, b, Alin, Alog, pi, d_lens, out, ll, U, T, N, NP, RS, RPS, lvl, ctx->nan_flag.as<unsigned>() + 3);
    else if (ub == 2)
        LN_LAUNCH(ctx, BACKWARD ? "backward_stream_u2" : "forward_stream_u2", (fwdbwd_stream<BACKWARD, 2>), dim3((U + 1) / 2),
                  dim3(NP * RS), shm, b, Alin, Alog, pi, d_lens, out, ll, U, T, N, NP, RS, RPS, lvl, ctx->nan_flag.as<unsigned>() + 3);
    else
        LN_LAUNCH(ctx, BACKWARD ? "backward_stream" : "forward_stream", (fwdbwd_stream<BACKWARD, 4>), dim3((U + 3) / 4),
                  dim3(NP * RS), shm, b, Alin, Alog, pi, d_lens, out, ll, U, T, N, NP, RS, RPS, lvl, ctx->nan_flag.as<unsigned>() + 3);
    return 0;
}

int fwdbwd_stream_device(lnasr_ctx* ctx, bool backward, const double* b, int U, int T, const int32_t* d_lens, double* out,
                         double* loglik) {
    return backward ? fwdbwd_stream_launch<true>(ctx, b, U, T, d_lens, out, loglik)
                    : fwdbwd_stream_launch<false>(ctx, b, U, T, d_lens, out, loglik);
}

}  // namespace lnasr
