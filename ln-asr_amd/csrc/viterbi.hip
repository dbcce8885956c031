// Viterbi recursion (hmm.py:133-172), kernels for N <= 128 and the dispatch over every family.
//
// Parallelism: utterances are independent; the frame recursion is sequential.  A block owns
// UB utterances and walks their frames in lock-step (one or two barriers per frame).
//
// Viterbi keeps NO back-pointers in the recursion: v[t] is stored and the pointer actually
// needed, bt[t+1, path[t+1]] = argmax_i(v[t,i] + A[i, path[t+1]]) (hmm.py:165,171), is
// recomputed during the backtrace (backtrace.hip) with the same fp64 additions, so the result is
// bit-identical while the inner loop is one add + one max per (i, j).  The full bt matrix is
// produced by a separate, fully parallel kernel only when the caller asks for it.
//
// Families: viterbi_generic (any N, A from L2), viterbi_reg / viterbi_reg1b (N <= 128, A in registers),
// viterbi_tiny (N <= 16, one wave per utterance); 128 < N <= 256: trellis256.hip; N > 256: bign.hip.
#include "common.h"
#include "devmath.h"
#include "trellis.h"

namespace lnasr {

// ------------------------------------------------------------------------------------------
// Viterbi, generic: any N, A streamed from L2
// ------------------------------------------------------------------------------------------
template <typename TS, int UB>
__global__ void viterbi_generic(const TS* __restrict__ b, const double* __restrict__ A, const double* __restrict__ pi,
                                const int32_t* __restrict__ lens, TS* __restrict__ v, int U, int T, int N) {
    extern __shared__ double Vs[];            // [2][UB][N]
    const int u0 = blockIdx.x * UB;
    int len[UB], maxlen = 0;
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        len[k] = (u0 + k < U) ? lens[u0 + k] : 0;
        maxlen = len[k] > maxlen ? len[k] : maxlen;
    }
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            double val = LN_NEG_INF;
            if (len[k] > 0) {
                size_t o = ((size_t)(u0 + k) * T) * N + j;
                val = rnd<TS>(pi[j] + ld(b, o));
                st(v, o, val);
            }
            Vs[(0 * UB + k) * N + j] = val;
        }
    }
    __syncthreads();
    for (int t = 1; t < maxlen; ++t) {
        const double* Vc = Vs + ((t - 1) & 1) * UB * N;
        double* Vn = Vs + (t & 1) * UB * N;
        for (int j = threadIdx.x; j < N; j += blockDim.x) {
            double acc[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) acc[k] = LN_NEG_INF;
            for (int i = 0; i < N; ++i) {
                const double a = A[(size_t)i * N + j];
#pragma unroll
                for (int k = 0; k < UB; ++k) acc[k] = fmax(acc[k], Vc[k * N + i] + a);
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                double val = Vc[k * N + j];
                if (t < len[k]) {
                    size_t o = ((size_t)(u0 + k) * T + t) * N + j;
                    val = rnd<TS>(acc[k] + ld(b, o));
                    st(v, o, val);
                }
                Vn[k * N + j] = val;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Viterbi, register-resident transition matrix (N <= 64*CPT, N <= NW*RPT)
// ------------------------------------------------------------------------------------------
// Thread (wave w, lane l) keeps A[w*RPT + r][l + 64*c] for r < RPT, c < CPT in registers
// (CPT*RPT doubles).  v[t-1] sits in LDS; each wave pulls its RPT-row slice into registers at
// the top of the frame (broadcast reads) and every value feeds CPT add+max pairs.  Row slices
// are combined through LDS (NW > 1); the CPT*UB (utterance, column-block) finalisation jobs are
// dealt round-robin to the waves, each of which adds b (prefetched one frame ahead), stores
// v[t] and publishes it for the next frame.  WPE = waves per SIMD the register budget targets.
template <int CPT, int NW, int RPT, int UB, int WPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void viterbi_reg(
    const double* __restrict__ b, const double* __restrict__ A, const double* __restrict__ pi,
    const int32_t* __restrict__ lens, double* __restrict__ v, int U, int T, int N) {
    constexpr int NP = 64 * CPT;
    constexpr int NJOB = CPT * UB;
    constexpr int JPW = (NJOB + NW - 1) / NW;
    __shared__ double Vs[2][UB][NP];
    __shared__ double Ps[NW > 1 ? NW : 1][UB][NP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u0 = blockIdx.x * UB;

    double areg[CPT][RPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = w * RPT + r, j = lane + 64 * c;
            areg[c][r] = (i < N && j < N) ? A[(size_t)i * N + j] : LN_NEG_INF;
        }

    int len[UB], maxlen = 0;
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        len[k] = (u0 + k < U) ? lens[u0 + k] : 0;
        maxlen = len[k] > maxlen ? len[k] : maxlen;
    }
    // t = 0
    for (int j = threadIdx.x; j < NP; j += 64 * NW) {
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            double val = LN_NEG_INF;
            if (j < N && len[k] > 0) {
                size_t o = ((size_t)(u0 + k) * T) * N + j;
                val = pi[j] + b[o];
                v[o] = val;
            }
            Vs[0][k][j] = val;
        }
    }
    // this wave's finalisation jobs and their emission stream (always a valid address: clamped)
    int jk[JPW], jc[JPW];
    const double* bp[JPW];
    double bnext[JPW];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int q = w + jj * NW;
        jk[jj] = q / CPT;
        jc[jj] = q % CPT;
        const int uu = (u0 + jk[jj] < U) ? u0 + jk[jj] : U - 1;
        const int j = lane + 64 * jc[jj];
        bp[jj] = b + (size_t)uu * T * N + (j < N ? j : N - 1);
        bnext[jj] = (q < NJOB) ? bp[jj][(size_t)(T > 1 ? 1 : 0) * N] : 0.0;   // wave-uniform: idle waves load nothing
    }
    __syncthreads();

    for (int t = 1; t < maxlen; ++t) {
        const int cur = (t - 1) & 1, nxt = t & 1;
        double bcur[JPW];
        const int tn = t + 1 < T ? t + 1 : T - 1;
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            bcur[jj] = bnext[jj];
            if (w + jj * NW < NJOB) bnext[jj] = bp[jj][(size_t)tn * N];
        }
        double vr[UB][RPT];
#pragma unroll
        for (int k = 0; k < UB; ++k)
#pragma unroll
            for (int r = 0; r < RPT; ++r) vr[k][r] = Vs[cur][k][w * RPT + r];

        double acc[UB][CPT];
#pragma unroll
        for (int k = 0; k < UB; ++k)
#pragma unroll
            for (int c = 0; c < CPT; ++c) acc[k][c] = vr[k][0] + areg[c][0];
#pragma unroll
        for (int r = 1; r < RPT; ++r)
#pragma unroll
            for (int k = 0; k < UB; ++k)
#pragma unroll
                for (int c = 0; c < CPT; ++c) acc[k][c] = fmax(acc[k][c], vr[k][r] + areg[c][r]);
        if (NW > 1) {
#pragma unroll
            for (int k = 0; k < UB; ++k)
#pragma unroll
                for (int c = 0; c < CPT; ++c) Ps[w][k][lane + 64 * c] = acc[k][c];
            __syncthreads();
        }
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const int q = w + jj * NW;
            if (q < NJOB) {
                const int k = jk[jj], c = jc[jj], j = lane + 64 * c;
                double m;
                if (NW > 1) {
                    m = Ps[0][k][j];
#pragma unroll
                    for (int ww = 1; ww < NW; ++ww) m = fmax(m, Ps[ww][k][j]);
                } else {
                    m = acc[k][c];
                }
                double val = Vs[cur][k][j];
                if (j < N && t < len[k]) {
                    val = m + bcur[jj];
                    v[((size_t)(u0 + k) * T + t) * N + j] = val;
                }
                Vs[nxt][k][j] = val;
            }
        }
        __syncthreads();
    }
}

// Single-barrier variant: b[t,j] is added to every row-slice maximum before the slices meet
// (rounding is monotonic, so max_i((v_i + A_ij) + b_j) == max_i(v_i + A_ij) + b_j bit for bit) and
// the slices are merged with LDS floating-point atomic max into a triple-buffered v row: one
// barrier per frame, no partial-maximum round trip.
// TS = storage type of b and v.  float: the arithmetic stays fp64 and every v is rounded to float where the
// reference's float32 array rounds it (hmm.py:165 with precision=np.float32) -- each row-slice maximum right
// before the slices meet, which equals rounding their maximum (rounding is monotonic).
// ALT: two of these blocks share a CU (UB = 2, U >= 512) and are not served alike -- the SIMD arbiter goes by priority,
// then AGE, so the older block ran nearly unimpeded and finished after 0.93 ms while the younger, fed the leftover
// slots, needed 1.40 ms: alone on the CU (and unable to fill it) for the difference, 83 % slot utilisation
// (tools/block_times.py).  With ALT the two halves of the grid raise their priority in alternate periods of the 100 MHz
// clock (alternate_priority, devmath.h; measured with periods of 10 us: 1.33 ms, 41 us: 1.27, 164 us: 1.25, 0.65 ms: 1.26,
// 1.3 ms: 1.34) and finish together: 1.37 -> 1.25 ms.  (Merging the two blocks behind ONE barrier
// -- 16 waves in lock-step -- was measured first: 2.68 ms; the independent blocks hide each other's per-frame latency.
// The opposite direction, after four-wave blocks had paid in fwdbwd_cols2: <2, 4, 32, 2, 2> -- two utterances per FOUR-wave
// block, 230 registers, two waves per SIMD -- 1.29 ms against 1.27; <2, 4, 32, 1, 3> -- one utterance, three blocks per CU,
// 168 registers and six of them spilled -- 1.99 ms.  This kernel is 76 % vector issue (SQ counters), not waits.)
// CHKB (lnasr_viterbi: emissions handed in by the caller): wave 0 tests the emissions it loads anyway for NaN -- the LDS atomic
// maximum drops a NaN candidate, so a NaN emission would otherwise surface only as a -inf cell of v, which a model with
// forbidden transitions (left-to-right: the normal ASR topology) has on perfectly clean data -- and raises word 1 of the
// context's flag itself; the back-trace then need not treat -inf cells as suspects.  Two compares per
// frame and column block in one of eight waves; the decode path (features tested by the emission kernel) runs CHKB = false.
template <int CPT, int NW, int RPT, int UB, int WPE, typename TS = double, bool ALT = false, bool CHKB = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void viterbi_reg1b(
    const TS* __restrict__ b, const double* __restrict__ A, const double* __restrict__ pi,
    const int32_t* __restrict__ lens, TS* __restrict__ v, int U, int T, int N, unsigned* __restrict__ nan_flag,
    unsigned nan_epoch) {
    constexpr int NP = 64 * CPT;
    __shared__ double Vs[3][UB][NP];
    bool nan0 = false;                             // CHKB, frame 0: per lane, folded into the scalar ballot word below
    LN_BLOCK_T0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u0 = blockIdx.x * UB;
    const unsigned cls = ALT && blockIdx.x >= (gridDim.x + 1) / 2 ? 1u : 0u;

    double areg[CPT][RPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = w * RPT + r, j = lane + 64 * c;
            areg[c][r] = (i < N && j < N) ? A[(size_t)i * N + j] : LN_NEG_INF;
        }
    int len[UB], maxlen = 0;
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        len[k] = (u0 + k < U) ? lens[u0 + k] : 0;
        maxlen = len[k] > maxlen ? len[k] : maxlen;
    }
    for (int j = threadIdx.x; j < NP; j += 64 * NW) {
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            double val = LN_NEG_INF;
            if (j < N && len[k] > 0) {
                size_t o = ((size_t)(u0 + k) * T) * N + j;
                const double b0 = ld(b, o);
                if (CHKB && b0 != b0) nan0 = true;
                val = rnd<TS>(pi[j] + b0);
                st(v, o, val);
            }
            Vs[0][k][j] = val;
            Vs[1][k][j] = LN_NEG_INF;
            Vs[2][k][j] = LN_NEG_INF;
        }
    }
    // emission streams of this wave's columns (every wave needs b for all UB*CPT column blocks): a SCALAR row base per
    // utterance (utterance and frame are wave-uniform: the 64-bit row offset is formed on the scalar unit) plus a per-lane 32-bit
    // column offset -- the load takes both.  (Until round 5 the row offset was added to per-lane 64-bit pointers: eight vector
    // instructions per frame and column block in a kernel whose bound is vector issue.)
    int jcol[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int j = lane + 64 * c;
        jcol[c] = j < N ? j : N - 1;
    }
    const TS* ub[UB];                              // b + u * T * N, wave-uniform
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int uk = __builtin_amdgcn_readfirstlane((u0 + k < U) ? u0 + k : U - 1);
        ub[k] = b + (size_t)uk * T * N;
    }
    double bnext[UB][CPT];
#pragma unroll
    for (int k = 0; k < UB; ++k)
#pragma unroll
        for (int c = 0; c < CPT; ++c) bnext[k][c] = ld(ub[k] + (size_t)(T > 1 ? 1 : 0) * N, jcol[c]);
    constexpr bool LAT = UB == 1;                  // one utterance per block: the latency instantiation (see below)
    // (LAT: per-lane pointers to the row after next, advanced by one row per turn -- one vector instruction per load and nothing
    // in front of it: 0.433 against 0.448 ms per 1000 frames; the two-utterance blocks have no registers for them)
    const TS* pb[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) pb[c] = ub[0] + (size_t)(T > 2 ? 2 : (T > 1 ? 1 : 0)) * N + jcol[c];
    // store duty: (utterance, column block) q = w + jj*NW written to global one frame late
    constexpr int NJOB = CPT * UB;
    __syncthreads();
    // (formed HERE, outside the divergent set-up loop: the word is wave-uniform and stays in scalar registers through the frames)
    unsigned long long saw_nan = CHKB ? __ballot(nan0) : 0ull;

    int cur = 0;                                   // buffer holding v[t-1]
    for (int t = 1; t < maxlen; ++t) {
        // (priority by PHASE instead -- loads, atomics and the barrier at 1, the add / max runs at 0, the scheme that pays in the
        // MFCC kernel -- was measured in round 4: 2.06 ms against 1.27; this kernel is issue-bound, its waves must not yield)
        if (ALT && (t & 7) == 0) alternate_priority(cls);     // devmath.h
        const int nxt = cur == 2 ? 0 : cur + 1, rst = nxt == 2 ? 0 : nxt + 1;
        double bcur[UB][CPT];
        if constexpr (LAT) {
            const long adv = t + 2 < T ? N : 0;      // stays on the last row once it is reached (that value is never used)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                bcur[0][c] = bnext[0][c];
                bnext[0][c] = (double)*pb[c];
                pb[c] += adv;
            }
        } else {
            const int tn = __builtin_amdgcn_readfirstlane(t + 1 < T ? t + 1 : T - 1);
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const TS* row = ub[k] + (size_t)tn * N;          // scalar
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    bcur[k][c] = bnext[k][c];
                    bnext[k][c] = ld(row, jcol[c]);
                }
            }
        }
        // v[t-1] of the columns this wave stores to global (complete since the last barrier): read now, with the other LDS
        // reads of the turn, stored AFTER the arithmetic -- the store's wait for its LDS read sat at the top of the frame of the
        // waves that have the duty, and the barrier waits for the slowest wave (round 5)
        double vst = 0.0;
        const bool duty = t > 1 && w < NJOB && lane + 64 * (w % CPT) < N && t - 1 < len[w / CPT < UB ? w / CPT : 0];
        if constexpr (LAT) {
            if (t > 1 && w < NJOB) vst = Vs[cur][w / CPT][lane + 64 * (w % CPT)];
        } else {
            if (duty) st(v, ((size_t)(u0 + w / CPT) * T + (t - 1)) * N + lane + 64 * (w % CPT), Vs[cur][w / CPT][lane + 64 * (w % CPT)]);
        }
        // reset the buffer that frame t+1 will accumulate into
        if (w == NW - 1) {
#pragma unroll
            for (int k = 0; k < UB; ++k)
#pragma unroll
                for (int c = 0; c < CPT; ++c) Vs[rst][k][lane + 64 * c] = LN_NEG_INF;
        }
        double acc[UB][CPT];
        // two half slices keep the live v values to UB*HR doubles; ONE utterance per block has the registers for the whole
        // slice, and its second batch of LDS reads no longer waits in the middle of the frame (round 5)
        constexpr int NH = UB == 1 ? 1 : 2, HR = RPT / NH;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            double vr[UB][HR];
#pragma unroll
            for (int k = 0; k < UB; ++k)
#pragma unroll
                for (int r = 0; r < HR; ++r) vr[k][r] = Vs[cur][k][w * RPT + h * HR + r];
#pragma unroll
            for (int r = 0; r < HR; ++r)
#pragma unroll
                for (int k = 0; k < UB; ++k)
#pragma unroll
                    for (int c = 0; c < CPT; ++c) {
                        const double cand = vr[k][r] + areg[c][h * HR + r];
                        acc[k][c] = (h == 0 && r == 0) ? cand : vmax_f64(acc[k][c], cand);     // (partial maxima per column -- four chains instead of two -- measured for one utterance: no gain)
                    }
        }
#pragma unroll
        for (int k = 0; k < UB; ++k)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const double cand = rnd<TS>(acc[k][c] + bcur[k][c]);
                // CHKB: one wave looks at the frame's candidates for the block -- v never holds a NaN (the atomic maximum drops
                // one), so a NaN candidate is a NaN emission (columns past N repeat column N - 1: no mask needed); tested HERE,
                // on a value that is live anyway (tested on bcur at the top of the frame the instantiation spilled 14 registers
                // into the loop: 2.07 ms against 1.25 for the metric shard, round 6)
                if (CHKB) saw_nan |= __ballot(cand != cand) & (t < len[k] ? ~0ull : 0ull);
                __hip_atomic_fetch_max(&Vs[nxt][k][lane + 64 * c], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        if constexpr (LAT) {
            if (duty) st(v, ((size_t)(u0 + w / CPT) * T + (t - 1)) * N + lane + 64 * (w % CPT), vst);
        }
        __syncthreads();
        cur = nxt;
    }
    // last frame
    if (maxlen > 1 && w < NJOB) {
        const int k = w / CPT, c = w % CPT, j = lane + 64 * c;
        if (j < N && maxlen - 1 < len[k]) st(v, ((size_t)(u0 + k) * T + (maxlen - 1)) * N + j, Vs[cur][k][j]);
    }
    if (CHKB && __ballot(saw_nan != 0) != 0 && lane == 0) *nan_flag = nan_epoch;
    LN_BLOCK_T1;
}
LN_BLOCK_TIMES_EXPORT(lnasr_dbg_block_times_trellis)

// ------------------------------------------------------------------------------------------
// Tiny models (N <= 16): the character-segmentation HMM of scripts/seg/seg.py (n = 4, one sequence of tens of
// thousands of symbols) and the toy models of the reference's tests.  One WAVE per utterance, no LDS, no barrier:
// lane j holds v[t-1, j]; v[t-1, i] reaches the others as a scalar (v_readlane), the column A[:, j] sits in
// registers, and the winning i is kept as a byte -- with N this small the back-pointers cost N bytes per frame and
// turn the backtrace into a chain of scalar table look-ups instead of an arg-max over a re-added column.
// Same additions and the same first-index tie rule as every other kernel here: bit-identical v, bt, path.
//   bp [U][T][NMAX] bytes: bp[u, t, j] = argmax_i(v[t-1, i] + A[i, j]),  bp[u, 0, :] = 0
// ------------------------------------------------------------------------------------------
template <typename TS, int NMAX>
__global__ __launch_bounds__(256) void viterbi_tiny_kernel(const TS* __restrict__ b, const double* __restrict__ A,
                                                           const double* __restrict__ pi, const int32_t* __restrict__ lens,
                                                           TS* __restrict__ v, uint8_t* __restrict__ bp, int U, int T, int N) {
    // Global memory is touched 64 frames at a time (emissions in, v and back-pointers out, all through the wave's
    // LDS slice): with a load or a store in every frame the recursion ran at the latency of the memory counter
    // (vmcnt is in order: the wait for an emission also waited for the store of the frame before), 275 ns a frame.
    constexpr int C = 64;
    __shared__ double bch[4][C * NMAX];                   // emissions of the chunk, [frame][state] with stride N
    __shared__ double vch[4][C * NMAX + 1];               // v of the chunk, same layout (+ a slot for idle lanes)
    __shared__ uint8_t bpch[4][C * NMAX + 4];             // back-pointers, stride NMAX (+ a slot for idle lanes)
    constexpr int VDUMP = C * NMAX, PDUMP = C * NMAX;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = blockIdx.x * (blockDim.x >> 6) + w;
    if (u >= U) return;
    const int L = lens[u];
    if (L <= 0) return;
    const bool act = lane < N;
    const int j = act ? lane : N - 1;
    double a[NMAX];                                       // A[i, j]; rows beyond N never win
#pragma unroll
    for (int i = 0; i < NMAX; ++i) a[i] = i < N ? A[(size_t)i * N + j] : LN_NEG_INF;
    const TS* bu = b + (size_t)u * T * N;
    TS* vu = v + (size_t)u * T * N;
    uint8_t* bpu = bp + (size_t)u * T * NMAX;
    double* bc = bch[w];
    double* vc = vch[w];
    uint8_t* pc = bpch[w];
    constexpr int EPL = (C * NMAX + 63) / 64;             // chunk elements per lane (at most)
    double stage[EPL];
    auto fetch = [&](int t0) {                            // emissions of frames t0 .. t0 + C - 1 (clamped) -> registers
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            const size_t g = (size_t)t0 * N + idx;
            stage[e] = (idx < C * N && g < (size_t)L * N) ? ld(bu, g) : 0.0;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            if (idx < C * N) bc[idx] = stage[e];
        }
    };
    fetch(0);
    stash();
    __builtin_amdgcn_wave_barrier();
    // lanes with nothing to write aim at a spare slot behind the chunk instead of branching around the store: the
    // frame loop below is one straight block (a lone wave pays dearly for every taken branch)
    double vcur = rnd<TS>(pi[j] + bc[j]);                 // frame 0
    for (int t0 = 0; t0 < L; t0 += C) {
        const int nfr = L - t0 < C ? L - t0 : C;
        if (t0 + C < L) fetch(t0 + C);                    // next chunk's emissions travel during this one
        int k0 = 0;
        if (t0 == 0) {
            if (act) vc[j] = vcur;
            if (lane < NMAX) pc[lane] = 0;
            k0 = 1;
        }
        double bnx = bc[(k0 < C ? k0 : 0) * N + j];
        for (int k = k0; k < nfr; ++k) {
            const double bt_ = bnx;
            bnx = bc[(k + 1 < C ? k + 1 : k) * N + j];    // the next frame's emission is read a frame ahead
            // v[t-1, i] comes from lane i of the DPP row (the utterance sits in lanes 0..N-1 of row 0); the NMAX
            // candidates meet in a tree whose upper operand wins only when strictly greater: the first index
            // among equal maxima (numpy.argmax), log2(NMAX) dependent steps instead of NMAX
            double cnd[NMAX];
            int idx[NMAX];
            static_for<0, NMAX>([&](auto I) {
                constexpr int i = decltype(I)::value;
                cnd[i] = mov_bcast<i>(vcur) + a[i];
                idx[i] = i;
            });
#pragma unroll
            for (int wd = NMAX / 2; wd >= 1; wd /= 2)
#pragma unroll
                for (int i = 0; i < wd; ++i) {
                    const bool win = cnd[2 * i + 1] > cnd[2 * i];
                    cnd[i] = win ? cnd[2 * i + 1] : cnd[2 * i];
                    idx[i] = win ? idx[2 * i + 1] : idx[2 * i];
                }
            vcur = rnd<TS>(cnd[0] + bt_);
            vc[act ? k * N + j : VDUMP] = vcur;
            pc[lane < NMAX ? k * NMAX + lane : PDUMP] = (uint8_t)idx[0];
        }
        __builtin_amdgcn_wave_barrier();
        // the chunk's v and back-pointers leave in whole rows
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            if (idx < nfr * N) st(vu, (size_t)t0 * N + idx, vc[idx]);
        }
        {
            const uint32_t* pw = reinterpret_cast<const uint32_t*>(pc);
            uint32_t* gw = reinterpret_cast<uint32_t*>(bpu + (size_t)t0 * NMAX);
#pragma unroll
            for (int e = 0; e < (C * NMAX / 4 + 63) / 64; ++e) {
                const int idx = lane + 64 * e;
                if (idx < nfr * NMAX / 4) gw[idx] = pw[idx];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (t0 + C < L) stash();
        __builtin_amdgcn_wave_barrier();
    }
}


int viterbi_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* v, bool* b_tested) {
    LN_DEVPTR(b); LN_DEVPTR(d_lens); LN_DEVPTR(v);
    // b_tested given: the caller wants the kernel to look for NaN emissions itself when it can (*b_tested = it did, and
    // raised word 1 of ctx->nan_flag with the call's epoch if it met one)
    const bool chk = b_tested != nullptr;
    if (b_tested) *b_tested = false;
    unsigned* const nflag = ctx->nan_flag.as<unsigned>() + 1;
    const unsigned epoch = ctx->nan_epoch;
    Model& md = ctx->model;
    const int N = md.N;
    const double* A = md.logA.as<double>();
    const double* pi = md.logpi.as<double>();
    ctx->bp_v = nullptr;
    if (N <= dispatch::kTinyMaxN && !sw().no_tinyvit) {
        // one wave per utterance, byte back-pointers for the backtrace (viterbi_tiny_kernel)
        const int nmax = N <= 4 ? 4 : (N <= 8 ? 8 : 16);
        LN_TRY(ctx->ws_bp.reserve((size_t)U * T * nmax));
        uint8_t* bp = ctx->ws_bp.as<uint8_t>();
        const dim3 grid((U + 3) / 4), block(256);
#define LN_TINY(TS, NM)                                                                                          \
    LN_LAUNCH(ctx, "viterbi_tiny", (viterbi_tiny_kernel<TS, NM>), grid, block, 0, (const TS*)b, A, pi, d_lens, (TS*)v, bp, \
              U, T, N)
        if (dtype == LNASR_F32) {
            if (nmax == 4) LN_TINY(float, 4);
            else if (nmax == 8) LN_TINY(float, 8);
            else LN_TINY(float, 16);
        } else {
            if (nmax == 4) LN_TINY(double, 4);
            else if (nmax == 8) LN_TINY(double, 8);
            else LN_TINY(double, 16);
        }
#undef LN_TINY
        ctx->bp_v = v;
        ctx->bp_U = U;
        ctx->bp_T = T;
        ctx->bp_nmax = nmax;
        return 0;
    }
    if (dtype == LNASR_F64 && !sw().no_regvit) {
        const char vsel = sw().vit ? sw().vit : 'e';
#define LN_VIT(name, CPT, NW, RPT, UB, WPE)                                                                  \
    do {                                                                                                     \
        LN_LAUNCH(ctx, name, (viterbi_reg<CPT, NW, RPT, UB, WPE>), dim3((U + UB - 1) / UB), dim3(64 * NW), 0, \
                  (const double*)b, A, pi, d_lens, (double*)v, U, T, N);                                     \
        return 0;                                                                                            \
    } while (0)
        // 17 ... 64 states: the two-wave blocks (two utterances each, many per CU) for batches; while every utterance can have a CU
        // of its own the eight-wave single-barrier kernel below walks a frame in 0.51 us against 0.69 (its 128-column tiles half
        // empty: the frame is latency, not work) -- tools/bench_rows.py midn, round 5
        if (N <= dispatch::kReg64MaxN && !(U <= ctx->prop.multiProcessorCount && vsel != 'a')) LN_VIT("viterbi_reg_n64", 1, 2, 32, 2, 4);
        if (N <= dispatch::kRegMaxN) {
            // default: single-barrier kernel (LDS atomic max); LNASR_VIT=a selects the two-barrier one
            if (vsel == 'a' && U < dispatch::kViterbiPairFromU) LN_VIT("viterbi_reg_n128_u1", 2, 8, 16, 1, 4);
            if (vsel == 'a') LN_VIT("viterbi_reg_n128", 2, 8, 16, 2, 4);
            if (chk) *b_tested = true;
            if (U < dispatch::kViterbiPairFromU) {
                if (chk)
                    LN_LAUNCH(ctx, "viterbi_reg1b_n128_u1", (viterbi_reg1b<2, 8, 16, 1, 4, double, false, true>), dim3(U), dim3(512),
                              0, (const double*)b, A, pi, d_lens, (double*)v, U, T, N, nflag, epoch);
                else
                    LN_LAUNCH(ctx, "viterbi_reg1b_n128_u1", (viterbi_reg1b<2, 8, 16, 1, 4>), dim3(U), dim3(512), 0,
                              (const double*)b, A, pi, d_lens, (double*)v, U, T, N, nflag, epoch);
            } else {
                if (chk)
                    LN_LAUNCH(ctx, "viterbi_reg1b_n128", (viterbi_reg1b<2, 8, 16, 2, 4, double, true, true>), dim3((U + 1) / 2),
                              dim3(512), 0, (const double*)b, A, pi, d_lens, (double*)v, U, T, N, nflag, epoch);
                else
                    LN_LAUNCH(ctx, "viterbi_reg1b_n128", (viterbi_reg1b<2, 8, 16, 2, 4, double, true>), dim3((U + 1) / 2), dim3(512), 0,
                              (const double*)b, A, pi, d_lens, (double*)v, U, T, N, nflag, epoch);
            }
            return 0;
        }
#undef LN_VIT
    }
    if (dtype == LNASR_F32 && N <= dispatch::kRegMaxN && !sw().no_regvit && !sw().f32_generic) {
        // precision=np.float32: the same register-resident kernel, fp64 arithmetic, float storage and rounding
        if (chk) *b_tested = true;
        if (U < dispatch::kViterbiPairFromU) {
            if (chk)
                LN_LAUNCH(ctx, "viterbi_reg1b_n128_u1_f32", (viterbi_reg1b<2, 8, 16, 1, 4, float, false, true>), dim3(U), dim3(512),
                          0, (const float*)b, A, pi, d_lens, (float*)v, U, T, N, nflag, epoch);
            else
                LN_LAUNCH(ctx, "viterbi_reg1b_n128_u1_f32", (viterbi_reg1b<2, 8, 16, 1, 4, float>), dim3(U), dim3(512), 0,
                          (const float*)b, A, pi, d_lens, (float*)v, U, T, N, nflag, epoch);
        } else {
            if (chk)
                LN_LAUNCH(ctx, "viterbi_reg1b_n128_f32", (viterbi_reg1b<2, 8, 16, 2, 4, float, true, true>), dim3((U + 1) / 2),
                          dim3(512), 0, (const float*)b, A, pi, d_lens, (float*)v, U, T, N, nflag, epoch);
            else
                LN_LAUNCH(ctx, "viterbi_reg1b_n128_f32", (viterbi_reg1b<2, 8, 16, 2, 4, float, true>), dim3((U + 1) / 2), dim3(512), 0,
                          (const float*)b, A, pi, d_lens, (float*)v, U, T, N, nflag, epoch);
        }
        return 0;
    }
    constexpr int UB = 4;
    using namespace dispatch;
    if (N > kRegMaxN && N <= kOnChipMaxN && !sw().no_t256 && (dtype == LNASR_F64 || !sw().f32_generic))
        return trellis256_device(ctx, 0, b, U, T, d_lens, dtype, v, nullptr);    // float: float storage and rounding
    if (dtype == LNASR_F64 && N > kOnChipMaxN && viterbi_framepar(N, U))
        return viterbi_frames_device(ctx, (const double*)b, U, T, d_lens, (double*)v);
    if (dtype == LNASR_F64 && N > kRegMaxN && N <= kStreamMaxN && !sw().no_stream)
        return viterbi_stream_device(ctx, (const double*)b, U, T, d_lens, (double*)v);
    int nt = ((N + 63) / 64) * 64;
    if (nt > 1024) nt = 1024;
    const size_t shmem = (size_t)2 * UB * N * 8;
    LN_REQUIRE(shmem <= 160 * 1024, "N=%d too large for the Viterbi kernel", N);
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "viterbi_generic_f32", (viterbi_generic<float, UB>), dim3((U + UB - 1) / UB), dim3(nt), shmem,
                  (const float*)b, A, pi, d_lens, (float*)v, U, T, N);
    else
        LN_LAUNCH(ctx, "viterbi_generic", (viterbi_generic<double, UB>), dim3((U + UB - 1) / UB), dim3(nt), shmem,
                  (const double*)b, A, pi, d_lens, (double*)v, U, T, N);
    return 0;
}

}  // namespace lnasr
