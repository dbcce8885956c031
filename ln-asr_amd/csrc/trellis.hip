// HMM trellises: Viterbi (+ lazy back-pointers / backtrace), forward, backward, Baum-Welch
// posteriors (hmm.py:47-200, 268-283).
//
// Parallelism: utterances are independent; the frame recursion is sequential.  A block owns
// UB utterances and walks their frames in lock-step (one or two barriers per frame).
//
// Viterbi keeps NO back-pointers in the recursion: v[t] is stored and the pointer actually
// needed, bt[t+1, path[t+1]] = argmax_i(v[t,i] + A[i, path[t+1]]) (hmm.py:165,171), is
// recomputed during the backtrace with the same fp64 additions, so the result is bit-identical
// while the inner loop is one add + one max per (i, j).  The full bt matrix is produced by a
// separate, fully parallel kernel only when the caller asks for it.
#include <utility>

#include "common.h"
#include "devmath.h"

namespace lnasr {

int emit_device(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm);

__global__ void fill_lens_kernel(int32_t* lens, int U, int T) {
    int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < U) lens[u] = T;
}

// HMM._map_b (hmm.py:47-61): b[u,t,:] = B[:, obs[u,t]]
template <typename TS>
__global__ void discrete_map_b_kernel(const double* __restrict__ logB, int N, int K, const int32_t* __restrict__ obs,
                                      long F, TS* __restrict__ b) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F * N) return;
    long f = idx / N;
    int j = (int)(idx % N);
    int k = obs[f];
    b[idx] = (TS)((k >= 0 && k < K) ? logB[(size_t)j * K + k] : __builtin_nan(""));
}

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

// ------------------------------------------------------------------------------------------
// backtrace: one wave per utterance
// ------------------------------------------------------------------------------------------
template <typename TS>
__global__ void backtrace_kernel(const TS* __restrict__ v, const double* __restrict__ AT,
                                 const int32_t* __restrict__ lens, uint32_t* __restrict__ path,
                                 double* __restrict__ best, int U, int T, int N) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = blockIdx.x * (blockDim.x >> 6) + w;
    if (u >= U) return;
    const int L = lens[u];
    if (L <= 0) return;
    const TS* vu = v + (size_t)u * T * N;
    uint32_t* pu = path + (size_t)u * T;
    double bv = LN_NEG_INF;
    int bi = kNoIndex;
    for (int j = lane; j < N; j += 64) argmax_combine(bv, bi, ld(vu, (size_t)(L - 1) * N + j), j);
    wave_argmax(bv, bi);
    int p = bi;
    if (lane == 0) {
        pu[L - 1] = (uint32_t)p;
        if (best) best[u] = bv;
    }
    if (N <= 128) {
        // two values per lane, next row prefetched while the dependent argmax chain runs
        const int j0 = lane, j1 = lane + 64;
        double n0 = LN_NEG_INF, n1 = LN_NEG_INF;
        if (L >= 2) {
            if (j0 < N) n0 = ld(vu, (size_t)(L - 2) * N + j0);
            if (j1 < N) n1 = ld(vu, (size_t)(L - 2) * N + j1);
        }
        for (int t = L - 2; t >= 0; --t) {
            const double c0 = n0, c1 = n1;
            if (t > 0) {
                if (j0 < N) n0 = ld(vu, (size_t)(t - 1) * N + j0);
                if (j1 < N) n1 = ld(vu, (size_t)(t - 1) * N + j1);
            }
            const double* col = AT + (size_t)p * N;
            double x = LN_NEG_INF;
            int xi = kNoIndex;
            if (j0 < N) { x = c0 + col[j0]; xi = j0; }
            if (j1 < N) argmax_combine(x, xi, c1 + col[j1], j1);
            wave_argmax(x, xi);
            p = xi;
            if (lane == 0) pu[t] = (uint32_t)p;
        }
    } else {
        for (int t = L - 2; t >= 0; --t) {
            const double* col = AT + (size_t)p * N;
            double x = LN_NEG_INF;
            int xi = kNoIndex;
            for (int j = lane; j < N; j += 64) argmax_combine(x, xi, ld(vu, (size_t)t * N + j) + col[j], j);
            wave_argmax(x, xi);
            p = xi;
            if (lane == 0) pu[t] = (uint32_t)p;
        }
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
template <int CPT, int NW, int RPT, int UB, int WPE, typename TS = double, bool ALT = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void viterbi_reg1b(
    const TS* __restrict__ b, const double* __restrict__ A, const double* __restrict__ pi,
    const int32_t* __restrict__ lens, TS* __restrict__ v, int U, int T, int N) {
    constexpr int NP = 64 * CPT;
    __shared__ double Vs[3][UB][NP];
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
                val = rnd<TS>(pi[j] + ld(b, o));
                st(v, o, val);
            }
            Vs[0][k][j] = val;
            Vs[1][k][j] = LN_NEG_INF;
            Vs[2][k][j] = LN_NEG_INF;
        }
    }
    // emission streams of this wave's columns (every wave needs b for all UB*CPT column blocks):
    // one per-lane pointer per column block, utterance/frame offsets stay scalar
    const TS* pl[CPT];
    size_t uoff[UB];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int j = lane + 64 * c;
        pl[c] = b + (j < N ? j : N - 1);
    }
#pragma unroll
    for (int k = 0; k < UB; ++k) uoff[k] = (size_t)((u0 + k < U) ? u0 + k : U - 1) * T;
    double bnext[UB][CPT];
#pragma unroll
    for (int k = 0; k < UB; ++k)
#pragma unroll
        for (int c = 0; c < CPT; ++c) bnext[k][c] = ld(pl[c], (uoff[k] + (T > 1 ? 1 : 0)) * N);
    // store duty: (utterance, column block) q = w + jj*NW written to global one frame late
    constexpr int NJOB = CPT * UB;
    __syncthreads();

    int cur = 0;                                   // buffer holding v[t-1]
    for (int t = 1; t < maxlen; ++t) {
        if (ALT && (t & 7) == 0) alternate_priority(cls);     // devmath.h
        const int nxt = cur == 2 ? 0 : cur + 1, rst = nxt == 2 ? 0 : nxt + 1;
        double bcur[UB][CPT];
        const int tn = t + 1 < T ? t + 1 : T - 1;
#pragma unroll
        for (int k = 0; k < UB; ++k)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                bcur[k][c] = bnext[k][c];
                bnext[k][c] = ld(pl[c], (uoff[k] + tn) * N);
            }
        // v[t-1] of the columns this wave stores to global (complete since the last barrier)
        if (t > 1 && w < NJOB) {
            const int k = w / CPT, c = w % CPT, j = lane + 64 * c;
            if (j < N && t - 1 < len[k]) st(v, ((size_t)(u0 + k) * T + (t - 1)) * N + j, Vs[cur][k][j]);
        }
        // reset the buffer that frame t+1 will accumulate into
        if (w == NW - 1) {
#pragma unroll
            for (int k = 0; k < UB; ++k)
#pragma unroll
                for (int c = 0; c < CPT; ++c) Vs[rst][k][lane + 64 * c] = LN_NEG_INF;
        }
        double acc[UB][CPT];
        constexpr int HR = RPT / 2;                // two half slices keep the live v values to UB*HR doubles
#pragma unroll
        for (int h = 0; h < 2; ++h) {
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
                        acc[k][c] = (h == 0 && r == 0) ? cand : vmax_f64(acc[k][c], cand);
                    }
        }
#pragma unroll
        for (int k = 0; k < UB; ++k)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                __hip_atomic_fetch_max(&Vs[nxt][k][lane + 64 * c], rnd<TS>(acc[k][c] + bcur[k][c]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        cur = nxt;
    }
    // last frame
    if (maxlen > 1 && w < NJOB) {
        const int k = w / CPT, c = w % CPT, j = lane + 64 * c;
        if (j < N && maxlen - 1 < len[k]) st(v, ((size_t)(u0 + k) * T + (maxlen - 1)) * N + j, Vs[cur][k][j]);
    }
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

template <typename TS, int NMAX>
__global__ __launch_bounds__(256) void backtrace_tiny_kernel(const TS* __restrict__ v, const uint8_t* __restrict__ bp,
                                                             const int32_t* __restrict__ lens, uint32_t* __restrict__ path,
                                                             double* __restrict__ best, int U, int T, int N) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (u >= U) return;
    const int L = lens[u];
    if (L <= 0) return;
    // last frame: first index among equal maxima
    const double x = lane < N ? ld(v, ((size_t)u * T + (L - 1)) * N + lane) : LN_NEG_INF;
    const double mx = wave_max_dpp(x);
    const unsigned long long mk = __ballot(x == mx);
    int p = __builtin_amdgcn_readfirstlane(mk ? __ffsll((long long)mk) - 1 : 0);
    if (lane == 0 && best) best[u] = mx;
    uint32_t* pu = path + (size_t)u * T;
    const uint32_t* rows = reinterpret_cast<const uint32_t*>(bp + (size_t)u * T * NMAX);   // NMAX / 4 words per frame
    constexpr int WPR = NMAX / 4;
    uint32_t mine = 0;
    if (lane == ((L - 1) & 63)) mine = (uint32_t)p;
    if (((L - 1) & 63) == 0) pu[L - 1] = (uint32_t)p;
    // frames are walked 64 at a time: lane l holds the back-pointer row of frame tb - l; path[t - 1] = row(t)[p]
    for (int tb = L - 1; tb >= 1; tb -= 64) {
        uint32_t w[WPR];
        const int tl = tb - lane > 0 ? tb - lane : 0;
#pragma unroll
        for (int q = 0; q < WPR; ++q) w[q] = rows[(size_t)tl * WPR + q];
        auto step = [&](int k) {
            uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)w[0], k);
#pragma unroll
            for (int q = 1; q < WPR; ++q) {
                const uint32_t wq = (uint32_t)__builtin_amdgcn_readlane((int)w[q], k);
                word = (p >> 2) == q ? wq : word;
            }
            p = (int)((word >> (8 * (p & 3))) & 0xffu);
            const int tp = tb - k - 1;
            if (lane == (tp & 63)) mine = (uint32_t)p;
            if ((tp & 63) == 0 && tp + lane < L) pu[tp + lane] = mine;          // frames tp .. tp + 63 are complete
        };
        if (tb >= 64) {                                   // a whole chunk: 64 steps, lanes known at compile time
#pragma unroll
            for (int k = 0; k < 64; ++k) step(k);
        } else {
            for (int k = 0; k < tb; ++k) step(k);         // frames tb .. 1
        }
    }
}

// Backtrace for N <= 128 (fp64): A^T lives in LDS so the only dependent memory access per step is
// an LDS row read; arg-max = DPP wave max, then the first lane holding the max (ballot + ffs),
// which is numpy.argmax's first-index rule because lane l holds states l and l+64.  Decoded
// states are collected 64 frames at a time and written as one coalesced 256-byte row.
template <int CPL, bool LDSAT, typename TS = double, bool FULL = false>
__global__ __launch_bounds__(256) void backtrace_lds_kernel(const TS* __restrict__ v,
                                                           const double* __restrict__ AT,
                                                           const int32_t* __restrict__ lens,
                                                           uint32_t* __restrict__ path, double* __restrict__ best,
                                                           int U, int T, int N, unsigned* __restrict__ suspect_flag,
                                                           unsigned epoch) {
    // lane l holds states l, l+64, ... (CPL of them).  LDSAT: A^T [N][N] (ATs[p*N + i] = A[i][p]) lives in LDS
    // (N <= 128); otherwise the one row a step needs is read from L2 (128 < N <= 256: 512 KB does not fit)
    extern __shared__ double ATs[];
    if (LDSAT) {
        for (int e = threadIdx.x; e < N * N; e += blockDim.x) ATs[e] = AT[e];
        __syncthreads();
    }
    const double* ATb = LDSAT ? ATs : AT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = blockIdx.x * (blockDim.x >> 6) + w;
    if (u >= U) return;
    const int L = lens[u];
    if (L <= 0) return;
    const TS* vu = v + (size_t)u * T * N;
    uint32_t* pu = path + (size_t)u * T;
    bool ok[CPL];                                            // FULL: N = 64 CPL, every lane owns CPL real states and the
    int jj[CPL];                                             // selects below fold away; else clamped: every load is a valid address
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        ok[c] = FULL || lane + 64 * c < N;
        jj[c] = ok[c] ? lane + 64 * c : 0;
    }
    // first index among equal maxima (numpy.argmax): lowest column block first, lowest lane within it
    auto pick = [&](const double (&x)[CPL], double& mx) -> int {
        double m = x[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) m = vmax_f64(m, x[c]);
        mx = wave_max_dpp(m);
        int r = 0;                                            // scalar selects, highest block first: the lowest wins
#pragma unroll
        for (int c = CPL - 1; c >= 0; --c) {
            const unsigned long long mk = __ballot(x[c] == mx);
            r = mk ? 64 * c + __ffsll((long long)mk) - 1 : r;
        }
        return r;
    };
    // A NaN emission leaves NaN or -inf in its cell of v (the Viterbi kernels' maximum drops a NaN operand) where numpy
    // would have turned every later frame into NaN.  This kernel reads every cell anyway and has issue slots to spare:
    // it raises a flag when it meets such a cell, and only then is b scanned for NaNs (nanfix.hip) -- a model with -inf
    // transitions pays that scan for nothing, a dense one never pays it
    unsigned long long suspect = 0;
    double x[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const double v0 = ld(vu, (size_t)(L - 1) * N + jj[c]);
        suspect |= __ballot(!(v0 > LN_NEG_INF));
        x[c] = ok[c] ? v0 : LN_NEG_INF;
    }
    double bestv;
    int p = pick(x, bestv);
    if (lane == 0 && best) best[u] = bestv;
    uint32_t mine = 0;
    if (lane == ((L - 1) & 63)) mine = (uint32_t)p;
    if (((L - 1) & 63) == 0) pu[L - 1] = (uint32_t)p;      // single-entry top block
    // v rows are streamed from HBM: keep PF frames in flight per wave (register ring) so that the
    // dependent chain below only ever waits on the A^T row
    constexpr int PF = CPL <= 2 ? 8 : (CPL <= 4 ? 4 : (CPL <= 8 ? 2 : 1));
    double cur[PF][CPL], nxt[PF][CPL];
    auto load_rows = [&](int tstart, double (&r)[PF][CPL]) {     // frames tstart, tstart-1, ... (clamped at 0)
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int tt = tstart - k > 0 ? tstart - k : 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) r[k][c] = ld(vu, (size_t)tt * N + jj[c]);
        }
    };
    load_rows(L - 2 > 0 ? L - 2 : 0, cur);
    for (int tb = L - 2; tb >= 0; tb -= PF) {
        load_rows(tb - PF > 0 ? tb - PF : 0, nxt);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = tb - k;
            if (t >= 0) {                                   // wave-uniform
                const double* col = ATb + (size_t)p * N;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const double sum = cur[k][c] + col[jj[c]];          // unconditional (clamped index), then a select
                    suspect |= __ballot(!(cur[k][c] > LN_NEG_INF));
                    x[c] = ok[c] ? sum : LN_NEG_INF;
                }
                double mx;
                p = pick(x, mx);
                if (lane == (t & 63)) mine = (uint32_t)p;
                if ((t & 63) == 0) {
                    // frames t .. min(t+63, L-1) are complete: lane l holds frame t + l
                    if (t + lane < L) pu[t + lane] = mine;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k)
#pragma unroll
            for (int c = 0; c < CPL; ++c) cur[k][c] = nxt[k][c];
    }
    if (suspect != 0 && lane == 0) *suspect_flag = epoch;
}

// full back-pointer matrix (hmm.py:165): bt[u,t,j] = argmax_i(v[u,t-1,i] + A[i,j]); bt[u,0,:] = 0
template <typename TS>
__global__ void bt_kernel(const TS* __restrict__ v, const double* __restrict__ A, const int32_t* __restrict__ lens,
                          uint32_t* __restrict__ bt, int T, int N) {
    extern __shared__ double vs[];            // [N]
    const int u = blockIdx.y, t = blockIdx.x;
    if (t >= lens[u]) return;
    uint32_t* out = bt + ((size_t)u * T + t) * N;
    if (t == 0) {
        for (int j = threadIdx.x; j < N; j += blockDim.x) out[j] = 0;
        return;
    }
    const TS* vp = v + ((size_t)u * T + (t - 1)) * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) vs[i] = ld(vp, i);
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        double best = vs[0] + A[j];
        int bi = 0;
        for (int i = 1; i < N; ++i) {
            double val = vs[i] + A[(size_t)i * N + j];
            // numpy.argmax: the first NaN wins over everything (v repaired by viterbi_nan_repair holds the reference's NaNs)
            if (val > best || (val != val && best == best)) { best = val; bi = i; }
        }
        out[j] = (uint32_t)bi;
    }
}

// ------------------------------------------------------------------------------------------
// forward / backward (log domain)
// ------------------------------------------------------------------------------------------
// alpha[t,j] = lse_i(alpha[t-1,i] + A[i,j]) + b[t,j]                       (hmm.py:97-99)
// Evaluated as s + log(sum_i exp(alpha[t-1,i] - s) * expA[i,j]) with s = max_i alpha[t-1,i]:
// one exp per state per frame and an FMA per (i,j) instead of a transcendental per (i,j).
// Columns whose linear-domain sum falls below kSafeSum (every contributor underflowed, or the
// column is unreachable) are redone with the exact max-shifted log-domain sum, so the result
// equals the reference's logaddexp fold to rounding for every input, including -inf entries.
constexpr double kSafeSum = 1e-290;

template <typename TS, bool BACKWARD>
__global__ void fwdbwd_generic(const TS* __restrict__ b, const double* __restrict__ Alog,
                               const double* __restrict__ Aexp, const double* __restrict__ pi,
                               const int32_t* __restrict__ lens, TS* __restrict__ out, double* __restrict__ loglik,
                               int U, int T, int N) {
    // FORWARD : Alog = logA [i][j] (row i, col j), Aexp = expA;  thread j sums over i: Aexp[i*N+j]
    // BACKWARD: Alog = logA^T [j][i], Aexp = expA^T;             thread i sums over j: Aexp[j*N+i]
    extern __shared__ double sm[];
    double* prev = sm;                 // [N]  previous row (forward: alpha[t-1]; backward: beta[t+1]+b[t+1])
    double* ex = sm + N;               // [N]  exp(prev - s)
    double* scratch = sm + 2 * N;      // [32]
    const int u = blockIdx.x;
    const int L = lens[u];
    if (L <= 0) return;
    const TS* bu = b + (size_t)u * T * N;
    TS* ou = out ? out + (size_t)u * T * N : nullptr;

    if (!BACKWARD) {
        for (int j = threadIdx.x; j < N; j += blockDim.x) {
            double val = rnd<TS>(pi[j] + ld(bu, j));
            if (ou) st(ou, j, val);
            prev[j] = val;
        }
    } else {
        for (int j = threadIdx.x; j < N; j += blockDim.x) {
            if (ou) st(ou, (size_t)(L - 1) * N + j, 0.0);
            if (L >= 2) prev[j] = 0.0 + ld(bu, (size_t)(L - 1) * N + j);
        }
    }
    __syncthreads();
    const int steps = L - 1;
    for (int sidx = 0; sidx < steps; ++sidx) {
        const int t = BACKWARD ? (L - 2 - sidx) : (1 + sidx);
        double lm = LN_NEG_INF;
        for (int i = threadIdx.x; i < N; i += blockDim.x) lm = fmax(lm, prev[i]);
        const double s = block_max(lm, scratch);
        for (int i = threadIdx.x; i < N; i += blockDim.x) ex[i] = (s == LN_NEG_INF) ? 0.0 : exp(prev[i] - s);
        __syncthreads();
        double newv[4];                // up to 4 states per thread (N <= 4 * blockDim)
        int cnt = 0;
        for (int j = threadIdx.x; j < N; j += blockDim.x, ++cnt) {
            double sum = 0.0;
            for (int i = 0; i < N; ++i) sum += ex[i] * Aexp[(size_t)i * N + j];
            double val;
            if (sum >= kSafeSum && s != LN_NEG_INF) {
                val = s + log(sum);
            } else {
                double mx = LN_NEG_INF;
                for (int i = 0; i < N; ++i) mx = fmax(mx, prev[i] + Alog[(size_t)i * N + j]);
                if (mx == LN_NEG_INF) {
                    val = mx;
                } else {
                    double s2 = 0.0;
                    for (int i = 0; i < N; ++i) s2 += exp(prev[i] + Alog[(size_t)i * N + j] - mx);
                    val = mx + log(s2);
                }
            }
            if (!BACKWARD) {
                val = rnd<TS>(val + ld(bu, (size_t)t * N + j));
                if (ou) st(ou, (size_t)t * N + j, val);
                newv[cnt] = val;
            } else {
                val = rnd<TS>(val);
                if (ou) st(ou, (size_t)t * N + j, val);
                newv[cnt] = val + ld(bu, (size_t)t * N + j);     // feeds step t-1: beta[t] + b[t]
            }
        }
        __syncthreads();
        cnt = 0;
        for (int j = threadIdx.x; j < N; j += blockDim.x, ++cnt) prev[j] = newv[cnt];
        __syncthreads();
    }
    if (!BACKWARD && loglik) {
        // calc_prob: lse(alpha[L-1])  (hmm.py:274)
        double lm = LN_NEG_INF;
        for (int i = threadIdx.x; i < N; i += blockDim.x) lm = fmax(lm, prev[i]);
        const double s = block_max(lm, scratch);
        double ls = 0.0;
        for (int i = threadIdx.x; i < N; i += blockDim.x) ls += (s == LN_NEG_INF) ? 0.0 : exp(prev[i] - s);
        const double tot = block_sum(ls, scratch);
        if (threadIdx.x == 0) loglik[u] = (s == LN_NEG_INF) ? s : s + log(tot);
    }
}

// ------------------------------------------------------------------------------------------
// forward / backward, register-resident exp(A) (N <= 64*CPT, N <= NW*RPT; fp64 storage)
// ------------------------------------------------------------------------------------------
// Same blocking as viterbi_reg: thread (wave w, lane l) keeps expA[w*RPT + r][l + 64c] (forward) or
// expA^T (backward) in registers, UB utterances walk in lock-step, row slices are combined through
// LDS and the CPT*UB (utterance, column-block) jobs are finalised by one wave each:
//     prev_t[j] = S + log(sum_i e[i] * expA[i,j])  (+ b[t,j] for forward)
//     e_t[j]    = exp(prev_t[j] - S'),   S' = max_j prev_{t-1}[j]
// The shift S' lags one frame (the max of the frame just consumed), so no extra block-wide
// reduction sits on the critical path: any shift is exact in floating point as long as nothing
// over/underflows, and sums outside [1e-290, 1e290] (or NaN) fall back to the exact max-shifted
// log-domain evaluation of that column, as in fwdbwd_generic.
template <bool BACKWARD, int CPT, int NW, int RPT, int UB, int WPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void fwdbwd_reg(
    const double* __restrict__ b, const double* __restrict__ Aexp, const double* __restrict__ Alog,
    const double* __restrict__ pi, const int32_t* __restrict__ lens, double* __restrict__ out,
    double* __restrict__ loglik, int U, int T, int N) {
    constexpr int NP = 64 * CPT;
    constexpr int NJOB = CPT * UB;
    constexpr int JPW = (NJOB + NW - 1) / NW;
    __shared__ double Es[2][UB][NP];      // exp(prev - shift) of the frame being consumed
    __shared__ double Cs[2][UB][NP];      // prev itself (log domain) for the exact fallback
    __shared__ double Ps[NW > 1 ? NW : 1][UB][NP];
    __shared__ double Ms[2][UB][CPT];     // per column-block max of prev
    __shared__ double Ss[2][UB];          // shift the e values of this buffer were computed with
    __shared__ double etab[kExpTab];      // table-driven exp / log (devmath.h)
    __shared__ double ltab[2 * kLogTab];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    exp_table_init(etab, threadIdx.x, 64 * NW);
    log_table_init(ltab, threadIdx.x, 64 * NW);   // visible after the first __syncthreads below
    const int u0 = blockIdx.x * UB;

    double areg[CPT][RPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = w * RPT + r, j = lane + 64 * c;
            areg[c][r] = (i < N && j < N) ? Aexp[(size_t)i * N + j] : 0.0;
        }

    int len[UB], maxlen = 0;
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        len[k] = (u0 + k < U) ? lens[u0 + k] : 0;
        maxlen = len[k] > maxlen ? len[k] : maxlen;
    }
    // jobs of this wave
    int jk[JPW], jc[JPW], jlen[JPW];
    const double* bp[JPW];
    double* op[JPW];
    bool jvalid[JPW];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int q = w + jj * NW;
        jvalid[jj] = q < NJOB;
        jk[jj] = jvalid[jj] ? q / CPT : 0;
        jc[jj] = q % CPT;
        const int uu = (u0 + jk[jj] < U) ? u0 + jk[jj] : U - 1;
        const int j = lane + 64 * jc[jj];
        bp[jj] = b + (size_t)uu * T * N + (j < N ? j : N - 1);
        op[jj] = out ? out + (size_t)uu * T * N + (j < N ? j : N - 1) : nullptr;
        jlen[jj] = 0;
#pragma unroll
        for (int k = 0; k < UB; ++k)
            if (k == jk[jj]) jlen[jj] = len[k];
    }
    // frame of utterance-job jj at step s:  forward t = s;  backward t = L-1-s
    auto frame_of = [&](int jj, int step) { return BACKWARD ? jlen[jj] - 1 - step : step; };

    // step 0: prev_0 and its per-block max
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        if (jvalid[jj]) {
            const int k = jk[jj], c = jc[jj], j = lane + 64 * c;
            double prev = LN_NEG_INF;
            if (j < N && jlen[jj] > 0) {
                const int t0 = frame_of(jj, 0);
                const double bv = bp[jj][(size_t)t0 * N];
                if (BACKWARD) {
                    if (op[jj]) op[jj][(size_t)t0 * N] = 0.0;
                    prev = 0.0 + bv;
                } else {
                    prev = pi[j] + bv;
                    if (op[jj]) op[jj][0] = prev;
                }
            }
            Cs[0][k][j] = prev;
            const double m = wave_max_dpp(prev);
            if (lane == 0) Ms[0][k][c] = m;
        }
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        if (jvalid[jj]) {
            const int k = jk[jj], c = jc[jj], j = lane + 64 * c;
            double S = Ms[0][k][0];
#pragma unroll
            for (int cc = 1; cc < CPT; ++cc) S = fmax(S, Ms[0][k][cc]);
            const double prev = Cs[0][k][j];
            Es[0][k][j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
            if (c == 0 && lane == 0) Ss[0][k] = S;
        }
    }
    double bnext[JPW];
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        int t1 = frame_of(jj, 1);
        t1 = t1 < 0 ? 0 : (t1 > T - 1 ? T - 1 : t1);
        bnext[jj] = jvalid[jj] ? bp[jj][(size_t)t1 * N] : 0.0;
    }
    __syncthreads();

    for (int step = 1; step < maxlen; ++step) {
        const int cur = (step - 1) & 1, nxt = step & 1;
        double bcur[JPW];
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            bcur[jj] = bnext[jj];
            int tn = frame_of(jj, step + 1);
            tn = tn < 0 ? 0 : (tn > T - 1 ? T - 1 : tn);
            if (jvalid[jj]) bnext[jj] = bp[jj][(size_t)tn * N];
        }
        // the wave's RPT e-values of an utterance sit in RPT/16 VGPR pairs, 16 per DPP row (lane l holds
        // e[16 g + (l & 15)]); v_fmac_f64_dpp row_newbcast:n feeds value n to every lane, so a transition is
        // ONE vector instruction and the LDS sees one read per utterance instead of one per row (the broadcast
        // reads made this kernel LDS-bound: 64 of them per frame against 128 FMAs)
        static_assert(RPT % 16 == 0, "row slices are whole DPP rows");
        double ek[UB][RPT / 16];
#pragma unroll
        for (int k = 0; k < UB; ++k)
#pragma unroll
            for (int g = 0; g < RPT / 16; ++g) ek[k][g] = Es[cur][k][w * RPT + 16 * g + (lane & 15)];
        double acc[UB][CPT];
#pragma unroll
        for (int k = 0; k < UB; ++k)
#pragma unroll
            for (int c = 0; c < CPT; ++c) acc[k][c] = 0.0;
        static_for<0, RPT>([&](auto r) {
#pragma unroll
            for (int k = 0; k < UB; ++k)
#pragma unroll
                for (int c = 0; c < CPT; ++c) fmac_bcast<(r.value & 15)>(acc[k][c], ek[k][r.value >> 4], areg[c][r.value]);
        });
        if (NW > 1) {
#pragma unroll
            for (int k = 0; k < UB; ++k)
#pragma unroll
                for (int c = 0; c < CPT; ++c) Ps[w][k][lane + 64 * c] = acc[k][c];
            __syncthreads();
        }
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            if (jvalid[jj]) {
                const int k = jk[jj], c = jc[jj], j = lane + 64 * c;
                if (step < jlen[jj]) {
                    double sum;
                    if (NW > 1) {
                        sum = Ps[0][k][j];
#pragma unroll
                        for (int ww = 1; ww < NW; ++ww) sum += Ps[ww][k][j];
                    } else {
                        sum = acc[k][c];
                    }
                    const double Sc = Ss[cur][k];
                    double val;
                    if (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF) {
                        val = Sc + log_sum(sum, ltab);
                    } else {                       // exact log-domain column (rare)
                        val = exact_column_lse(&Cs[cur][k][0], Alog + (j < N ? j : N - 1), N);
                    }
                    const int t = frame_of(jj, step);
                    double prev;
                    if (BACKWARD) {
                        if (j < N && op[jj]) op[jj][(size_t)t * N] = val;
                        prev = val + bcur[jj];
                    } else {
                        prev = val + bcur[jj];
                        if (j < N && op[jj]) op[jj][(size_t)t * N] = prev;
                    }
                    if (j >= N) prev = LN_NEG_INF;
                    double S = Ms[cur][k][0];
#pragma unroll
                    for (int cc = 1; cc < CPT; ++cc) S = fmax(S, Ms[cur][k][cc]);
                    Cs[nxt][k][j] = prev;
                    Es[nxt][k][j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
                    const double m = wave_max_dpp(prev);
                    if (lane == 0) Ms[nxt][k][c] = m;
                    if (c == 0 && lane == 0) Ss[nxt][k] = S;
                } else {                           // this utterance has ended: carry its state
                    Cs[nxt][k][j] = Cs[cur][k][j];
                    Es[nxt][k][j] = Es[cur][k][j];
                    if (lane == 0) Ms[nxt][k][c] = Ms[cur][k][c];
                    if (c == 0 && lane == 0) Ss[nxt][k] = Ss[cur][k];
                }
            }
        }
        __syncthreads();
    }
    // calc_prob: lse_j alpha[L-1, j]  (hmm.py:274)
    if (!BACKWARD && loglik && w == 0) {
        const int fin = (maxlen > 0 ? maxlen - 1 : 0) & 1;
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            if (u0 + k < U && len[k] > 0) {
                double mx = Ms[fin][k][0];
#pragma unroll
                for (int cc = 1; cc < CPT; ++cc) mx = fmax(mx, Ms[fin][k][cc]);
                double sacc = 0.0;
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    const double pv = Cs[fin][k][lane + 64 * c];
                    sacc += (mx == LN_NEG_INF || pv == LN_NEG_INF) ? 0.0 : exp(pv - mx);
                }
                sacc = wave_sum(sacc);
                if (lane == 0) loglik[u0 + k] = (mx == LN_NEG_INF) ? mx : mx + log(sacc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward / backward, 64 < N <= 128, four utterances per block: COLUMN-owned blocking.
// Wave w owns the 16 columns j = 16 w + (lane & 15); the four 16-lane DPP rows of the wave split the
// 128 source states into groups of 32, so lane (g, c) keeps expA[32 g + r][16 w + c], r < 32, in
// registers.  A transition is still one v_fmac_f64_dpp (the e value comes from lane n of the row), but
// the row-group partial sums of a column now sit in the SAME wave: two v_permlane32_swap and one
// v_permlane16_swap (gfx950) turn the four utterances' partials into "row k holds utterance k's sum",
// so the whole finalisation (log, emission, store, exp) of 4 utterances x 16 columns runs on all 64
// lanes of the wave that produced the sums -- no partial sums through the LDS, ONE barrier per frame
// (fwdbwd_reg: partials through the LDS, two barriers).  Same shift rule, same exact fallback.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap32_f64(double& a, double& b) {     // a[63:32] <-> b[31:0]
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap16_f64(double& a, double& b) {     // odd rows of a <-> even rows of b
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double row_max_dpp(double x) {             // max over the 16 lanes of a DPP row
    x = vmax_f64(x, dpp_f64<0xB1>(x));
    x = vmax_f64(x, dpp_f64<0x4E>(x));
    x = vmax_f64(x, dpp_f64<0x141>(x));
    return vmax_f64(x, dpp_f64<0x140>(x));
}

template <bool BACKWARD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwdbwd_cols(
    const double* __restrict__ b, const double* __restrict__ Aexp, const double* __restrict__ Alog,
    const double* __restrict__ pi, const int32_t* __restrict__ lens, double* __restrict__ out,
    double* __restrict__ loglik, int U, int T, int N) {
    constexpr int NP = 128, UB = 4, NW = 8;
    __shared__ double Es[2][UB][NP];      // exp(prev - shift) of the frame being consumed
    __shared__ double Cs[2][UB][NP];      // prev itself (log domain) for the exact fallback
    __shared__ double Ms[2][UB][NW];      // per wave (16 columns) max of prev
    __shared__ double Ss[2][UB];          // shift the e values of this buffer were computed with
    __shared__ double etab[kExpTab];
    __shared__ double ltab[2 * kLogTab];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    exp_table_init(etab, threadIdx.x, 64 * NW);
    log_table_init(ltab, threadIdx.x, 64 * NW);   // visible after the first __syncthreads below
    const int u0 = blockIdx.x * UB;
    const int g = lane >> 4, c = lane & 15;
    const int j = 16 * w + c;                      // the lane's column, products and finalisation alike
    const int jc = j < N ? j : N - 1;

    double areg[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int i = 32 * g + r;
        areg[r] = (i < N && j < N) ? Aexp[(size_t)i * N + j] : 0.0;
    }
    int maxlen = 0;
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int l = (u0 + k < U) ? lens[u0 + k] : 0;
        maxlen = l > maxlen ? l : maxlen;
    }
    // finalisation role: row g of the wave is utterance k = g
    const int k = g;
    const int uu = (u0 + k < U) ? u0 + k : U - 1;
    const int klen = (u0 + k < U) ? lens[u0 + k] : 0;
    const double* bp = b + (size_t)uu * T * N + jc;
    double* op = out ? out + (size_t)uu * T * N + jc : nullptr;
    auto frame_of = [&](int step) { return BACKWARD ? klen - 1 - step : step; };
    auto clampt = [&](int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };

    // step 0
    {
        double prev = LN_NEG_INF;
        if (j < N && klen > 0) {
            const int t0 = frame_of(0);
            const double bv = bp[(size_t)t0 * N];
            if (BACKWARD) {
                if (op) op[(size_t)t0 * N] = 0.0;
                prev = 0.0 + bv;
            } else {
                prev = pi[j] + bv;
                if (op) op[0] = prev;
            }
        }
        Cs[0][k][j] = prev;
        const double m = row_max_dpp(prev);
        if (c == 0) Ms[0][k][w] = m;
    }
    __syncthreads();
    {
        double S = Ms[0][k][0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) S = fmax(S, Ms[0][k][ww]);
        const double prev = Cs[0][k][j];
        Es[0][k][j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
        if (w == 0 && c == 0) Ss[0][k] = S;
    }
    double bnext = bp[(size_t)clampt(frame_of(1)) * N];
    drain_vmem();                                  // areg resident: no vmcnt guards in the loop
    __syncthreads();

    for (int step = 1; step < maxlen; ++step) {
        const int cur = (step - 1) & 1, nxt = step & 1;
        const double bcur = bnext;
        bnext = bp[(size_t)clampt(frame_of(step + 1)) * N];
        // e values of the four utterances for this lane's row group: lane (g, c) holds e[32 g + 16 q + c]
        double ek[UB][2];
#pragma unroll
        for (int kk = 0; kk < UB; ++kk)
#pragma unroll
            for (int q = 0; q < 2; ++q) ek[kk][q] = Es[cur][kk][32 * g + 16 * q + c];
        double acc[UB];
#pragma unroll
        for (int kk = 0; kk < UB; ++kk) acc[kk] = 0.0;
        static_for<0, 32>([&](auto r) {
#pragma unroll
            for (int kk = 0; kk < UB; ++kk) fmac_bcast<(r.value & 15)>(acc[kk], ek[kk][r.value >> 4], areg[r.value]);
        });
        // row groups -> utterances: after the swaps row k of `sum` holds utterance k's total,
        // ((g0 + g2) + (g1 + g3)) in every lane
        swap32_f64(acc[0], acc[2]);
        swap32_f64(acc[1], acc[3]);
        double s02 = acc[0] + acc[2], s13 = acc[1] + acc[3];
        swap16_f64(s02, s13);
        const double sum = s02 + s13;

        if (step < klen) {
            const double Sc = Ss[cur][k];
            double val;
            if (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF) {
                val = Sc + log_sum(sum, ltab);
            } else {                               // exact log-domain column (rare)
                val = exact_column_lse(&Cs[cur][k][0], Alog + jc, N);
            }
            const int t = frame_of(step);
            double prev;
            if (BACKWARD) {
                if (j < N && op) op[(size_t)t * N] = val;
                prev = val + bcur;
            } else {
                prev = val + bcur;
                if (j < N && op) op[(size_t)t * N] = prev;
            }
            if (j >= N) prev = LN_NEG_INF;
            double S = Ms[cur][k][0];
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) S = fmax(S, Ms[cur][k][ww]);
            Cs[nxt][k][j] = prev;
            Es[nxt][k][j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
            const double m = row_max_dpp(prev);
            if (c == 0) Ms[nxt][k][w] = m;
            if (w == 0 && c == 0) Ss[nxt][k] = S;
        } else {                                   // this utterance has ended: carry its state
            Cs[nxt][k][j] = Cs[cur][k][j];
            Es[nxt][k][j] = Es[cur][k][j];
            if (c == 0) Ms[nxt][k][w] = Ms[cur][k][w];
            if (w == 0 && c == 0) Ss[nxt][k] = Ss[cur][k];
        }
        __syncthreads();
    }
    // calc_prob: lse_j alpha[L-1, j]  (hmm.py:274)
    if (!BACKWARD && loglik && w == 0) {
        const int fin = (maxlen > 0 ? maxlen - 1 : 0) & 1;
#pragma unroll
        for (int kk = 0; kk < UB; ++kk) {
            if (u0 + kk < U && lens[u0 + kk] > 0) {
                double mx = Ms[fin][kk][0];
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) mx = fmax(mx, Ms[fin][kk][ww]);
                double sacc = 0.0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double pv = Cs[fin][kk][lane + 64 * h];
                    sacc += (mx == LN_NEG_INF || pv == LN_NEG_INF) ? 0.0 : exp(pv - mx);
                }
                sacc = wave_sum(sacc);
                if (lane == 0) loglik[u0 + kk] = (mx == LN_NEG_INF) ? mx : mx + log(sacc);
            }
        }
    }
}

// The same column-owned scheme with TWO utterances per block of FOUR waves (round 3).  A wave owns 32 columns (two per
// lane: 64 transition entries = 128 registers), so a block is one wave per SIMD and two blocks share a CU: two waves per SIMD
// as before, but from blocks that do not share a barrier.  With four utterances behind one barrier all eight waves were in
// the same phase -- SQ counters of fwdbwd_cols: vector issue 63 % of the kernel, 40 % of every wave's time in s_waitcnt (the e
// vector right after the barrier, the log / exp table look-ups of the finalisation) with its SIMD partner waiting on the
// same thing.  Independent blocks drift apart and one's product phase runs under the other's waits (what merging the two
// Viterbi blocks of a CU behind one barrier had shown from the other side: 1.37 -> 2.68 ms).  The four accumulators of a
// lane are now (utterance, column half) instead of four utterances; the swap network is unchanged and leaves combination
// m = 2 k + h in row m: row g of a wave finalises utterance g >> 1, column 32 w + 16 (g & 1) + c.
template <bool BACKWARD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwdbwd_cols2(
    const double* __restrict__ b, const double* __restrict__ Aexp, const double* __restrict__ Alog,
    const double* __restrict__ pi, const int32_t* __restrict__ lens, double* __restrict__ out,
    double* __restrict__ loglik, int U, int T, int N, unsigned* __restrict__ nan_flag, unsigned nan_epoch) {
    constexpr int NP = 128, UB = 2, NW = 4, NM = 2 * NW;   // NM: partial maxima per utterance (wave x column half)
    // a NaN emission: np.logaddexp carries it into every later (earlier) frame, the maxima here drop it.  The value the
    // lanes finalise is NaN exactly then; one compare into the scalar unit per frame raises the flag for nanfix.hip
    unsigned long long saw_nan = 0;
    __shared__ double Es[2][UB][NP];      // exp(prev - shift) of the frame being consumed
    __shared__ double Cs[2][UB][NP];      // prev itself (log domain) for the exact fallback
    __shared__ double Ms[2][UB][NM];      // per DPP row (16 columns) max of prev
    __shared__ double Ss[2][UB];          // shift the e values of this buffer were computed with
    __shared__ double etab[kExpTab];
    __shared__ double ltab[2 * kLogTab];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    exp_table_init(etab, threadIdx.x, 64 * NW);
    log_table_init(ltab, threadIdx.x, 64 * NW);   // visible after the first __syncthreads below
    const int u0 = blockIdx.x * UB;
    const unsigned cls = blockIdx.x >= (gridDim.x + 1) / 2 ? 1u : 0u;      // the two blocks of a CU: alternate_priority
    const int g = lane >> 4, c = lane & 15;

    double areg[2][32];                            // product role: columns 32 w + 16 q + c, source rows 32 g + r
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int i = 32 * g + r, jq = 32 * w + 16 * q + c;
            areg[q][r] = (i < N && jq < N) ? Aexp[(size_t)i * N + jq] : 0.0;
        }
    int maxlen = 0;
#pragma unroll
    for (int kk = 0; kk < UB; ++kk) {
        const int l = (u0 + kk < U) ? lens[u0 + kk] : 0;
        maxlen = l > maxlen ? l : maxlen;
    }
    // finalisation role: row g of the wave is utterance k, column half h
    const int k = g >> 1, h = g & 1;
    const int j = 32 * w + 16 * h + c;
    const int jc = j < N ? j : N - 1;
    const int uu = (u0 + k < U) ? u0 + k : U - 1;
    const int klen = (u0 + k < U) ? lens[u0 + k] : 0;
    const double* bp = b + (size_t)uu * T * N + jc;
    double* op = out ? out + (size_t)uu * T * N + jc : nullptr;
    auto frame_of = [&](int step) { return BACKWARD ? klen - 1 - step : step; };
    auto clampt = [&](int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };

    // step 0
    {
        double prev = LN_NEG_INF;
        if (j < N && klen > 0) {
            const int t0 = frame_of(0);
            const double bv = bp[(size_t)t0 * N];
            if (BACKWARD) {
                if (op) op[(size_t)t0 * N] = 0.0;
                prev = 0.0 + bv;
            } else {
                prev = pi[j] + bv;
                if (op) op[0] = prev;
            }
        }
        saw_nan |= __ballot(prev != prev);
        Cs[0][k][j] = prev;
        const double m = row_max_dpp(prev);
        if (c == 0) Ms[0][k][2 * w + h] = m;
    }
    __syncthreads();
    {
        double S = Ms[0][k][0];
#pragma unroll
        for (int e = 1; e < NM; ++e) S = fmax(S, Ms[0][k][e]);
        const double prev = Cs[0][k][j];
        Es[0][k][j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
        if (w == 0 && h == 0 && c == 0) Ss[0][k] = S;
    }
    double bnext = bp[(size_t)clampt(frame_of(1)) * N];
    drain_vmem();                                  // areg resident: no vmcnt guards in the loop
    __syncthreads();

    for (int step = 1; step < maxlen; ++step) {
        if ((step & 7) == 0) alternate_priority(cls);
        const int cur = (step - 1) & 1, nxt = step & 1;
        const double bcur = bnext;
        bnext = bp[(size_t)clampt(frame_of(step + 1)) * N];
        // e values of both utterances for this lane's row group: lane (g, c) holds e[32 g + 16 q + c]
        double ek[UB][2];
#pragma unroll
        for (int kk = 0; kk < UB; ++kk)
#pragma unroll
            for (int q = 0; q < 2; ++q) ek[kk][q] = Es[cur][kk][32 * g + 16 * q + c];
        double acc[UB][2];                         // [utterance][column half]
#pragma unroll
        for (int kk = 0; kk < UB; ++kk)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[kk][q] = 0.0;
        static_for<0, 32>([&](auto r) {
#pragma unroll
            for (int kk = 0; kk < UB; ++kk)
#pragma unroll
                for (int q = 0; q < 2; ++q) fmac_bcast<(r.value & 15)>(acc[kk][q], ek[kk][r.value >> 4], areg[q][r.value]);
        });
        // row groups -> (utterance, column half): after the swaps row m of `sum` holds combination m = 2 k + h,
        // ((g0 + g2) + (g1 + g3)) in every lane
        swap32_f64(acc[0][0], acc[1][0]);
        swap32_f64(acc[0][1], acc[1][1]);
        double s02 = acc[0][0] + acc[1][0], s13 = acc[0][1] + acc[1][1];
        swap16_f64(s02, s13);
        const double sum = s02 + s13;

        bool prev_nan = false;
        if (step < klen) {
            const double Sc = Ss[cur][k];
            double val;
            if (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF) {
                val = Sc + log_sum(sum, ltab);
            } else {                               // exact log-domain column (rare)
                val = exact_column_lse(&Cs[cur][k][0], Alog + jc, N);
            }
            const int t = frame_of(step);
            double prev;
            if (BACKWARD) {
                if (j < N && op) op[(size_t)t * N] = val;
                prev = val + bcur;
            } else {
                prev = val + bcur;
                if (j < N && op) op[(size_t)t * N] = prev;
            }
            if (j >= N) prev = LN_NEG_INF;
            prev_nan = prev != prev;
            double S = Ms[cur][k][0];
#pragma unroll
            for (int e = 1; e < NM; ++e) S = fmax(S, Ms[cur][k][e]);
            Cs[nxt][k][j] = prev;
            Es[nxt][k][j] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
            const double m = row_max_dpp(prev);
            if (c == 0) Ms[nxt][k][2 * w + h] = m;
            if (w == 0 && h == 0 && c == 0) Ss[nxt][k] = S;
        } else {                                   // this utterance has ended: carry its state
            Cs[nxt][k][j] = Cs[cur][k][j];
            Es[nxt][k][j] = Es[cur][k][j];
            if (c == 0) Ms[nxt][k][2 * w + h] = Ms[cur][k][2 * w + h];
            if (w == 0 && h == 0 && c == 0) Ss[nxt][k] = Ss[cur][k];
        }
        saw_nan |= __ballot(prev_nan);             // outside the branch: the rows of a wave end at different frames
        __syncthreads();
    }
    if (saw_nan != 0 && lane == 0) *nan_flag = nan_epoch;
    // calc_prob: lse_j alpha[L-1, j]  (hmm.py:274)
    if (!BACKWARD && loglik && w == 0) {
        const int fin = (maxlen > 0 ? maxlen - 1 : 0) & 1;
#pragma unroll
        for (int kk = 0; kk < UB; ++kk) {
            if (u0 + kk < U && lens[u0 + kk] > 0) {
                double mx = Ms[fin][kk][0];
#pragma unroll
                for (int e = 1; e < NM; ++e) mx = fmax(mx, Ms[fin][kk][e]);
                double sacc = 0.0;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const double pv = Cs[fin][kk][lane + 64 * hh];
                    sacc += (mx == LN_NEG_INF || pv == LN_NEG_INF) ? 0.0 : exp(pv - mx);
                }
                sacc = wave_sum(sacc);
                if (lane == 0) loglik[u0 + kk] = (mx == LN_NEG_INF) ? mx : mx + log(sacc);
            }
        }
    }
}

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

template <bool BACKWARD, int UB>
__global__ __launch_bounds__(1024) void fwdbwd_stream(const double* __restrict__ b, const double* __restrict__ Aexp,
                                                      const double* __restrict__ Alog, const double* __restrict__ pi,
                                                      const int32_t* __restrict__ lens, double* __restrict__ out,
                                                      double* __restrict__ loglik, int U, int T, int N, int NP, int RS,
                                                      int RPS) {
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
                if (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF) {
                    val = Sc + log(sum);
                } else {
                    const int jl = j < N ? j : N - 1;
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
                                                           double* __restrict__ Mring, int U, int T, int N, int s) {
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
        const double hprev = Mring[(size_t)((s >= 2 ? s - 2 : 0) & 3) * U + u];       // shift of E_{s-1}
        const double hcur = Mring[(size_t)((s - 1) & 3) * U + u];                     // shift for E_s
        double val;
        if (sum >= kSafeSum && sum <= 1e290 && hprev != LN_NEG_INF) {
            val = hprev + log_sum(sum, ltab);
        } else {
            val = exact_column_lse(P + ((size_t)cur * U + u) * N, Alog + j, N);
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
    LN_LAUNCH(ctx, "fwdbwd_fill", fill_value_kernel, dim3((unsigned)((4 * (size_t)U + 255) / 256)), dim3(256), 0, Mring,
              (size_t)4 * U, LN_NEG_INF);
    LN_LAUNCH(ctx, BACKWARD ? "backward_frame0" : "forward_frame0", fwdbwd_frame0a_kernel<BACKWARD>,
              dim3((unsigned)((un + 255) / 256)), dim3(256), 0, b, md.logpi.as<double>(), d_lens, out, P, Mring, U, T, N);
    LN_LAUNCH(ctx, BACKWARD ? "backward_frame0" : "forward_frame0", fwdbwd_frame0b_kernel, dim3((unsigned)((un + 255) / 256)),
              dim3(256), 0, (const double*)P, (const double*)Mring, E, U, N);
    for (int s = 1; s < T; ++s)
        LN_LAUNCH(ctx, BACKWARD ? "backward_frame" : "forward_frame", (fwdbwd_frame_kernel<UT, BACKWARD>),
                  dim3(ctiles, utiles), dim3(256), shm, b, Aexp, Alog, d_lens, out, E, P, Mring, U, T, N, s);
    if (!BACKWARD && loglik)
        LN_LAUNCH(ctx, "forward_loglik", final_loglik_kernel, dim3((U + 3) / 4), dim3(256), 0, (const double*)out, d_lens,
                  loglik, U, T, N);
    return 0;
}

// ------------------------------------------------------------------------------------------
// 128 < N <= 256: the transition matrix (512 KB at N = 256) is exactly the register file of one CU.
// One 1024-thread block per utterance keeps it ON CHIP: thread (column jc, row slice rs) owns 64 rows
// of its column, most of them in registers and the rest in LDS (136 KB for the block), so a frame costs
// 64 FMAs (or mov+add+max) per thread and no L2 traffic at all (17 ms -> 2.6 ms for a forward pass at
// N = 256, T = 2000, U = 64).  MODE 0: Viterbi, 1: forward, 2: backward (same arithmetic, shift rule and
// exact fallback as the register-resident kernels).
// ------------------------------------------------------------------------------------------

// LDS map of the on-chip kernel (doubles)
constexpr int kT256FinReg = 32, kT256CmpReg = 52;                       // A rows held in registers per thread
constexpr int kT256FinLds = 64 - kT256FinReg, kT256CmpLds = 64 - kT256CmpReg;
constexpr int kT256Al0 = 0;                                              // [kT256FinLds][256]     slice 0
constexpr int kT256Al1 = kT256Al0 + kT256FinLds * 256;                   // [kT256CmpLds][3][256]  slices 1..3
constexpr int kT256Vs = kT256Al1 + kT256CmpLds * 3 * 256;                // [2][256] v[t-1] or e = exp(prev - shift)
constexpr int kT256Cs = kT256Vs + 2 * 256;                               // [2][256] prev in the log domain
constexpr int kT256Ps = kT256Cs + 2 * 256;                               // [3][256] partials of slices 1..3
constexpr int kT256Ms = kT256Ps + 3 * 256;                               // [3][4]   per 64-column block max (ring of three frames)
constexpr int kT256Ss = kT256Ms + 12;                                    // [2]      shift of each e buffer
constexpr int kT256Et = kT256Ss + 2;                                     // [kExpTab]
constexpr int kT256Lt = kT256Et + kExpTab;                               // [2 * kLogTab]
constexpr int kT256End = kT256Lt + 2 * kLogTab;

// One role of the on-chip kernel.  FIN = true: the four waves of row slice 0, which also finalise a frame
// (combine the slices, log / exp, emission, store) and therefore keep fewer rows of A in registers;
// FIN = false: the twelve waves of slices 1..3, which only multiply.  The two roles are separate
// instantiations so that each gets its own register allocation; both execute the same barriers.
// TS = float (Viterbi only): `precision=np.float32` -- b and v are float arrays, every sum is formed in fp64 and
// rounded to float where the reference's float32 array rounds it, i.e. at the store of v[t, j] (hmm.py:160-165).
//
// (A variant that took log / exp off the frame's critical path -- e <- sum * F with F formed during the multiply
// phase -- was built and measured slower in round 2, DESIGN 4.5, and removed in round 3.)
template <int MODE, bool FIN, typename TS = double>
__device__ __forceinline__ void trellis256_role(double* sm, const TS* __restrict__ bu, const double* __restrict__ Amat,
                                                const double* __restrict__ Alog, const double* __restrict__ pi,
                                                TS* __restrict__ ou, double* __restrict__ loglik_u, int L, int T, int N) {
    static_assert(MODE == 0 || std::is_same<TS, double>::value, "float storage: Viterbi only");
    constexpr bool VIT = MODE == 0, BACKWARD = MODE == 2;
    constexpr int NP = 256, RPS = 64;
    constexpr int NREG = FIN ? kT256FinReg : kT256CmpReg, NLDS = 64 - NREG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int jc = tid & (NP - 1), rs = tid >> 8;
    double* Al = FIN ? sm + kT256Al0 + jc : sm + kT256Al1 + (rs - 1) * NP + jc;
    constexpr int ALS = FIN ? NP : 3 * NP;                               // stride between LDS rows of A
    double* Vs = sm + kT256Vs;
    double* Cs = sm + kT256Cs;
    double* Ps = sm + kT256Ps;
    double* Ms = sm + kT256Ms;
    double* Ss = sm + kT256Ss;
    const double* etab = sm + kT256Et;
    const double* ltab = sm + kT256Lt;
    const double pad = VIT ? LN_NEG_INF : 0.0;
    double areg[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int i = rs * RPS + r;
        areg[r] = (i < N && jc < N) ? Amat[(size_t)i * N + jc] : pad;
    }
#pragma unroll
    for (int r = 0; r < NLDS; ++r) {
        const int i = rs * RPS + NREG + r;
        Al[r * ALS] = (i < N && jc < N) ? Amat[(size_t)i * N + jc] : pad;
    }
    const int jcl = jc < N ? jc : N - 1;
    auto frame_of = [&](int step) { return BACKWARD ? L - 1 - step : step; };
    auto clampt = [&](int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };

    // step 0
    double bnext = 0.0;
    if (FIN) {
        double prev = LN_NEG_INF;
        if (jc < N) {
            const int t0 = frame_of(0);
            const double bv = (double)bu[(size_t)t0 * N + jc];
            if (BACKWARD) {
                if (ou) ou[(size_t)t0 * N + jc] = (TS)0.0;
                prev = 0.0 + bv;
            } else {
                prev = rnd<TS>(pi[jc] + bv);
                if (ou) ou[(size_t)t0 * N + jc] = (TS)prev;
            }
        }
        if (VIT) {
            Vs[jc] = prev;
        } else {
            Cs[jc] = prev;
            const double m = wave_max_dpp(prev);
            if (lane == 0) Ms[jc >> 6] = m;
        }
        bnext = (double)bu[(size_t)clampt(frame_of(1)) * N + jcl];
    }
    __syncthreads();
    if (FIN && !VIT) {
        const double S = fmax(fmax(Ms[0], Ms[1]), fmax(Ms[2], Ms[3]));
        const double prev = Cs[jc];
        Vs[jc] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
        if (jc == 0) Ss[0] = S;
    }
    __syncthreads();

    for (int step = 1; step < L; ++step) {
        const int cur = (step - 1) & 1, nxt = step & 1;
        const double* Vc = Vs + cur * NP + rs * RPS + (lane & 15);
        double bcur = bnext;
        if (FIN) bnext = (double)bu[(size_t)clampt(frame_of(step + 1)) * N + jcl];
        // The 64 previous-frame values of this row slice sit in four VGPR pairs, 16 per DPP row
        // (lane l holds V[16k + (l & 15)]); row_newbcast:n hands value n of the pair to every lane,
        // so a transition costs one v_fmac_f64_dpp and no LDS broadcast read (the LDS pipe would
        // otherwise be the bound: 64 reads x 16 waves x 4 cycles per frame).
        double Vk[4], acc[4], part;
#pragma unroll
        for (int k = 0; k < 4; ++k) Vk[k] = Vc[16 * k];
        if (VIT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = LN_NEG_INF;
            static_for<0, NREG>([&](auto r) {
                constexpr int k = r.value >> 4, n = r.value & 15;
                acc[k] = vmax_f64(acc[k], mov_bcast<n>(Vk[k]) + areg[r.value]);
            });
            static_for<0, NLDS>([&](auto r) {
                constexpr int k = (NREG + r.value) >> 4, n = (NREG + r.value) & 15;
                acc[k] = vmax_f64(acc[k], mov_bcast<n>(Vk[k]) + Al[r.value * ALS]);
            });
            part = vmax_f64(vmax_f64(acc[0], acc[1]), vmax_f64(acc[2], acc[3]));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = 0.0;
            static_for<0, NREG>([&](auto r) {
                constexpr int k = r.value >> 4, n = r.value & 15;
                fmac_bcast<n>(acc[k], Vk[k], areg[r.value]);
            });
            static_for<0, NLDS>([&](auto r) {
                constexpr int k = (NREG + r.value) >> 4, n = (NREG + r.value) & 15;
                fmac_bcast<n>(acc[k], Vk[k], Al[r.value * ALS]);
            });
            part = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        if (!FIN) Ps[(rs - 1) * NP + jc] = part;
        __syncthreads();
        if (FIN) {
            const int t = frame_of(step);
            if (VIT) {
                const double m = vmax_f64(vmax_f64(part, Ps[jc]), vmax_f64(Ps[NP + jc], Ps[2 * NP + jc]));
                double val = LN_NEG_INF;
                if (jc < N) {
                    val = rnd<TS>(m + bcur);
                    ou[(size_t)t * N + jc] = (TS)val;
                }
                Vs[nxt * NP + jc] = val;
            } else {
                const double sum = (part + Ps[jc]) + (Ps[NP + jc] + Ps[2 * NP + jc]);
                const double Sc = Ss[cur];
                double val;
                if (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF) {
                    val = Sc + log_sum(sum, ltab);
                } else {
                    val = exact_column_lse(Cs + cur * NP, Alog + jcl, N);
                }
                double prev = LN_NEG_INF;
                if (jc < N) {
                    if (BACKWARD) {
                        if (ou) ou[(size_t)t * N + jc] = (TS)val;
                        prev = val + bcur;
                    } else {
                        prev = val + bcur;
                        if (ou) ou[(size_t)t * N + jc] = (TS)prev;
                    }
                }
                const double S = fmax(fmax(Ms[cur * 4], Ms[cur * 4 + 1]), fmax(Ms[cur * 4 + 2], Ms[cur * 4 + 3]));
                Cs[nxt * NP + jc] = prev;
                Vs[nxt * NP + jc] = (S == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - S, etab);
                const double m = wave_max_dpp(prev);
                if (lane == 0) Ms[nxt * 4 + (jc >> 6)] = m;
                if (jc == 0) Ss[nxt] = S;
            }
        }
        __syncthreads();
    }
    const int mfin = (L - 1) & 1;                             // slot of the last frame's maxima
    if (FIN && MODE == 1 && loglik_u && tid < 64) {
        const int fin = (L - 1) & 1;
        const double mx = fmax(fmax(Ms[mfin * 4], Ms[mfin * 4 + 1]), fmax(Ms[mfin * 4 + 2], Ms[mfin * 4 + 3]));
        double sacc = 0.0;
        for (int j = lane; j < NP; j += 64) {
            const double pv = Cs[fin * NP + j];
            sacc += (mx == LN_NEG_INF || pv == LN_NEG_INF) ? 0.0 : exp(pv - mx);
        }
        sacc = wave_sum(sacc);
        if (lane == 0) *loglik_u = (mx == LN_NEG_INF) ? mx : mx + log(sacc);
    }
}

template <int MODE, typename TS = double>
__global__ __launch_bounds__(1024) void trellis256_kernel(const TS* __restrict__ b,
                                                          const double* __restrict__ Amat,   // logA / expA / expA^T
                                                          const double* __restrict__ Alog,   // logA or logA^T (fallback)
                                                          const double* __restrict__ pi,
                                                          const int32_t* __restrict__ lens, TS* __restrict__ out,
                                                          double* __restrict__ loglik, int U, int T, int N) {
    extern __shared__ double sm[];
    const int u = blockIdx.x;
    const int L = lens[u];
    if (L <= 0) return;
    if (MODE != 0) {
        exp_table_init(sm + kT256Et, threadIdx.x, 1024);
        log_table_init(sm + kT256Lt, threadIdx.x, 1024);
    }
    const TS* bu = b + (size_t)u * T * N;
    TS* ou = out ? out + (size_t)u * T * N : nullptr;
    // wave-uniform role split (threadIdx.x >> 8 is constant within a wave); the barriers of the two roles pair up
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0)
        trellis256_role<MODE, true, TS>(sm, bu, Amat, Alog, pi, ou, loglik ? loglik + u : nullptr, L, T, N);
    else
        trellis256_role<MODE, false, TS>(sm, bu, Amat, Alog, pi, ou, nullptr, L, T, N);
}

static inline size_t trellis256_lds() { return (size_t)kT256End * 8; }

// ------------------------------------------------------------------------------------------
// Baum-Welch posteriors (hmm.py:186-199): one block per (t, u)
// ------------------------------------------------------------------------------------------
template <typename TS>
__global__ void baumwelch_kernel(const TS* __restrict__ b, const TS* __restrict__ alpha, const TS* __restrict__ beta,
                                 const double* __restrict__ A, const int32_t* __restrict__ lens, TS* __restrict__ xi,
                                 TS* __restrict__ gamma, int T, int N) {
    extern __shared__ double sm[];
    double* al = sm;                   // [N] alpha[t]
    double* bb = sm + N;               // [N] b[t+1] , beta[t+1]
    double* be = sm + 2 * N;
    double* scratch = sm + 3 * N;      // [32]
    const int t = blockIdx.x, u = blockIdx.y;
    const int L = lens[u];
    if (t >= L) return;
    const size_t row = (size_t)u * T + t;
    TS* xo = xi ? xi + row * N * N : nullptr;
    TS* go = gamma + row * N;
    const int NN = N * N;
    if (t == L - 1) {                  // hmm.py:190,199: last frame stays -inf
        if (xo) for (int e = threadIdx.x; e < NN; e += blockDim.x) st(xo, e, LN_NEG_INF);
        for (int i = threadIdx.x; i < N; i += blockDim.x) st(go, i, LN_NEG_INF);
        return;
    }
    bool has_nan = false;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        al[i] = ld(alpha, row * N + i);
        bb[i] = ld(b, (row + 1) * N + i);
        be[i] = ld(beta, (row + 1) * N + i);
        has_nan |= al[i] != al[i] || bb[i] != bb[i] || be[i] != be[i];
    }
    // one NaN among alpha[t], b[t+1], beta[t+1]: lse(numer) is NaN (np.logaddexp), so every xi[t, i, j] and gamma[t, i]
    // is (hmm.py:192-199); xi follows from the arithmetic below, the row maxima of gamma would drop it
    has_nan = __syncthreads_or(has_nan);
    auto numer = [&](int e) {
        const int i = e / N, j = e - i * N;
        return ((al[i] + A[e]) + bb[j]) + be[j];                           // hmm.py:192-195 order
    };
    double lm = LN_NEG_INF;
    for (int e = threadIdx.x; e < NN; e += blockDim.x) lm = fmax(lm, numer(e));
    const double mx = block_max(lm, scratch);
    double ls = 0.0;
    if (mx != LN_NEG_INF)
        for (int e = threadIdx.x; e < NN; e += blockDim.x) ls += exp(numer(e) - mx);
    const double tot = block_sum(ls, scratch);
    const double denom = (mx == LN_NEG_INF) ? mx : mx + log(tot);
    if (xo)
        for (int e = threadIdx.x; e < NN; e += blockDim.x) st(xo, e, numer(e) - denom);
    // gamma[i] = lse_j xi[i,j]  (hmm.py:199); one wave per row
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = w; i < N; i += nw) {
        double m = LN_NEG_INF;
        for (int j = lane; j < N; j += 64) m = fmax(m, rnd<TS>(numer(i * N + j) - denom));
        m = wave_max(m);
        double s = 0.0;
        if (m != LN_NEG_INF)
            for (int j = lane; j < N; j += 64) s += exp(rnd<TS>(numer(i * N + j) - denom) - m);
        s = wave_sum(s);
        if (lane == 0) st(go, i, has_nan ? __builtin_nan("") : ((m == LN_NEG_INF || m != m) ? m : m + log(s)));
    }
}

// ------------------------------------------------------------------------------------------
// host-side drivers shared with other translation units
// ------------------------------------------------------------------------------------------
int device_lens(lnasr_ctx* ctx, const int32_t* lens, int U, int T, const int32_t** out) {
    LN_TRY(ctx->ws_lens.reserve((size_t)(U > 0 ? U : 1) * 4));
    int32_t* d = ctx->ws_lens.as<int32_t>();
    if (lens) {
        if (is_device_ptr(lens)) {
            *out = lens;
            return 0;
        }
        for (int u = 0; u < U; ++u)
            LN_REQUIRE(lens[u] >= 1 && lens[u] <= T, "lens[%d]=%d out of range [1,%d]", u, lens[u], T);
        LN_HIP(hipMemcpyAsync(d, lens, (size_t)U * 4, hipMemcpyHostToDevice, ctx->stream));
        LN_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        LN_LAUNCH(ctx, "fill_lens", fill_lens_kernel, dim3((U + 255) / 256), dim3(256), 0, d, U, T);
    }
    *out = d;
    return 0;
}

// v [U,T,N] must be device memory
// N > 256: with few utterances the frame-parallel kernels (one launch per frame over the whole batch) beat the
// per-utterance-block streaming kernels 3-16x; with ~1000 utterances the streaming kernels already fill the chip
constexpr size_t kStreamLdsMax = 160 * 1024;     // dynamic LDS a streaming block may ask for
// N > 256: frame-parallel kernels (one launch per frame over the whole batch) or per-utterance streaming kernels.
// Measured on MI355X at T = 100 (tools/bign.py, ms, frame-parallel / streaming):
//   Viterbi  N=320: U=64 1.2/2.1  U=512 2.2/2.2  U=1024 3.0/2.4    N=512: 1.1/4.5  3.0/4.6  4.2/5.1
//            N=1024: 1.8/32  6.5/29  11.6/29
//   forward  N=320: 1.8/1.5  2.2/2.0  2.8/2.2    N=512: 2.6/3.1  3.4/4.3  5.0/4.4    N=1024: 4.7/24  10.8/25  17/50
// LNASR_FRAMEPAR_MAXU (developer/test switch) replaces the rule by "U <= value".
static bool framepar_env(int U, bool* use) {
    if (sw().no_framepar) { *use = false; return true; }
    if (sw().framepar_maxu >= 0) { *use = U <= sw().framepar_maxu; return true; }
    return false;
}
static bool viterbi_framepar(int N, int U) {
    bool use;
    if (framepar_env(U, &use)) return use;
    return N > 384 || U <= 512;
}
static bool fwdbwd_framepar(int N, int U) {
    bool use;
    if (framepar_env(U, &use)) return use;
    return N > 768 || (N > 384 && U <= 512);
}

int viterbi_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* v) {
    Model& md = ctx->model;
    const int N = md.N;
    const double* A = md.logA.as<double>();
    const double* pi = md.logpi.as<double>();
    ctx->bp_v = nullptr;
    if (N <= 16 && !sw().no_tinyvit) {
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
        if (N <= 64) LN_VIT("viterbi_reg_n64", 1, 2, 32, 2, 4);
        if (N <= 128) {
            // default: single-barrier kernel (LDS atomic max); LNASR_VIT=a selects the two-barrier one
            if (vsel == 'a' && U < 512) LN_VIT("viterbi_reg_n128_u1", 2, 8, 16, 1, 4);
            if (vsel == 'a') LN_VIT("viterbi_reg_n128", 2, 8, 16, 2, 4);
            if (U < 512) {
                LN_LAUNCH(ctx, "viterbi_reg1b_n128_u1", (viterbi_reg1b<2, 8, 16, 1, 4>), dim3(U), dim3(512), 0,
                          (const double*)b, A, pi, d_lens, (double*)v, U, T, N);
            } else {
                LN_LAUNCH(ctx, "viterbi_reg1b_n128", (viterbi_reg1b<2, 8, 16, 2, 4, double, true>), dim3((U + 1) / 2), dim3(512), 0,
                          (const double*)b, A, pi, d_lens, (double*)v, U, T, N);
            }
            return 0;
        }
#undef LN_VIT
    }
    if (dtype == LNASR_F32 && N <= 128 && !sw().no_regvit && !sw().f32_generic) {
        // precision=np.float32: the same register-resident kernel, fp64 arithmetic, float storage and rounding
        if (U < 512)
            LN_LAUNCH(ctx, "viterbi_reg1b_n128_u1_f32", (viterbi_reg1b<2, 8, 16, 1, 4, float>), dim3(U), dim3(512), 0,
                      (const float*)b, A, pi, d_lens, (float*)v, U, T, N);
        else
            LN_LAUNCH(ctx, "viterbi_reg1b_n128_f32", (viterbi_reg1b<2, 8, 16, 2, 4, float, true>), dim3((U + 1) / 2), dim3(512), 0,
                      (const float*)b, A, pi, d_lens, (float*)v, U, T, N);
        return 0;
    }
    constexpr int UB = 4;
    if (dtype == LNASR_F64 && N > 128 && N <= 256 && !sw().no_t256) {
        LN_LAUNCH(ctx, "viterbi_onchip256", trellis256_kernel<0>, dim3(U), dim3(1024), trellis256_lds(), (const double*)b, A,
                  A, pi, d_lens, (double*)v, (double*)nullptr, U, T, N);
        return 0;
    }
    if (dtype == LNASR_F32 && N > 128 && N <= 256 && !sw().no_t256 && !sw().f32_generic) {
        // precision=np.float32 beyond 128 states: the on-chip kernel with float storage and rounding
        LN_LAUNCH(ctx, "viterbi_onchip256_f32", (trellis256_kernel<0, float>), dim3(U), dim3(1024), trellis256_lds(),
                  (const float*)b, A, A, pi, d_lens, (float*)v, (double*)nullptr, U, T, N);
        return 0;
    }
    if (dtype == LNASR_F64 && N > 256 && viterbi_framepar(N, U)) {
        // one launch per frame over the whole batch (see viterbi_frame_kernel)
        constexpr int UT = 8;
        const int CPT = U > 128 ? 4 : 2;             // wide tiles when the batch alone fills the chip
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
            LN_LAUNCH(ctx, "viterbi_fill", fill_value_kernel, dim3(2048), dim3(256), 0, (double*)v, cells, LN_NEG_INF);
        LN_LAUNCH(ctx, "viterbi_frame0", viterbi_frame0_kernel, dim3((unsigned)(((size_t)U * N + 255) / 256)), dim3(256), 0,
                  (const double*)b, pi, d_lens, (double*)v, U, T, N);
        for (int t = 1; t < T; ++t) {
            if (CPT == 4)
                LN_LAUNCH(ctx, "viterbi_frame", (viterbi_frame_kernel<UT, 4>), dim3(ctiles, utiles, RS), dim3(256), shm,
                          (const double*)b, A, d_lens, (double*)v, U, T, N, t, RPS, RS > 1 ? 1 : 0);
            else
                LN_LAUNCH(ctx, "viterbi_frame", (viterbi_frame_kernel<UT, 2>), dim3(ctiles, utiles, RS), dim3(256), shm,
                          (const double*)b, A, d_lens, (double*)v, U, T, N, t, RPS, RS > 1 ? 1 : 0);
        }
        return 0;
    }
    if (dtype == LNASR_F64 && N > 128 && N <= 1024 && !sw().no_stream) {
        const int NP = ((N + 63) / 64) * 64, RS = 1024 / NP, RPS = (N + RS - 1) / RS;
        const size_t shm = (size_t)(2 + RS) * UB * NP * 8;
        LN_LAUNCH(ctx, "viterbi_stream", (viterbi_stream<UB>), dim3((U + UB - 1) / UB), dim3(NP * RS), shm,
                  (const double*)b, A, pi, d_lens, (double*)v, U, T, N, NP, RS, RPS);
        return 0;
    }
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

int backtrace_device(lnasr_ctx* ctx, const void* v, int U, int T, const int32_t* d_lens, int dtype, uint32_t* path,
                     double* best, bool* v_tested = nullptr) {
    // *v_tested: the kernel looked at every cell of v and raised the context's "suspect" word on NaN / -inf (nanfix.hip)
    Model& md = ctx->model;
    const int wpb = 4;
    unsigned* const sflag = ctx->nan_flag.as<unsigned>() + 1;
    const unsigned epoch = ctx->nan_epoch;
    if (v_tested) *v_tested = false;
    if (ctx->bp_v == v && ctx->bp_U == U && ctx->bp_T == T && v != nullptr) {
        // v came from viterbi_tiny_kernel: its byte back-pointers are in ws_bp
        const uint8_t* bp = ctx->ws_bp.as<uint8_t>();
        const int nmax = ctx->bp_nmax;
        ctx->bp_v = nullptr;
        const dim3 grid((U + wpb - 1) / wpb), block(64 * wpb);
#define LN_TINY(TS, NM)                                                                                        \
    LN_LAUNCH(ctx, "backtrace_tiny", (backtrace_tiny_kernel<TS, NM>), grid, block, 0, (const TS*)v, bp, d_lens, path, best, \
              U, T, md.N)
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
        return 0;
    }
    if (dtype == LNASR_F64 && md.N <= 128 && !sw().no_ldsbt) {
        if (v_tested) *v_tested = true;
        if (md.N == 128)
            LN_LAUNCH(ctx, "backtrace_lds", (backtrace_lds_kernel<2, true, double, true>), dim3((U + wpb - 1) / wpb),
                      dim3(64 * wpb), (size_t)md.N * md.N * 8, (const double*)v, md.logAT.as<double>(), d_lens, path, best, U,
                      T, md.N, sflag, epoch);
        else
            LN_LAUNCH(ctx, "backtrace_lds", (backtrace_lds_kernel<2, true>), dim3((U + wpb - 1) / wpb), dim3(64 * wpb),
                      (size_t)md.N * md.N * 8, (const double*)v, md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);
        return 0;
    }
    if (dtype == LNASR_F64 && md.N <= 1024 && !sw().no_ldsbt) {
        if (v_tested) *v_tested = true;
        // one wave per block (with few utterances the waves should sit on different CUs), N / 64 states per lane
        if (md.N <= 256)
            LN_LAUNCH(ctx, "backtrace_wave256", (backtrace_lds_kernel<4, false>), dim3(U), dim3(64), 0, (const double*)v,
                      md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);
        else if (md.N <= 512)
            LN_LAUNCH(ctx, "backtrace_wave512", (backtrace_lds_kernel<8, false>), dim3(U), dim3(64), 0, (const double*)v,
                      md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);
        else
            LN_LAUNCH(ctx, "backtrace_wave1024", (backtrace_lds_kernel<16, false>), dim3(U), dim3(64), 0, (const double*)v,
                      md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);
        return 0;
    }
    if (dtype == LNASR_F32 && md.N <= 128 && !sw().no_ldsbt && !sw().f32_generic) {
        if (v_tested) *v_tested = true;
        LN_LAUNCH(ctx, "backtrace_lds_f32", (backtrace_lds_kernel<2, true, float>), dim3((U + wpb - 1) / wpb), dim3(64 * wpb),
                  (size_t)md.N * md.N * 8, (const float*)v, md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);
        return 0;
    }
    if (dtype == LNASR_F32 && md.N <= 256 && !sw().no_ldsbt && !sw().f32_generic) {
        if (v_tested) *v_tested = true;
        LN_LAUNCH(ctx, "backtrace_wave256_f32", (backtrace_lds_kernel<4, false, float>), dim3(U), dim3(64), 0, (const float*)v,
                  md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);
        return 0;
    }
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "backtrace_f32", backtrace_kernel<float>, dim3((U + wpb - 1) / wpb), dim3(64 * wpb), 0,
                  (const float*)v, md.logAT.as<double>(), d_lens, path, best, U, T, md.N);
    else
        LN_LAUNCH(ctx, "backtrace", backtrace_kernel<double>, dim3((U + wpb - 1) / wpb), dim3(64 * wpb), 0,
                  (const double*)v, md.logAT.as<double>(), d_lens, path, best, U, T, md.N);
    return 0;
}

int forward_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* alpha,
                   double* loglik) {
    Model& md = ctx->model;
    const int N = md.N;
    ctx->fwd_nan_tested = false;
    if (dtype == LNASR_F32 && f32_via_f64()) {
        const size_t cells = (size_t)U * T * N;
        double *b64, *a64 = nullptr;
        LN_TRY(widen_f32(ctx, b, cells, ctx->ws_w1, &b64));
        if (alpha) {
            LN_TRY(ctx->ws_w2.reserve(cells * 8));
            a64 = ctx->ws_w2.as<double>();
        }
        LN_TRY(forward_device(ctx, b64, U, T, d_lens, LNASR_F64, a64, loglik));
        return alpha ? narrow_f64(ctx, a64, cells, alpha) : 0;
    }
    if (dtype == LNASR_F64 && N <= 128 && !sw().no_regfwd) {
        if (N <= 64)
            LN_LAUNCH(ctx, "forward_reg_n64", (fwdbwd_reg<false, 1, 2, 32, 2, 3>), dim3((U + 1) / 2), dim3(128), 0,
                      (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                      (double*)alpha, loglik, U, T, N);
        else
            // column-owned waves, two utterances per four-wave block, two blocks per CU (fwdbwd_cols2): measured against the
            // eight-wave kernels at every batch size -- 16 ... 512 utterances 0.83-0.87 ms against 0.96-1.02 (fwdbwd_reg, two
            // per block), 1000 utterances 1.23 against 1.45 (fwdbwd_cols, four per block)
            if (!sw().no_colfwd && sw().fwd != 'c')
            {
                LN_LAUNCH(ctx, "forward_cols2_n128", (fwdbwd_cols2<false>), dim3((U + 1) / 2), dim3(256), 0,
                          (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                          (double*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                ctx->fwd_nan_tested = true;
            }
            else if (!sw().no_colfwd)            // LNASR_FWD=c: four per eight-wave block
                LN_LAUNCH(ctx, "forward_cols_n128", (fwdbwd_cols<false>), dim3((U + 3) / 4), dim3(512), 0,
                          (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                          (double*)alpha, loglik, U, T, N);
            else if (U < 4 * ctx->prop.multiProcessorCount)      // row-owned waves, partial sums through the LDS
                LN_LAUNCH(ctx, "forward_reg_n128", (fwdbwd_reg<false, 2, 8, 16, 2, 4>), dim3((U + 1) / 2), dim3(512), 0,
                          (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                          (double*)alpha, loglik, U, T, N);
            else
                LN_LAUNCH(ctx, "forward_reg_n128", (fwdbwd_reg<false, 2, 8, 16, 4, 2>), dim3((U + 3) / 4), dim3(512), 0,
                          (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                          (double*)alpha, loglik, U, T, N);
        return 0;
    }
    if (dtype == LNASR_F64 && N > 128 && N <= 256 && !sw().no_t256) {
        LN_LAUNCH(ctx, "forward_onchip256", (trellis256_kernel<1, double>), dim3(U), dim3(1024), trellis256_lds(),
                  (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                  (double*)alpha, loglik, U, T, N);
        return 0;
    }
    if (dtype == LNASR_F64 && N > 256 && fwdbwd_framepar(N, U)) {
        double* a = (double*)alpha;
        if (!a) {                                         // log-likelihood only: alpha still has to exist somewhere
            LN_TRY(ctx->ws_alpha.reserve((size_t)U * T * N * 8));
            a = ctx->ws_alpha.as<double>();
        }
        return fwdbwd_frames<false>(ctx, (const double*)b, md.expA.as<double>(), md.logA.as<double>(), U, T, d_lens, a,
                                    loglik);
    }
    if (dtype == LNASR_F64 && N > 128 && N <= 1024 && !sw().no_stream) {
        const int NP = ((N + 63) / 64) * 64, RS = 1024 / NP, RPS = (N + RS - 1) / RS, NB = NP / 64;
        // few utterances: one per block keeps more CUs busy (the frame chain is the critical path);
        // many: four per block share every read of exp(A)
        if (U < 2 * ctx->prop.multiProcessorCount) {
            constexpr int UB = 1;
            const size_t shm = ((size_t)(4 + RS) * UB * NP + 2 * UB * NB + 2 * UB) * 8;
            LN_LAUNCH(ctx, "forward_stream_u1", (fwdbwd_stream<false, UB>), dim3(U), dim3(NP * RS), shm, (const double*)b,
                      md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens, (double*)alpha,
                      loglik, U, T, N, NP, RS, RPS);
        } else if (((size_t)(4 + RS) * 4 * NP + 8 * NB + 8) * 8 > kStreamLdsMax) {   // N > 960: four do not fit
            constexpr int UB = 2;
            const size_t shm = ((size_t)(4 + RS) * UB * NP + 2 * UB * NB + 2 * UB) * 8;
            LN_LAUNCH(ctx, "forward_stream_u2", (fwdbwd_stream<false, UB>), dim3((U + UB - 1) / UB), dim3(NP * RS), shm,
                      (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                      (double*)alpha, loglik, U, T, N, NP, RS, RPS);
        } else {
            constexpr int UB = 4;
            const size_t shm = ((size_t)(4 + RS) * UB * NP + 2 * UB * NB + 2 * UB) * 8;
            LN_LAUNCH(ctx, "forward_stream", (fwdbwd_stream<false, UB>), dim3((U + UB - 1) / UB), dim3(NP * RS), shm,
                      (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                      (double*)alpha, loglik, U, T, N, NP, RS, RPS);
        }
        return 0;
    }
    int nt = ((N + 63) / 64) * 64;
    if (nt > 1024) nt = 1024;
    LN_REQUIRE(N <= 4 * nt, "N=%d too large for the forward kernel", N);
    const size_t shmem = (size_t)(2 * N + 32) * 8;
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "forward_f32", (fwdbwd_generic<float, false>), dim3(U), dim3(nt), shmem, (const float*)b,
                  md.logA.as<double>(), md.expA.as<double>(), md.logpi.as<double>(), d_lens, (float*)alpha, loglik,
                  U, T, N);
    else
        LN_LAUNCH(ctx, "forward", (fwdbwd_generic<double, false>), dim3(U), dim3(nt), shmem, (const double*)b,
                  md.logA.as<double>(), md.expA.as<double>(), md.logpi.as<double>(), d_lens, (double*)alpha, loglik,
                  U, T, N);
    return 0;
}

int backward_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* beta) {
    Model& md = ctx->model;
    const int N = md.N;
    ctx->fwd_nan_tested = false;
    if (dtype == LNASR_F32 && f32_via_f64()) {
        const size_t cells = (size_t)U * T * N;
        double* b64;
        LN_TRY(widen_f32(ctx, b, cells, ctx->ws_w1, &b64));
        LN_TRY(ctx->ws_w2.reserve(cells * 8));
        LN_TRY(backward_device(ctx, b64, U, T, d_lens, LNASR_F64, ctx->ws_w2.as<double>()));
        return narrow_f64(ctx, ctx->ws_w2.as<double>(), cells, beta);
    }
    const double* expAT = md.expAT.as<double>();
    if (dtype == LNASR_F64 && N <= 128 && !sw().no_regfwd) {
        if (N <= 64)
            LN_LAUNCH(ctx, "backward_reg_n64", (fwdbwd_reg<true, 1, 2, 32, 2, 3>), dim3((U + 1) / 2), dim3(128), 0,
                      (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                      (double*)nullptr, U, T, N);
        else
            if (!sw().no_colfwd && sw().fwd != 'c')
            {
                LN_LAUNCH(ctx, "backward_cols2_n128", (fwdbwd_cols2<true>), dim3((U + 1) / 2), dim3(256), 0,
                          (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                          (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                ctx->fwd_nan_tested = true;
            }
            else if (!sw().no_colfwd)
                LN_LAUNCH(ctx, "backward_cols_n128", (fwdbwd_cols<true>), dim3((U + 3) / 4), dim3(512), 0,
                          (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                          (double*)nullptr, U, T, N);
            else if (U < 4 * ctx->prop.multiProcessorCount)
                LN_LAUNCH(ctx, "backward_reg_n128", (fwdbwd_reg<true, 2, 8, 16, 2, 4>), dim3((U + 1) / 2), dim3(512), 0,
                          (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                          (double*)nullptr, U, T, N);
            else
                LN_LAUNCH(ctx, "backward_reg_n128", (fwdbwd_reg<true, 2, 8, 16, 4, 2>), dim3((U + 3) / 4), dim3(512), 0,
                          (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                          (double*)nullptr, U, T, N);
        return 0;
    }
    if (dtype == LNASR_F64 && N > 128 && N <= 256 && !sw().no_t256) {
        LN_LAUNCH(ctx, "backward_onchip256", (trellis256_kernel<2, double>), dim3(U), dim3(1024), trellis256_lds(),
                  (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                  (double*)nullptr, U, T, N);
        return 0;
    }
    if (dtype == LNASR_F64 && N > 256 && fwdbwd_framepar(N, U))
        return fwdbwd_frames<true>(ctx, (const double*)b, expAT, md.logAT.as<double>(), U, T, d_lens, (double*)beta, nullptr);
    if (dtype == LNASR_F64 && N > 128 && N <= 1024 && !sw().no_stream) {
        const int NP = ((N + 63) / 64) * 64, RS = 1024 / NP, RPS = (N + RS - 1) / RS, NB = NP / 64;
        if (U < 2 * ctx->prop.multiProcessorCount) {
            constexpr int UB = 1;
            const size_t shm = ((size_t)(4 + RS) * UB * NP + 2 * UB * NB + 2 * UB) * 8;
            LN_LAUNCH(ctx, "backward_stream_u1", (fwdbwd_stream<true, UB>), dim3(U), dim3(NP * RS), shm, (const double*)b,
                      expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta, (double*)nullptr, U,
                      T, N, NP, RS, RPS);
        } else if (((size_t)(4 + RS) * 4 * NP + 8 * NB + 8) * 8 > kStreamLdsMax) {
            constexpr int UB = 2;
            const size_t shm = ((size_t)(4 + RS) * UB * NP + 2 * UB * NB + 2 * UB) * 8;
            LN_LAUNCH(ctx, "backward_stream_u2", (fwdbwd_stream<true, UB>), dim3((U + UB - 1) / UB), dim3(NP * RS), shm,
                      (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                      (double*)nullptr, U, T, N, NP, RS, RPS);
        } else {
            constexpr int UB = 4;
            const size_t shm = ((size_t)(4 + RS) * UB * NP + 2 * UB * NB + 2 * UB) * 8;
            LN_LAUNCH(ctx, "backward_stream", (fwdbwd_stream<true, UB>), dim3((U + UB - 1) / UB), dim3(NP * RS), shm,
                      (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                      (double*)nullptr, U, T, N, NP, RS, RPS);
        }
        return 0;
    }
    int nt = ((N + 63) / 64) * 64;
    if (nt > 1024) nt = 1024;
    LN_REQUIRE(N <= 4 * nt, "N=%d too large for the backward kernel", N);
    const size_t shmem = (size_t)(2 * N + 32) * 8;
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "backward_f32", (fwdbwd_generic<float, true>), dim3(U), dim3(nt), shmem, (const float*)b,
                  md.logAT.as<double>(), expAT, md.logpi.as<double>(), d_lens, (float*)beta, (double*)nullptr, U, T,
                  N);
    else
        LN_LAUNCH(ctx, "backward", (fwdbwd_generic<double, true>), dim3(U), dim3(nt), shmem, (const double*)b,
                  md.logAT.as<double>(), expAT, md.logpi.as<double>(), d_lens, (double*)beta, (double*)nullptr, U, T,
                  N);
    return 0;
}

int baumwelch_device(lnasr_ctx* ctx, const void* b, const void* alpha, const void* beta, int U, int T,
                     const int32_t* d_lens, int dtype, void* xi, void* gamma) {
    Model& md = ctx->model;
    const int N = md.N;
    const size_t shmem = (size_t)(3 * N + 32) * 8;
    LN_REQUIRE(U <= 65535, "U=%d too large for one Baum-Welch launch", U);
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "baumwelch_f32", baumwelch_kernel<float>, dim3(T, U), dim3(256), shmem, (const float*)b,
                  (const float*)alpha, (const float*)beta, md.logA.as<double>(), d_lens, (float*)xi, (float*)gamma, T,
                  N);
    else
        LN_LAUNCH(ctx, "baumwelch", baumwelch_kernel<double>, dim3(T, U), dim3(256), shmem, (const double*)b,
                  (const double*)alpha, (const double*)beta, md.logA.as<double>(), d_lens, (double*)xi,
                  (double*)gamma, T, N);
    return 0;
}

}  // namespace lnasr

using namespace lnasr;

#define LN_CHECK_BATCH(ctx, U, T, dtype)                                                      \
    LN_REQUIRE(ctx, "null ctx");                                                              \
    LN_REQUIRE((U) >= 0 && (T) >= 0, "negative batch shape");                                 \
    LN_REQUIRE((dtype) == LNASR_F64 || (dtype) == LNASR_F32, "bad dtype %d", (dtype));        \
    if (!(ctx)->model.has_hmm) return fail(LNASR_E_NOMODEL, "no HMM transition model uploaded")

extern "C" {

int lnasr_discrete_map_b(lnasr_ctx* ctx, const double* logB, int N, int K, const int32_t* obs, int U, int T,
                         int dtype, void* b) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_REQUIRE(N >= 1 && K >= 1 && U >= 0 && T >= 0, "bad shape");
    LN_REQUIRE(dtype == LNASR_F64 || dtype == LNASR_F32, "bad dtype %d", dtype);
    const long F = (long)U * T;
    if (F == 0) return 0;
    if (!is_device_ptr(obs))
        for (long f = 0; f < F; ++f)
            LN_REQUIRE(obs[f] >= 0 && obs[f] < K, "obs[%ld]=%d outside the symbol set [0,%d)", f, obs[f], K);
    const void *d_B, *d_obs;
    void* d_b;
    const size_t es = dtype_size(dtype);
    LN_TRY(stage_in(ctx, logB, (size_t)N * K * 8, ctx->ws_in2, &d_B));
    LN_TRY(stage_in(ctx, obs, (size_t)F * 4, ctx->ws_in, &d_obs));
    LN_TRY(stage_out(ctx, b, (size_t)F * N * es, ctx->ws_out, &d_b));
    const unsigned grid = (unsigned)((F * N + 255) / 256);
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "discrete_map_b_f32", discrete_map_b_kernel<float>, dim3(grid), dim3(256), 0,
                  (const double*)d_B, N, K, (const int32_t*)d_obs, F, (float*)d_b);
    else
        LN_LAUNCH(ctx, "discrete_map_b", discrete_map_b_kernel<double>, dim3(grid), dim3(256), 0, (const double*)d_B,
                  N, K, (const int32_t*)d_obs, F, (double*)d_b);
    return finish_out(ctx, b, d_b, (size_t)F * N * es);
}

int lnasr_viterbi(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* lens, int dtype, void* v_or_null,
                  uint32_t* bt_or_null, uint32_t* path, double* best_score_or_null) {
    LN_ENTER(ctx);
    LN_CHECK_BATCH(ctx, U, T, dtype);
    if (U == 0 || T == 0) return 0;
    LN_REQUIRE(b && path, "null b/path");
    const int N = ctx->model.N;
    const size_t es = dtype_size(dtype), cells = (size_t)U * T * N;
    const int32_t* d_lens;
    LN_TRY(device_lens(ctx, lens, U, T, &d_lens));
    const void* d_b;
    void *d_v, *d_path, *d_bt = nullptr, *d_best = nullptr;
    LN_TRY(stage_in(ctx, b, cells * es, ctx->ws_in, &d_b));
    LN_TRY(stage_out(ctx, v_or_null, cells * es, ctx->ws_v, &d_v));
    LN_TRY(stage_out(ctx, path, (size_t)U * T * 4, ctx->ws_out, &d_path));
    if (best_score_or_null) LN_TRY(stage_out(ctx, best_score_or_null, (size_t)U * 8, ctx->ws_out2, &d_best));
    // float32 storage stays on the storage-templated kernels here: rounding v to f32 after every frame is what
    // makes v, bt and path bit-identical to the reference run with precision=np.float32 (golden gmmhmm_small_f32);
    // forward / backward / emission are tolerance-checked and take the widened f64 route (common.h)
    bool nan_tested = false;
    LN_TRY(viterbi_device(ctx, d_b, U, T, d_lens, dtype, d_v));
    LN_TRY(backtrace_device(ctx, d_v, U, T, d_lens, dtype, (uint32_t*)d_path, (double*)d_best, &nan_tested));
    // emissions handed in by the caller: a NaN among them propagates as numpy's max / argmax do (nanfix.hip); kernels
    // that did not look for one themselves cost a scan of b here
    LN_TRY(viterbi_nan_repair(ctx, d_b, U, T, d_lens, dtype, d_v, (uint32_t*)d_path, (double*)d_best, !nan_tested));
    if (bt_or_null) {
        LN_TRY(stage_out(ctx, bt_or_null, cells * 4, ctx->ws_out3, &d_bt));
        LN_REQUIRE(U <= 65535, "U=%d too large for the bt kernel", U);
        int nt = ((N + 63) / 64) * 64;
        if (nt > 256) nt = 256;
        if (dtype == LNASR_F32)
            LN_LAUNCH(ctx, "viterbi_bt_f32", bt_kernel<float>, dim3(T, U), dim3(nt), (size_t)N * 8, (const float*)d_v,
                      ctx->model.logA.as<double>(), d_lens, (uint32_t*)d_bt, T, N);
        else
            LN_LAUNCH(ctx, "viterbi_bt", bt_kernel<double>, dim3(T, U), dim3(nt), (size_t)N * 8, (const double*)d_v,
                      ctx->model.logA.as<double>(), d_lens, (uint32_t*)d_bt, T, N);
    }
    if (v_or_null) LN_TRY(finish_out(ctx, v_or_null, d_v, cells * es));
    if (bt_or_null) LN_TRY(finish_out(ctx, bt_or_null, d_bt, cells * 4));
    LN_TRY(finish_out(ctx, path, d_path, (size_t)U * T * 4));
    if (best_score_or_null) LN_TRY(finish_out(ctx, best_score_or_null, d_best, (size_t)U * 8));
    return 0;
}

int lnasr_decode(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, uint32_t* path,
                 double* best_score_or_null) {
    LN_ENTER(ctx);
    LN_CHECK_BATCH(ctx, U, T, LNASR_F64);
    if (!ctx->model.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    if (U == 0 || T == 0) return 0;
    LN_REQUIRE(obs && path, "null obs/path");
    Model& md = ctx->model;
    const size_t cells = (size_t)U * T * md.N;
    const int32_t* d_lens;
    LN_TRY(device_lens(ctx, lens, U, T, &d_lens));
    void *d_path, *d_best = nullptr;
    LN_TRY(ctx->ws_b.reserve(cells * 8));
    LN_TRY(ctx->ws_v.reserve(cells * 8));
    LN_TRY(stage_out(ctx, path, (size_t)U * T * 4, ctx->ws_out, &d_path));
    if (best_score_or_null) LN_TRY(stage_out(ctx, best_score_or_null, (size_t)U * 8, ctx->ws_out2, &d_best));
    // three launches in series on one stream (cutting the batch into chunks whose trellis work runs on a side stream
    // beside the next chunk's emission was measured slower, DESIGN 4.8, and removed in round 3)
    const double* d_x = nullptr;
    bool scanned = false, v_tested = false;
    // (full covariance: the emission pass re-scores frames with a non-finite feature itself -- an infinity gives -inf or
    // NaN per component there -- and the trellis is repaired from the emissions, as for lnasr_viterbi)
    const bool by_obs = !nan_rows_rescored(md);
    LN_TRY(emit_batch(ctx, obs, U, T, ctx->ws_b.as<double>(), nullptr, &d_x, by_obs ? &scanned : nullptr));
    LN_TRY(viterbi_device(ctx, ctx->ws_b.p, U, T, d_lens, LNASR_F64, ctx->ws_v.p));
    LN_TRY(backtrace_device(ctx, ctx->ws_v.p, U, T, d_lens, LNASR_F64, (uint32_t*)d_path, (double*)d_best, &v_tested));
    // a NaN or an infinite feature makes its frame's emissions NaN in the reference, and np.max carries that through every
    // later frame: one launch that returns at once unless the emission kernel raised the flag (nanfix.hip)
    if (by_obs)
        LN_TRY(viterbi_nan_repair_obs(ctx, d_x, U, T, d_lens, ctx->ws_v.as<double>(), (uint32_t*)d_path, (double*)d_best,
                                      !scanned));
    else
        LN_TRY(viterbi_nan_repair(ctx, ctx->ws_b.p, U, T, d_lens, LNASR_F64, ctx->ws_v.p, (uint32_t*)d_path, (double*)d_best,
                                  !v_tested));
    LN_TRY(finish_out(ctx, path, d_path, (size_t)U * T * 4));
    if (best_score_or_null) LN_TRY(finish_out(ctx, best_score_or_null, d_best, (size_t)U * 8));
    return 0;
}

int lnasr_forward(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* lens, int dtype, void* alpha_or_null,
                  double* loglik_or_null) {
    LN_ENTER(ctx);
    LN_CHECK_BATCH(ctx, U, T, dtype);
    if (U == 0 || T == 0) return 0;
    LN_REQUIRE(b, "null b");
    const int N = ctx->model.N;
    const size_t es = dtype_size(dtype), cells = (size_t)U * T * N;
    const int32_t* d_lens;
    LN_TRY(device_lens(ctx, lens, U, T, &d_lens));
    const void* d_b;
    void *d_alpha = nullptr, *d_ll = nullptr;
    LN_TRY(stage_in(ctx, b, cells * es, ctx->ws_in, &d_b));
    if (alpha_or_null) LN_TRY(stage_out(ctx, alpha_or_null, cells * es, ctx->ws_alpha, &d_alpha));
    if (loglik_or_null) LN_TRY(stage_out(ctx, loglik_or_null, (size_t)U * 8, ctx->ws_out2, &d_ll));
    LN_TRY(forward_device(ctx, d_b, U, T, d_lens, dtype, d_alpha, (double*)d_ll));
    // a NaN among the caller's emissions: alpha and the likelihood as np.logaddexp leaves them (nanfix.hip)
    LN_TRY(trellis_nan_repair(ctx, d_b, U, T, d_lens, dtype, d_alpha, (double*)d_ll, nullptr, !ctx->fwd_nan_tested));
    if (alpha_or_null) LN_TRY(finish_out(ctx, alpha_or_null, d_alpha, cells * es));
    if (loglik_or_null) LN_TRY(finish_out(ctx, loglik_or_null, d_ll, (size_t)U * 8));
    return 0;
}

int lnasr_backward(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* lens, int dtype, void* beta) {
    LN_ENTER(ctx);
    LN_CHECK_BATCH(ctx, U, T, dtype);
    if (U == 0 || T == 0) return 0;
    LN_REQUIRE(b && beta, "null b/beta");
    const int N = ctx->model.N;
    const size_t es = dtype_size(dtype), cells = (size_t)U * T * N;
    const int32_t* d_lens;
    LN_TRY(device_lens(ctx, lens, U, T, &d_lens));
    const void* d_b;
    void* d_beta;
    LN_TRY(stage_in(ctx, b, cells * es, ctx->ws_in, &d_b));
    LN_TRY(stage_out(ctx, beta, cells * es, ctx->ws_beta, &d_beta));
    LN_TRY(backward_device(ctx, d_b, U, T, d_lens, dtype, d_beta));
    LN_TRY(trellis_nan_repair(ctx, d_b, U, T, d_lens, dtype, nullptr, nullptr, d_beta, !ctx->fwd_nan_tested));
    return finish_out(ctx, beta, d_beta, cells * es);
}

int lnasr_calc_prob(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, double* loglik) {
    return lnasr_calc_prob_p(ctx, obs, U, T, lens, LNASR_F64, loglik);
}

int lnasr_calc_prob_p(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, int dtype, double* loglik) {
    LN_ENTER(ctx);
    LN_CHECK_BATCH(ctx, U, T, LNASR_F64);
    LN_REQUIRE(dtype == LNASR_F64 || dtype == LNASR_F32, "bad dtype %d", dtype);
    if (!ctx->model.has_gmm) return fail(LNASR_E_NOMODEL, "no GMM-HMM model uploaded");
    if (U == 0 || T == 0) return 0;
    LN_REQUIRE(obs && loglik, "null obs/loglik");
    Model& md = ctx->model;
    const size_t cells = (size_t)U * T * md.N;
    const int32_t* d_lens;
    LN_TRY(device_lens(ctx, lens, U, T, &d_lens));
    void* d_ll;
    LN_TRY(ctx->ws_b.reserve(cells * 8));
    LN_TRY(stage_out(ctx, loglik, (size_t)U * 8, ctx->ws_out2, &d_ll));
    const double* d_x = nullptr;
    bool scanned = false;
    const bool by_obs = !nan_rows_rescored(md);
    if (dtype == LNASR_F32 && f32_stats_ok(md)) {
        // the emission of the float32 statistics pass (lnasr_bw_accumulate_p): same b, hence the same likelihood
        const void* d_obs;
        LN_TRY(stage_in(ctx, obs, (size_t)U * T * md.D * 8, ctx->ws_in, &d_obs));
        LN_TRY(ctx->ws_bm.reserve((cells + (size_t)kMixPadFrames * md.N) * md.M * 4));
        LN_TRY(emit_device_f32(ctx, (const double*)d_obs, (long)U * T, ctx->ws_b.as<double>(), ctx->ws_bm.as<float>()));
        d_x = (const double*)d_obs;                         // emit_diag_mfma_f32 has no fused test (f32stats.hip)
    } else {
        LN_TRY(emit_batch(ctx, obs, U, T, ctx->ws_b.as<double>(), nullptr, &d_x, by_obs ? &scanned : nullptr));
    }
    LN_TRY(forward_device(ctx, ctx->ws_b.p, U, T, d_lens, LNASR_F64, nullptr, (double*)d_ll));
    // an utterance with a NaN or an infinite feature scores NaN (np.logaddexp carries it to the last frame, hmm.py:97-99);
    // full covariance: an utterance whose re-scored emissions hold a NaN does (an infinity may leave only -inf there)
    if (by_obs)
        LN_TRY(loglik_nan_repair(ctx, d_x, U, T, d_lens, (double*)d_ll, !scanned));
    else
        LN_TRY(trellis_nan_repair(ctx, ctx->ws_b.p, U, T, d_lens, LNASR_F64, nullptr, (double*)d_ll, nullptr,
                                  !ctx->fwd_nan_tested));
    return finish_out(ctx, loglik, d_ll, (size_t)U * 8);
}

int lnasr_baumwelch(lnasr_ctx* ctx, const void* b, const void* alpha, const void* beta, int U, int T,
                    const int32_t* lens, int dtype, void* xi, void* gamma) {
    LN_ENTER(ctx);
    LN_CHECK_BATCH(ctx, U, T, dtype);
    if (U == 0 || T == 0) return 0;
    LN_REQUIRE(b && alpha && beta && gamma, "null argument");
    const int N = ctx->model.N;
    const size_t es = dtype_size(dtype), cells = (size_t)U * T * N;
    const int32_t* d_lens;
    LN_TRY(device_lens(ctx, lens, U, T, &d_lens));
    const void *d_b, *d_al, *d_be;
    void *d_xi = nullptr, *d_ga;
    LN_TRY(stage_in(ctx, b, cells * es, ctx->ws_in, &d_b));
    LN_TRY(stage_in(ctx, alpha, cells * es, ctx->ws_in2, &d_al));
    LN_TRY(stage_in(ctx, beta, cells * es, ctx->ws_in3, &d_be));
    if (xi) LN_TRY(stage_out(ctx, xi, cells * N * es, ctx->ws_out, &d_xi));
    LN_TRY(stage_out(ctx, gamma, cells * es, ctx->ws_out2, &d_ga));
    LN_TRY(baumwelch_device(ctx, d_b, d_al, d_be, U, T, d_lens, dtype, d_xi, d_ga));
    if (xi) LN_TRY(finish_out(ctx, xi, d_xi, cells * N * es));
    return finish_out(ctx, gamma, d_ga, cells * es);
}

}  // extern "C"
