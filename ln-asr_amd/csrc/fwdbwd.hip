// Forward / backward recursions (hmm.py:63-131), kernels for N <= 128 and the dispatch over every family.
// 128 < N <= 256: trellis256.hip; N > 256: bign.hip.
#include "common.h"
#include "devmath.h"
#include "trellis.h"

namespace lnasr {

// ------------------------------------------------------------------------------------------
// forward / backward (log domain)
// ------------------------------------------------------------------------------------------
// alpha[t,j] = lse_i(alpha[t-1,i] + A[i,j]) + b[t,j]                       (hmm.py:97-99)
// Evaluated as s + log(sum_i exp(alpha[t-1,i] - s) * expA[i,j]) with s = max_i alpha[t-1,i]:
// one exp per state per frame and an FMA per (i,j) instead of a transcendental per (i,j).
// Columns whose linear-domain sum falls below kSafeSum (every contributor underflowed, or the
// column is unreachable) are redone with the exact max-shifted log-domain sum, so the result
// equals the reference's logaddexp fold to rounding for every input, including -inf entries.

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
    double* __restrict__ loglik, int U, int T, int N, unsigned* __restrict__ fb_count) {
    constexpr int NP = 64 * CPT;
    constexpr int NJOB = CPT * UB;
    constexpr int JPW = (NJOB + NW - 1) / NW;
    __shared__ double Es[2][UB][NP];      // exp(prev - shift) of the frame being consumed
    __shared__ double Cs[2][UB][NP];      // prev itself (log domain) for the exact fallback
    __shared__ double Ps[NW > 1 ? NW : 1][UB][NP];
    __shared__ double Ms[2][UB][CPT];     // per column-block max of prev
    __shared__ float Lb[UB][CPT * 4];     // per DPP row max of the emissions of the frame being finalised (see fwdbwd_cols2)
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
        static_assert(NW > 1, "the level partials below travel with the first barrier");
        // level of this frame's emission row, for the shift (fwdbwd_cols2 has the story): the lagged maximum alone assumes
        // emissions of a steady size
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            if (jvalid[jj]) {
                const float lv = row_max_f32(lane + 64 * jc[jj] < N ? (float)bcur[jj] : -__builtin_huge_valf());
                if ((lane & 15) == 0) Lb[jk[jj]][jc[jj] * 4 + (lane >> 4)] = lv;
            }
        }
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
                    if (j >= N || (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF)) {   // (a padding column: nothing to redo)
                        val = Sc + log_sum(sum, ltab);
                    } else {                       // exact log-domain column (rare)
                        val = exact_column_lse(&Cs[cur][k][0], Alog + (j < N ? j : N - 1), N);
                        atomicAdd(fb_count, 1u);   // diagnostic (lnasr_debug_fallback_columns)
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
                    {
                        float lvl = Lb[k][0];
#pragma unroll
                        for (int e = 1; e < CPT * 4; ++e) lvl = fmaxf(lvl, Lb[k][e]);
                        S += (double)lvl;          // the lagged maximum, carried to this frame's emission level
                    }
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
// forward / backward, 64 < N <= 128: COLUMN-owned blocking (fwdbwd_cols2 below; round 1 introduced it with four utterances per
// eight-wave block -- wave w owning 16 columns -- which round 3's two-utterance four-wave blocks beat at every batch size and
// round 5 removed).  A wave owns its columns; the 16-lane DPP rows of the wave split the 128 source states into groups of 32, so a
// transition is one v_fmac_f64_dpp (the e value comes from lane n of the row) and the row-group partial sums of a column sit in
// the SAME wave: v_permlane32_swap / v_permlane16_swap (gfx950) turn them into totals, and the whole finalisation (log,
// emission, store, exp) runs on the wave that produced the sums -- no partial sums through the LDS, ONE barrier per frame
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

// The same column-owned scheme with TWO utterances per block of FOUR waves (round 3).  A wave owns 32 columns (two per
// lane: 64 transition entries = 128 registers), so a block is one wave per SIMD and two blocks share a CU: two waves per SIMD
// as before, but from blocks that do not share a barrier.  With four utterances behind one barrier all eight waves were in
// the same phase -- SQ counters of fwdbwd_cols: vector issue 63 % of the kernel, 40 % of every wave's time in s_waitcnt (the e
// vector right after the barrier, the log / exp table look-ups of the finalisation) with its SIMD partner waiting on the
// same thing.  Independent blocks drift apart and one's product phase runs under the other's waits (what merging the two
// Viterbi blocks of a CU behind one barrier had shown from the other side: 1.37 -> 2.68 ms).  The four accumulators of a
// lane are now (utterance, column half) instead of four utterances; the swap network is unchanged and leaves combination
// m = 2 k + h in row m: row g of a wave finalises utterance g >> 1, column 32 w + 16 (g & 1) + c.
// TS = storage type of b and alpha / beta.  float (`precision=np.float32`): the arithmetic stays fp64 -- A and pi are
// float64 arrays in the reference, so alpha[t-1] + A[:, j] is a float64 sum there too -- and every alpha[t, j] / beta[t, i]
// is rounded to float where the reference's float32 array rounds it, at the store (hmm.py:93-99, 124-130); the next
// frame continues from the ROUNDED value, as the reference reads its array back.
// UB = 1 (round 5): ONE utterance per block, for batches that leave CUs idle anyway (a single utterance through the reference's
// calc_prob / _forward / train, hmm.py:268-313).  The frame is a chain of dependent instructions walked by a lone wave per
// SIMD, so its duration is its instruction count: without the second utterance's 64 products per lane and its half of the
// swap network a frame is ~150 instructions instead of ~230.  Rows 0-1 of a wave then hold column half 0, rows 2-3 half 1
// (the odd rows repeat the even ones and store nothing); sums are associated as with two utterances: the same bits.
template <bool BACKWARD, typename TS = double, int UB = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwdbwd_cols2(
    const TS* __restrict__ b, const double* __restrict__ Aexp, const double* __restrict__ Alog,
    const double* __restrict__ pi, const int32_t* __restrict__ lens, TS* __restrict__ out,
    double* __restrict__ loglik, int U, int T, int N, unsigned* __restrict__ nan_flag, unsigned nan_epoch) {
    constexpr int NP = 128, NW = 4, NM = 2 * NW;   // NM: partial maxima per utterance (wave x column half)
    // a NaN emission: np.logaddexp carries it into every later (earlier) frame, the maxima here drop it.  The value the
    // lanes finalise is NaN exactly then; one compare into the scalar unit per frame raises the flag for nanfix.hip
    unsigned long long saw_nan = 0;
    __shared__ double Es[2][UB][NP];      // exp(prev - shift) of the frame being consumed
    __shared__ double Cs[2][UB][NP];      // prev itself (log domain) for the exact fallback
    __shared__ double Ms[2][UB][NM];      // per DPP row (16 columns) max of prev
    __shared__ float Lb[2][UB][NM];       // per DPP row max of the EMISSIONS of the frame whose e row goes into this buffer
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
    const int k = UB == 2 ? g >> 1 : 0, h = UB == 2 ? g & 1 : g >> 1;
    const bool primary = UB == 2 || (g & 1) == 0;             // UB = 1: the odd rows duplicate the even ones
    const int j = 32 * w + 16 * h + c;
    const int jc = j < N ? j : N - 1;
    const int uu = (u0 + k < U) ? u0 + k : U - 1;
    const int klen = (u0 + k < U) ? lens[u0 + k] : 0;
    const TS* bp = b + (size_t)uu * T * N + jc;
    TS* op = out ? out + (size_t)uu * T * N + jc : nullptr;
    auto frame_of = [&](int step) { return BACKWARD ? klen - 1 - step : step; };
    auto clampt = [&](int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };

    // step 0
    {
        double prev = LN_NEG_INF;
        if (j < N && klen > 0) {
            const int t0 = frame_of(0);
            const double bv = (double)bp[(size_t)t0 * N];
            if (BACKWARD) {
                if (op && primary) op[(size_t)t0 * N] = (TS)0.0;
                prev = 0.0 + bv;
            } else {
                prev = rnd<TS>(pi[j] + bv);
                if (op && primary) op[0] = (TS)prev;
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
    double bnext = (double)bp[(size_t)clampt(frame_of(1)) * N];
    double bnext2 = (double)bp[(size_t)clampt(frame_of(2)) * N];       // emissions are prefetched TWO frames ahead (see below)
    // The shift of a row lags one frame (the maximum of the row just consumed: no reduction on the frame's path) -- which
    // assumed that consecutive frames' emissions have about the same SIZE.  On data a model does not fit they do not (MFCCs of
    // a test signal against a random model: row maxima around -2000 with a frame-to-frame spread of 800 nats), every e row
    // then underflowed or overflowed and every column took the exact fallback: 35 us a frame instead of 0.5 (round 5,
    // tools/mismatch.py).  The level of the emission row of the frame being produced -- max_j b[t, j], to float accuracy,
    // formed one frame ahead from the prefetched row and combined through the LDS like the row maxima -- is therefore added
    // to the lagged shift: what is left to the exponent range is the spread WITHIN a row.
    {
        const float lv = row_max_f32(j < N ? (float)bnext : -__builtin_huge_valf());
        if (c == 0) Lb[0][k][2 * w + h] = lv;
    }
    drain_vmem();                                  // areg resident: no vmcnt guards in the loop
    __syncthreads();

    for (int step = 1; step < maxlen; ++step) {
        // (priority by phase -- finalisation ahead of the partner block's products, as in the MFCC kernel -- was measured in
        // round 4 and LOSES here: 1.364 / 1.322 ms against 1.319 / 1.263 with the alternating periods below)
        if ((step & 7) == 0) alternate_priority(cls);
        const int cur = (step - 1) & 1, nxt = step & 1;
        const double bcur = bnext;
        bnext = bnext2;
        bnext2 = (double)bp[(size_t)clampt(frame_of(step + 2)) * N];
        {
            // level of the row after this one, for the next turn's shift: from a value that was requested a whole frame ago
            // (taken from the row requested in this turn, the wait for it sat in every frame: forward 1.31 -> 1.68 ms)
            const float lv = row_max_f32(j < N ? (float)bnext : -__builtin_huge_valf());
            if (c == 0) Lb[nxt][k][2 * w + h] = lv;
        }
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
        double s02, s13;
        if constexpr (UB == 2) {
            swap32_f64(acc[0][0], acc[1][0]);
            swap32_f64(acc[0][1], acc[1][1]);
            s02 = acc[0][0] + acc[1][0];
            s13 = acc[0][1] + acc[1][1];
        } else {                                   // rows 0-1: half 0 (g0 + g2, g1 + g3), rows 2-3: half 1
            swap32_f64(acc[0][0], acc[0][1]);
            s02 = acc[0][0] + acc[0][1];
            s13 = s02;
        }
        swap16_f64(s02, s13);
        const double sum = s02 + s13;

        bool prev_nan = false;
        if (step < klen) {
            const double Sc = Ss[cur][k];
            double val;
            if (j >= N || (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF)) {   // (a padding column: nothing to redo)
                val = Sc + log_sum(sum, ltab);
            } else {                               // exact log-domain column (rare)
                val = exact_column_lse(&Cs[cur][k][0], Alog + jc, N);
                atomicAdd(nan_flag + 1, 1u);       // diagnostic count of such columns (lnasr_debug_fallback_columns)
            }
            const int t = frame_of(step);
            double prev;
            if (BACKWARD) {
                val = rnd<TS>(val);
                if (j < N && op && primary) op[(size_t)t * N] = (TS)val;
                prev = val + bcur;
            } else {
                prev = rnd<TS>(val + bcur);
                if (j < N && op && primary) op[(size_t)t * N] = (TS)prev;
            }
            if (j >= N) prev = LN_NEG_INF;
            prev_nan = prev != prev;
            double S = Ms[cur][k][0];
            float lvl = Lb[cur][k][0];
#pragma unroll
            for (int e = 1; e < NM; ++e) {
                S = fmax(S, Ms[cur][k][e]);
                lvl = fmaxf(lvl, Lb[cur][k][e]);
            }
            S += (double)lvl;                      // the lagged maximum, carried to this frame's emission level
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


// One utterance per FOUR-wave block with the logarithm and the exponential taken OFF the frame's dependent chain (fp64
// storage; round 5).  SQ counters of fwdbwd_cols2<.., 1> on one utterance (profiles/r05k_*): 164 vector instructions per wave
// and frame, 657 per CU -- 656 cycles of issue -- in a frame of 1,758 cycles: a lone wave per SIMD walking
//     e row from the LDS -> products -> swaps -> log (table) -> + b -> exp (table) -> LDS -> barrier
// issues 37 % of the time.  (Eight waves of 16 columns -- built, measured, removed -- shorten each wave's chain but finalise every column
// four times over: 1,050 instructions per CU, 0.72 ms against 0.75.)  Here the chain loses its log / exp half:
//     e[t, j] = sum[t, j] * G[t, j],   G[t, j] = exp((Sh[t-1] - Sh[t]) + b[t, j]),
// where Sh[t] -- the shift the row e[t, :] is expressed in -- is the maximum of the log-domain row of frame t - 2 plus the levels
// of the emission rows t - 1 and t: known when the frame starts, so G is formed beside the products and the chain is  LDS -> products -> swaps -> one multiply -> LDS ->
// barrier.  The log-domain value  Sh[t-1] + log(sum[t, j]) + b[t, j]  (the result, the input of the exact fallback and of the
// shift two frames on) is evaluated one iteration LATER, in the same straight-line block as the next frame's products, where
// it fills the issue slots their dependences leave.  A sum outside [1e-290, 1e290] (or a non-finite one) is redone from the
// log-domain row of the frame before, exactly as in the other kernels -- that row is complete by then, a barrier lies in
// between -- so whatever a degenerate e did to later sums is caught by the same test; results agree with the log-round-trip
// kernels to the rounding of one multiplication.  (Round 2 measured the idea at 1024 utterances and dropped it: with the
// chip full the chain's instructions still have to issue.  With one utterance per CU only the chain's LENGTH counts.)
template <bool BACKWARD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fwdbwd_lat4(
    const double* __restrict__ b, const double* __restrict__ Aexp, const double* __restrict__ Alog,
    const double* __restrict__ pi, const int32_t* __restrict__ lens, double* __restrict__ out,
    double* __restrict__ loglik, int U, int T, int N, unsigned* __restrict__ nan_flag, unsigned nan_epoch) {
    constexpr int NP = 128, NW = 4, NM = 2 * NW;
    unsigned long long saw_nan = 0;
    __shared__ double Es[2][NP];          // e row (linear domain) of the frame being consumed / produced
    __shared__ double Cs[2][NP];          // log-domain row by frame parity (exact fallback, likelihood)
    __shared__ double Ms[2][NM];          // per DPP row (16 columns) maximum of the log-domain row, by frame parity
    __shared__ float Lb[2][NM];           // per DPP row maximum of the EMISSIONS of the frame whose e row goes into this buffer
    __shared__ double etab[kExpTab];
    __shared__ double ltab[2 * kLogTab];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    exp_table_init(etab, threadIdx.x, 64 * NW);
    log_table_init(ltab, threadIdx.x, 64 * NW);   // visible after the first __syncthreads below
    const int u = blockIdx.x;
    const int g = lane >> 4, c = lane & 15;
    const int h = g >> 1;                          // finalisation role: rows 0-1 column half 0, rows 2-3 half 1
    const bool primary = (g & 1) == 0;             // the odd rows repeat the even ones and store nothing
    const int j = 32 * w + 16 * h + c;
    const int jc = j < N ? j : N - 1;

    double areg[2][32];                            // product role: columns 32 w + 16 q + c, source rows 32 g + r
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int i = 32 * g + r, jq = 32 * w + 16 * q + c;
            areg[q][r] = (i < N && jq < N) ? Aexp[(size_t)i * N + jq] : 0.0;
        }
    const int klen = lens[u];
    if (klen <= 0) return;
    const double* bp = b + (size_t)u * T * N + jc;
    double* op = (out && j < N && primary) ? out + (size_t)u * T * N + jc : nullptr;
    auto frame_of = [&](int step) { return BACKWARD ? klen - 1 - step : step; };
    auto clampt = [&](int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };
    auto row_maxima = [&](int buf) {
        double S = Ms[buf][0];
#pragma unroll
        for (int e = 1; e < NM; ++e) S = fmax(S, Ms[buf][e]);
        return S;
    };

    // step 0: the log-domain row, its maximum, e in that shift
    {
        double prev = LN_NEG_INF;
        if (j < N) {
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
        Cs[0][j] = prev;
        const double m = row_max_dpp(prev);
        if (c == 0 && primary) Ms[0][2 * w + h] = m;
    }
    __syncthreads();
    double sh_e;                                   // Sh[step - 1]: the shift of the e row an iteration consumes
    {
        sh_e = row_maxima(0);
        const double prev = Cs[0][j];
        Es[0][j] = (sh_e == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp_clamped(prev - sh_e, etab);
    }
    double bnext = bp[(size_t)clampt(frame_of(1)) * N];
    double bnext2 = bp[(size_t)clampt(frame_of(2)) * N];              // emissions are prefetched two frames ahead: the level of a
    {                                                                  // row is formed from a value requested a whole frame ago
        const float lv = row_max_f32(j < N ? (float)bnext : -__builtin_huge_valf());     // level of the next emission row
        if (c == 0 && primary) Lb[0][2 * w + h] = lv;
    }
    drain_vmem();                                  // areg resident: no vmcnt guards in the loop
    __syncthreads();

    // One straight-line block per iteration: the e row of frame `step` (the chain) and, beside it, the log-domain evaluation
    // of frame step - 1.  `eval` = 0 in the first turn (frame 0 is complete): the evaluation runs on a harmless sum and its
    // stores are masked.  The last turn (step == klen) only evaluates frame klen - 1.
    // A wave issues in order and is alone on its SIMD, so what overlaps is what the instruction stream interleaves -- and left
    // to itself the scheduler put the whole evaluation between the products and the swaps and the exponential of G after them
    // (0.62 ms: the chain as long as before).  The stream is therefore laid out by hand: sixteen groups of four products, a
    // slice of the other work (table exp and log spelled out: devmath.h's exp_nonpos / log_sum step by step) after each,
    // pinned with sched_barrier.
#define LN_PROD(R)                                                        \
    fmac_bcast<R>(acc[0][0], ek[0], areg[0][R]);                         \
    fmac_bcast<R>(acc[0][1], ek[1], areg[0][16 + R]);                    \
    fmac_bcast<R>(acc[1][0], ek[0], areg[1][R]);                         \
    fmac_bcast<R>(acc[1][1], ek[1], areg[1][16 + R])
#define LN_PIN __builtin_amdgcn_sched_barrier(0)
    constexpr double kInvStep = 369.32993046757463, kStepHi = 2.7076061740622863e-03, kStepLo = 9.058776616587108e-20;
    constexpr double kMagic = 6755399441055744.0, kLn2 = 0.69314718055994530942;
    double sum_prev = 1.0, b_prev = 0.0, sh_pp = 0.0;
    float lvl_prev = 0.0f;
    for (int step = 1; step <= klen; ++step) {
        const int cur = (step - 1) & 1, nxt = step & 1;
        const bool eval = step >= 2;
        const double bcur = bnext;
        bnext = bnext2;
        bnext2 = bp[(size_t)clampt(frame_of(step + 2)) * N];
        double ek[2];
        ek[0] = Es[cur][32 * g + c];
        ek[1] = Es[cur][32 * g + 16 + c];
        // the shift of the row this iteration produces: the maximum of the log-domain row two frames back (frame 0 for step 1)
        double mrow[NM];
        float lrow[NM];
#pragma unroll
        for (int e = 0; e < NM; ++e) {
            mrow[e] = Ms[eval ? nxt : 0][e];
            lrow[e] = Lb[cur][e];
        }

        // ---- evaluation of frame f = step - 1, part 1 (registers only): range test, mantissa, table address
        const int fb = cur, pb = nxt;              // parity buffers of frame f and of frame f - 1
        const bool safe = j >= N || (sum_prev >= kSafeSum && sum_prev <= 1e290 && sh_pp != LN_NEG_INF);   // (padding columns: nothing to redo)
        const double ls = safe ? sum_prev : 1.0;
        const int lhi = __double2hiint(ls);
        const int le = (lhi >> 20) - 1023;
        const int lj = (lhi >> 13) & (kLogTab - 1);
        const double lm = __hiloint2double((lhi & 0x000FFFFF) | 0x3FF00000, __double2loint(ls));
        const double lt0 = ltab[2 * lj], lt1 = ltab[2 * lj + 1];
        LN_PIN;
        double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        LN_PROD(0);
        double m01 = vmax_f64(mrow[0], mrow[1]), m23 = vmax_f64(mrow[2], mrow[3]);     // (a NaN row surfaces through prev != prev)
        LN_PIN;
        LN_PROD(1);
        double m45 = vmax_f64(mrow[4], mrow[5]), m67 = vmax_f64(mrow[6], mrow[7]);
        {
            const float lv = row_max_f32(j < N ? (float)bnext : -__builtin_huge_valf());     // level of the row after this one
            if (c == 0 && primary) Lb[nxt][2 * w + h] = lv;
        }
        LN_PIN;
        LN_PROD(2);
        m01 = vmax_f64(m01, m23);
        m45 = vmax_f64(m45, m67);
        LN_PIN;
        LN_PROD(3);
        // ... carried forward by the LEVELS of the emission rows in between (max_j b[t, j] to float accuracy: formed a frame
        // ahead from the prefetched row, combined through the LDS like the maxima): whatever the size of the emissions and however
        // it jumps from frame to frame (MFCCs of a test signal against a random model: -2000 +- 800 nats), e stays near 1 and
        // the whole exponent range is left for the spread WITHIN a row.  Any value works as a shift as long as every lane computes
        // the same one -- these all read the same words.
        const double m2 = vmax_f64(m01, m45);
        const float l01 = fmaxf(fmaxf(lrow[0], lrow[1]), fmaxf(lrow[2], lrow[3]));
        const float l45 = fmaxf(fmaxf(lrow[4], lrow[5]), fmaxf(lrow[6], lrow[7]));
        const float lvl = fmaxf(l01, l45);
        const double sh_new = m2 + ((double)lvl + (eval ? (double)lvl_prev : 0.0));
        lvl_prev = lvl;
        const bool sh_ok = sh_new != LN_NEG_INF && sh_e != LN_NEG_INF && j < N && sh_new == sh_new;
        const double gx0 = sh_ok ? (sh_e - sh_new) + bcur : 0.0;
        LN_PIN;
        // ---- G = exp((Sh[t-1] - Sh[t]) + b[t, j]) (0 where a shift is -inf: the exact fallback takes over), part 1
        LN_PROD(4);
        double gx = gx0;
        asm("v_min_f64 %0, %1, %2" : "=v"(gx) : "v"(gx0), "v"(690.0));
        gx = vmax_f64(gx, -700.0);
        const double gkp = fma(gx, kInvStep, kMagic);
        LN_PIN;
        LN_PROD(5);
        const int gk = __double2loint(gkp);
        const double gt0 = etab[gk & (kExpTab - 1)];
        const double gkd = gkp - kMagic;
        LN_PIN;
        LN_PROD(6);
        // ---- evaluation, part 2: the logarithm (log_sum), the log-domain value, its store
        const double lr = fma(lm, lt0, -1.0);
        double gr = fma(-gkd, kStepHi, gx);
        LN_PIN;
        LN_PROD(7);
        double lq = fma(lr, -1.0 / 6.0, 0.2);
        gr = fma(-gkd, kStepLo, gr);
        LN_PIN;
        LN_PROD(8);
        lq = fma(lq, lr, -0.25);
        double gq = fma(gr, 1.0 / 24.0, 1.0 / 6.0);
        LN_PIN;
        LN_PROD(9);
        lq = fma(lq, lr, 1.0 / 3.0);
        gq = fma(gq, gr, 0.5);
        const double grr = gr * gr;
        LN_PIN;
        LN_PROD(10);
        lq = fma(lq, lr, -0.5);
        const double gp = fma(gq, grr, gr);
        LN_PIN;
        LN_PROD(11);
        lq = fma(lq, lr, 1.0);
        const double gt = __hiloint2double(__double2hiint(gt0) + ((gk >> 8) << 20), __double2loint(gt0));
        double Gv = fma(gt, gp, gt);
        LN_PIN;
        LN_PROD(12);
        double val = sh_pp + fma((double)le, kLn2, fma(lq, lr, lt1));
        Gv = gx0 > 690.0 ? __builtin_huge_val() : Gv;
        const double G = sh_ok ? Gv : 0.0;
        LN_PIN;
        LN_PROD(13);
        double prev = j < N ? val + b_prev : LN_NEG_INF;
        double* const opf = eval ? op : nullptr;
        const size_t oidx = (size_t)frame_of(eval ? step - 1 : 0) * N;
        if (opf) opf[oidx] = BACKWARD ? val : prev;
        if (eval) Cs[fb][j] = prev;
        LN_PIN;
        LN_PROD(14);
        double m = vmax_f64(prev, dpp_f64<0xB1>(prev));
        m = vmax_f64(m, dpp_f64<0x4E>(m));
        LN_PIN;
        LN_PROD(15);
        m = vmax_f64(m, dpp_f64<0x141>(m));
        m = vmax_f64(m, dpp_f64<0x140>(m));
        if (eval && c == 0 && primary) Ms[fb][2 * w + h] = m;
        LN_PIN;
        // row groups -> column halves: rows 0-1 half 0, rows 2-3 half 1, ((g0 + g2) + (g1 + g3))
        double a0 = acc[0][0] + acc[0][1], a1 = acc[1][0] + acc[1][1];
        swap32_f64(a0, a1);
        double s02 = a0 + a1, s13 = s02;
        swap16_f64(s02, s13);
        const double sum = s02 + s13;
        Es[nxt][j] = sum * G;                      // (the last turn writes a row nobody reads)
        // once in a blue moon: a column whose sum left the safe range is redone from the log-domain row of the frame before
        // (complete: a barrier ago), and what the straight line stored for it is overwritten before anybody reads it
        if (__builtin_expect(eval && __any(!safe), 0)) {
            if (!safe) {
                val = exact_column_lse(&Cs[pb][0], Alog + jc, N);
                atomicAdd(nan_flag + 1, 1u);       // diagnostic count of such columns (lnasr_debug_fallback_columns)
                prev = j < N ? val + b_prev : LN_NEG_INF;
                if (opf) opf[oidx] = BACKWARD ? val : prev;
                Cs[fb][j] = prev;
            }
            m = row_max_dpp(prev);
            if (c == 0 && primary) Ms[fb][2 * w + h] = m;
        }
        saw_nan |= __ballot(eval && prev != prev);
        sum_prev = sum;
        b_prev = bcur;
        sh_pp = sh_e;
        sh_e = sh_new;
        __syncthreads();
    }
#undef LN_PROD
#undef LN_PIN
    if (saw_nan != 0 && lane == 0) *nan_flag = nan_epoch;
    // calc_prob: lse_j alpha[L-1, j]  (hmm.py:274)
    if (!BACKWARD && loglik && w == 0) {
        const int fin = (klen - 1) & 1;
        const double mx = row_maxima(fin);
        double sacc = 0.0;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const double pv = Cs[fin][lane + 64 * hh];
            sacc += (mx == LN_NEG_INF || pv == LN_NEG_INF) ? 0.0 : exp(pv - mx);
        }
        sacc = wave_sum(sacc);
        if (lane == 0) loglik[u] = (mx == LN_NEG_INF) ? mx : mx + log(sacc);
    }
}

// ------------------------------------------------------------------------------------------
// Tiny models (N <= 16; round 4): the toy HMMs of the reference's own test scripts (test/hmm-test.py n = 4,
// test/gmmhmm-test.py n = 5) and segmentation-sized discrete HMMs (scripts/seg/seg.py, n = 4) through `train` /
// `calc_prob`.  The block kernels above spend two barriers and an LDS round trip per frame whatever N is: 2 us a frame,
// i.e. 0.38 ms for one forward pass over T = 190 -- three quarters of a Baum-Welch iteration of hmm-test.py.  Here, as in
// viterbi_tiny_kernel: one WAVE per utterance, no barrier; lane j (of DPP row 0) holds prev[j] and the column
// expA[:, j] in registers, the e values reach every lane through v_fmac_f64_dpp row_newbcast (one instruction per
// transition), the maximum is four DPP steps inside the row, emissions and results move through the wave's LDS slice 64
// frames at a time.  Same shift rule, same exact log-domain fallback for columns whose linear sum leaves the safe range,
// same rounding at the store for float storage as the kernels above.
//   FORWARD : Aexp = expA  [i][j], Alog = logA;   lane j:  a[i] = expA[i][j]
//   BACKWARD: Aexp = expA^T[j][i], Alog = logA^T; lane i:  a[j] = expA[i][j]   (prev[j] = beta[t+1, j] + b[t+1, j])
// ------------------------------------------------------------------------------------------
template <typename TS, bool BACKWARD, int NMAX>
__global__ __launch_bounds__(256) void fwdbwd_tiny_kernel(const TS* __restrict__ b, const double* __restrict__ Aexp,
                                                         const double* __restrict__ Alog, const double* __restrict__ pi,
                                                         const int32_t* __restrict__ lens, TS* __restrict__ out,
                                                         double* __restrict__ loglik, int U, int T, int N,
                                                         unsigned* __restrict__ nan_flag, unsigned nan_epoch) {
    constexpr int C = 64;
    __shared__ double bch[4][C * NMAX];                   // emissions of the chunk, [frame - lo][state], stride N
    __shared__ double och[4][C * NMAX + 1];               // results of the chunk, same layout (+ a slot for idle lanes)
    __shared__ double etab[kExpTab];
    __shared__ double ltab[2 * kLogTab];
    exp_table_init(etab, threadIdx.x, 256);
    log_table_init(ltab, threadIdx.x, 256);
    __syncthreads();                                      // tables; from here on the waves never synchronise
    constexpr int ODUMP = C * NMAX;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = blockIdx.x * 4 + w;
    if (u >= U) return;
    const int L = lens[u];
    if (L <= 0) return;
    const bool act = lane < N;
    const int j = act ? lane : N - 1;
    double a[NMAX], al[NMAX];                             // linear and log transition weights into this lane's state
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        a[i] = (i < N && act) ? Aexp[(size_t)i * N + j] : 0.0;
        al[i] = (i < N && act) ? Alog[(size_t)i * N + j] : LN_NEG_INF;
    }
    const TS* bu = b + (size_t)u * T * N;
    TS* ou = out ? out + (size_t)u * T * N : nullptr;
    double* bc = bch[w];
    double* oc = och[w];
    constexpr int EPL = (C * NMAX + 63) / 64;
    double stage[EPL];
    // chunk c covers steps 64 c .. 64 c + nfr - 1; step s is frame s (forward) or L - 1 - s (backward); lo = its lowest frame
    auto chunk_lo = [&](int s0, int nfr) { return BACKWARD ? L - s0 - nfr : s0; };
    auto fetch = [&](int s0) {
        const int nfr = L - s0 < C ? L - s0 : C;
        const int lo = chunk_lo(s0, nfr);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            stage[e] = idx < nfr * N ? ld(bu, (size_t)lo * N + idx) : 0.0;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int idx = lane + 64 * e;
            if (idx < C * N) bc[idx] = stage[e];
        }
    };
    unsigned long long saw_nan = 0;
    double prev = LN_NEG_INF;                             // log domain; lanes >= N stay at -inf
    fetch(0);
    stash();
    __builtin_amdgcn_wave_barrier();
    for (int s0 = 0; s0 < L; s0 += C) {
        const int nfr = L - s0 < C ? L - s0 : C;
        const int lo = chunk_lo(s0, nfr);
        if (s0 + C < L) fetch(s0 + C);                    // the next chunk's emissions travel during this one
        for (int k = 0; k < nfr; ++k) {
            const int fl = BACKWARD ? nfr - 1 - k : k;    // frame - lo
            const double bt_ = bc[fl * N + j];
            double val;
            if (s0 + k == 0) {                            // first step: alpha[0] = pi + b[0];  beta[L-1] = 0
                val = BACKWARD ? 0.0 : rnd<TS>(pi[j] + bt_);
            } else {
                double S = act ? prev : LN_NEG_INF;
                S = vmax_f64(S, dpp_f64<0xB1>(S));
                S = vmax_f64(S, dpp_f64<0x4E>(S));
                S = vmax_f64(S, dpp_f64<0x141>(S));
                S = vmax_f64(S, dpp_f64<0x140>(S));
                const bool dead = S == LN_NEG_INF;
                const double e = (dead || prev == LN_NEG_INF || !act) ? 0.0 : exp_clamped(prev - S, etab);
                double sum = 0.0;
                static_for<0, NMAX>([&](auto I) { fmac_bcast<decltype(I)::value>(sum, e, a[decltype(I)::value]); });
                const bool safe = sum >= kSafeSum && sum <= 1e290 && !dead;
                val = S + log_sum(safe ? sum : 1.0, ltab);
                if (__any(!safe && act)) {                // rare, wave-uniform: the exact log-domain column for every lane
                    double mx = LN_NEG_INF;
                    static_for<0, NMAX>([&](auto I) {
                        mx = fmax(mx, mov_bcast<decltype(I)::value>(prev) + al[decltype(I)::value]);
                    });
                    double s2 = 0.0;
                    static_for<0, NMAX>([&](auto I) {
                        const double z = mov_bcast<decltype(I)::value>(prev) + al[decltype(I)::value];
                        s2 += (mx == LN_NEG_INF || z == LN_NEG_INF) ? 0.0 : exp(z - mx);
                    });
                    const double ex = (mx == LN_NEG_INF || mx != mx) ? mx : mx + log(s2);
                    val = safe ? val : ex;
                }
                val = BACKWARD ? rnd<TS>(val) : rnd<TS>(val + bt_);
            }
            oc[act ? fl * N + j : ODUMP] = val;
            prev = act ? (BACKWARD ? val + bt_ : val) : LN_NEG_INF;   // idle lanes stay at -inf: they are broadcast too
            saw_nan |= __ballot(prev != prev);
        }
        __builtin_amdgcn_wave_barrier();
        if (ou) {                                         // the chunk's rows leave whole
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int idx = lane + 64 * e;
                if (idx < nfr * N) st(ou, (size_t)lo * N + idx, oc[idx]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (s0 + C < L) stash();
        __builtin_amdgcn_wave_barrier();
    }
    if (saw_nan != 0 && lane == 0) *nan_flag = nan_epoch;
    if (!BACKWARD && loglik) {                            // calc_prob: lse_j alpha[L-1, j]  (hmm.py:274)
        double mx = act ? prev : LN_NEG_INF;
        mx = vmax_f64(mx, dpp_f64<0xB1>(mx));
        mx = vmax_f64(mx, dpp_f64<0x4E>(mx));
        mx = vmax_f64(mx, dpp_f64<0x141>(mx));
        mx = vmax_f64(mx, dpp_f64<0x140>(mx));
        double sacc = (!act || mx == LN_NEG_INF || prev == LN_NEG_INF) ? 0.0 : exp(prev - mx);
        sacc += dpp_f64<0xB1>(sacc);
        sacc += dpp_f64<0x4E>(sacc);
        sacc += dpp_f64<0x141>(sacc);
        sacc += dpp_f64<0x140>(sacc);
        const double nanv = act ? prev : 0.0;             // a NaN alpha anywhere makes the likelihood NaN
        const bool any_nan = __ballot(nanv != nanv) != 0;
        if (lane == 0) loglik[u] = any_nan ? __builtin_nan("") : ((mx == LN_NEG_INF) ? mx : mx + log(sacc));
    }
}

template <bool BACKWARD>
static int fwdbwd_tiny_launch(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* out,
                              double* loglik) {
    Model& md = ctx->model;
    const int N = md.N;
    const double* Alin = BACKWARD ? md.expAT.as<double>() : md.expA.as<double>();
    const double* Alog = BACKWARD ? md.logAT.as<double>() : md.logA.as<double>();
    const double* pi = md.logpi.as<double>();
    unsigned* flag = ctx->nan_flag.as<unsigned>() + 2;
    const dim3 grid((U + 3) / 4), block(256);
    const int nmax = N <= 4 ? 4 : (N <= 8 ? 8 : 16);
#define LN_FT(TS, NM)                                                                                                \
    LN_LAUNCH(ctx, BACKWARD ? "backward_tiny" : "forward_tiny", (fwdbwd_tiny_kernel<TS, BACKWARD, NM>), grid, block, 0, \
              (const TS*)b, Alin, Alog, pi, d_lens, (TS*)out, BACKWARD ? (double*)nullptr : loglik, U, T, N, flag,   \
              ctx->nan_epoch)
    if (dtype == LNASR_F32) {
        if (nmax == 4) LN_FT(float, 4);
        else if (nmax == 8) LN_FT(float, 8);
        else LN_FT(float, 16);
    } else {
        if (nmax == 4) LN_FT(double, 4);
        else if (nmax == 8) LN_FT(double, 8);
        else LN_FT(double, 16);
    }
#undef LN_FT
    ctx->fwd_nan_tested = true;
    return 0;
}

int forward_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* alpha,
                   double* loglik) {
    LN_DEVPTR(b); LN_DEVPTR(d_lens); LN_DEVPTR(alpha); LN_DEVPTR(loglik);
    Model& md = ctx->model;
    const int N = md.N;
    ctx->fwd_nan_tested = false;
    if (N <= dispatch::kTinyMaxN && !sw().no_tinyfwd && (dtype == LNASR_F64 || !sw().f32_generic))
        return fwdbwd_tiny_launch<false>(ctx, b, U, T, d_lens, dtype, alpha, loglik);
    // precision=np.float32: float storage, every alpha rounded at its store and the recursion continued from the
    // rounded value, as the reference's float32 array does (hmm.py:93-99).  N <= 128: the column-owned kernel with float
    // storage; 128 < N <= 256: the on-chip kernel with float storage; otherwise (and under LNASR_F32_GENERIC /
    // LNASR_NO_REGFWD / LNASR_NO_COLFWD) the storage-templated generic kernel below -- the same rounding everywhere.
    if (dtype == LNASR_F32 && !sw().f32_generic) {
        if (N <= dispatch::kRegMaxN && !sw().no_regfwd && !sw().no_colfwd) {
            if (dispatch::fwdbwd_one_per_block(U, ctx->prop.multiProcessorCount))
                LN_LAUNCH(ctx, "forward_cols1_f32", (fwdbwd_cols2<false, float, 1>), dim3(U), dim3(256), 0, (const float*)b,
                          md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens, (float*)alpha, loglik, U,
                          T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
            else
                LN_LAUNCH(ctx, "forward_cols2_f32", (fwdbwd_cols2<false, float>), dim3((U + 1) / 2), dim3(256), 0,
                          (const float*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                          (float*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
            ctx->fwd_nan_tested = true;
            return 0;
        }
        if (N > dispatch::kRegMaxN && N <= dispatch::kOnChipMaxN && !sw().no_t256)
            return trellis256_device(ctx, 1, b, U, T, d_lens, LNASR_F32, alpha, loglik);
    }
    if (dtype == LNASR_F64 && N <= dispatch::kRegMaxN && !sw().no_regfwd) {
        // (17 ... 64 states: the two-barrier blocks for batches; a few utterances take the column-owned latency kernels below,
        // whose 128-column tiles run half empty but whose frame is 0.5 us against 2-4 -- round 5)
        if (N <= dispatch::kReg64MaxN && !(dispatch::fwdbwd_latency_tier(U, ctx->prop.multiProcessorCount) && !sw().no_latfwd &&
                                           !sw().no_colfwd))
            LN_LAUNCH(ctx, "forward_reg_n64", (fwdbwd_reg<false, 1, 2, 32, 2, 3>), dim3((U + 1) / 2), dim3(128), 0,
                      (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                      (double*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
        else
            // column-owned waves, two utterances per four-wave block, two blocks per CU (fwdbwd_cols2): measured against the
            // eight-wave kernels at every batch size -- 16 ... 512 utterances 0.83-0.87 ms against 0.96-1.02 (fwdbwd_reg, two
            // per block), 1000 utterances 1.23 against 1.45 (four per eight-wave block: removed in round 5)
            if (!sw().no_colfwd)
            {
                if (dispatch::fwdbwd_latency_tier(U, ctx->prop.multiProcessorCount) && !sw().no_latfwd)
                    LN_LAUNCH(ctx, "forward_lat4_n128", (fwdbwd_lat4<false>), dim3(U), dim3(256), 0,
                              (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                              (double*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                else if (dispatch::fwdbwd_one_per_block(U, ctx->prop.multiProcessorCount))
                    LN_LAUNCH(ctx, "forward_cols1_n128", (fwdbwd_cols2<false, double, 1>), dim3(U), dim3(256), 0,
                              (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                              (double*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                else
                    LN_LAUNCH(ctx, "forward_cols2_n128", (fwdbwd_cols2<false>), dim3((U + 1) / 2), dim3(256), 0,
                              (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                              (double*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                ctx->fwd_nan_tested = true;
            }
            else if (!dispatch::fwdbwd_reg_four_per_block(U, ctx->prop.multiProcessorCount))      // row-owned waves, partial sums through the LDS
                LN_LAUNCH(ctx, "forward_reg_n128", (fwdbwd_reg<false, 2, 8, 16, 2, 4>), dim3((U + 1) / 2), dim3(512), 0,
                          (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                          (double*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
            else
                LN_LAUNCH(ctx, "forward_reg_n128", (fwdbwd_reg<false, 2, 8, 16, 4, 2>), dim3((U + 3) / 4), dim3(512), 0,
                          (const double*)b, md.expA.as<double>(), md.logA.as<double>(), md.logpi.as<double>(), d_lens,
                          (double*)alpha, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
        return 0;
    }
    if (dtype == LNASR_F64 && N > dispatch::kRegMaxN && N <= dispatch::kOnChipMaxN && !sw().no_t256)
        return trellis256_device(ctx, 1, b, U, T, d_lens, LNASR_F64, alpha, loglik);
    if (dtype == LNASR_F64 && N > dispatch::kOnChipMaxN && dispatch::fwdbwd_framepar(N, U))
        return fwdbwd_frames_device(ctx, false, (const double*)b, U, T, d_lens, (double*)alpha, loglik);
    if (dtype == LNASR_F64 && N > dispatch::kRegMaxN && N <= dispatch::kStreamMaxN && !sw().no_stream)
        return fwdbwd_stream_device(ctx, false, (const double*)b, U, T, d_lens, (double*)alpha, loglik);
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
    LN_DEVPTR(b); LN_DEVPTR(d_lens); LN_DEVPTR(beta);
    Model& md = ctx->model;
    const int N = md.N;
    ctx->fwd_nan_tested = false;
    const double* expAT = md.expAT.as<double>();
    if (N <= dispatch::kTinyMaxN && !sw().no_tinyfwd && (dtype == LNASR_F64 || !sw().f32_generic))
        return fwdbwd_tiny_launch<true>(ctx, b, U, T, d_lens, dtype, beta, nullptr);
    if (dtype == LNASR_F32 && !sw().f32_generic) {       // float storage, rounded at every store (hmm.py:124-130); see forward_device
        if (N <= dispatch::kRegMaxN && !sw().no_regfwd && !sw().no_colfwd) {
            if (dispatch::fwdbwd_one_per_block(U, ctx->prop.multiProcessorCount))
                LN_LAUNCH(ctx, "backward_cols1_f32", (fwdbwd_cols2<true, float, 1>), dim3(U), dim3(256), 0, (const float*)b,
                          expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (float*)beta, (double*)nullptr, U, T, N,
                          ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
            else
                LN_LAUNCH(ctx, "backward_cols2_f32", (fwdbwd_cols2<true, float>), dim3((U + 1) / 2), dim3(256), 0,
                          (const float*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (float*)beta,
                          (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
            ctx->fwd_nan_tested = true;
            return 0;
        }
        if (N > dispatch::kRegMaxN && N <= dispatch::kOnChipMaxN && !sw().no_t256)
            return trellis256_device(ctx, 2, b, U, T, d_lens, LNASR_F32, beta, nullptr);
    }
    if (dtype == LNASR_F64 && N <= dispatch::kRegMaxN && !sw().no_regfwd) {
        if (N <= dispatch::kReg64MaxN && !(dispatch::fwdbwd_latency_tier(U, ctx->prop.multiProcessorCount) && !sw().no_latfwd &&
                                           !sw().no_colfwd))
            LN_LAUNCH(ctx, "backward_reg_n64", (fwdbwd_reg<true, 1, 2, 32, 2, 3>), dim3((U + 1) / 2), dim3(128), 0,
                      (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                      (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
        else
            if (!sw().no_colfwd)
            {
                if (dispatch::fwdbwd_latency_tier(U, ctx->prop.multiProcessorCount) && !sw().no_latfwd)
                    LN_LAUNCH(ctx, "backward_lat4_n128", (fwdbwd_lat4<true>), dim3(U), dim3(256), 0,
                              (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                              (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                else if (dispatch::fwdbwd_one_per_block(U, ctx->prop.multiProcessorCount))
                    LN_LAUNCH(ctx, "backward_cols1_n128", (fwdbwd_cols2<true, double, 1>), dim3(U), dim3(256), 0,
                              (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                              (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                else
                    LN_LAUNCH(ctx, "backward_cols2_n128", (fwdbwd_cols2<true>), dim3((U + 1) / 2), dim3(256), 0,
                              (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                              (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 2, ctx->nan_epoch);
                ctx->fwd_nan_tested = true;
            }
            else if (!dispatch::fwdbwd_reg_four_per_block(U, ctx->prop.multiProcessorCount))
                LN_LAUNCH(ctx, "backward_reg_n128", (fwdbwd_reg<true, 2, 8, 16, 2, 4>), dim3((U + 1) / 2), dim3(512), 0,
                          (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                          (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
            else
                LN_LAUNCH(ctx, "backward_reg_n128", (fwdbwd_reg<true, 2, 8, 16, 4, 2>), dim3((U + 3) / 4), dim3(512), 0,
                          (const double*)b, expAT, md.logAT.as<double>(), md.logpi.as<double>(), d_lens, (double*)beta,
                          (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
        return 0;
    }
    if (dtype == LNASR_F64 && N > dispatch::kRegMaxN && N <= dispatch::kOnChipMaxN && !sw().no_t256)
        return trellis256_device(ctx, 2, b, U, T, d_lens, LNASR_F64, beta, nullptr);
    if (dtype == LNASR_F64 && N > dispatch::kOnChipMaxN && dispatch::fwdbwd_framepar(N, U))
        return fwdbwd_frames_device(ctx, true, (const double*)b, U, T, d_lens, (double*)beta, nullptr);
    if (dtype == LNASR_F64 && N > dispatch::kRegMaxN && N <= dispatch::kStreamMaxN && !sw().no_stream)
        return fwdbwd_stream_device(ctx, true, (const double*)b, U, T, d_lens, (double*)beta, nullptr);
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

}  // namespace lnasr
