// Non-finite data, the reference's way.
//
// numpy's max / argmax / logaddexp PROPAGATE NaN; the kernels' v_max_f64 / ds_max_f64 are IEEE maxNum and return the
// other operand, and the table-driven exp clamps its argument, so a NaN that enters a trellis here stays in its own
// cell while the reference turns every later frame into NaN.  What the reference returns is a closed form of where
// the first NaN sits, so the tuned kernels stay as they are and a repair pass rewrites their outputs:
//
//   * emission (gmm.py:59-68, gmmhmm.py:64-67): a NaN OR an infinite feature in frame t makes every component
//     density of that frame NaN -- the reference multiplies (x - mu) with the FULL inverse matrix, whose zero
//     off-diagonals meet the infinity (inf * 0) -- hence bm[:, :, t] = b[:, t] = NaN (golden `gmmhmm_nonfinite`).  With a
//     genuinely full Sigma, or D = 1, an infinity meets no zero and the reference's components come out -inf or NaN by
//     the signs of the inverse's entries: those frames are re-scored in the reference's arithmetic (nan_rows_full_kernel);
//   * Viterbi (hmm.py:162-171): with t1 the first frame whose emissions hold a NaN and j1 the first such state,
//     v[t1, j] = NaN for the NaN states, v[t > t1, :] = NaN, bt[t1+1, :] = j1, bt[t > t1+1, :] = 0, so
//     path[t > t1] = 0, path[t1] = j1 and path[t < t1] is the back-trace from (t1, j1) over the untouched frames;
//   * forward / backward (hmm.py:97-99, 128-130): np.logaddexp carries a NaN the same way -- alpha[t1, j] NaN for the NaN
//     states and alpha[t > t1, :] NaN; beta[t] folds b[:, t+1], so beta[t < tl, :] is NaN for tl the LAST bad frame;
//   * likelihood (hmm.py:268-274): NaN for an utterance that holds a NaN frame.
//
// Cost on clean data: nothing is scanned by default.  The hot kernels test what they read anyway and store the running
// API call's number (ctx->nan_epoch, LN_ENTER) into a word of ctx->nan_flag -- emit_diag_mfma and emit_single_direct the
// features they stage (word 0: "a NaN or an infinite feature"; the single-GMM kernel also produces its NaN rows itself),
// the back-trace kernel the cells of v (word 1: "a NaN or -inf cell", which is all a NaN emission can leave there),
// fwdbwd_cols2 the value its lanes finalise (word 2: NaN exactly when the emission is) -- and
// every kernel here starts by comparing its word with the call's epoch and returns.  A word is "raised" only while it
// EQUALS the running epoch, so nothing is ever reset.  Paths whose kernels have no fused test (`force`) scan their input
// once at memory speed: b by nan_locate_kernel, the observations by nan_rows_kernel, which raises word 0 for the kernels
// that walk one utterance per block.  Every fused test was A/B-measured before it stayed: inside viterbi_reg1b (at its
// 128-VGPR limit) it spilled seven registers, 1.26 -> 1.47 ms; inside the float32 emission kernel it cost 0.25 ms of 1.87;
// turning bad features into NaN inside emit_diag_mfma (rows for free afterwards) cost 0.02-0.03 ms that decode would
// pay for rows it never reads.  None of those stayed.
#include "common.h"
#include "devmath.h"

namespace lnasr {

namespace {

constexpr unsigned long long kNoBad = ~0ull;
constexpr int kLocateMaxChunks = 64;

template <typename T>
__device__ __forceinline__ T quiet_nan();
template <>
__device__ __forceinline__ double quiet_nan<double>() { return __builtin_nan(""); }
template <>
__device__ __forceinline__ float quiet_nan<float>() { return __builtin_nanf(""); }

// Frames with a non-finite feature: their rows of b [F,N] and bm [F,N,M] become NaN.  Flat coalesced scan of obs;
// the wave that meets a bad element fills that frame's rows (several bad elements of a frame: filled again, harmless).
template <typename TB>
__global__ __launch_bounds__(256) void nan_rows_kernel(const double* __restrict__ obs, long F, int D, int N, int M,
                                                       TB* __restrict__ b, TB* __restrict__ bm, unsigned* flag,
                                                       unsigned epoch, int force) {
    if (!force && *flag != epoch) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const long E = F * (long)D;
    const TB nan = quiet_nan<TB>();
    for (long base = (long)blockIdx.x * 256; base < E; base += (long)gridDim.x * 256) {
        const long e = base + tid;
        const bool bad = e < E && !__builtin_isfinite(obs[e]);
        unsigned long long mask = __ballot(bad);
        while (mask) {                                           // wave-uniform
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const long f = (base + (tid & ~63) + l) / D;
            if (b)
                for (int j = lane; j < N; j += 64) b[f * N + j] = nan;
            if (bm)
                for (long j = lane; j < (long)N * M; j += 64) bm[f * N * M + j] = nan;
            if (lane == 0) *flag = epoch;
        }
    }
}

// np.logaddexp (the fold of utils.lse): equal operands (infinities of one sign included) x + ln 2; a NaN difference is NaN
__device__ __forceinline__ double logaddexp_np(double x, double y) {
    if (x == y) return x + 0.69314718055994530942;
    const double d = x - y;
    if (d > 0) return x + log1p(exp(-d));
    if (d <= 0) return y + log1p(exp(d));
    return d;
}

// Genuinely full covariance: an INFINITE feature meets no zeros in the inverse, and the reference's quadratic form
// ((x - mu) P) . (x - mu) comes out +inf or NaN component by component (inf - inf where the signs of P's entries and of
// the other coordinates disagree); and with D = 1 there is no off-diagonal at all, diagonal or not: the density is -inf.
// The frames that hold a non-finite feature are therefore re-scored here in the
// reference's own arithmetic and order (gmm.py:64-68, gmmhmm.py:64-67) instead of being filled with NaN: whether a sum of
// infinities is an infinity or a NaN does not depend on the order of its terms, so the pattern is the reference's.
template <typename TB>
__global__ __launch_bounds__(256) void nan_rows_full_kernel(const double* __restrict__ obs, long F, int D, int N, int M,
                                                            const double* __restrict__ mu, const double* __restrict__ prec,
                                                            const double* __restrict__ ivar1,
                                                            const double* __restrict__ gconst,
                                                            const double* __restrict__ logw, TB* __restrict__ b,
                                                            TB* __restrict__ bm, unsigned* flag, unsigned epoch, int force) {
    if (!force && *flag != epoch) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const long E = F * (long)D;
    for (long base = (long)blockIdx.x * 256; base < E; base += (long)gridDim.x * 256) {
        const long e = base + tid;
        const bool bad = e < E && !__builtin_isfinite(obs[e]);
        unsigned long long mask = __ballot(bad);
        while (mask) {                                           // wave-uniform
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const long f = (base + (tid & ~63) + l) / D;
            const double* x = obs + f * D;
            for (int j = lane; j < N; j += 64) {
                double r = 0.0;
                for (int m = 0; m < M; ++m) {
                    const size_t c = (size_t)j * M + m;
                    const double* mc = mu + c * D;
                    double q = 0.0;
                    if (prec) {
                        const double* pc = prec + c * D * D;
                        for (int d = 0; d < D; ++d) {
                            double t = 0.0;                      // ((x - mu) P)[d]
                            for (int k = 0; k < D; ++k) t += (x[k] - mc[k]) * pc[(size_t)k * D + d];
                            q += t * (x[d] - mc[d]);
                        }
                    } else {                                     // D = 1: the 1 x 1 inverse has no zero to meet the infinity
                        q = ((x[0] - mc[0]) * ivar1[c]) * (x[0] - mc[0]);
                    }
                    const double val = rnd<TB>(gconst[c] - 0.5 * q);
                    if (bm) st<TB>(bm, ((size_t)f * N + j) * M + m, val);
                    r = m == 0 ? logw[c] + val : logaddexp_np(r, logw[c] + val);
                }
                if (b) st<TB>(b, (size_t)f * N + j, r);
            }
            if (lane == 0) *flag = epoch;
        }
    }
}

// First NaN of every utterance's emissions b [U,T,N] in row-major order -- i.e. (first bad frame, first bad state in
// it) as ONE linear index.  grid (U, C): block c scans its share of the utterance's lens[u] * N values and leaves its
// minimum in slot[u * C + c]; no atomics across blocks, nothing to initialise.
template <typename TS>
__global__ __launch_bounds__(256) void nan_locate_kernel(const TS* __restrict__ b, const int32_t* __restrict__ lens,
                                                         int T, int N, unsigned long long* __restrict__ slot,
                                                         unsigned long long* __restrict__ slot_last,
                                                         unsigned* flag, unsigned epoch, int force) {
    if (!force && *flag != epoch) return;
    const int u = blockIdx.x, c = blockIdx.y, C = gridDim.y, tid = threadIdx.x;
    const long total = (long)lens[u] * N;
    const long per = (total + C - 1) / C;
    const long e0 = (long)c * per, e1 = e0 + per < total ? e0 + per : total;
    const TS* bu = b + (size_t)u * T * N;
    unsigned long long first = kNoBad;
    long last = -1;                                              // (the backward pass needs the LAST bad frame)
    for (long e = e0 + tid; e < e1; e += 256) {
        const TS x = bu[e];
        if (x != x) {                                            // ascending per thread: its first hit is its minimum
            if (first == kNoBad) first = (unsigned long long)e;
            if (!slot_last) break;
            last = e;
        }
    }
    __shared__ unsigned long long sm;
    __shared__ long long sl;
    if (tid == 0) {
        sm = kNoBad;
        sl = -1;
    }
    __syncthreads();
    if (first != kNoBad) atomicMin(&sm, first);
    if (last >= 0) atomicMax(&sl, (long long)last);
    __syncthreads();
    if (tid == 0) {
        slot[(size_t)u * C + c] = sm;
        if (slot_last) slot_last[(size_t)u * C + c] = sl < 0 ? kNoBad : (unsigned long long)sl;
        if (sm != kNoBad) *flag = epoch;
    }
}

// First and last NaN of one utterance's emissions, found by the block itself (all 256 threads call): the path of the
// repair kernels when a hot kernel raised the flag -- then no separate scan was launched, and on clean data the repair
// kernel is the only launch there is
template <typename TS>
__device__ __forceinline__ void block_nan_span(const TS* __restrict__ bu, long total, unsigned long long& first,
                                               long long& last) {
    __shared__ unsigned long long sf;
    __shared__ long long sl;
    const int tid = threadIdx.x;
    unsigned long long mf = kNoBad;
    long long ml = -1;
    for (long e = tid; e < total; e += 256) {
        const TS x = bu[e];
        if (x != x) {
            if (mf == kNoBad) mf = (unsigned long long)e;
            ml = e;
        }
    }
    if (tid == 0) {
        sf = kNoBad;
        sl = -1;
    }
    __syncthreads();
    if (mf != kNoBad) atomicMin(&sf, mf);
    if (ml >= 0) atomicMax(&sl, ml);
    __syncthreads();
    first = sf;
    last = sl;
}

// Forward pass (hmm.py:97-99): np.logaddexp carries a NaN emission of frame t1 into every state of every later frame:
// alpha[t1, j] = NaN for the NaN states, alpha[t > t1, :] = NaN, the likelihood NaN.  Backward (hmm.py:128-130):
// beta[t] folds b[:, t + 1], so with tl the LAST frame holding a NaN, beta[t < tl, :] = NaN and the rest stands.
template <typename TS>
__global__ __launch_bounds__(256) void alpha_nan_repair_kernel(const TS* __restrict__ b, TS* __restrict__ alpha,
                                                               double* __restrict__ loglik,
                                                               const int32_t* __restrict__ lens, int T, int N, int C,
                                                               const unsigned long long* __restrict__ slot,
                                                               const unsigned* flag, unsigned epoch) {
    if (*flag != epoch) return;
    const int u = blockIdx.x, tid = threadIdx.x;
    const int L = lens[u];
    unsigned long long first = kNoBad;
    if (slot) {
        for (int c = 0; c < C; ++c) {
            const unsigned long long s = slot[(size_t)u * C + c];
            first = s < first ? s : first;
        }
    } else {
        long long last;
        block_nan_span(b + (size_t)u * T * N, (long)L * N, first, last);
    }
    if (first == kNoBad) return;
    const int t1 = (int)(first / (unsigned long long)N);
    const TS nan = quiet_nan<TS>();
    if (alpha) {
        TS* au = alpha + (size_t)u * T * N;
        for (int j = tid; j < N; j += 256) {
            const TS x = b[((size_t)u * T + t1) * N + j];
            if (x != x) au[(size_t)t1 * N + j] = nan;
        }
        for (long e = (long)(t1 + 1) * N + tid; e < (long)L * N; e += 256) au[e] = nan;
    }
    if (loglik && tid == 0) loglik[u] = quiet_nan<double>();
}

template <typename TS>
__global__ __launch_bounds__(256) void beta_nan_repair_kernel(const TS* __restrict__ b, TS* __restrict__ beta,
                                                              const int32_t* __restrict__ lens, int T, int N, int C,
                                                              const unsigned long long* __restrict__ slot_last,
                                                              const unsigned* flag, unsigned epoch) {
    if (*flag != epoch) return;
    const int u = blockIdx.x, tid = threadIdx.x;
    long long last = -1;
    if (slot_last) {
        for (int c = 0; c < C; ++c) {
            const unsigned long long s = slot_last[(size_t)u * C + c];
            if (s != kNoBad && (long long)s > last) last = (long long)s;
        }
    } else {
        unsigned long long first;
        block_nan_span(b + (size_t)u * T * N, (long)lens[u] * N, first, last);
    }
    if (last < 0) return;
    const long tl = last / N;                                    // beta[t] reads b[:, t + 1]: frames below tl
    TS* bu = beta + (size_t)u * T * N;
    const TS nan = quiet_nan<TS>();
    for (long e = tid; e < tl * N; e += 256) bu[e] = nan;
}

// Same for observations [U,T,D] (calc_prob: only "does the utterance hold a bad frame" is needed)
__global__ __launch_bounds__(256) void nan_utt_obs_kernel(const double* __restrict__ obs, const int32_t* __restrict__ lens,
                                                          int T, int D, double* __restrict__ loglik,
                                                          const unsigned* flag, unsigned epoch) {
    if (*flag != epoch) return;
    const int u = blockIdx.x, tid = threadIdx.x;
    const long total = (long)lens[u] * D;
    const double* ou = obs + (size_t)u * T * D;
    bool bad = false;
    for (long e = tid; e < total; e += 256) bad |= !__builtin_isfinite(ou[e]);
    if (__syncthreads_or(bad) && tid == 0) loglik[u] = quiet_nan<double>();
}

// Baum-Welch statistics (gmmhmm.py:69-156 over a batch): an utterance with a bad frame has NaN in alpha from that frame
// on and in beta before it, so every posterior is NaN and so is every sum it enters -- the whole statistics block
__global__ __launch_bounds__(256) void nan_stats_kernel(const double* __restrict__ obs, const int32_t* __restrict__ lens,
                                                        int T, int D, double* __restrict__ stats, long n,
                                                        const unsigned* flag, unsigned epoch) {
    if (*flag != epoch) return;
    const int u = blockIdx.x, tid = threadIdx.x;
    const long total = (long)lens[u] * D;
    const double* ou = obs + (size_t)u * T * D;
    bool bad = false;
    for (long e = tid; e < total; e += 256) bad |= !__builtin_isfinite(ou[e]);
    if (!__syncthreads_or(bad)) return;
    for (long e = tid; e < n; e += 256) stats[e] = quiet_nan<double>();
}

// One block per utterance: rewrite v, path and the best score of an utterance whose emissions hold a NaN (header).
// Where the first NaN sits comes from the slots of nan_locate_kernel (emissions handed in by the caller), or -- obs
// given: decode, whose emissions are its own -- from the block's own scan for the first frame with a NaN or an infinite
// feature, every state of which is NaN then (j1 = 0); that scan only runs when the flag is raised (by the emission kernel,
// or by the flat scan that stands in for it where a kernel has no fused test).
template <typename TS>
__global__ __launch_bounds__(256) void viterbi_nan_repair_kernel(const TS* __restrict__ b, TS* __restrict__ v,
                                                                 const double* __restrict__ AT,
                                                                 const int32_t* __restrict__ lens, int T, int N, int C,
                                                                 const unsigned long long* __restrict__ slot,
                                                                 const double* __restrict__ obs, int D,
                                                                 const unsigned* flag, unsigned epoch,
                                                                 uint32_t* __restrict__ path, double* __restrict__ best) {
    if (*flag != epoch) return;
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int L = lens[u];
    int t1, j1;
    if (obs) {
        const double* ou = obs + (size_t)u * T * D;
        const long total = (long)L * D;
        unsigned long long mine = kNoBad;
        for (long e = tid; e < total; e += 256)
            if (!__builtin_isfinite(ou[e])) {
                mine = (unsigned long long)e;
                break;
            }
        __shared__ unsigned long long sm;
        if (tid == 0) sm = kNoBad;
        __syncthreads();
        if (mine != kNoBad) atomicMin(&sm, mine);
        __syncthreads();
        if (sm == kNoBad) return;                                // block-uniform
        t1 = (int)(sm / (unsigned long long)D);
        j1 = 0;
    } else {
        unsigned long long first = kNoBad;
        if (slot) {
            for (int c = 0; c < C; ++c) {
                const unsigned long long s = slot[(size_t)u * C + c];
                first = s < first ? s : first;
            }
        } else {
            long long last;
            block_nan_span(b + (size_t)u * T * N, (long)L * N, first, last);
        }
        if (first == kNoBad) return;                             // block-uniform
        t1 = (int)(first / (unsigned long long)N);
        j1 = (int)(first % (unsigned long long)N);
    }
    uint32_t* pu = path + (size_t)u * T;
    const TS nan = quiet_nan<TS>();
    {
        TS* vu = v + (size_t)u * T * N;
        for (int j = tid; j < N; j += 256) {
            bool hit = obs != nullptr;                           // a bad feature: every state of the frame
            if (!hit) {
                const TS x = b[((size_t)u * T + t1) * N + j];
                hit = x != x;
            }
            if (hit) vu[(size_t)t1 * N + j] = nan;
        }
        for (long e = (long)(t1 + 1) * N + tid; e < (long)L * N; e += 256) vu[e] = nan;
    }
    for (int t = t1 + 1 + tid; t < L; t += 256) pu[t] = 0u;
    if (tid == 0) {
        pu[t1] = (uint32_t)j1;
        if (best) best[u] = quiet_nan<double>();
    }
    // back-trace from (t1, j1) over the frames the NaN never reached: the very additions and the first-index rule of
    // the back-trace kernels (argmax_i(v[t-1, i] + A[i, p]), hmm.py:165,170)
    const TS* vu = v + (size_t)u * T * N;
    __shared__ double sx[4];
    __shared__ int si[4];
    int p = j1;
    for (int t = t1; t >= 1; --t) {
        const double* col = AT + (size_t)p * N;
        double x = LN_NEG_INF;
        int xi = kNoIndex;
        for (int i = tid; i < N; i += 256) argmax_combine(x, xi, ld(vu, (size_t)(t - 1) * N + i) + col[i], i);
        wave_argmax(x, xi);
        if (lane == 0) {
            sx[w] = x;
            si[w] = xi;
        }
        __syncthreads();
        x = sx[0];
        xi = si[0];
        for (int k = 1; k < 4; ++k) argmax_combine(x, xi, sx[k], si[k]);
        __syncthreads();
        p = xi;
        if (tid == 0) pu[t - 1] = (uint32_t)p;
    }
}

int locate_chunks(int U) {
    int C = (2048 + U - 1) / U;
    return C < 1 ? 1 : (C > kLocateMaxChunks ? kLocateMaxChunks : C);
}

}  // namespace

bool nan_rows_rescored(const Model& md) { return !md.diag || md.D == 1; }

int nan_rows_device(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm, bool force) {
    LN_DEVPTR(obs); LN_DEVPTR(b); LN_DEVPTR(bm);
    if (F <= 0) return 0;
    const Model& md = ctx->model;
    const long E = F * (long)md.D;
    long grid = (E + 255) / 256;
    if (grid > 2048) grid = 2048;
    unsigned* flag = ctx->nan_flag.as<unsigned>();
    if (nan_rows_rescored(md) && (b || bm)) {         // full covariance, or D = 1: re-score the bad frames
        const double* prec = md.diag ? nullptr : (const double*)md.prec.as<double>();
        const double* ivar1 = md.diag ? (const double*)md.ivar.as<double>() : nullptr;
        if (dtype == LNASR_F32)
            LN_LAUNCH(ctx, "nan_rows_full_f32", nan_rows_full_kernel<float>, dim3((unsigned)grid), dim3(256), 0, obs, F, md.D,
                      md.N, md.M, (const double*)md.mu.as<double>(), prec, ivar1,
                      (const double*)md.gconst.as<double>(), (const double*)md.logw.as<double>(), (float*)b, (float*)bm, flag,
                      ctx->nan_epoch, force ? 1 : 0);
        else
            LN_LAUNCH(ctx, "nan_rows_full", nan_rows_full_kernel<double>, dim3((unsigned)grid), dim3(256), 0, obs, F, md.D,
                      md.N, md.M, (const double*)md.mu.as<double>(), prec, ivar1,
                      (const double*)md.gconst.as<double>(), (const double*)md.logw.as<double>(), (double*)b, (double*)bm,
                      flag, ctx->nan_epoch, force ? 1 : 0);
        return 0;
    }
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "nan_rows_f32", nan_rows_kernel<float>, dim3((unsigned)grid), dim3(256), 0, obs, F, md.D, md.N, md.M,
                  (float*)b, (float*)bm, flag, ctx->nan_epoch, force ? 1 : 0);
    else
        LN_LAUNCH(ctx, "nan_rows", nan_rows_kernel<double>, dim3((unsigned)grid), dim3(256), 0, obs, F, md.D, md.N, md.M,
                  (double*)b, (double*)bm, flag, ctx->nan_epoch, force ? 1 : 0);
    return 0;
}

int viterbi_nan_repair(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* v,
                       uint32_t* path, double* best, bool force) {
    LN_DEVPTR(b); LN_DEVPTR(d_lens); LN_DEVPTR(v); LN_DEVPTR(path); LN_DEVPTR(best);
    if (U <= 0 || T <= 0) return 0;
    const int N = ctx->model.N;
    const int C = locate_chunks(U);
    LN_REQUIRE(v, "the repair pass needs the Viterbi lattice");
    unsigned* flag = ctx->nan_flag.as<unsigned>() + 1;      // the "suspect cell in v" word (backtrace_lds_kernel)
    const double* AT = ctx->model.logAT.as<double>();
    const double* no_obs = nullptr;
    // force: the scan is launched (grid (U, C), memory speed) and raises the word itself; otherwise the back-trace kernel
    // has already looked, and the repair kernel -- the only launch on clean data -- finds its utterance's NaN by itself
    unsigned long long* slot = nullptr;
    if (force) {
        LN_TRY(ctx->nan_slots.reserve((size_t)U * C * 8));
        slot = ctx->nan_slots.as<unsigned long long>();
    }
    if (dtype == LNASR_F32) {
        if (force)
            LN_LAUNCH(ctx, "nan_locate_f32", nan_locate_kernel<float>, dim3(U, C), dim3(256), 0, (const float*)b, d_lens, T,
                      N, slot, (unsigned long long*)nullptr, flag, ctx->nan_epoch, 1);
        LN_LAUNCH(ctx, "viterbi_nan_repair_f32", viterbi_nan_repair_kernel<float>, dim3(U), dim3(256), 0, (const float*)b,
                  (float*)v, AT, d_lens, T, N, C, (const unsigned long long*)slot, no_obs, 0, flag, ctx->nan_epoch, path, best);
    } else {
        if (force)
            LN_LAUNCH(ctx, "nan_locate", nan_locate_kernel<double>, dim3(U, C), dim3(256), 0, (const double*)b, d_lens, T, N,
                      slot, (unsigned long long*)nullptr, flag, ctx->nan_epoch, 1);
        LN_LAUNCH(ctx, "viterbi_nan_repair", viterbi_nan_repair_kernel<double>, dim3(U), dim3(256), 0, (const double*)b,
                  (double*)v, AT, d_lens, T, N, C, (const unsigned long long*)slot, no_obs, 0, flag, ctx->nan_epoch, path, best);
    }
    return 0;
}

// caller-supplied emissions through the forward / backward entry points.  Word 2 of ctx->nan_flag: "a forward / backward
// kernel met a NaN" (fwdbwd_cols2 tests the value it finalises; the other kernels leave the scan to `force`)
int trellis_nan_repair(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* alpha,
                       double* loglik, void* beta, bool force) {
    LN_DEVPTR(b); LN_DEVPTR(d_lens);
    if (U <= 0 || T <= 0) return 0;
    const int N = ctx->model.N;
    const int C = locate_chunks(U);
    unsigned* flag = ctx->nan_flag.as<unsigned>() + 2;
    unsigned long long *slot = nullptr, *slot_last = nullptr;      // see viterbi_nan_repair
    if (force) {
        LN_TRY(ctx->nan_slots.reserve((size_t)U * C * 8 * 2));
        slot = ctx->nan_slots.as<unsigned long long>();
        slot_last = slot + (size_t)U * C;
    }
    if (dtype == LNASR_F32) {
        if (force)
            LN_LAUNCH(ctx, "nan_locate_f32", nan_locate_kernel<float>, dim3(U, C), dim3(256), 0, (const float*)b, d_lens, T,
                      N, slot, beta ? slot_last : nullptr, flag, ctx->nan_epoch, 1);
        if (alpha || loglik)
            LN_LAUNCH(ctx, "alpha_nan_repair_f32", alpha_nan_repair_kernel<float>, dim3(U), dim3(256), 0, (const float*)b,
                      (float*)alpha, loglik, d_lens, T, N, C, (const unsigned long long*)slot, flag, ctx->nan_epoch);
        if (beta)
            LN_LAUNCH(ctx, "beta_nan_repair_f32", beta_nan_repair_kernel<float>, dim3(U), dim3(256), 0, (const float*)b,
                      (float*)beta, d_lens, T, N, C, (const unsigned long long*)slot_last, flag, ctx->nan_epoch);
    } else {
        if (force)
            LN_LAUNCH(ctx, "nan_locate", nan_locate_kernel<double>, dim3(U, C), dim3(256), 0, (const double*)b, d_lens, T, N,
                      slot, beta ? slot_last : nullptr, flag, ctx->nan_epoch, 1);
        if (alpha || loglik)
            LN_LAUNCH(ctx, "alpha_nan_repair", alpha_nan_repair_kernel<double>, dim3(U), dim3(256), 0, (const double*)b,
                      (double*)alpha, loglik, d_lens, T, N, C, (const unsigned long long*)slot, flag, ctx->nan_epoch);
        if (beta)
            LN_LAUNCH(ctx, "beta_nan_repair", beta_nan_repair_kernel<double>, dim3(U), dim3(256), 0, (const double*)b,
                      (double*)beta, d_lens, T, N, C, (const unsigned long long*)slot_last, flag, ctx->nan_epoch);
    }
    return 0;
}

// force for the kernels that walk ONE utterance per block: a flat scan of the observations at memory speed raises the
// flag first (it does not know the utterances' lengths: padding may raise it for nothing), the slow walk follows only then
static int raise_flag_if_bad(lnasr_ctx* ctx, const double* obs, int U, int T) {
    return nan_rows_device(ctx, obs, (long)U * T, LNASR_F64, nullptr, nullptr, true);
}

int viterbi_nan_repair_obs(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* d_lens, double* v,
                           uint32_t* path, double* best, bool force) {
    LN_DEVPTR(obs); LN_DEVPTR(d_lens); LN_DEVPTR(v); LN_DEVPTR(path); LN_DEVPTR(best);
    if (U <= 0 || T <= 0) return 0;
    LN_REQUIRE(obs && v, "the repair pass needs the observations and the Viterbi lattice");
    if (force) LN_TRY(raise_flag_if_bad(ctx, obs, U, T));
    const unsigned long long* no_slot = nullptr;
    LN_LAUNCH(ctx, "viterbi_nan_repair", viterbi_nan_repair_kernel<double>, dim3(U), dim3(256), 0, (const double*)nullptr,
              v, (const double*)ctx->model.logAT.as<double>(), d_lens, T, ctx->model.N, 0, no_slot, obs, ctx->model.D,
              (const unsigned*)ctx->nan_flag.as<unsigned>(), ctx->nan_epoch, path, best);
    return 0;
}

int stats_nan_repair(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* d_lens, double* stats, long n,
                     bool force) {
    LN_DEVPTR(obs); LN_DEVPTR(d_lens); LN_DEVPTR(stats);
    if (U <= 0 || T <= 0 || n <= 0) return 0;
    if (force) LN_TRY(raise_flag_if_bad(ctx, obs, U, T));
    LN_LAUNCH(ctx, "stats_nan_repair", nan_stats_kernel, dim3(U), dim3(256), 0, obs, d_lens, T, ctx->model.D, stats, n,
              (const unsigned*)ctx->nan_flag.as<unsigned>(), ctx->nan_epoch);
    return 0;
}

int loglik_nan_repair(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* d_lens, double* loglik, bool force) {
    LN_DEVPTR(obs); LN_DEVPTR(d_lens); LN_DEVPTR(loglik);
    if (U <= 0 || T <= 0) return 0;
    if (force) LN_TRY(raise_flag_if_bad(ctx, obs, U, T));
    LN_LAUNCH(ctx, "loglik_nan_repair", nan_utt_obs_kernel, dim3(U), dim3(256), 0, obs, d_lens, T, ctx->model.D, loglik,
              (const unsigned*)ctx->nan_flag.as<unsigned>(), ctx->nan_epoch);
    return 0;
}

}  // namespace lnasr
