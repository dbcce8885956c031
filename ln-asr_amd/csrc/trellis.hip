// HMM trellis entry points of the C-ABI (hmm.py:47-200, 268-283): discrete emission, Baum-Welch posteriors, and the
// host side of lnasr_viterbi / lnasr_decode / lnasr_forward / lnasr_backward / lnasr_calc_prob / lnasr_baumwelch.
// The kernels live by family in viterbi.hip, backtrace.hip, fwdbwd.hip, trellis256.hip and bign.hip; trellis.h holds
// their launchers and the dispatch thresholds.
#include <utility>

#include "common.h"
#include "devmath.h"
#include "trellis.h"

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
    const void* before = ctx->ws_lens.p;
    LN_TRY(ctx->ws_lens.reserve((size_t)(U > 0 ? U : 1) * 4));
    if (ctx->ws_lens.p != before) ctx->lens_fill_T = -1;
    int32_t* d = ctx->ws_lens.as<int32_t>();
    if (lens) {
        if (is_device_ptr(lens)) {
            *out = lens;
            return 0;
        }
        for (int u = 0; u < U; ++u)
            LN_REQUIRE(lens[u] >= 1 && lens[u] <= T, "lens[%d]=%d out of range [1,%d]", u, lens[u], T);
        ctx->lens_fill_T = -1;
        LN_HIP(hipMemcpyAsync(d, lens, (size_t)U * 4, hipMemcpyHostToDevice, ctx->stream));
        LN_HIP(hipStreamSynchronize(ctx->stream));
    } else if (!(ctx->lens_fill_T == T && ctx->lens_fill_U >= U)) {
        // "every utterance is T frames long": written once per (U, T), not once per call (a step of the bench, a decode() of the
        // mirror classes: the same shape over and over)
        LN_LAUNCH(ctx, "fill_lens", fill_lens_kernel, dim3((U + 255) / 256), dim3(256), 0, d, U, T);
        ctx->lens_fill_T = T;
        ctx->lens_fill_U = U;
    }
    *out = d;
    return 0;
}

int baumwelch_device(lnasr_ctx* ctx, const void* b, const void* alpha, const void* beta, int U, int T,
                     const int32_t* d_lens, int dtype, void* xi, void* gamma) {
    LN_DEVPTR(b); LN_DEVPTR(alpha); LN_DEVPTR(beta); LN_DEVPTR(d_lens); LN_DEVPTR(xi); LN_DEVPTR(gamma);
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
    // emissions handed in by the caller: a NaN among them propagates as numpy's max / argmax do (nanfix.hip).  The
    // register-resident Viterbi kernels test the emissions they load (b_tested) and raise the flag themselves -- a -inf cell of v is
    // then just a forbidden state (left-to-right models) and no suspect; otherwise the back-trace kernel flags NaN and -inf cells of
    // v, and kernels that did not look at all cost a scan of b here
    bool b_tested = false, v_tested = false;
    LN_TRY(viterbi_device(ctx, d_b, U, T, d_lens, dtype, d_v, &b_tested));
    LN_TRY(backtrace_device(ctx, d_v, U, T, d_lens, dtype, (uint32_t*)d_path, (double*)d_best, &v_tested, !b_tested));
    LN_TRY(viterbi_nan_repair(ctx, d_b, U, T, d_lens, dtype, d_v, (uint32_t*)d_path, (double*)d_best, !(b_tested || v_tested)));
    if (bt_or_null) {
        LN_TRY(stage_out(ctx, bt_or_null, cells * 4, ctx->ws_out3, &d_bt));
        LN_TRY(bt_matrix_device(ctx, d_v, U, T, d_lens, dtype, (uint32_t*)d_bt));
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
    bool scanned = false, v_tested = false, b_tested = false;
    // (full covariance: the emission pass re-scores frames with a non-finite feature itself -- an infinity gives -inf or
    // NaN per component there -- and the trellis is repaired from the emissions, as for lnasr_viterbi)
    const bool by_obs = !nan_rows_rescored(md);
    LN_TRY(emit_batch(ctx, obs, U, T, ctx->ws_b.as<double>(), nullptr, &d_x, by_obs ? &scanned : nullptr));
    // (diagonal models: the emission kernel tested the features, the Viterbi kernel runs without its own test of b)
    LN_TRY(viterbi_device(ctx, ctx->ws_b.p, U, T, d_lens, LNASR_F64, ctx->ws_v.p, by_obs ? nullptr : &b_tested));
    LN_TRY(backtrace_device(ctx, ctx->ws_v.p, U, T, d_lens, LNASR_F64, (uint32_t*)d_path, (double*)d_best, &v_tested,
                            !by_obs && !b_tested));
    // a NaN or an infinite feature makes its frame's emissions NaN in the reference, and np.max carries that through every
    // later frame: one launch that returns at once unless the emission kernel raised the flag (nanfix.hip)
    if (by_obs)
        LN_TRY(viterbi_nan_repair_obs(ctx, d_x, U, T, d_lens, ctx->ws_v.as<double>(), (uint32_t*)d_path, (double*)d_best,
                                      !scanned));
    else
        LN_TRY(viterbi_nan_repair(ctx, ctx->ws_b.p, U, T, d_lens, LNASR_F64, ctx->ws_v.p, (uint32_t*)d_path, (double*)d_best,
                                  !(b_tested || v_tested)));
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
