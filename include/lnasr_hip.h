/*
 * lnasr_hip.h -- C-ABI of the MI355X (gfx950) GMM-HMM acoustic-scoring / decoding path.
 *
 * Drop-in boundary for yehuohan/ln-asr's numpy hot path.  The reference has no
 * FFI of its own on this path: its boundary is the Python class /
 * function surface.  Each entry point below therefore cites the reference
 * function (file:line under the reference checkout) whose arithmetic it
 * replaces; `ln-asr_amd/lnasr/_ffi.py` is the only caller (ctypes) and
 * `INTEGRATION.md` shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = error (LNASR_E_*); the
 *     message is available from lnasr_last_error() (thread-local).  No C++
 *     exception crosses this boundary.
 *   - one context = one process = one GPU (one HIP device, one compute stream).
 *     Calls on one context are not re-entrant.
 *   - data pointers are plain pointers to contiguous row-major arrays and may be
 *     HOST or DEVICE memory (device memory from lnasr_dev_alloc); the library
 *     detects which (hipPointerGetAttributes).  Host buffers are staged through
 *     the context's workspace; device buffers are used in place.  Calls are
 *     synchronous with respect to host buffers; with device buffers they only
 *     enqueue on the context's stream (use lnasr_sync()).
 *   - all probabilities are natural-log (hmm.py:27-33).  A[i*N+j] is i -> j.
 *   - BATCH-FIRST, FRAME-MAJOR layouts (U utterances, T frames, N states,
 *     M mixtures, D dims):   obs [U,T,D]   b [U,T,N]   bm [U,T,N,M]
 *     alpha/beta/v [U,T,N]   bt [U,T,N] u32   path [U,T] u32   xi [U,T,N,N]
 *     The reference's single-utterance arrays are views of these with U = 1
 *     (its `b` [N,T] is b[0].T, its `bm` [N,M,T] is bm[0].transpose(1,2,0)).
 *   - `lens` (optional, may be NULL): int32 [U] true length of each utterance
 *     (1 <= lens[u] <= T); rows t >= lens[u] of the outputs are left untouched.
 *   - `dtype`: storage type of b/bm/alpha/beta/v/xi/gamma buffers, LNASR_F64 or
 *     LNASR_F32.  As in the reference (`precision=`, hmm.py:38) arithmetic is
 *     fp64 either way (A, pi, w are fp64 there); F32 rounds at every store and
 *     the recursions continue from the rounded value, exactly as the reference
 *     reads its float32 arrays back: v / bt / path (hmm.py:160-171), alpha
 *     (hmm.py:93-99) and beta (hmm.py:124-130) equal the reference's float32
 *     arrays bit for bit (a value whose fp64 sum lies within ~1e-15 of a float
 *     rounding boundary may differ by one ulp).  The log-likelihood of
 *     lnasr_forward is the fp64 log-sum-exp of the rounded alpha[T-1] (the
 *     reference folds that one in float32: agreement to ~1e-6 relative).
 *     obs, model parameters, statistics and MFCC outputs are always fp64.
 */
#ifndef LNASR_HIP_H
#define LNASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LNASR_ABI_VERSION 1

#define LNASR_F64 0
#define LNASR_F32 1

#define LNASR_OK 0
#define LNASR_E_INVALID (-1)   /* bad argument / shape */
#define LNASR_E_HIP (-2)       /* a HIP runtime call failed */
#define LNASR_E_NOMODEL (-3)   /* no model uploaded for this call */
#define LNASR_E_NODEVICE (-4)  /* no usable gfx950 device */
#define LNASR_E_COMM (-5)      /* RCCL failure */
#define LNASR_E_UNSUPPORTED (-6)

typedef struct lnasr_ctx lnasr_ctx;

/* ---- runtime ---------------------------------------------------------------- */
int lnasr_abi_version(void);
const char* lnasr_last_error(void);
int lnasr_device_count(int* out);
int lnasr_ctx_create(int device_id, lnasr_ctx** out);
void lnasr_ctx_destroy(lnasr_ctx* ctx);
int lnasr_sync(lnasr_ctx* ctx);
/* every stream of the context's device has drained (hipDeviceSynchronize): the bracket of a timed region */
int lnasr_device_sync(lnasr_ctx* ctx);
/* LNASR_* test / ablation switches are read once per process; this re-reads them from the environment.  Test-only:
 * not to be called while another thread is inside the library. */
int lnasr_reload_switches(void);
int lnasr_device_info(lnasr_ctx* ctx, char* name, int name_cap, int* cu_count, size_t* hbm_bytes);

/* device memory for HBM-resident inputs/outputs */
int lnasr_dev_alloc(lnasr_ctx* ctx, size_t bytes, void** out);
int lnasr_dev_free(lnasr_ctx* ctx, void* p);
int lnasr_memcpy_h2d(lnasr_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int lnasr_memcpy_d2h(lnasr_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int lnasr_memset(lnasr_ctx* ctx, void* dst_dev, int byte, size_t bytes);
/* Page-locked host memory (hipHostMalloc).  The reference hands host arrays to decode() (hmm.py:276-283); a host
 * buffer from here is recognised by every entry point and moves at PCIe rate with no staging copy, a pageable one
 * (plain numpy) goes through the context's pinned staging ring.
 * A page-locked INPUT is read by asynchronous copies: when every output of a call is a device pointer the call
 * returns with those copies possibly still in flight, so the buffer must stay unchanged until lnasr_sync() (a
 * pageable input has been consumed when the call returns; a call with a host output always has).
 * lnasr_host_free: ctx may be NULL for a block that outlives its context (device-wide wait, then freed). */
int lnasr_host_alloc(lnasr_ctx* ctx, size_t bytes, void** out);
int lnasr_host_free(lnasr_ctx* ctx, void* p);

/* The check the kernel launchers apply to every pointer they are handed when LNASR_CHECK_PTRS=1 (always in the
 * host-sanitizer build): LNASR_E_INVALID unless p is null or device memory.  Exposed so that the CPU suite can prove
 * the guard fires on a heap pointer (tests/test_host_asan.py). */
int lnasr_debug_require_device_ptr(const void* p);
/* Diagnostic: the number of columns (state x frame) the scaled linear-domain forward / backward kernels (every family
 * above 16 states) found outside the exponent range and redid in the exact log-domain form (hmm.py:97-99's logaddexp fold,
 * a state at a time) since the counter was last read; reading clears it.  The results do not depend on it, the time does:
 * the parity suite uses it to pin that emissions whose size jumps from frame to frame -- data a model does not fit -- stay
 * on the fast path (tests/test_gpu_trellis.py). */
int lnasr_debug_fallback_columns(lnasr_ctx* ctx, unsigned* count);
/* Diagnostic: the shader clock (GHz) directly behind whatever the context's stream ran last -- a one-wave kernel counts
 * shader-clock cycles over 20 us of the constant 100 MHz counter.  bench.py reports it behind the integer emission kernel,
 * behind an fp64 kernel and after idling: the integer matrix kernels are power-limited and leave the chip at a lower
 * clock.  Nothing in the reference corresponds to it. */
int lnasr_debug_clock(lnasr_ctx* ctx, double* ghz);

/* 128-bit fingerprint of a HOST buffer (no context, no device).  The reference's methods re-read self.A / self.mu /
 * self.si on every call (hmm.py:82-99, gmmhmm.py:58-67): the mirror classes fingerprint the parameters per call and
 * upload only what changed.  Buffers of 2 MB and more are hashed by a small pool of host threads (25 MB of `si`:
 * ~0.2 ms instead of ~1.5 ms on one thread); the result depends on the bytes, their count and the seed only. */
int lnasr_hash128(const void* data, size_t bytes, uint64_t seed, uint64_t* out2);
/* The same fingerprint of n buffers (buffer k with seed + k) in the background: _begin returns at once, _end waits and
 * writes two words per buffer (out may be NULL to discard).  Lets a caller enqueue work on the model the device already
 * holds while the check that it is still current runs beside it.  The buffers must stay alive and unchanged until _end;
 * every job begun must be ended. */
int lnasr_hash128_begin(const void* const* ptrs, const size_t* bytes, int n, uint64_t seed, void** job_out);
int lnasr_hash128_end(void* job, uint64_t* out);

/* per-kernel HIP-event timing on the context's own stream (bench.py roofline) */
int lnasr_prof_enable(lnasr_ctx* ctx, int on);
int lnasr_prof_reset(lnasr_ctx* ctx);
int lnasr_prof_count(lnasr_ctx* ctx, int* n_kernels);
int lnasr_prof_get(lnasr_ctx* ctx, int i, char* name, int name_cap, int64_t* launches, double* total_ms);

/* ---- models ----------------------------------------------------------------- */
/* GMM-HMM parameters: GMMHMM.__init__ (gmmhmm.py:34-44).  sigma_kind 0: full
 * covariance [N,M,D,D] exactly as the reference stores `si` (inverted + log-det
 * on the device with LU semantics, gmm.py:62-63; if every off-diagonal entry is
 * exactly zero the diagonal kernels are used); sigma_kind 1: diagonal [N,M,D]. */
int lnasr_model_upload(lnasr_ctx* ctx, int N, int M, int D,
                       const double* logA, const double* logpi, const double* logw,
                       const double* mu, const double* sigma, int sigma_kind);
/* same, but never take the diagonal fast path (keeps the reference's full-Sigma arithmetic) */
int lnasr_model_upload_full(lnasr_ctx* ctx, int N, int M, int D,
                            const double* logA, const double* logpi, const double* logw,
                            const double* mu, const double* sigma);
/* transition-only model for the trellis entry points (HMM.__init__, hmm.py:36-45) */
int lnasr_hmm_upload(lnasr_ctx* ctx, int N, const double* logA, const double* logpi);
int lnasr_model_is_diag(lnasr_ctx* ctx, int* out);
/* Conditioning of the uploaded mixture model for the matrix-core emission kernels, which evaluate the expanded
 * quadratic on observations centred on the mean of the component means: kappa = max over components of
 * (mu - centre)^T Sigma^-1 (mu - centre), the size of the terms that expansion cancels (absolute error of a
 * log-density ~ 3e-16 * kappa; the reference's direct (x - mu) form, gmm.py:59-68, has ~1e-14).  direct_form = 1:
 * kappa exceeds 4096 (or is not finite) and the model is scored by the direct-form kernels instead. */
int lnasr_model_conditioning(lnasr_ctx* ctx, double* kappa, int* direct_form);
/* Range of the uploaded diagonal model as the integer matrix-core bank's fixed point sees it (csrc/gmm_i8.hip): that bank
 * keeps 56 bits below each component's largest scaled coefficient and below each frame's largest scaled feature, so its
 * absolute error on a component's log-density is ~3 * 2^-53 * A_c * X_f.  rho = max over components of A_c * X_c, X_c taken
 * at frames within four standard deviations of the component's own mean: a component whose own variances (relative to the
 * model's scale in each dimension) span decades has a large rho, one sharp component among broad ones does not.
 * integer_bank = 1: the model's shape takes that bank and rho <= 2048 (error <= 1e-12); 0: large batches are scored by the
 * fp64 matrix-core bank as small ones are (shape not taken: rho = 0).  A full-covariance model reports the same two values
 * for its own integer bank (csrc/gmm_full_i8.hip: K = D (D + 1) / 2 + D features, gate 4096, measured in
 * tests/test_gpu_full_i8.py).  Replaces nothing in the reference, whose direct form
 * (gmm.py:59-68) has no such limit -- this is how the limit of the fast path is made visible. */
int lnasr_model_range(lnasr_ctx* ctx, double* rho, int* integer_bank);

/* ---- emission --------------------------------------------------------------- */
/* HMM._map_b (hmm.py:47-61): b[u,t,:] = B[:, obs[u,t]];  B is [N,K] */
int lnasr_discrete_map_b(lnasr_ctx* ctx, const double* logB, int N, int K,
                         const int32_t* obs, int U, int T, int dtype, void* b);
/* GMMHMM._map_b (gmmhmm.py:46-67): b and (optionally) the per-mixture bm.  A frame with a NaN or an infinite feature
 * gets what the reference's product with the full inverse matrix gives (gmm.py:63-68): NaN rows (inf * 0 for a diagonal
 * Sigma), or -inf / NaN component by component where an infinity meets no zero (a genuinely full Sigma, D = 1).
 * Which kernel scores a call depends on the shape and on the model: 4096 frames or more of a diagonal bank of 48 states or
 * more with 14 <= D <= 39 and at least 5 mixtures take the integer matrix cores (a 56-bit fixed-point product, csrc/gmm_i8.hip)
 * when the model's range allows it (lnasr_model_range: rho <= 2048), the fp64 matrix cores otherwise; models whose means lie
 * many standard deviations apart (lnasr_model_conditioning) take the direct form.  The gates are set so that every route
 * returns the reference's b to 1e-12 * max(1, |b|) at the frames the model describes (measured: tests/test_gpu_i8.py; frames
 * far outside the model lose accuracy in proportion to their distance on every expanded-form route), and two routes agree
 * to ~1e-13 -- the b of a small batch and of a large one that contains it are therefore not guaranteed to be the same bits. */
int lnasr_gmm_logprob(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens,
                      int dtype, void* b, void* bm_or_null);
/* free functions of gmm.py (one Gaussian / one mixture, no HMM):
 * gaussian_distribution{,_log} (gmm.py:16-30) */
int lnasr_gauss1d(lnasr_ctx* ctx, const double* x, size_t L, double mu, double sigma2, int want_log, double* out);
/* gaussian_multi_distribution{,_log} (gmm.py:32-68); Sigma [D,D], LU semantics */
int lnasr_gauss_multi(lnasr_ctx* ctx, const double* x, size_t L, int D, const double* mu, const double* Sigma,
                      int want_log, double* out);
/* gaussian_mixture_distribution{,_log} (gmm.py:70-100); c is log-weights when want_log */
int lnasr_gauss_mixture(lnasr_ctx* ctx, const double* c, int M, const double* x, size_t L, int D,
                        const double* mu, const double* Sigma, int want_log, double* out);

/* ---- trellises (use the uploaded logA / logpi) ---------------------------------- */
/* HMM._viterbi (hmm.py:133-172): v, bt optional; path required.  best_score [U] optional.
 * NaN emissions propagate as numpy's max / argmax propagate them (hmm.py:164-171): from the first frame t1 that holds one,
 * v[t > t1] is NaN, path[t > t1] = 0, path[t1] = the first NaN state, path[t < t1] the back-trace from there. */
int lnasr_viterbi(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* lens, int dtype,
                  void* v_or_null, uint32_t* bt_or_null, uint32_t* path, double* best_score_or_null);
/* GMMHMM.decode (hmm.py:276-283 -> gmmhmm.py:46-67 + hmm.py:133-172): emission, trellis and
 * backtrace as three launches on the context's stream; b and v stay in HBM workspaces of the
 * context (they are not on chip: 1 KB per frame each way, far from the binding resource) and
 * only the paths come back.  NaN / infinite features: the reference's paths (see lnasr_viterbi), best_score NaN */
int lnasr_decode(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens,
                 uint32_t* path, double* best_score_or_null);
/* HMM._forward (hmm.py:63-100) and calc_prob (hmm.py:268-274): alpha optional, loglik [U] optional.  A NaN emission
 * propagates as np.logaddexp propagates it: alpha NaN from that frame on (forward), beta NaN before it (backward).
 * Kernels are chosen by model AND batch size (csrc/trellis.h): Viterbi v / bt / path are the same bits whatever the batch
 * size; fp64 alpha / beta agree between batch sizes to ~1e-13 relative (a handful of utterances take kernels that form the
 * next frame's linear-domain row by one multiplication instead of a log / exp round trip), and to the reference's within
 * 1e-11 either way. */
int lnasr_forward(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* lens, int dtype,
                  void* alpha_or_null, double* loglik_or_null);
/* HMM._backward (hmm.py:102-131) */
int lnasr_backward(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* lens, int dtype, void* beta);
/* GMMHMM.calc_prob: emission + forward, loglik [U] (NaN for an utterance with a NaN or an infinite feature) */
int lnasr_calc_prob(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, double* loglik);
/* the same under the model's `precision=`: LNASR_F32 scores the emission exactly as lnasr_bw_accumulate_p(…, F32)
 * does (fp32 matrix cores where the model allows it), so the two likelihoods of the reference's stopping rule
 * (hmm.py:301-313) come from one arithmetic */
int lnasr_calc_prob_p(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, int dtype, double* loglik);

/* ---- Baum-Welch --------------------------------------------------------------- */
/* HMM._baumwelch (hmm.py:174-200): xi [U,T,N,N], gamma [U,T,N]; frame lens-1 is -inf */
int lnasr_baumwelch(lnasr_ctx* ctx, const void* b, const void* alpha, const void* beta,
                    int U, int T, const int32_t* lens, int dtype, void* xi, void* gamma);
/* HMM._estimate (hmm.py:202-225), single utterance: new logA [N,N], logB [N,K], logpi [N] */
int lnasr_estimate_discrete(lnasr_ctx* ctx, const int32_t* obs, int T, int N, int K, int dtype,
                            const void* xi, const void* gamma,
                            void* logA_out, void* logB_out, void* logpi_out);
/* GMMHMM._estimate (gmmhmm.py:69-156), single utterance, using the uploaded (old) w / mu:
 * outputs logA [N,N], logpi [N], logw [N,M] in `dtype`; mu [N,M,D], si [N,M,D,D] fp64 */
int lnasr_estimate_gmm(lnasr_ctx* ctx, const double* obs, int T, int dtype,
                       const void* bm, const void* alpha, const void* beta,
                       const void* xi, const void* gamma, double min_std,
                       void* logA_out, void* logpi_out, void* logw_out, double* mu_out, double* si_out);

/* Multi-utterance sufficient statistics (new): emission + forward + backward +
 * posteriors accumulated on the device in the linear domain.  Packed fp64 layout of `stats`
 * (length lnasr_bw_stats_len): xi_sum[N*N] gamma_sum[N] gamma0[N] mix[N*M] mix_x[N*M*D]
 * mix_xx[N*M*D] (diagonal model) or [N*M*D*D] (full), loglik_sum[1], n_utt[1], n_frames[1]. */
int lnasr_bw_stats_len(lnasr_ctx* ctx, size_t* n_doubles);
int lnasr_bw_accumulate(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, double* stats);
/* The same pass under the reference's `precision=` (hmm.py:38, gmmhmm.py:34-44).  LNASR_F32: component log-densities
 * and posterior-weighted moments run on the fp32 matrix cores and are handed over as float (diagonal, well
 * conditioned models with N % 4 == 0, M in {4, 8, 16, 32}, D <= 39; any other model takes the fp64 pass -- more
 * precise than asked for, never less); the recursions, xi and the packed statistics stay fp64.  Results agree with
 * the fp64 pass to float32 accuracy (~1e-5 relative). */
int lnasr_bw_accumulate_p(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* lens, int dtype, double* stats);
/* M-step from (all-reduced) statistics: new logA, logpi, logw, mu, sigma (diag [N,M,D] or full) */
int lnasr_bw_mstep(lnasr_ctx* ctx, const double* stats, double min_std,
                   double* logA, double* logpi, double* logw, double* mu, double* sigma);

/* ---- numeric utils ------------------------------------------------------------- */
/* lse / lse2 (utils.py:140-204) along one axis of a 3-D view [outer, n, inner] -> [outer, inner] */
int lnasr_lse(lnasr_ctx* ctx, const double* x, size_t outer, size_t n, size_t inner, int base2, double* out);

/* ---- MFCC ------------------------------------------------------------------------ */
typedef struct lnasr_mfcc_cfg {
    int fs;            /* MFCC.__init__ (mfcc.py:23-48) */
    int frame_len;     /* int(fs*frame_T)      = 400 */
    int frame_step;    /* int(fs*frame_stride) = 160 */
    int fft_n;         /* 512 (power of two, <= 4096) */
    int n_mel;         /* 40 */
    int n_cep;         /* 12 */
    double hpf_alpha;  /* 0.97 */
} lnasr_mfcc_cfg;
/* split_frames frame count (utils.py:119) */
int lnasr_num_frames(int64_t n_samples, int frame_len, int frame_step, int64_t* out);
/* MFCC.calc_high_pass_filter (mfcc.py:50-64): y[0]=x[0], y[t]=x[t]-alpha*x[t-1] */
int lnasr_preemphasis(lnasr_ctx* ctx, const double* x, int64_t n, double alpha, double* y);
/* split_frames (utils.py:106-130): frames [Nf, L] fp64 */
int lnasr_split_frames(lnasr_ctx* ctx, const double* signal, int64_t n, int L, int S, double* frames);
/* MFCC.calc_mfcc (mfcc.py:108-175) over a batch of utterances concatenated in `signal`;
 * offsets [U+1] (samples).  sample_kind 0: fp64 samples, 1: int16 PCM (read_pcm, utils.py:100-104).
 * window [frame_len] and fbank [n_mel, fft_n/2+1] are the model constants (create_hamming
 * utils.py:132-138, create_filter_bank mfcc.py:78-106).  Outputs are concatenated over
 * utterances: power [sumNf, fft_n/2+1] (optional), cepstrum [sumNf, n_mel] (optional),
 * features [sumNf, 3*(n_cep+1)]. */
int lnasr_mfcc(lnasr_ctx* ctx, const void* signal, int sample_kind, const int64_t* offsets, int U,
               const lnasr_mfcc_cfg* cfg, const double* window, const double* fbank,
               double* power_or_null, double* cepstrum_or_null, double* features);

/* Ragged -> rectangular: rows row_offsets[u] .. row_offsets[u+1] of packed [sumNf, W] (e.g. the features of
 * lnasr_mfcc) become rows 0 .. of padded[u] in padded [U, pad_rows, W]; the remaining rows are zero.  This is
 * the layout lnasr_decode / lnasr_calc_prob / lnasr_bw_accumulate take together with lens[u] = row count
 * (variable-length utterances batched by an offsets array).  row_offsets is a host array. */
int lnasr_pad_rows(lnasr_ctx* ctx, const double* packed, const int64_t* row_offsets, int U, int W, int64_t pad_rows,
                   double* padded);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) --------------------------------- */
#define LNASR_UNIQUE_ID_BYTES 128
int lnasr_comm_unique_id(void* id_out /* LNASR_UNIQUE_ID_BYTES */);
int lnasr_comm_init(lnasr_ctx* ctx, const void* id, int rank, int nranks);
int lnasr_comm_destroy(lnasr_ctx* ctx);
/* rank and size as RCCL reports them for the live communicator (ncclCommUserRank / ncclCommCount) */
int lnasr_comm_info(lnasr_ctx* ctx, int* rank, int* nranks);
/* gather of decoded state paths: every rank contributes count u32, receives nranks*count */
int lnasr_allgather_u32(lnasr_ctx* ctx, const uint32_t* local, size_t count, uint32_t* all);
/* same gather for state paths of a model with n_states states: indices below 256 travel as bytes (a quarter
 * of the xGMI traffic), the result is the u32 array lnasr_allgather_u32 would give */
int lnasr_allgather_paths(lnasr_ctx* ctx, const uint32_t* local, size_t count, int n_states, uint32_t* all);
/* sum all-reduce of packed Baum-Welch statistics, in place */
int lnasr_allreduce_f64(lnasr_ctx* ctx, double* buf, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* LNASR_HIP_H */
