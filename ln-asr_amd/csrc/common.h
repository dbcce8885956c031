// Shared internals of liblnasr_hip.so (gfx950 only).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/lnasr_hip.h"

namespace lnasr {

// ---- errors -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define LN_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess)                                                         \
            return lnasr::fail(LNASR_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                               hipGetErrorString(e__));                                \
    } while (0)

#define LN_TRY(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__ != 0) return rc__; \
    } while (0)

#define LN_REQUIRE(cond, ...)                                      \
    do {                                                           \
        if (!(cond)) return lnasr::fail(LNASR_E_INVALID, __VA_ARGS__); \
    } while (0)

// First statement of every C-ABI entry point that takes a context: the HIP "current device" is per thread, so a
// call from a thread other than the one that created the context (or after another library moved the thread's device) would
// otherwise allocate workspaces and launch on device 0.  hipSetDevice is a cheap thread-local store.
#define LN_ENTER(ctx)                                          \
    do {                                                       \
        if (ctx) {                                             \
            LN_HIP(hipSetDevice((ctx)->device));               \
            if (++(ctx)->nan_epoch == 0) (ctx)->nan_epoch = 1; \
        }                                                      \
    } while (0)

// ---- grow-only device buffer -----------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- uploaded model -----------------------------------------------------------------------
struct Model {
    int N = 0, M = 0, D = 0;
    bool has_hmm = false;   // logA / logpi valid
    bool has_gmm = false;   // mixture parameters valid
    bool diag = false;      // diagonal fast path
    DevBuf logA, logAT, logpi;   // [N,N], transposed [N,N], [N]
    DevBuf expA, expAT;          // exp(logA) [N,N] and its transpose (linear-domain forward/backward)
    DevBuf logw;                 // [N,M]
    DevBuf mu;                   // [N,M,D]
    DevBuf ivar;                 // diag: 1/var [N,M,D]
    DevBuf var;                  // diag: var [N,M,D]
    DevBuf prec;                 // full: Sigma^-1 [N,M,D,D]
    DevBuf gconst;               // [N,M]: -0.5 (D ln 2pi + ln det Sigma)
    DevBuf sigma_raw;            // full upload staging [N,M,D,D]
    DevBuf lu_bad;               // [N,M] int: components the register LU declined (redone by the LDS kernel)
    DevBuf wpack, wpack_nw;      // MFMA-packed coefficient fragments (diag): with / without log w folded in
    int wpack_Q = 0, wpack_KS = 0, wpack_NG = 0;
    DevBuf wdirect;              // N = 1: coefficient fragments of the single mixture (gmm.hip: emit_single_direct)
    int wdirect_KS = 0;
    DevBuf wfull, wfull_nw, pairtab;   // full covariance: quadratic-feature coefficients in MFMA fragment order
    int wfull_Q = 0, wfull_NCH = 0, wfull_virt = 1;
    int wpack_virt = 1;          // M = 16*virt mixtures are scored as `virt` virtual 16-mixture states, then merged
    // shapes the matrix-core bank kernel does not take directly (N not a multiple of 4, M not in {4, 8, 16, 16k}) are
    // scored on a padded copy of the parameters (padding components have weight -inf) and compacted afterwards
    DevBuf pad_mu, pad_ivar, pad_gconst, pad_logw, pad_prec;
    int pad_N = 0, pad_M = 0;    // padded sizes, 0 = the model is used as it is
    DevBuf flags;                // small scratch: [0] int flag (diag detection / degenerate variance), [8] kappa bits, [16] rho bits
    // The matrix-core emission kernels evaluate the EXPANDED quadratic  x'^T S x' + L.x' + K  on centred observations
    // x' = x - centre (centre[d] = mean of the component means): the expansion cancels terms of size kappa =
    // max_c (mu_c - centre)^T Sigma_c^-1 (mu_c - centre), so its absolute error is ~ 3e-16 * kappa where the direct
    // (x - mu) form the reference uses (gmm.py:59-68) has ~1e-14.  Models with kappa > kKappaMax (means many standard
    // deviations apart: MFMA result no longer good to 1e-12) are scored by the direct-form kernels instead.
    DevBuf centre;               // [D]
    DevBuf mu_c;                 // [N,M,D] mu - centre: what the coefficient packers see
    double kappa = 0.0;
    bool direct_form = false;    // kappa > kKappaMax (or not finite): expanded form refused
    bool centred = false;        // centre != 0 (the model was not well conditioned about the origin)
    // fp32 matrix-core pass (f32stats.hip): float coefficient fragments, packed on first use after an upload
    DevBuf wpack32;
    unsigned long upload_serial = 0, wpack32_for = ~0ul;
    // integer matrix-core bank (gmm_i8.hip): digit planes of the coefficients, row scales, component constants
    DevBuf i8_g, i8_pow2g, i8_h, i8_wq, i8_const, i8_const_nw;   // i8_h [Nv, mix] int: the components' exponents
    double i8_rho = 0.0;         // the model's range as the fixed point sees it (gmm_i8.hip); beyond kI8RhoMax: the fp64 bank
    // full-covariance bank on the integer matrix cores (gmm_full_i8.hip): W planes in A-operand order, scales, constants
    DevBuf fi8_gl, fi8_pow2gl, fi8_h, fi8_const, fi8_wq;
    int fi8_nk = 0, fi8_Nv = 0;  // k-steps of 64 features (0: not packed), (virtual, padded) states
    double fi8_rho = 0.0;
    const double* i8_logw = nullptr;   // the (padded) log weights the planes were packed with: [Nv,16]
    int i8_NG = 0;               // state groups packed (0: none)
    int i8_mix = 16;             // mixtures per (virtual) state of the packed planes: 16 or 8
    void release();
};

constexpr double kKappaMax = 4096.0;
// The integer matrix-core bank is a fixed-point product: its absolute error on a component's log-density is ~3 * 2^-53 * rho at
// the frames that component describes (gmm_i8.hip header; rho = 26 for the metric's model).  2048 -> <= 1e-12; beyond it the
// fp64 matrix-core bank scores the model.
constexpr double kI8RhoMax = 2048.0;
// its four-plane instantiation (the float32 statistics pass's hand-off) keeps 32 bits: ~3 * 2^-30 * rho, 512 -> <= 1.5e-6 of
// the 8e-6 that pass promises
constexpr double kI8RhoMaxF32 = 512.0;
// The fp32 matrix-core statistics pass (f32stats.hip) evaluates the same expanded quadratic in FLOAT: its
// cancellation error is ~6e-8 * kappa in every component log-density, so it gets a tighter gate than the fp64
// kernels (128 -> <= 8e-6 absolute, the float32 accuracy include/lnasr_hip.h promises); above it the fp64 pass runs.
constexpr double kKappaMaxF32 = 128.0;

// Test / ablation switches (LNASR_* environment variables).  Read ONCE per process, on first use;
// lnasr_reload_switches() re-reads them (the test-suite toggles switches inside one process).
struct Switches {
    bool no_mfma = false, no_pad = false, no_tinyvit = false, no_regvit = false, no_regfwd = false, no_colfwd = false,
         no_t256 = false, no_framepar = false, no_stream = false, no_ldsbt = false, no_mfma_stats = false,
         no_fft512 = false, no_pipeline = false, f32_generic = false, no_f32_stats = false, no_pinned_ring = false,
         no_tinyfwd = false, no_reglu = false, no_tilepack = false, no_latfwd = false, no_i8emit = false;
    char vit = 0;                 // LNASR_VIT: first character, 0 = unset
    int framepar_maxu = -1;       // LNASR_FRAMEPAR_MAXU, -1 = unset (measured rule)
    int chase_maxu = -1;          // LNASR_CHASE_MAXU: back-pointer chase instead of the arg-max back-trace up to this batch size
    int fwd_lat_maxu = -1;        // LNASR_FWD_LAT_MAXU: the chain-shortened one-utterance forward / backward (fwdbwd_lat4) up to this batch size
    long i8_min_frames = -1;      // LNASR_I8_MIN_FRAMES: the integer matrix-core bank from this many frames per launch (-1: measured rule)
    int fwd_one_maxu = -1;        // LNASR_FWD_ONE_MAXU: one utterance per block in the column-owned forward / backward up to this batch size
    double kappa_max = kKappaMax; // LNASR_KAPPA_MAX
    double i8_rho_max = kI8RhoMax; // LNASR_I8_RHO_MAX (set: it replaces the gates of BOTH integer banks, diagonal and full)
    bool i8_rho_max_set = false;
    int h2d_threads = 4;          // LNASR_H2D_THREADS: host threads filling the pinned staging ring
    bool check_ptrs = false;      // LNASR_CHECK_PTRS: LN_DEVPTR verifies every pointer a launch helper is handed
    char emit = 0;                // LNASR_EMIT (ablation builds only)
};
const Switches& sw();

struct ProfRec {
    int name_id;
    hipEvent_t e0, e1;
};

}  // namespace lnasr

struct lnasr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // side stream: the backward pass runs beside the forward pass
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<hipEvent_t> copy_events;      // one per upload chunk of emit_batch
    // pinned staging ring of the upload pipeline (pageable host observations -> pinned slots -> HBM); grow-only
    static constexpr int kRingSlots = 4;
    void* ring[kRingSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t ring_cap = 0;                      // bytes per slot
    hipEvent_t ring_ev[kRingSlots] = {nullptr, nullptr, nullptr, nullptr};   // the copy that last read the slot
    bool ring_busy[kRingSlots] = {false, false, false, false};               // ring_ev[s] was recorded and not waited for yet
    hipDeviceProp_t prop;
    lnasr::Model model;
    // workspace
    lnasr::DevBuf ws_in, ws_in2, ws_in3, ws_in4, ws_in5;   // staged host inputs
    lnasr::DevBuf ws_out, ws_out2, ws_out3;                // staged outputs
    lnasr::DevBuf ws_b, ws_bm, ws_v, ws_alpha, ws_beta, ws_misc, ws_misc2, ws_lens, ws_stats;
    int lens_fill_U = 0, lens_fill_T = -1;                 // ws_lens holds "every utterance is T frames long" for this many utterances
    lnasr::DevBuf ws_p, ws_q, ws_raw;                      // Baum-Welch statistics: xi factors, raw mixture moments
    lnasr::DevBuf ws_w1, ws_w2, ws_w3;                     // float32 storage: widened inputs / outputs of the f64 kernels
    lnasr::DevBuf ws_pad;                                  // bm of a padded model before compaction
    lnasr::DevBuf ws_bp;                                   // byte back-pointers of the tiny-model Viterbi (N <= 16)
    lnasr::DevBuf ws_lvl, ws_lvl2;                         // emission-row levels of the N > 256 forward / backward kernels (bign.hip)
    lnasr::DevBuf ws_fs;                                   // digit planes of the features (gmm_i8.hip)
    lnasr::DevBuf ws_btb;                                  // byte back-pointer rows of the small-batch back-trace (backtrace.hip)
    lnasr::DevBuf ws_clock;                                // two words of the clock probe (lnasr_debug_clock)
    // non-finite data (nanfix.hip): a device word the hot kernels set to the running call's epoch when they meet a NaN or
    // an infinity, and the per-utterance slots of the scan that follows
    lnasr::DevBuf nan_flag, nan_slots;
    unsigned nan_epoch = 0;                                // number of the running API call (LN_ENTER); never 0
    bool fwd_nan_tested = false;                           // the last forward / backward kernel tested its emissions itself
    const void* bp_v = nullptr;                            // the v array ws_bp belongs to (set by viterbi_device,
    int bp_U = 0, bp_T = 0, bp_nmax = 0;                   //   consumed by backtrace_device)
    // MFCC constants (twiddles, DCT matrix), kept on the device between calls with the same fft_n / n_mel;
    // the per-call frame offsets live here too so that the asynchronous upload never reads a dead host buffer
    lnasr::DevBuf mfcc_const;
    int mfcc_nfft = 0, mfcc_nmel = 0;
    std::vector<int64_t> mfcc_foff;
    int64_t* mfcc_meta = nullptr;                          // pinned: sample offsets [U+1] then frame offsets [U+1] of the call
    size_t mfcc_meta_cap = 0;                              // ... in elements
    hipEvent_t mfcc_meta_ev = nullptr;                     // the upload that last read mfcc_meta
    // profiling
    bool prof_on = false;
    std::vector<std::string> prof_names;
    std::vector<lnasr::ProfRec> prof_recs;
    std::vector<hipEvent_t> event_pool;
    std::vector<int64_t> prof_launches;
    std::vector<double> prof_ms;
    // comm
    void* comm = nullptr;
    int rank = 0, nranks = 1;
};

namespace lnasr {

// LN_LAUNCH (and its profiling events) follow ctx->stream: device helpers are sent to another stream of the context
// by this guard, which puts the stream back on EVERY way out of the scope (an early return through LN_TRY included)
struct StreamScope {
    lnasr_ctx* c;
    hipStream_t saved;
    StreamScope(lnasr_ctx* ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { ctx->stream = s; }
    ~StreamScope() { c->stream = saved; }
    StreamScope(const StreamScope&) = delete;
    StreamScope& operator=(const StreamScope&) = delete;
};
int prof_begin(lnasr_ctx* ctx, const char* name);   // returns record index or -1
void prof_end(lnasr_ctx* ctx, int rec);

// Launch wrapper: optional HIP-event bracket on the context's stream + launch error check.
#define LN_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                                 \
    do {                                                                                      \
        int rec__ = (ctx)->prof_on ? lnasr::prof_begin((ctx), (name)) : -1;                   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);           \
        if (rec__ >= 0) lnasr::prof_end((ctx), rec__);                                        \
        hipError_t le__ = hipGetLastError();                                                  \
        if (le__ != hipSuccess)                                                               \
            return lnasr::fail(LNASR_E_HIP, "launch %s failed: %s", (name), hipGetErrorString(le__)); \
    } while (0)

// Is `p` device memory?  (host pointers from numpy are unregistered)
bool is_device_ptr(const void* p);
// Guard of the launch helpers (`*_device`, `*_nan_repair`, emit_*): they take DEVICE pointers only -- the entry points
// stage host buffers first (stage_in / stage_out).  A host pointer that slips through (round 3: a side-stream
// experiment handed the caller's numpy `obs` to the moments kernel instead of its staged copy) is a GPU memory fault
// at best.  Under LNASR_CHECK_PTRS=1 -- always in the --host-asan build -- every such pointer is verified with
// hipPointerGetAttributes before anything is launched and the call fails with LNASR_E_INVALID instead.
int require_device_ptr(const void* p, const char* what, const char* where);
#define LN_DEVPTR(p)                                                                    \
    do {                                                                                \
        if (lnasr::sw().check_ptrs) LN_TRY(lnasr::require_device_ptr((p), #p, __func__)); \
    } while (0)
// Is `p` page-locked host memory (lnasr_host_alloc / hipHostMalloc / hipHostRegister)?  Such buffers are copied
// asynchronously at PCIe rate without a staging copy.
bool is_pinned_host_ptr(const void* p);

// Bring an input to the device (returns the device pointer to use; stages through `ws` if host).
int stage_in(lnasr_ctx* ctx, const void* p, size_t bytes, DevBuf& ws, const void** out);
// Prepare an output: returns device pointer to write; if host, `ws` is used and finish_out copies back.
int stage_out(lnasr_ctx* ctx, void* p, size_t bytes, DevBuf& ws, void** out);
int finish_out(lnasr_ctx* ctx, void* user, const void* dev, size_t bytes);

// Baum-Welch statistics on the matrix cores (bwstats.hip)
int xi_sum_device(lnasr_ctx* ctx, const double* alpha, const double* beta, const double* b, const double* loglik,
                  const int32_t* d_lens, int U, int T, double* xi_sum);
bool mix_stats_mfma_ok(const Model& md);
// phase (mix_stats_device, mix_stats_device_f32): kMixBegin zeroes the raw sums, kMixAdd adds the batch given,
// kMixEnd turns the raw sums into mix / mix_x / mix_xx.  One call with kMixAll does everything; the pipelined pass
// (estimate.hip) begins once, adds chunk after chunk and ends once.
constexpr int kMixPadFrames = 64;   // slack (frames) behind bm / q32: the moments kernels prefetch past their range
constexpr int kMixBegin = 1, kMixAdd = 2, kMixEnd = 4, kMixAll = 7;
int mix_stats_device(lnasr_ctx* ctx, const double* obs, const double* alpha, const double* beta, const double* b,
                     const double* bm, const double* c1, const int32_t* d_lens, int U, int T, double* mix,
                     double* mix_x, double* mix_xx, int phase = kMixAll);

bool full_stats_mfma_ok(const Model& md);

// `precision=np.float32` statistics pass on the fp32 matrix cores (f32stats.hip)
bool f32_stats_ok(const Model& md);
int emit_device_f32(lnasr_ctx* ctx, const double* obs, long F, double* b, float* q32);
int mix_stats_device_f32(lnasr_ctx* ctx, const double* obs, const double* alpha, const double* beta, const double* b,
                         const float* q32, const double* c1, const int32_t* d_lens, int U, int T, double* mix,
                         double* mix_x, double* mix_xx, int phase = kMixAll);
int emit_device(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm);
// gmm_i8.hip: the diagonal bank on the integer matrix cores (fixed-point digit planes)
bool i8_emit_shape_ok(int Nv, int Mv, int D);
bool i8_model_ok(const Model& md);          // planes packed, range within the gate, not switched off
bool i8_route(const Model& md, long F);     // ... and a batch of F frames is large enough to take it
int pack_i8_model(lnasr_ctx* ctx, const double* mu_s, const double* iv_s, const double* gc_s, const double* lw_s, int Nv, int Mv);
int emit_diag_i8_device(lnasr_ctx* ctx, const double* obs, long F, double* b, double* bm, float* q32 = nullptr);
// gmm_full_i8.hip: the full-covariance bank on the integer matrix cores (b only)
bool full_i8_model_ok(const Model& md);
bool full_i8_route(const Model& md, long F);
int pack_full_i8_model(lnasr_ctx* ctx, const double* mu_p, const double* prec_p, const double* gconst_p, const double* logw_p,
                       int Np, int Mp, int Nv);
int emit_full_i8_device(lnasr_ctx* ctx, const double* obs, long F, double* b);
// emit_device without the pass over non-finite frames; *scanned: the kernel tested the features it staged and raised
// ctx->nan_flag itself (emit_batch runs the kernels chunk by chunk and the pass once)
int emit_device_raw(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm, bool* scanned,
                    bool* rows_done_or_null);      // *rows_done: b / bm already hold NaN rows for such frames
int full_stats_device(lnasr_ctx* ctx, const double* obs, const double* alpha, const double* beta, const double* b,
                      const double* bm, const double* c1, const int32_t* d_lens, int U, int T, double* mix,
                      double* mix_x, double* mix_xx);

// Emission of a batch whose observations may still be in host memory: the upload is cut into chunks that a helper
// thread pushes through the side stream while the emission kernel already runs on the chunks that have arrived
// (device-resident observations: one plain launch).  b [U*T, N], bm [U*T, N, M] or null are device pointers.
// scanned_or_null: given, the pass over non-finite frames (nan_rows_device) is left to the caller -- decode and calc_prob
// repair their own outputs from the observations -- and *scanned says whether every kernel tested its features itself.
int emit_batch(lnasr_ctx* ctx, const double* obs, int U, int T, double* b, double* bm, const double** d_obs_out,
               bool* scanned_or_null = nullptr);

// float32 storage (`precision=np.float32`) of the EMISSION kernels: f32 outputs are produced by the f64 kernels and
// rounded on the way out (gmm.hip; bm first, b from the rounded bm as the reference does, gmmhmm.py:64-67).
// LNASR_F32_GENERIC=1 selects the storage-templated scalar kernels instead.  The trellises do not take this route:
// their float-storage kernels round at every store and continue from the rounded value (viterbi.hip, fwdbwd.hip,
// trellis256.hip), which is what the reference's float32 arrays do.
bool f32_via_f64();
int widen_f32(lnasr_ctx* ctx, const void* src, size_t n, DevBuf& ws, double** out);
int narrow_f64(lnasr_ctx* ctx, const double* src, size_t n, void* dst);

// Non-finite data, the reference's way (nanfix.hip).  force: scan the input; else act only when a kernel of this
// API call raised ctx->nan_flag.
int nan_rows_device(lnasr_ctx* ctx, const double* obs, long F, int dtype, void* b, void* bm, bool force);
// full covariance, or D = 1: an infinite feature meets no zero of the inverse, the reference's densities are -inf or NaN
// component by component; nan_rows_device re-scores such frames and the trellises are repaired from the emissions
bool nan_rows_rescored(const Model& md);
int viterbi_nan_repair(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* v,
                       uint32_t* path, double* best, bool force);          // the caller's emissions
int viterbi_nan_repair_obs(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* d_lens, double* v,
                           uint32_t* path, double* best, bool force);       // decode: keyed on the observations
int loglik_nan_repair(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* d_lens, double* loglik, bool force);
int trellis_nan_repair(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* alpha_or_null,
                       double* loglik_or_null, void* beta_or_null, bool force);     // lnasr_forward / lnasr_backward
int stats_nan_repair(lnasr_ctx* ctx, const double* obs, int U, int T, const int32_t* d_lens, double* stats, long n,
                     bool force);

inline size_t dtype_size(int dtype) { return dtype == LNASR_F32 ? 4 : 8; }

}  // namespace lnasr
