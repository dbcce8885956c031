// Runtime: context, device memory, staging, per-kernel HIP-event profiling, errors.
#include <algorithm>
#include <atomic>
#include <mutex>

#include "common.h"

namespace lnasr {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap && p) return 0;
    if (p) {
        hipError_t e = hipFree(p);
        p = nullptr;
        cap = 0;
        if (e != hipSuccess) return fail(LNASR_E_HIP, "hipFree: %s", hipGetErrorString(e));
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        return fail(LNASR_E_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
    }
    cap = want;
    return 0;
}

void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

void Model::release() {
    DevBuf* all[] = {&logA, &logAT, &logpi, &expA, &expAT, &logw, &mu, &ivar, &var, &prec, &gconst, &sigma_raw, &wpack, &wpack_nw, &wdirect, &pad_mu, &pad_ivar, &pad_gconst, &pad_logw, &pad_prec, &wfull, &wfull_nw, &pairtab, &flags, &centre, &mu_c, &wpack32, &lu_bad, &i8_g, &i8_pow2g, &i8_h, &i8_wq, &i8_const, &i8_const_nw, &fi8_gl, &fi8_pow2gl, &fi8_h, &fi8_const, &fi8_wq};
    for (DevBuf* b : all) b->release();
    has_hmm = has_gmm = false;
}

static Switches g_sw;
static std::atomic<bool> g_sw_loaded{false};      // release after g_sw is filled, acquire before it is read
static std::mutex g_sw_mu;

static void load_switches(Switches& s) {
    auto on = [](const char* name) { return getenv(name) != nullptr; };
    s = Switches();
    s.no_mfma = on("LNASR_NO_MFMA");
    s.no_pad = on("LNASR_NO_PAD");
    s.no_tinyvit = on("LNASR_NO_TINYVIT");
    s.no_tinyfwd = on("LNASR_NO_TINYFWD");
    s.no_reglu = on("LNASR_NO_REGLU");
    s.no_tilepack = on("LNASR_NO_TILEPACK");
    s.no_latfwd = on("LNASR_NO_LATFWD");
    s.no_i8emit = on("LNASR_NO_I8EMIT");
    if (const char* v = getenv("LNASR_I8_MIN_FRAMES")) s.i8_min_frames = atol(v);
    s.no_regvit = on("LNASR_NO_REGVIT");
    s.no_regfwd = on("LNASR_NO_REGFWD");
    s.no_colfwd = on("LNASR_NO_COLFWD");
    s.no_t256 = on("LNASR_NO_T256");
    s.no_framepar = on("LNASR_NO_FRAMEPAR");
    s.no_stream = on("LNASR_NO_STREAM");
    s.no_ldsbt = on("LNASR_NO_LDSBT");
    s.no_mfma_stats = on("LNASR_NO_MFMA_STATS");
    s.no_fft512 = on("LNASR_NO_FFT512");
    s.no_pipeline = on("LNASR_NO_PIPELINE");
    s.f32_generic = on("LNASR_F32_GENERIC");
    s.no_f32_stats = on("LNASR_NO_F32_STATS");
    s.no_pinned_ring = on("LNASR_NO_PINNED_RING");
#ifdef LNASR_DEBUG_GUARDS
    s.check_ptrs = true;                                   // the --host-asan build: always
#endif
    if (const char* e = getenv("LNASR_CHECK_PTRS")) s.check_ptrs = e[0] != '0';
    if (const char* e = getenv("LNASR_VIT")) s.vit = e[0];
    if (const char* e = getenv("LNASR_EMIT")) s.emit = e[0];
    if (const char* e = getenv("LNASR_FRAMEPAR_MAXU")) s.framepar_maxu = atoi(e);
    if (const char* e = getenv("LNASR_FWD_ONE_MAXU")) s.fwd_one_maxu = atoi(e);
    if (const char* e = getenv("LNASR_CHASE_MAXU")) s.chase_maxu = atoi(e);
    if (const char* e = getenv("LNASR_FWD_LAT_MAXU")) s.fwd_lat_maxu = atoi(e);
    if (const char* e = getenv("LNASR_KAPPA_MAX")) s.kappa_max = atof(e);
    if (const char* e = getenv("LNASR_I8_RHO_MAX")) {
        s.i8_rho_max = atof(e);
        s.i8_rho_max_set = true;
    }
    if (const char* e = getenv("LNASR_H2D_THREADS")) {
        s.h2d_threads = atoi(e);
        if (s.h2d_threads < 1) s.h2d_threads = 1;
        if (s.h2d_threads > 16) s.h2d_threads = 16;
    }
}

const Switches& sw() {
    if (!g_sw_loaded.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_sw_mu);
        if (!g_sw_loaded.load(std::memory_order_relaxed)) {
            load_switches(g_sw);
            g_sw_loaded.store(true, std::memory_order_release);
        }
    }
    return g_sw;
}

bool is_pinned_host_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // clear sticky "invalid value" for plain host memory
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int require_device_ptr(const void* p, const char* what, const char* where) {
    if (!p || is_device_ptr(p)) return 0;             // optional outputs may be null
    return fail(LNASR_E_INVALID, "%s: `%s` (%p) is not device memory -- an unstaged host pointer reached a kernel launcher",
                where, what, p);
}

int stage_in(lnasr_ctx* ctx, const void* p, size_t bytes, DevBuf& ws, const void** out) {
    if (bytes == 0) {
        LN_TRY(ws.reserve(256));
        *out = ws.p;
        return 0;
    }
    if (!p) return fail(LNASR_E_INVALID, "null input pointer");
    if (is_device_ptr(p)) {
        *out = p;
        return 0;
    }
    LN_TRY(ws.reserve(bytes));
    LN_HIP(hipMemcpyAsync(ws.p, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = ws.p;
    return 0;
}

int stage_out(lnasr_ctx* ctx, void* p, size_t bytes, DevBuf& ws, void** out) {
    if (p && is_device_ptr(p)) {
        *out = p;
        return 0;
    }
    LN_TRY(ws.reserve(bytes));
    *out = ws.p;
    return 0;
}

__global__ void widen_kernel(const float* __restrict__ s, double* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (double)s[i];
}
__global__ void narrow_kernel(const double* __restrict__ s, float* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
bool f32_via_f64() { return !sw().f32_generic; }
int widen_f32(lnasr_ctx* ctx, const void* src, size_t n, DevBuf& ws, double** out) {
    LN_TRY(ws.reserve(n * 8));
    *out = ws.as<double>();
    if (n == 0) return 0;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 65535 * 4);
    LN_LAUNCH(ctx, "widen_f32", widen_kernel, dim3(grid), dim3(256), 0, (const float*)src, *out, n);
    return 0;
}
int narrow_f64(lnasr_ctx* ctx, const double* src, size_t n, void* dst) {
    if (n == 0) return 0;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 65535 * 4);
    LN_LAUNCH(ctx, "narrow_f64", narrow_kernel, dim3(grid), dim3(256), 0, src, (float*)dst, n);
    return 0;
}

int finish_out(lnasr_ctx* ctx, void* user, const void* dev, size_t bytes) {
    if (!user || user == dev || bytes == 0) return 0;
    LN_HIP(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int prof_begin(lnasr_ctx* ctx, const char* name) {
    int id = -1;
    for (size_t i = 0; i < ctx->prof_names.size(); ++i)
        if (ctx->prof_names[i] == name) { id = (int)i; break; }
    if (id < 0) {
        id = (int)ctx->prof_names.size();
        ctx->prof_names.emplace_back(name);
        ctx->prof_launches.push_back(0);
        ctx->prof_ms.push_back(0.0);
    }
    hipEvent_t e[2];
    for (int k = 0; k < 2; ++k) {
        if (!ctx->event_pool.empty()) {
            e[k] = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else if (hipEventCreate(&e[k]) != hipSuccess) {
            return -1;
        }
    }
    if (hipEventRecord(e[0], ctx->stream) != hipSuccess) return -1;
    ctx->prof_recs.push_back({id, e[0], e[1]});
    return (int)ctx->prof_recs.size() - 1;
}

void prof_end(lnasr_ctx* ctx, int rec) { (void)hipEventRecord(ctx->prof_recs[rec].e1, ctx->stream); }

static int prof_collect(lnasr_ctx* ctx) {
    if (ctx->prof_recs.empty()) return 0;
    LN_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            ctx->prof_ms[r.name_id] += ms;
            ctx->prof_launches[r.name_id] += 1;
        }
        ctx->event_pool.push_back(r.e0);
        ctx->event_pool.push_back(r.e1);
    }
    ctx->prof_recs.clear();
    return 0;
}

}  // namespace lnasr

using namespace lnasr;

extern "C" {

int lnasr_abi_version(void) { return LNASR_ABI_VERSION; }

const char* lnasr_last_error(void) { return g_err; }

int lnasr_device_count(int* out) {
    LN_REQUIRE(out, "null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out = n;
    return 0;
}

int lnasr_ctx_create(int device_id, lnasr_ctx** out) {
    LN_REQUIRE(out, "null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(LNASR_E_NODEVICE, "no HIP device visible (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    }
    LN_REQUIRE(device_id >= 0 && device_id < n, "device_id %d out of range [0,%d)", device_id, n);
    LN_HIP(hipSetDevice(device_id));
    lnasr_ctx* ctx = new lnasr_ctx();
    ctx->device = device_id;
    LN_HIP(hipGetDeviceProperties(&ctx->prop, device_id));
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = ctx->prop.gcnArchName;
        delete ctx;
        return fail(LNASR_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_id, arch.c_str());
    }
    LN_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    LN_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    LN_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    LN_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    if (ctx->nan_flag.reserve(16) != 0 || hipMemset(ctx->nan_flag.p, 0, 16) != hipSuccess) {
        lnasr_ctx_destroy(ctx);
        return fail(LNASR_E_HIP, "allocating the context's flag word failed");
    }
    *out = ctx;
    return 0;
}

void lnasr_ctx_destroy(lnasr_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    lnasr_comm_destroy(ctx);
    ctx->model.release();
    DevBuf* all[] = {&ctx->ws_in, &ctx->ws_in2, &ctx->ws_in3, &ctx->ws_in4, &ctx->ws_in5, &ctx->ws_out, &ctx->ws_out2,
                     &ctx->ws_out3, &ctx->ws_b, &ctx->ws_bm, &ctx->ws_v, &ctx->ws_alpha, &ctx->ws_beta,
                     &ctx->ws_misc, &ctx->ws_misc2, &ctx->ws_lens, &ctx->ws_stats, &ctx->ws_p, &ctx->ws_q, &ctx->ws_raw, &ctx->ws_w1, &ctx->ws_w2, &ctx->ws_w3, &ctx->ws_pad, &ctx->ws_bp, &ctx->ws_btb, &ctx->ws_clock, &ctx->ws_fs, &ctx->ws_lvl, &ctx->ws_lvl2, &ctx->mfcc_const, &ctx->nan_flag, &ctx->nan_slots};
    for (DevBuf* b : all) b->release();
    for (auto& r : ctx->prof_recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->copy_events) (void)hipEventDestroy(e);
    for (int k = 0; k < lnasr_ctx::kRingSlots; ++k) {
        if (ctx->ring_ev[k]) (void)hipEventDestroy(ctx->ring_ev[k]);
        if (ctx->ring[k]) (void)hipHostFree(ctx->ring[k]);
    }
    if (ctx->mfcc_meta_ev) (void)hipEventDestroy(ctx->mfcc_meta_ev);
    if (ctx->mfcc_meta) (void)hipHostFree(ctx->mfcc_meta);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int lnasr_sync(lnasr_ctx* ctx) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int lnasr_device_sync(lnasr_ctx* ctx) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_HIP(hipDeviceSynchronize());
    return 0;
}

// TEST-ONLY: rewrites the switch table in place.  Must not run while another thread is inside the library (readers
// hold a reference to the table without the lock); the test-suite is single-threaded around it.
int lnasr_reload_switches(void) {
    std::lock_guard<std::mutex> lk(g_sw_mu);
    load_switches(g_sw);
    g_sw_loaded.store(true, std::memory_order_release);
    return 0;
}

int lnasr_debug_fallback_columns(lnasr_ctx* ctx, unsigned* out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && out, "null argument");
    unsigned* word = ctx->nan_flag.as<unsigned>() + 3;
    LN_HIP(hipStreamSynchronize(ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream2));
    LN_HIP(hipMemcpy(out, word, 4, hipMemcpyDeviceToHost));
    LN_HIP(hipMemset(word, 0, 4));
    return 0;
}

// Shader clock right now, as the GPU counts it: one wave spins for 20 us of the constant 100 MHz counter (s_memrealtime) and
// reports how many shader-clock cycles (s_memtime) went by.  Enqueued on the context's stream, so it runs directly behind
// whatever was launched last: the clock a kernel LEAVES the chip at (the power-limited integer kernels leave it
// lower than the fp64 ones).
__global__ void clock_probe_kernel(unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < 2000ull) r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

int lnasr_debug_clock(lnasr_ctx* ctx, double* ghz) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && ghz, "null argument");
    LN_TRY(ctx->ws_clock.reserve(64));
    unsigned long long* d = ctx->ws_clock.as<unsigned long long>();
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, d);
    LN_HIP(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    LN_HIP(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream));
    *ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    return 0;
}

int lnasr_debug_require_device_ptr(const void* p) { return require_device_ptr(p, "p", "lnasr_debug_require_device_ptr"); }

int lnasr_host_alloc(lnasr_ctx* ctx, size_t bytes, void** out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    LN_HIP(hipHostMalloc(out, bytes ? bytes : 256, hipHostMallocDefault));
    return 0;
}

int lnasr_host_free(lnasr_ctx* ctx, void* p) {
    LN_ENTER(ctx);
    if (!p) return 0;
    // the context is only needed to wait for copies that may still read the block; a block that outlives its context
    // (ctx == NULL) is freed after a device-wide wait instead of being leaked
    if (ctx) {
        LN_HIP(hipStreamSynchronize(ctx->stream));
        LN_HIP(hipStreamSynchronize(ctx->stream2));
    } else {
        LN_HIP(hipDeviceSynchronize());
    }
    LN_HIP(hipHostFree(p));
    return 0;
}

int lnasr_device_info(lnasr_ctx* ctx, char* name, int name_cap, int* cu_count, size_t* hbm_bytes) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    // (this pool's driver reports an empty marketing name: say what the device is rather than print " (gfx950...)")
    if (name && name_cap > 0)
        snprintf(name, name_cap, "%s (%s)", ctx->prop.name[0] ? ctx->prop.name : "AMD Instinct MI355X-class GPU", ctx->prop.gcnArchName);
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return 0;
}

int lnasr_dev_alloc(lnasr_ctx* ctx, size_t bytes, void** out) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    LN_HIP(hipMalloc(out, bytes ? bytes : 256));
    return 0;
}

int lnasr_dev_free(lnasr_ctx* ctx, void* p) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (!p) return 0;
    LN_HIP(hipStreamSynchronize(ctx->stream));
    LN_HIP(hipFree(p));
    return 0;
}

int lnasr_memcpy_h2d(lnasr_ctx* ctx, void* dst, const void* src, size_t bytes) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && (bytes == 0 || (dst && src)), "null argument");
    if (!bytes) return 0;
    LN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int lnasr_memcpy_d2h(lnasr_ctx* ctx, void* dst, const void* src, size_t bytes) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && (bytes == 0 || (dst && src)), "null argument");
    if (!bytes) return 0;
    LN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int lnasr_memset(lnasr_ctx* ctx, void* dst, int byte, size_t bytes) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && (bytes == 0 || dst), "null argument");
    if (!bytes) return 0;
    LN_HIP(hipMemsetAsync(dst, byte, bytes, ctx->stream));
    return 0;
}

int lnasr_prof_enable(lnasr_ctx* ctx, int on) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (!on) LN_TRY(prof_collect(ctx));
    ctx->prof_on = on != 0;
    return 0;
}

int lnasr_prof_reset(lnasr_ctx* ctx) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_TRY(prof_collect(ctx));
    for (auto& v : ctx->prof_launches) v = 0;
    for (auto& v : ctx->prof_ms) v = 0.0;
    return 0;
}

int lnasr_prof_count(lnasr_ctx* ctx, int* n) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && n, "null argument");
    LN_TRY(prof_collect(ctx));
    *n = (int)ctx->prof_names.size();
    return 0;
}

int lnasr_prof_get(lnasr_ctx* ctx, int i, char* name, int name_cap, int64_t* launches, double* total_ms) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_TRY(prof_collect(ctx));
    LN_REQUIRE(i >= 0 && i < (int)ctx->prof_names.size(), "index %d out of range", i);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", ctx->prof_names[i].c_str());
    if (launches) *launches = ctx->prof_launches[i];
    if (total_ms) *total_ms = ctx->prof_ms[i];
    return 0;
}

}  // extern "C"
