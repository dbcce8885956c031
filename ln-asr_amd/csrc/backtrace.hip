// Back-trace of the Viterbi trellis (hmm.py:168-171) and the full back-pointer matrix (hmm.py:165).
// The pointer needed at each step, argmax_i(v[t,i] + A[i, path[t+1]]), is recomputed from the stored v with the
// same fp64 additions as the recursion made (first index on ties, numpy.argmax's rule).
#include "common.h"
#include "devmath.h"
#include "trellis.h"

namespace lnasr {

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
// SUSPECT = false: nobody will read the suspect word (decode on a diagonal model: the emission kernel tested the features;
// lnasr_viterbi when the Viterbi kernel tested the emissions itself) -- no cell of v is tested at all.
template <int CPL, bool LDSAT, typename TS = double, bool FULL = false, bool SUSPECT = true>
__global__ __launch_bounds__(1024) void backtrace_lds_kernel(const TS* __restrict__ v,
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
        if (SUSPECT) suspect |= __ballot(!(v0 > LN_NEG_INF));
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
                    if (SUSPECT) suspect |= __ballot(!(cur[k][c] > LN_NEG_INF));
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
    if (SUSPECT && suspect != 0 && lane == 0) *suspect_flag = epoch;
}

// full back-pointer matrix (hmm.py:165): bt[u,t,j] = argmax_i(v[u,t-1,i] + A[i,j]); bt[u,0,:] = 0.
// OUT = uint32_t, pitch N: the array HMM._viterbi returns.  OUT = uint8_t (N <= 256), pitch a multiple of 16: the rows the
// small-batch back-trace chases through (backtrace_chase_kernel below).
template <typename TS, typename OUT>
__global__ void bt_kernel(const TS* __restrict__ v, const double* __restrict__ A, const int32_t* __restrict__ lens,
                          OUT* __restrict__ bt, int T, int N, int pitch) {
    extern __shared__ double vs[];            // [N]
    const int u = blockIdx.y, t = blockIdx.x;
    if (t >= lens[u]) return;
    OUT* out = bt + ((size_t)u * T + t) * pitch;
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
        out[j] = (OUT)bi;
    }
}

// Small batches (round 5; the reference's decode(obs) is ONE utterance, hmm.py:276-283): the arg-max chain of
// backtrace_lds_kernel costs ~550 cycles a frame and nothing shortens a dependent chain -- but with a handful of utterances
// the chip is idle, so the back-pointers of EVERY state are formed first, all frames at once (bt_kernel above, bytes), and
// the chain that is left is one LDS byte read per frame: path[t-1] = bt[t][path[t]].  One wave per utterance; the rows
// travel HBM -> registers -> LDS FR frames at a time, the next chunk in flight while the current one is walked.
// Same additions, same first-index rule as the recomputing kernels: the same path.  (Batches: the N^2 work per frame is the
// Viterbi kernel's all over again -- from ~32 utterances on the recomputing kernel is cheaper, dispatch::backtrace_chase.)
template <typename TS, int NPV>   // NPV: 16-byte pieces per lane and chunk
__global__ __launch_bounds__(64) void backtrace_chase_kernel(const TS* __restrict__ v, const uint8_t* __restrict__ btb,
                                                             const int32_t* __restrict__ lens, uint32_t* __restrict__ path,
                                                             double* __restrict__ best, int U, int T, int N, int NP, int FR) {
    extern __shared__ uint8_t rows[];                         // [2][NPV * 1024] (>= FR * NP)
    const int lane = threadIdx.x, u = blockIdx.x;
    const int L = lens[u];
    if (L <= 0) return;
    const TS* vu = v + (size_t)u * T * N;
    uint32_t* pu = path + (size_t)u * T;
    const uint8_t* bu = btb + (size_t)u * T * NP;
    // last frame: first index among equal maxima (lane l holds states l, l + 64, ...), as backtrace_lds_kernel picks it
    int p;
    {
        double x[4], m = LN_NEG_INF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = lane + 64 * c;
            x[c] = j < N ? ld(vu, (size_t)(L - 1) * N + j) : LN_NEG_INF;
            m = vmax_f64(m, x[c]);
        }
        const double mx = wave_max_dpp(m);
        p = 0;
#pragma unroll
        for (int c = 3; c >= 0; --c) {
            const unsigned long long mk = __ballot(x[c] == mx);
            p = mk ? 64 * c + __ffsll((long long)mk) - 1 : p;
        }
        if (lane == 0 && best) best[u] = mx;
    }
    uint32_t mine = 0;
    if (lane == ((L - 1) & 63)) mine = (uint32_t)p;
    if (((L - 1) & 63) == 0) pu[L - 1] = (uint32_t)p;
    if (L == 1) return;
    // rows 1 .. L-1 are walked downwards in chunks of FR: chunk k = rows hi_k - FR + 1 .. hi_k (clipped at 1)
    constexpr int chunk_bytes = NPV * 1024;                   // >= FR * NP: 16 bytes x 64 lanes x NPV pieces
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 stage[NPV];
    auto fetch = [&](int hi) {
        const int lo = hi - FR + 1 > 1 ? hi - FR + 1 : 1;
        const uint8_t* src = bu + (size_t)lo * NP;
        const int nbytes = (hi - lo + 1) * NP;
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            const int off = (q * 64 + lane) * 16;
            stage[q] = off < nbytes ? *reinterpret_cast<const u32x4*>(src + off) : u32x4{0, 0, 0, 0};
        }
    };
    fetch(L - 1);
    int buf = 0;
    for (int hi = L - 1; hi >= 1; hi -= FR, buf ^= 1) {
        const int lo = hi - FR + 1 > 1 ? hi - FR + 1 : 1;
        uint8_t* dst = rows + (size_t)buf * chunk_bytes;
#pragma unroll
        for (int q = 0; q < NPV; ++q) *reinterpret_cast<u32x4*>(dst + (q * 64 + lane) * 16) = stage[q];
        __syncthreads();                                      // one wave: orders the LDS writes before the reads below
        if (lo > 1) fetch(lo - 1);                            // the next chunk travels while this one is walked
        for (int t = hi; t >= lo; --t) {
            p = dst[(t - lo) * NP + p];                       // path[t-1] = bt[t][path[t]]
            const int tp = t - 1;
            if (lane == (tp & 63)) mine = (uint32_t)p;
            if ((tp & 63) == 0 && tp + lane < L) pu[tp + lane] = mine;      // frames tp .. tp + 63 are complete
        }
    }
}

int backtrace_device(lnasr_ctx* ctx, const void* v, int U, int T, const int32_t* d_lens, int dtype, uint32_t* path,
                     double* best, bool* v_tested, bool want_suspect) {
    LN_DEVPTR(v); LN_DEVPTR(d_lens); LN_DEVPTR(path); LN_DEVPTR(best);
    // *v_tested: the kernel looked at every cell of v and raised the context's "suspect" word on NaN / -inf (nanfix.hip)
    Model& md = ctx->model;
    // waves (= utterances) per block of the LDS kernels: four while the batch fits one round of blocks (one per CU: A^T takes
    // 128 KB of its LDS), more -- up to sixteen, four chains per SIMD -- when it does not: the chains are latency-bound and
    // interleave, so 8192 utterances on one GPU take two rounds instead of eight (2.17 -> 1.51 ms; 2048: 0.52 -> 0.43)
    const int wpb = dispatch::backtrace_waves_per_block(U, ctx->prop.multiProcessorCount);
    unsigned* const sflag = ctx->nan_flag.as<unsigned>() + 1;
    const unsigned epoch = ctx->nan_epoch;
    if (v_tested) *v_tested = false;
    if (ctx->bp_v == v && ctx->bp_U == U && ctx->bp_T == T && v != nullptr) {
        // v came from viterbi_tiny_kernel: its byte back-pointers are in ws_bp
        const uint8_t* bp = ctx->ws_bp.as<uint8_t>();
        const int nmax = ctx->bp_nmax;
        ctx->bp_v = nullptr;
        const dim3 grid((U + 3) / 4), block(256);           // four waves per block: the kernel's LDS slices are [4]
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
    // a handful of utterances whose cells nobody has to test: all back-pointers at once, then a chain of LDS byte reads
    if (!want_suspect && md.N <= 256 && dispatch::backtrace_chase(U) && !sw().no_ldsbt) {
        const int NP = (md.N + 15) / 16 * 16;
        const int FR = NP <= 128 ? 128 : 64;                  // 16 KB (NP = 128) ... 16 KB (NP = 256) per chunk
        const int npv = FR * NP / 1024;                       // 16-byte pieces per lane: 2 ... 16
        LN_TRY(ctx->ws_btb.reserve((size_t)U * T * NP + 1024));
        uint8_t* btb = ctx->ws_btb.as<uint8_t>();
        int nt = ((md.N + 63) / 64) * 64;
        if (nt > 256) nt = 256;
#define LN_CHASE(TSv, tag)                                                                                          \
    do {                                                                                                            \
        LN_LAUNCH(ctx, "viterbi_bt_bytes" tag, (bt_kernel<TSv, uint8_t>), dim3(T, U), dim3(nt), (size_t)md.N * 8,    \
                  (const TSv*)v, md.logA.as<double>(), d_lens, btb, T, md.N, NP);                                   \
        if (npv <= 2) LN_LAUNCH(ctx, "backtrace_chase" tag, (backtrace_chase_kernel<TSv, 2>), dim3(U), dim3(64), 2 * 2 * 1024, (const TSv*)v, btb, d_lens, path, best, U, T, md.N, NP, FR); \
        else if (npv <= 4) LN_LAUNCH(ctx, "backtrace_chase" tag, (backtrace_chase_kernel<TSv, 4>), dim3(U), dim3(64), 2 * 4 * 1024, (const TSv*)v, btb, d_lens, path, best, U, T, md.N, NP, FR); \
        else if (npv <= 8) LN_LAUNCH(ctx, "backtrace_chase" tag, (backtrace_chase_kernel<TSv, 8>), dim3(U), dim3(64), 2 * 8 * 1024, (const TSv*)v, btb, d_lens, path, best, U, T, md.N, NP, FR); \
        else LN_LAUNCH(ctx, "backtrace_chase" tag, (backtrace_chase_kernel<TSv, 16>), dim3(U), dim3(64), 2 * 16 * 1024, (const TSv*)v, btb, d_lens, path, best, U, T, md.N, NP, FR); \
    } while (0)
        if (dtype == LNASR_F32) LN_CHASE(float, "_f32");
        else LN_CHASE(double, "");
#undef LN_CHASE
        return 0;
    }
    // SUSPECT variant only when somebody will read the word (want_suspect: the caller's emissions were not tested upstream)
#define LN_BT(name, CPLv, LDSv, TSv, FULLv, grid, block, shm)                                                          \
    do {                                                                                                               \
        if (want_suspect)                                                                                               \
            LN_LAUNCH(ctx, name, (backtrace_lds_kernel<CPLv, LDSv, TSv, FULLv, true>), grid, block, shm, (const TSv*)v,  \
                      md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);                            \
        else                                                                                                           \
            LN_LAUNCH(ctx, name, (backtrace_lds_kernel<CPLv, LDSv, TSv, FULLv, false>), grid, block, shm, (const TSv*)v, \
                      md.logAT.as<double>(), d_lens, path, best, U, T, md.N, sflag, epoch);                            \
        if (v_tested) *v_tested = want_suspect;                                                                         \
        return 0;                                                                                                      \
    } while (0)
    const dim3 g4((U + wpb - 1) / wpb), b4(64 * wpb), g1(U), b1(64);
    const size_t at_lds = (size_t)md.N * md.N * 8;
    if (dtype == LNASR_F64 && md.N <= 128 && !sw().no_ldsbt) {
        if (md.N == 128) LN_BT("backtrace_lds", 2, true, double, true, g4, b4, at_lds);
        LN_BT("backtrace_lds", 2, true, double, false, g4, b4, at_lds);
    }
    if (dtype == LNASR_F64 && md.N <= 1024 && !sw().no_ldsbt) {
        // one wave per block (with few utterances the waves should sit on different CUs), N / 64 states per lane
        if (md.N <= 256) LN_BT("backtrace_wave256", 4, false, double, false, g1, b1, 0);
        if (md.N <= 512) LN_BT("backtrace_wave512", 8, false, double, false, g1, b1, 0);
        LN_BT("backtrace_wave1024", 16, false, double, false, g1, b1, 0);
    }
    if (dtype == LNASR_F32 && md.N <= 128 && !sw().no_ldsbt && !sw().f32_generic)
        LN_BT("backtrace_lds_f32", 2, true, float, false, g4, b4, at_lds);
    if (dtype == LNASR_F32 && md.N <= 256 && !sw().no_ldsbt && !sw().f32_generic)
        LN_BT("backtrace_wave256_f32", 4, false, float, false, g1, b1, 0);
#undef LN_BT
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "backtrace_f32", backtrace_kernel<float>, dim3((U + 3) / 4), dim3(256), 0,
                  (const float*)v, md.logAT.as<double>(), d_lens, path, best, U, T, md.N);
    else
        LN_LAUNCH(ctx, "backtrace", backtrace_kernel<double>, dim3((U + 3) / 4), dim3(256), 0,
                  (const double*)v, md.logAT.as<double>(), d_lens, path, best, U, T, md.N);
    return 0;
}

// full back-pointer matrix bt [U,T,N] from the stored v (only when the caller asks for it: HMM._viterbi returns it)
int bt_matrix_device(lnasr_ctx* ctx, const void* v, int U, int T, const int32_t* d_lens, int dtype, uint32_t* bt) {
    LN_DEVPTR(v); LN_DEVPTR(d_lens); LN_DEVPTR(bt);
    const int N = ctx->model.N;
    LN_REQUIRE(U <= 65535, "U=%d too large for the bt kernel", U);
    int nt = ((N + 63) / 64) * 64;
    if (nt > 256) nt = 256;
    if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "viterbi_bt_f32", (bt_kernel<float, uint32_t>), dim3(T, U), dim3(nt), (size_t)N * 8, (const float*)v,
                  ctx->model.logA.as<double>(), d_lens, bt, T, N, N);
    else
        LN_LAUNCH(ctx, "viterbi_bt", (bt_kernel<double, uint32_t>), dim3(T, U), dim3(nt), (size_t)N * 8, (const double*)v,
                  ctx->model.logA.as<double>(), d_lens, bt, T, N, N);
    return 0;
}

}  // namespace lnasr
