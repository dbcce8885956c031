// `precision=np.float32` for the multi-utterance Baum-Welch pass (BASELINE config 5 asks for an fp32 match): the two
// GEMM-shaped stages -- component log-densities (gmmhmm.py:46-67) and posterior-weighted moments
// (gmmhmm.py:95-140) -- on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: twice the fp64 rate), with the weighted
// component log-densities handed from one to the other as FLOAT (half the HBM round trip of the fp64 pass).
// Forward, backward, xi and the M-step stay in fp64: the recursions are where fp32 loses its digits (alpha ~ -1e5
// at T = 2000), and they are not the expensive part.
//
//   emit_diag_mfma_f32   q[f,j,m] = log w + log N(o; mu, var) (float) and b[f,j] = lse_m q (double)
//   mix_mfma_f32         S[(j,m), col] += sum_f exp(alpha + beta - c1 - b + q) * [x - c, (x - c)^2, 1]
//
// Layouts of v_mfma_f32_16x16x4_f32 (wave64): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[i = 4 (lane >> 4) + r][j = lane & 15] for accumulator register r = 0..3 (the fp64 instruction has i = (lane >> 4) + 4 r).
#include "common.h"
#include "devmath.h"

namespace lnasr {

typedef float float4_t __attribute__((ext_vector_type(4)));

// wpack32[((sg*Q + q)*KS + ks)*64 + lane]: coefficient of (state 4 sg + (row >> 2), mixture 4 q + (row & 3)), row = lane & 15,
// for K index 4 ks + (lane >> 4): same rows of the expanded quadratic as pack_wfrag_kernel (gmm.hip), log w folded in,
// means centred -- but the row <-> (state, mixture) map of the fp32 instruction's D layout.
__global__ void pack_wfrag32_kernel(const double* __restrict__ mu_c, const double* __restrict__ ivar,
                                    const double* __restrict__ gconst, const double* __restrict__ logw,
                                    float* __restrict__ wpack, int N, int M, int D, int Q, int KS) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(N / 4) * Q * KS * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t r1 = idx >> 6;
    const int ks = (int)(r1 % KS);
    r1 /= KS;
    const int q = (int)(r1 % Q);
    const int sg = (int)(r1 / Q);
    const int row = lane & 15, k = 4 * ks + (lane >> 4);
    const int state = 4 * sg + (row >> 2), mix = 4 * q + (row & 3);
    const size_t c = (size_t)state * M + mix;
    double val = 0.0;
    if (k < D) {
        val = -0.5 * ivar[c * D + k];
    } else if (k < 2 * D) {
        const int d = k - D;
        val = mu_c[c * D + d] * ivar[c * D + d];
    } else if (k == 2 * D) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            const double m = mu_c[c * D + d];
            s += m * m * ivar[c * D + d];
        }
        val = logw[c] + gconst[c] - 0.5 * s;
    }
    wpack[idx] = (float)val;
}

// Block = 8 waves; wave w owns state group sg = 8 y + w (4 (virtual) states x 4 Q mixtures), coefficient fragments
// resident in registers; the block walks 16-frame tiles whose feature matrix [x'^2 | x' | 1] (x' = x - centre, formed
// in fp64, stored as float) is built once in LDS and read as the B operand by all waves.  XCD-aware block decode as
// in emit_diag_mfma.
// PAIR (M = 32 scored as two virtual 16-mixture states per state): the halves of a state are the virtual states 2 j and 2 j + 1,
// i.e. adjacent lane groups of one wave -- they share their maximum and add their sums through lane ^ 16, and b [F, N / 2] holds
// the real states' values: no merge pass over b afterwards (q32 keeps its layout, which is the same memory either way).
template <int Q, int KS, bool PAIR = false>
__global__ __launch_bounds__(512) void emit_diag_mfma_f32(const double* __restrict__ obs, const float* __restrict__ wpack,
                                                          const double* __restrict__ centre, double* __restrict__ b,
                                                          float* __restrict__ q32, long F, int N, int D, int NG, int GY,
                                                          long ntiles, int tiles_per_block, long nchunks) {
    constexpr int KP = 4 * KS, NT = 512, SPB = 32, FS = 17, OS = SPB + 1;
    __shared__ float Fl[2][KP * FS];
    __shared__ double ob[2][16 * OS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long i = bid >> 3;
    const int y = (int)(i % GY);
    const long chunk = (i / GY) * 8 + xcd;
    if (chunk >= nchunks) return;
    const long tile0 = chunk * tiles_per_block;
    long tile1 = tile0 + tiles_per_block;
    if (tile1 > ntiles) tile1 = ntiles;
    const int sg = 8 * y + w;
    const bool active = sg < NG;
    const int sgc = active ? sg : 0;
    constexpr int M = 4 * Q;
    const int g = lane >> 4, fr = lane & 15;
    const int my_state = 4 * sgc + g;

    float wr[Q][KS];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wr[q][ks] = wpack[(((size_t)sgc * Q + q) * KS + ks) * 64 + lane];

    {   // constant rows of F (k >= 2D): 1 for k == 2D, 0 above
        const int rows = KP - 2 * D;
        for (int e = tid; e < 2 * rows * 16; e += NT) {
            const int bt = e / (rows * 16), r = e % (rows * 16);
            const int k = 2 * D + r / 16;
            Fl[bt][k * FS + (r % 16)] = (k == 2 * D) ? 1.0f : 0.0f;
        }
    }
    const int EL = 16 * D;
    constexpr int MAXLD = 2;                                 // 16 frames x D <= 64 over 512 threads
    const long lim = F * (long)D;
    double pre[MAXLD], cen[MAXLD];
    int foff[MAXLD];
#pragma unroll
    for (int r = 0; r < MAXLD; ++r) {
        const int e = tid + r * NT;
        const int f = e / D, d = e - f * D;
        foff[r] = e < EL ? d * FS + (f & 15) : -1;
        cen[r] = centre[d];
    }
    auto load_stage = [&](long tile) {
        const long base = tile * 16 * (long)D;
#pragma unroll
        for (int r = 0; r < MAXLD; ++r) {
            long ge = base + tid + r * NT;
            ge = ge < lim ? ge : lim - 1;
            pre[r] = obs[ge];
        }
    };
    // (No test for NaN / infinite features here, unlike emit_diag_mfma: the same compare-and-ballot per staged element cost
    // this kernel 0.25 ms of its 1.87 at config 5, three alternations on one box.  Its callers scan the observations -- 40 MB,
    // ~15 us -- instead: nanfix.hip, `force`.)
    auto store_stage = [&](int buf) {
        float* Fd = &Fl[buf][0];
        const int xo = D * FS;
#pragma unroll
        for (int r = 0; r < MAXLD; ++r) {
            if (foff[r] >= 0) {
                const double x = pre[r] - cen[r];
                Fd[foff[r]] = (float)(x * x);
                Fd[foff[r] + xo] = (float)x;
            }
        }
    };
    auto flush_stage = [&](int buf, long tile) {             // 16 frames x 32 states of b
        const int f = tid / SPB, st = tid % SPB;
        const long frame = tile * 16 + f;
        const int state = SPB * y + st;
        if (PAIR) {                                          // the even virtual state's slot holds the real state's value
            if (tile < tile1 && frame < F && state < N && !(st & 1)) b[frame * (N / 2) + state / 2] = ob[buf][f * OS + st];
        } else if (tile < tile1 && frame < F && state < N) {
            b[frame * N + state] = ob[buf][f * OS + st];
        }
    };

    load_stage(tile0);
    store_stage(0);
    drain_vmem();
    __syncthreads();
    int cur = 0;
    for (long tile = tile0; tile < tile1; ++tile, cur ^= 1) {
        const bool has_next = tile + 1 < tile1;
        if (has_next) load_stage(tile + 1);
        if (tile > tile0) flush_stage(cur ^ 1, tile - 1);
        float4_t acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = float4_t{0.f, 0.f, 0.f, 0.f};
        const float* Fc = &Fl[cur][0];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float bf = Fc[(4 * ks + g) * FS + fr];
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q][ks], bf, acc[q], 0, 0, 0);
        }
        // lane (g, fr): acc[q][r] = mixture 4 q + r of state 4 sg + g at frame fr
        const long frame = tile * 16 + fr;
        if (active && frame < F) {
            float4_t* dst = reinterpret_cast<float4_t*>(q32 + ((size_t)frame * N + my_state) * M);
#pragma unroll
            for (int q = 0; q < Q; ++q) dst[q] = acc[q];
        }
        float mx = acc[0][0];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (q + r > 0) mx = fmaxf(mx, acc[q][r]);
        if (PAIR) mx = fmaxf(mx, __shfl_xor(mx, 16));
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += __expf(acc[q][r] - mx);
        if (PAIR) s += __shfl_xor(s, 16);
        // mx = -inf: the differences are NaN, s is NaN; the state's value is -inf
        ob[cur][fr * OS + 4 * w + g] = (mx == -__builtin_huge_valf()) ? (double)mx : (double)mx + (double)__logf(s);
        if (has_next) store_stage(cur ^ 1);
        __syncthreads();
    }
    if (tile1 > tile0) flush_stage((int)((tile1 - tile0 - 1) & 1), tile1 - 1);
}

// b[cell] = lse_h bv[cell * k + h]   (M = 16 k mixtures scored as k virtual 16-mixture states)
__global__ void merge_virtual32_kernel(const double* __restrict__ bv, size_t cells, int k, double* __restrict__ b) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    Lse acc;
    for (int h = 0; h < k; ++h) acc.add(bv[idx * k + h]);
    b[idx] = acc.value();
}

// Mixture moments, fp32 matrix cores.  Same blocking as mix_mfma_kernel (bwstats.hip): block = 8 waves = 8 states over
// a contiguous frame range, the wave of state j turns q[f,j,:] into posteriors (A operand: mixture = lane & 15, frame
// = lane >> 4 of the k-step) and multiplies them with the block's feature rows from LDS.  Accumulators are float and
// are added into the fp64 sums R every kFlush frames, so no float sum runs over more than kFlush terms.
constexpr int kMix32Frames = 32, kMix32Waves = 8, kMix32NT = 64 * kMix32Waves, kMix32Flush = 4096;
constexpr int kMix32Ring = 8;        // q prefetch ring (k-steps; must divide kMix32Frames / 4)
// unclamped running pointer, as in bwstats.hip: the slack behind q32 must cover chunk rounding + prefetch depth
static_assert((kMix32Frames - 1) + 4 * (kMix32Ring - 1) + 3 < kMixPadFrames, "q32 prefetch ring outruns kMixPadFrames");
static_assert((kMix32Frames / 4) % kMix32Ring == 0, "kMix32Ring must divide kMix32Frames / 4");

template <int MT, int CT>
__global__ __launch_bounds__(kMix32NT) void mix_mfma_f32(const double* __restrict__ obs, const double* __restrict__ alpha,
                                                         const double* __restrict__ beta, const double* __restrict__ b,
                                                         const float* __restrict__ q32, const double* __restrict__ c1,
                                                         const double* __restrict__ centre,
                                                         const int32_t* __restrict__ lens, long F, long FPB, int T, int N,
                                                         int M, int D, double* __restrict__ R) {
    constexpr int CH = kMix32Frames, NFP = 16 * CT, FST = NFP + 1, MP = 16 * MT;
    constexpr int OPT = (CH * 39 + kMix32NT - 1) / kMix32NT;
    __shared__ float feat[2][CH * FST];
    __shared__ float sab[2][CH * kMix32Waves];
    __shared__ double cen[40];
    LN_BLOCK_T0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int j = blockIdx.x * kMix32Waves + w;
    const bool active = j < N;
    const int jc = active ? j : N - 1;
    const long fb0 = (long)blockIdx.y * FPB;
    long fb1 = fb0 + FPB;
    if (fb1 > F) fb1 = F;
    if (fb0 >= fb1) return;
    if (tid < D) cen[tid] = centre[tid];
    for (int e = tid; e < 2 * CH * (FST - 2 * D); e += kMix32NT) {
        const int bufi = e / (CH * (FST - 2 * D)), r = e % (CH * (FST - 2 * D));
        const int fr = r / (FST - 2 * D), col = 2 * D + r % (FST - 2 * D);
        feat[bufi][fr * FST + col] = (col == 2 * D) ? 1.0f : 0.0f;
    }
    double oreg[OPT];
    float s_ab = 0.f;
    auto stage_load = [&](long f0) {
        const long nfr = (fb1 - f0) < CH ? (fb1 - f0) : CH;
#pragma unroll
        for (int r = 0; r < OPT; ++r) {
            const int e = tid + r * kMix32NT;
            oreg[r] = e < nfr * D ? obs[(size_t)f0 * D + e] : 0.0;
        }
        if (tid < CH * kMix32Waves) {
            const int fr = tid / kMix32Waves, js = blockIdx.x * kMix32Waves + (tid % kMix32Waves);
            const long f = f0 + fr;
            s_ab = -__builtin_huge_valf();
            if (f < fb1 && js < N) {
                const int u = (int)(f / T), t = (int)(f % T);
                if (t < lens[u]) {
                    const size_t fj = (size_t)f * N + js;
                    s_ab = (float)((alpha[fj] + beta[fj] - c1[f]) - b[fj]);     // log state posterior - b, in fp64, then one rounding
                }
            }
        }
    };
    auto stage_store = [&](int bufi) {
#pragma unroll
        for (int r = 0; r < OPT; ++r) {
            const int e = tid + r * kMix32NT;
            if (e < CH * D) {
                const int fr = e / D, d = e % D;
                const float x = (float)(oreg[r] - cen[d]), xx = (float)((oreg[r] - cen[d]) * (oreg[r] - cen[d]));
                const bool fin = __builtin_isfinite(xx);        // padding frames: see mix_mfma_kernel (bwstats.hip)
                feat[bufi][fr * FST + d] = fin ? x : 0.0f;
                feat[bufi][fr * FST + D + d] = fin ? xx : 0.0f;
            }
        }
        if (tid < CH * kMix32Waves) sab[bufi][tid] = s_ab;
    };
    float4_t acc[MT][CT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = float4_t{0.f, 0.f, 0.f, 0.f};
    };
    auto flush_acc = [&]() {
        if (!active) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * mt + 4 * kk + r;                   // D layout of the fp32 instruction
                    const float v = acc[mt][ct][r];
                    if (v != 0.f) atomicAdd(&R[((size_t)j * MP + m) * NFP + 16 * ct + l15], (double)v);
                }
    };
    zero_acc();
    // q of the lane's (mixture, frame of the k-step): a running pointer, one 64-bit add per load.  The ring runs up to
    // 4 RM frames past the block's range; q32 is allocated with that much slack (kMixPadFrames, common.h), and what
    // is read there is never used (frames past the range carry a log posterior of -inf).
    const size_t qstride = (size_t)N * M;
    bool qok[MT];
    const float* qptr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        qok[mt] = 16 * mt + l15 < M;
        qptr[mt] = q32 + ((size_t)(fb0 + kk) * N + jc) * M + (qok[mt] ? 16 * mt + l15 : 0);
    }
    constexpr int RM = kMix32Ring;
    float ring[RM][MT];
    auto ring_load = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ring[slot][mt] = *qptr[mt];
            qptr[mt] += 4 * qstride;
        }
    };
    __syncthreads();
    stage_load(fb0);
    stage_store(0);
#pragma unroll
    for (int s = 0; s < RM - 1; ++s) ring_load(s);
    __syncthreads();
    int cur = 0;
    long since = 0;
    for (long f0 = fb0; f0 < fb1; f0 += CH, cur ^= 1) {
        const bool has_next = f0 + CH < fb1;
        if (has_next) stage_load(f0 + CH);
        if (active) {
#pragma unroll
            for (int ks = 0; ks < CH / 4; ++ks) {
                ring_load((ks + RM - 1) % RM);
                const int fk = 4 * ks + kk;
                const float ab = sab[cur][fk * kMix32Waves + w];
                float gq[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float z = ab + ring[ks % RM][mt];
                    const float ev = __expf(z);
                    gq[mt] = (qok[mt] && z >= -80.f) ? ev : 0.f;           // padding lanes, -inf, NaN -> 0
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float fv = feat[cur][fk * FST + 16 * ct + l15];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(gq[mt], fv, acc[mt][ct], 0, 0, 0);
                }
            }
        }
        since += CH;
        if (since >= kMix32Flush) {
            flush_acc();
            zero_acc();
            since = 0;
        }
        if (has_next) stage_store(cur ^ 1);
        __syncthreads();
    }
    flush_acc();
    LN_BLOCK_T1;
}
LN_BLOCK_TIMES_EXPORT(lnasr_dbg_block_times_f32stats)

// bwstats.hip
int mix_centre_launch(lnasr_ctx* ctx, double* centre);
int mix_finish_launch(lnasr_ctx* ctx, const double* R, const double* centre, int MP, int NFP, double* mix, double* mix_x,
                      double* mix_xx);

// Can this model take the fp32 matrix-core pass?  Diagonal, well conditioned, shapes the bank kernel takes without
// padding (N a multiple of 4; M = 4, 8, 16 or 16 k up to 32), D <= 39.  Anything else runs the fp64 pass: more
// precise than asked for, never less.
bool f32_stats_ok(const Model& md) {
    if (!md.diag || md.direct_form || md.wpack_Q == 0 || md.pad_N != 0) return false;
    // the float expanded quadratic loses ~6e-8 * kappa: its own, tighter gate (common.h); LNASR_KAPPA_MAX below the
    // default tightens it further, never loosens it
    if (!(md.kappa <= (sw().kappa_max < kKappaMaxF32 ? sw().kappa_max : kKappaMaxF32))) return false;
    if (md.D > 39 || md.M > 32 || md.wpack_KS > 20) return false;
    return !sw().no_mfma && !sw().no_mfma_stats && !sw().no_f32_stats;
}

static int pack32(lnasr_ctx* ctx) {
    Model& md = ctx->model;
    const int Q = md.wpack_Q, KS = md.wpack_KS, virt = md.wpack_virt;
    const int Nv = md.N * virt, Mv = md.M / virt;
    const size_t total = (size_t)(Nv / 4) * Q * KS * 64;
    if (md.wpack32.p && md.wpack32_for == md.upload_serial) return 0;
    LN_TRY(md.wpack32.reserve(total * 4));
    LN_LAUNCH(ctx, "pack_wfrag32", pack_wfrag32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
              (const double*)md.mu_c.as<double>(), (const double*)md.ivar.as<double>(), (const double*)md.gconst.as<double>(),
              (const double*)md.logw.as<double>(), md.wpack32.as<float>(), Nv, Mv, md.D, Q, KS);
    md.wpack32_for = md.upload_serial;
    return 0;
}

template <int Q, int KS>
static int launch_emit32(lnasr_ctx* ctx, const double* obs, long F, double* bv, float* q32) {
    Model& md = ctx->model;
    const int Nv = md.wpack_NG * 4, D = md.D;
    const int GY = (Nv + 31) / 32;
    const long ntiles = (F + 15) / 16;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    // The kernel needs 138 registers (three waves per SIMD, a block brings two: one resident block per CU); forced
    // down to 128 for two resident blocks it spills four W fragments into the MFMA chain (1.85 -> 2.05 ms at config
    // 5).  Twice as many blocks as CUs, each walking half the tiles, measured better than one block per CU (1.85 vs
    // 2.07 ms): the second round evens out the tail.
    long want_chunks = ((long)cus * 2) / GY;
    if (want_chunks < 8) want_chunks = 8;
    want_chunks = (want_chunks / 8) * 8;
    long tpb = (ntiles + want_chunks - 1) / want_chunks;
    if (tpb < 1) tpb = 1;
    const long nchunks = (ntiles + tpb - 1) / tpb;
    const long grid = ((nchunks + 7) / 8) * 8 * GY;
    if (Q == 4 && md.wpack_virt == 2)      // bv is b itself here (emit_device_f32): the kernel joins the halves
        LN_LAUNCH(ctx, "emit_diag_mfma_f32", (emit_diag_mfma_f32<Q, KS, true>), dim3((unsigned)grid), dim3(512), 0, obs,
                  (const float*)md.wpack32.as<float>(), (const double*)md.centre.as<double>(), bv, q32, F, Nv, D, md.wpack_NG,
                  GY, ntiles, (int)tpb, nchunks);
    else
        LN_LAUNCH(ctx, "emit_diag_mfma_f32", (emit_diag_mfma_f32<Q, KS>), dim3((unsigned)grid), dim3(512), 0, obs,
                  (const float*)md.wpack32.as<float>(), (const double*)md.centre.as<double>(), bv, q32, F, Nv, D, md.wpack_NG,
                  GY, ntiles, (int)tpb, nchunks);
    return 0;
}

// emission of the fp32 pass: b [F, N] double, q32 [F, N, M] float (weighted component log-densities)
int emit_device_f32(lnasr_ctx* ctx, const double* obs, long F, double* b, float* q32) {
    LN_DEVPTR(obs); LN_DEVPTR(b); LN_DEVPTR(q32);
    Model& md = ctx->model;
    const int Q = md.wpack_Q, KS = md.wpack_KS, virt = md.wpack_virt;
    // batches the integer matrix-core bank takes (gmm_i8.hip): its four-plane instantiation hands over the same b + q32
    // (four planes keep 32 bits below the scales: ~3 * 2^-30 * rho on a component's log-density -- its own, tighter range gate)
    if (((md.i8_mix == 16 && virt == 1) || md.i8_mix == 32) && i8_route(md, F) &&
        md.i8_rho <= (sw().i8_rho_max_set ? sw().i8_rho_max : kI8RhoMaxF32))
        return emit_diag_i8_device(ctx, obs, F, b, nullptr, q32);
    LN_TRY(pack32(ctx));
    double* bv = b;
    const bool joined = virt == 2 && Q == 4;              // the kernel joins the two halves of a state itself (PAIR)
    if (virt > 1 && !joined) {
        LN_TRY(ctx->ws_misc2.reserve((size_t)F * md.N * virt * 8));
        bv = ctx->ws_misc2.as<double>();
    }
    int rc = LNASR_E_UNSUPPORTED;
#define LN_CASE(q, ks) \
    if (Q == q && KS == ks) rc = launch_emit32<q, ks>(ctx, obs, F, bv, q32);
    LN_CASE(4, 20) LN_CASE(2, 20) LN_CASE(1, 20) LN_CASE(4, 14) LN_CASE(2, 14) LN_CASE(1, 14) LN_CASE(4, 7) LN_CASE(2, 7)
    LN_CASE(1, 7)
#undef LN_CASE
    LN_TRY(rc);
    if (virt > 1 && !joined) {
        const size_t cells = (size_t)F * md.N;
        LN_LAUNCH(ctx, "merge_virtual_states", merge_virtual32_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0,
                  (const double*)bv, cells, virt, b);
    }
    return 0;
}

int mix_stats_device_f32(lnasr_ctx* ctx, const double* obs, const double* alpha, const double* beta, const double* b,
                         const float* q32, const double* c1, const int32_t* d_lens, int U, int T, double* mix,
                         double* mix_x, double* mix_xx, int phase) {
    LN_DEVPTR(obs); LN_DEVPTR(alpha); LN_DEVPTR(beta); LN_DEVPTR(b); LN_DEVPTR(q32); LN_DEVPTR(c1); LN_DEVPTR(d_lens); LN_DEVPTR(mix); LN_DEVPTR(mix_x); LN_DEVPTR(mix_xx);
    Model& md = ctx->model;
    const int N = md.N, M = md.M, D = md.D;
    const int MT = (M + 15) / 16, CT = (2 * D + 1 + 15) / 16;
    const int MP = 16 * MT, NFP = 16 * CT;
    const long F = (long)U * T;
    const size_t nraw = (size_t)N * MP * NFP;
    LN_TRY(ctx->ws_raw.reserve((nraw + 64) * 8));
    double* R = ctx->ws_raw.as<double>();
    double* centre = R + nraw;
    if (phase & kMixBegin) {
        LN_HIP(hipMemsetAsync(R, 0, nraw * 8, ctx->stream));
        LN_TRY(mix_centre_launch(ctx, centre));
    }
    if (phase & kMixAdd) {
        const int SG = (N + kMix32Waves - 1) / kMix32Waves;
        const long nchunk = (F + kMix32Frames - 1) / kMix32Frames;
        long G = (2L * ctx->prop.multiProcessorCount + SG - 1) / SG;
        if (G > nchunk) G = nchunk;
        if (G < 1) G = 1;
        if (G > 65535) G = 65535;
        const long FPB = ((nchunk + G - 1) / G) * kMix32Frames;
        G = (F + FPB - 1) / FPB;
        const dim3 grid(SG, (unsigned)G);
#define LN_MIX32(mt, ct)                                                                                             \
    if (MT == mt && CT == ct) {                                                                                      \
        LN_LAUNCH(ctx, "mix_mfma_f32", (mix_mfma_f32<mt, ct>), grid, dim3(kMix32NT), 0, obs, alpha, beta, b, q32, c1,  \
                  (const double*)centre, d_lens, F, FPB, T, N, M, D, R);                                              \
    }
        LN_MIX32(1, 1) LN_MIX32(1, 2) LN_MIX32(1, 3) LN_MIX32(1, 4) LN_MIX32(1, 5)
        LN_MIX32(2, 1) LN_MIX32(2, 2) LN_MIX32(2, 3) LN_MIX32(2, 4) LN_MIX32(2, 5)
#undef LN_MIX32
    }
    if (phase & kMixEnd) return mix_finish_launch(ctx, R, centre, MP, NFP, mix, mix_x, mix_xx);
    return 0;
}

}  // namespace lnasr
