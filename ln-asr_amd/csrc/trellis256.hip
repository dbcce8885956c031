// 128 < N <= 256: Viterbi / forward / backward with the transition matrix held ON CHIP by one 1024-thread block
// per utterance (BASELINE configs[4]: N = 256).
#include "common.h"
#include "devmath.h"
#include "trellis.h"

namespace lnasr {

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
constexpr int kT256Lb = kT256Ss + 2;                                     // [16] floats (8 doubles): emission-row level partials
constexpr int kT256Et = kT256Lb + 8;                                     // [kExpTab]
constexpr int kT256Lt = kT256Et + kExpTab;                               // [2 * kLogTab]
constexpr int kT256End = kT256Lt + 2 * kLogTab;

// One role of the on-chip kernel.  FIN = true: the four waves of row slice 0, which also finalise a frame
// (combine the slices, log / exp, emission, store) and therefore keep fewer rows of A in registers;
// FIN = false: the twelve waves of slices 1..3, which only multiply.  The two roles are separate
// instantiations so that each gets its own register allocation; both execute the same barriers.
// TS = float: `precision=np.float32` -- b and v / alpha / beta are float arrays, every sum is formed in fp64 and
// rounded to float where the reference's float32 array rounds it, i.e. at the store of v[t, j] / alpha[t, j] /
// beta[t, i] (hmm.py:160-165, 93-99, 124-130); the next frame continues from the rounded value.
//
// (A variant that took log / exp off the frame's critical path -- e <- sum * F with F formed during the multiply
// phase -- was built and measured slower in round 2, DESIGN 4.5, and removed in round 3.)
template <int MODE, bool FIN, typename TS = double>
__device__ __forceinline__ void trellis256_role(double* sm, const TS* __restrict__ bu, const double* __restrict__ Amat,
                                                const double* __restrict__ Alog, const double* __restrict__ pi,
                                                TS* __restrict__ ou, double* __restrict__ loglik_u, int L, int T, int N,
                                                unsigned* __restrict__ fb_count) {
    constexpr bool VIT = MODE == 0, BACKWARD = MODE == 2;
    constexpr int NP = 256, RPS = 64;
    constexpr int NREG = FIN ? kT256FinReg : kT256CmpReg, NLDS = 64 - NREG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int jc = tid & (NP - 1), rs = tid >> 8;
    // the four finalising waves (one per SIMD) are the frame's critical path: they multiply AND finalise while the other twelve
    // wait at the barrier -- favoured at the arbiter (round 4: forward 2.764 -> 2.732 ms, backward 2.797 -> 2.781 at config 5;
    // small, because the frame is a dependent chain rather than an issue contest: profiles/r04c_prio_experiment.log)
    if (FIN) __builtin_amdgcn_s_setprio(1);
    double* Al = FIN ? sm + kT256Al0 + jc : sm + kT256Al1 + (rs - 1) * NP + jc;
    constexpr int ALS = FIN ? NP : 3 * NP;                               // stride between LDS rows of A
    double* Vs = sm + kT256Vs;
    double* Cs = sm + kT256Cs;
    double* Ps = sm + kT256Ps;
    double* Ms = sm + kT256Ms;
    double* Ss = sm + kT256Ss;
    float* Lb = reinterpret_cast<float*>(sm + kT256Lb);
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
        if (FIN && !VIT) {
            // level of this frame's emission row (max_j b[t, j], float accuracy) for the shift below: the lagged maximum
            // alone assumes that consecutive frames' emissions have about the same size, and on data a model does not fit --
            // row maxima jumping by hundreds of nats -- every e row under- or overflowed and every column took the exact
            // fallback (fwdbwd.hip, round 5).  Partials of the sixteen DPP rows go through the LDS with the first barrier.
            const float lv = row_max_f32(jc < N ? (float)bcur : -__builtin_huge_valf());
            if ((lane & 15) == 0) Lb[tid >> 4] = lv;
        }
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
                if (jc >= N || (sum >= kSafeSum && sum <= 1e290 && Sc != LN_NEG_INF)) {   // (a padding column: nothing to redo)
                    val = Sc + log_sum(sum, ltab);
                } else {
                    val = exact_column_lse(Cs + cur * NP, Alog + jcl, N);
                    atomicAdd(fb_count, 1u);       // diagnostic (lnasr_debug_fallback_columns)
                }
                double prev = LN_NEG_INF;
                if (jc < N) {
                    if (BACKWARD) {
                        val = rnd<TS>(val);
                        if (ou) ou[(size_t)t * N + jc] = (TS)val;
                        prev = val + bcur;
                    } else {
                        prev = rnd<TS>(val + bcur);
                        if (ou) ou[(size_t)t * N + jc] = (TS)prev;
                    }
                }
                double S = fmax(fmax(Ms[cur * 4], Ms[cur * 4 + 1]), fmax(Ms[cur * 4 + 2], Ms[cur * 4 + 3]));
                {
                    float lvl = Lb[0];
#pragma unroll
                    for (int e = 1; e < 16; ++e) lvl = fmaxf(lvl, Lb[e]);
                    S += (double)lvl;              // the lagged maximum, carried to this frame's emission level
                }
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
                                                          double* __restrict__ loglik, int U, int T, int N,
                                                          unsigned* __restrict__ fb_count) {
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
        trellis256_role<MODE, true, TS>(sm, bu, Amat, Alog, pi, ou, loglik ? loglik + u : nullptr, L, T, N, fb_count);
    else
        trellis256_role<MODE, false, TS>(sm, bu, Amat, Alog, pi, ou, nullptr, L, T, N, fb_count);
}

static inline size_t trellis256_lds() { return (size_t)kT256End * 8; }


int trellis256_device(lnasr_ctx* ctx, int mode, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* out,
                      double* loglik) {
    Model& md = ctx->model;
    const int N = md.N;
    const dim3 grid(U), block(1024);
    const size_t lds = trellis256_lds();
    const double* pi = md.logpi.as<double>();
    if (mode == 0) {
        const double* A = md.logA.as<double>();
        if (dtype == LNASR_F32)      // precision=np.float32: float storage, v rounded where the reference's array rounds it
            LN_LAUNCH(ctx, "viterbi_onchip256_f32", (trellis256_kernel<0, float>), grid, block, lds, (const float*)b, A, A, pi,
                      d_lens, (float*)out, (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
        else
            LN_LAUNCH(ctx, "viterbi_onchip256", trellis256_kernel<0>, grid, block, lds, (const double*)b, A, A, pi, d_lens,
                      (double*)out, (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
        return 0;
    }
    if (mode == 1 && dtype == LNASR_F32)
        LN_LAUNCH(ctx, "forward_onchip256_f32", (trellis256_kernel<1, float>), grid, block, lds, (const float*)b,
                  md.expA.as<double>(), md.logA.as<double>(), pi, d_lens, (float*)out, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
    else if (mode == 1)
        LN_LAUNCH(ctx, "forward_onchip256", (trellis256_kernel<1, double>), grid, block, lds, (const double*)b,
                  md.expA.as<double>(), md.logA.as<double>(), pi, d_lens, (double*)out, loglik, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
    else if (dtype == LNASR_F32)
        LN_LAUNCH(ctx, "backward_onchip256_f32", (trellis256_kernel<2, float>), grid, block, lds, (const float*)b,
                  md.expAT.as<double>(), md.logAT.as<double>(), pi, d_lens, (float*)out, (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
    else
        LN_LAUNCH(ctx, "backward_onchip256", (trellis256_kernel<2, double>), grid, block, lds, (const double*)b,
                  md.expAT.as<double>(), md.logAT.as<double>(), pi, d_lens, (double*)out, (double*)nullptr, U, T, N, ctx->nan_flag.as<unsigned>() + 3);
    return 0;
}

}  // namespace lnasr
