// Internal interface between the trellis translation units (not part of the ABI):
//   viterbi.hip     Viterbi kernels for N <= 128 + viterbi_device (dispatch over all families)
//   backtrace.hip   back-trace kernels, the full back-pointer matrix + backtrace_device / bt_matrix_device
//   fwdbwd.hip      forward / backward kernels for N <= 128 + forward_device / backward_device
//   trellis256.hip  128 < N <= 256: transition matrix on chip, one block per utterance
//   bign.hip        N > 128 streaming kernels, N > 256 frame-parallel kernels
//   trellis.hip     discrete emission, Baum-Welch posteriors, device_lens, the C-ABI entry points
//
// and THE table of batch-size / model-size dispatch thresholds, each with the measurement that set it.
#pragma once
#include "common.h"

namespace lnasr {

// forward / backward in the scaled linear domain: a column sum below this is treated as an all -inf column
constexpr double kSafeSum = 1e-290;

// ---- dispatch table -------------------------------------------------------------------------------------------------
// Every threshold below was set by measurement on one MI355X (gfx950, 256 CUs, ROCm 7.2); the tool that reproduces
// the comparison and the round it was last run in are named.  `tools/tier_sweep.py` checks that every tier on either
// side of every threshold gives the results of the tier the oracle tests pin (round 3, 2026-10-04: green).
namespace dispatch {

// Model size -> kernel family (structural limits, not tuning: what fits where)
constexpr int kTinyMaxN = 16;        // one wave per utterance, A column in registers (viterbi_tiny; scripts/seg/seg.py n = 4)
constexpr int kReg64MaxN = 64;       // A / exp(A) in the registers of a two-wave block (viterbi_reg, fwdbwd_reg <1,2,32,..>)
constexpr int kRegMaxN = 128;        // ... of an eight-wave block (viterbi_reg1b) / a four-wave block (fwdbwd_cols2)
constexpr int kOnChipMaxN = 256;     // A = the register file + 136 KB LDS of ONE CU (trellis256.hip)
constexpr int kStreamMaxN = 1024;    // streaming blocks: NP * RS <= 1024 threads, (4 + RS) * UB * NP doubles of LDS
constexpr size_t kStreamLdsMax = 160 * 1024;     // dynamic LDS a streaming block may ask for

// Viterbi, 64 < N <= 128: one utterance per block below this batch size, two per block (with alternating
// wave priority) from it on.  tools/qbench.py + tools/block_times.py, round 2: at U >= 512 every CU
// holds two blocks either way and sharing the A registers wins (1.37 -> 1.25 ms at U = 1024); below, one utterance
// per block spreads the chains over more CUs.
constexpr int kViterbiPairFromU = 512;

// Forward / backward, 64 < N <= 128, eight-wave fallback kernels (LNASR_NO_COLFWD): four utterances per block once
// the batch holds four per CU.  Round 1 (profiles/r01k_rows.log).
inline bool fwdbwd_reg_four_per_block(int U, int n_cu) { return U >= 4 * n_cu; }

// Forward / backward, 64 < N <= 128, column-owned kernels: ONE utterance per four-wave block (fwdbwd_cols2<.., 1>) up to two
// blocks per CU, two utterances per block beyond.  tools/bench_rows.py fwdsweep, round 5 (profiles/r05e_fwdsweep.log), ms for
// T = 1000, forward, two per block / one per block: U=1 0.96 / 0.75, 64 0.88 / 0.77, 256 0.91 / 0.78, 512 0.95 / 0.89,
// 768 1.19 / 1.65, 1024 1.33 / 1.87 (backward alike).  LNASR_FWD_ONE_MAXU overrides (tests, sweeps).
inline bool fwdbwd_one_per_block(int U, int n_cu) {
    if (sw().fwd_one_maxu >= 0) return U <= sw().fwd_one_maxu;
    return U <= 2 * (n_cu > 0 ? n_cu : 256);
}

// ... and, fp64 storage, the kernel with the log / exp off the frame's chain (fwdbwd_lat4) while every block has a CU of its
// own: only there does the LENGTH of the chain, rather than the instruction count, set the frame time.  forward, U = 1,
// T = 1000: 0.757 ms (cols2<.., 1>) -> 0.621 ms.  LNASR_FWD_LAT_MAXU overrides; LNASR_NO_LATFWD disables.  Round 5.
inline bool fwdbwd_latency_tier(int U, int n_cu) {
    if (sw().fwd_lat_maxu >= 0) return U <= sw().fwd_lat_maxu;
    return U <= (n_cu > 0 ? n_cu : 256);
}

// N > 256: frame-parallel kernels (ONE launch per frame over the whole batch) against per-utterance streaming
// kernels.  tools/bign.py, T = 100, ms, frame-parallel / streaming, round 2:
//   Viterbi  N=320: U=64 1.2/2.1  U=512 2.2/2.2  U=1024 3.0/2.4    N=512: 1.1/4.5  3.0/4.6  4.2/5.1
//            N=1024: 1.8/32  6.5/29  11.6/29
//   forward  N=320: 1.8/1.5  2.2/2.0  2.8/2.2    N=512: 2.6/3.1  3.4/4.3  5.0/4.4    N=1024: 4.7/24  10.8/25  17/50
// With few utterances the frame-parallel kernels win 3-16x; with ~1000 the streaming kernels already fill the chip.
// LNASR_NO_FRAMEPAR / LNASR_FRAMEPAR_MAXU (test switches) override the rule.
inline bool framepar_env(int U, bool* use) {
    if (sw().no_framepar) { *use = false; return true; }
    if (sw().framepar_maxu >= 0) { *use = U <= sw().framepar_maxu; return true; }
    return false;
}
inline bool viterbi_framepar(int N, int U) {
    bool use;
    if (framepar_env(U, &use)) return use;
    return N > 384 || U <= 512;
}
inline bool fwdbwd_framepar(int N, int U) {
    bool use;
    if (framepar_env(U, &use)) return use;
    return N > 768 || (N > 384 && U <= 512);
}
// frame-parallel Viterbi: wide column tiles (4 columns per thread) once the batch alone fills the chip.  tools/bign.py, round 2.
inline int viterbi_frame_cpt(int U) { return U > 128 ? 4 : 2; }

// Streaming forward / backward (128 < N <= 1024 when the on-chip / frame-parallel kernels do not apply): utterances
// per block.  Few utterances: one per block keeps more CUs busy (the frame chain is the critical path); many: four per
// block share every read of exp(A); N > 960: four do not fit the LDS, two do.  tools/bign.py, round 2.
inline int fwdbwd_stream_ub(int N, int U, int n_cu) {
    const int NP = ((N + 63) / 64) * 64, RS = 1024 / NP, NB = NP / 64;
    if (U < 2 * n_cu) return 1;
    if (((size_t)(4 + RS) * 4 * NP + 8 * NB + 8) * 8 > kStreamLdsMax) return 2;
    return 4;
}

// Back-trace: waves (= utterances) per block of the LDS kernels: four (round 1) while U <= 4 x CUs, i.e. while one round of
// blocks -- one per CU, A^T fills its LDS -- covers the batch; beyond that as many as keep it to one round, at most sixteen
// (round 4, bench.py --utts 8192: the latency-bound chains of a SIMD's four waves interleave for free).
inline int backtrace_waves_per_block(int U, int n_cu) {
    if (n_cu <= 0) n_cu = 256;
    const int w = (U + n_cu - 1) / n_cu;
    return w < 4 ? 4 : (w > 16 ? 16 : w);
}

// Back-trace of a handful of utterances: every back-pointer first (all frames in parallel, bytes), then a chain of LDS byte
// reads (backtrace_chase_kernel) instead of an arg-max per frame.  The N^2 work per frame pays while the chip is idle.
// tools/bench_rows.py btsweep, round 5 (profiles/r05g_btsweep.log), decode at N = 128, T = 1000, ms, arg-max
// back-trace / chase: U=1 0.80 / 0.68, 4 0.83 / 0.71, 8 0.87 / 0.78, 16 0.93 / 0.88, 32 1.04 / 1.10, 64 1.24 / 1.52.
// LNASR_CHASE_MAXU overrides (tests, sweeps).
inline bool backtrace_chase(int U) {
    if (sw().chase_maxu >= 0) return U <= sw().chase_maxu;
    return U <= 16;
}

}  // namespace dispatch

// ---- launchers -------------------------------------------------------------------------------------------------------
// b, v, alpha, beta [U,T,N] and d_lens [U] are DEVICE pointers (LN_DEVPTR checks them under LNASR_CHECK_PTRS=1).
int device_lens(lnasr_ctx* ctx, const int32_t* lens, int U, int T, const int32_t** out);
// b_tested (optional): ask the kernel to look for NaN emissions itself; *b_tested = it did (word 1 of ctx->nan_flag raised
// with the call's epoch when it met one) -- then the back-trace need not treat -inf cells of v as suspects
int viterbi_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* v,
                   bool* b_tested = nullptr);
// want_suspect: the kernel looks at every cell of v it reads and raises the context's "suspect" word on NaN / -inf (nanfix.hip;
// *v_tested = it did).  false when nobody will read the word: the emissions were tested upstream (b_tested, or the features
// by the emission kernel in lnasr_decode) -- a -inf cell is then just a forbidden state, and the kernel tests nothing
int backtrace_device(lnasr_ctx* ctx, const void* v, int U, int T, const int32_t* d_lens, int dtype, uint32_t* path,
                     double* best, bool* v_tested = nullptr, bool want_suspect = true);
int bt_matrix_device(lnasr_ctx* ctx, const void* v, int U, int T, const int32_t* d_lens, int dtype, uint32_t* bt);
int forward_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* alpha,
                   double* loglik);
int backward_device(lnasr_ctx* ctx, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* beta);
int baumwelch_device(lnasr_ctx* ctx, const void* b, const void* alpha, const void* beta, int U, int T,
                     const int32_t* d_lens, int dtype, void* xi, void* gamma);

// trellis256.hip.  mode 0: Viterbi (out = v), 1: forward (out = alpha or null, loglik or null), 2: backward (out = beta)
int trellis256_device(lnasr_ctx* ctx, int mode, const void* b, int U, int T, const int32_t* d_lens, int dtype, void* out,
                      double* loglik);
// bign.hip (fp64 only)
int viterbi_frames_device(lnasr_ctx* ctx, const double* b, int U, int T, const int32_t* d_lens, double* v);
int viterbi_stream_device(lnasr_ctx* ctx, const double* b, int U, int T, const int32_t* d_lens, double* v);
int fwdbwd_frames_device(lnasr_ctx* ctx, bool backward, const double* b, int U, int T, const int32_t* d_lens, double* out,
                         double* loglik);
int fwdbwd_stream_device(lnasr_ctx* ctx, bool backward, const double* b, int U, int T, const int32_t* d_lens, double* out,
                         double* loglik);

}  // namespace lnasr
