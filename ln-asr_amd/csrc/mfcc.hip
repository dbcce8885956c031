// MFCC front end (mfcc.py:108-175, utils.py:106-138): pre-emphasis -> framing -> Hamming ->
// rFFT -> power -> mel filter bank -> 20 log10 -> mean norm -> DCT-II (ortho) -> cepstral mean
// norm -> log-energy -> delta / delta-delta.  fp64 throughout, utterances batched by offsets.
//
// Kernels
//   mfcc_frame512x4_kernel  fft_n = 512: four frames per wave, samples -> power [257], DCT of the log-mel row, log-energy
//   mfcc_frame_kernel       any other fft_n: one block per frame (radix-2), samples -> power, log-mel, log-energy
//   dct_kernel              (generic path) per frame: DCT-II of the log-mel row
//   col_sum_kernel          per-utterance column sums (both of the reference's utterance-level means)
//   feat_kernel             per frame: the mean subtractions (the second one applied to the returned cepstrum as
//                           well, as the reference does through a view, mfcc.py:155-156), energy, delta, delta-delta
#include <algorithm>

#include "common.h"
#include "devmath.h"

namespace lnasr {

constexpr int kMaxFft = 4096;

// Developer diagnostic (python ln-asr_amd/build.py is never run with it): -DLNASR_MFCC_STAMPS makes lane 0 of wave 0 of
// block 0 add the shader cycles (s_memtime) each phase of the fft-512 frame kernel took into g_mfcc_stamps, which
// tools/mfcc_stamps.py reads back.  The stamps drain the LDS counter at every phase boundary and cost ~10 %.
#ifdef LNASR_MFCC_STAMPS
__device__ unsigned long long g_mfcc_stamps[16];
__device__ unsigned long long g_mfcc_wave_times[2 * 4096];     // per wave: first and last 100 MHz tick (last launch)
#define LN_STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime()
#define LN_STAMP(i)                                                         \
    do {                                                                    \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime();     \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_mfcc_stamps[i] += st_now - st_prev; \
        st_prev = st_now;                                                   \
    } while (0)
#else
#define LN_STAMP_DECL
#define LN_STAMP(i)
#endif

__device__ __forceinline__ int find_utt(const int64_t* __restrict__ foff, int U, int64_t f) {
    int lo = 0, hi = U;                 // foff[lo] <= f < foff[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (foff[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

template <typename SAMPLE>
__device__ __forceinline__ double sample_at(const SAMPLE* x, int64_t i) { return (double)x[i]; }

// y[0] = x[0]; y[i] = x[i] - a x[i-1] (mfcc.py:63); zero beyond the utterance (utils.py:121-125)
template <typename SAMPLE>
__device__ __forceinline__ double preemph(const SAMPLE* __restrict__ x, int64_t n, int64_t i, double a) {
    if (i >= n) return 0.0;
    const double cur = sample_at(x, i);
    // separate multiply and subtract (no FMA contraction): bit-identical to numpy's x[1:] - a * x[:-1]
    return i == 0 ? cur : __dsub_rn(cur, __dmul_rn(a, sample_at(x, i - 1)));
}

// [lo, hi) of the non-zero taps of every mel filter (zeros add nothing to the dot product of mfcc.py:148)
__global__ void mel_range_kernel(const double* __restrict__ fbank, int n_mel, int nbin, int* __restrict__ range) {
    // one wave per filter: lanes scan the bins with a 64-stride, min / max over the wave
    const int m = blockIdx.x, lane = threadIdx.x;
    const double* fb = fbank + (size_t)m * nbin;
    int lo = nbin, hi = 0;
    for (int k = lane; k < nbin; k += 64)
        if (fb[k] != 0.0) {
            if (k < lo) lo = k;
            hi = k + 1;
        }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const int olo = __shfl_xor(lo, s, 64), ohi = __shfl_xor(hi, s, 64);
        lo = olo < lo ? olo : lo;
        hi = ohi > hi ? ohi : hi;
    }
    if (hi == 0) lo = 0;
    if (lane == 0) {
        range[2 * m] = lo;
        range[2 * m + 1] = hi;
    }
}

template <typename SAMPLE>
__global__ void mfcc_frame_kernel(const SAMPLE* __restrict__ signal, const int64_t* __restrict__ soff,
                                  const int64_t* __restrict__ foff, int U, int L, int S, int NFFT, int LOG2N,
                                  int n_mel, double alpha, const double* __restrict__ window,
                                  const double* __restrict__ fbank, const int* __restrict__ mel_range,
                                  const double* __restrict__ twid, double* __restrict__ power_out, double* __restrict__ logmel,
                                  double* __restrict__ energy) {
    extern __shared__ double sm[];
    double* re = sm;                    // [NFFT]
    double* im = sm + NFFT;             // [NFFT]
    double* pw = sm + 2 * NFFT;         // [NFFT/2 + 1]
    double* red = pw + NFFT / 2 + 1;    // [32]
    const int64_t f = blockIdx.x;
    const int u = find_utt(foff, U, f);
    const int64_t fl = f - foff[u];
    const SAMPLE* x = signal + soff[u];
    const int64_t n = soff[u + 1] - soff[u];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nbin = NFFT / 2 + 1;

    // windowed frame in bit-reversed order (np.fft.rfft(frames, NFFT) truncates / zero-pads to NFFT)
    for (int k = tid; k < NFFT; k += nt) {
        double v = 0.0;
        if (k < L) v = preemph(x, n, fl * S + k, alpha) * window[k];
        const int r = (int)(__brev((unsigned)k) >> (32 - LOG2N));
        re[r] = v;
        im[r] = 0.0;
    }
    __syncthreads();
    // radix-2 decimation in time; twid[k] = (cos, -sin)(2 pi k / NFFT)
    for (int s = 1; s <= LOG2N; ++s) {
        const int half = 1 << (s - 1);
        for (int bfly = tid; bfly < NFFT / 2; bfly += nt) {
            const int grp = bfly >> (s - 1), pos = bfly & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            const int tw = pos << (LOG2N - s);
            const double wr = twid[2 * tw], wi = twid[2 * tw + 1];
            const double xr = re[i1], xi = im[i1];
            const double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
            const double ur = re[i0], ui = im[i0];
            re[i0] = ur + tr;
            im[i0] = ui + ti;
            re[i1] = ur - tr;
            im[i1] = ui - ti;
        }
        __syncthreads();
    }
    // |X|^2 / NFFT  (mfcc.py:141-142: magnitude via np.abs, then squared)
    double esum = 0.0;
    for (int k = tid; k < nbin; k += nt) {
        const double mag = hypot(re[k], im[k]);
        const double p = (1.0 / NFFT) * (mag * mag);
        pw[k] = p;
        esum += p;
        if (power_out) power_out[f * nbin + k] = p;
    }
    const double etot = block_sum(esum, red);
    if (tid == 0) energy[f] = log(etot);                                   // mfcc.py:158-160
    __syncthreads();
    for (int m = tid; m < n_mel; m += nt) {
        const double* fb = fbank + (size_t)m * nbin;
        double s = 0.0;
        const int hi = mel_range[2 * m + 1];
        for (int k = mel_range[2 * m]; k < hi; ++k) s += pw[k] * fb[k];      // mfcc.py:148
        if (s == 0.0) s = 2.220446049250313e-16;                             // mfcc.py:149
        logmel[f * n_mel + m] = 20.0 * log10(s);                             // mfcc.py:150
    }
}

// A lane's constant share of the mel and DCT phases of the fft-512 kernels
struct Mel512Lane {
    int m, cnt, flags, cmax, n_mel, n_out, dq;
    const double* fb;          // the lane's taps in global memory (filter banks too dense for the LDS table)
    const double* C;           // DCT matrix C[k][m] in global memory (layout without the output split: n_out > 16)
    int lo;
    bool taps_in_lds, dct_split;
    // padded, lane-major tables (fast path): tapT[i * 64] = tap i of this lane's window of cp bins starting at lo2
    // (zero outside the lane's filter chunk), ctT[j * 64] = DCT coefficient of (output lane >> 2, input m0 + j),
    // zero beyond the lane's quarter -- no index is ever clamped or selected in the loops
    const double* tapT;
    const double* ctT;
    int cp, lo2, jp, m0;
};

// Block-level tables (three block barriers; waves are independent afterwards): DCT matrix transposed, the
// non-zero filter taps packed filter after filter, and the lane assignment of the mel phase -- a filter of
// cnt taps gets ceil(cnt / C) consecutive lanes (at most 3), C the smallest chunk for which 64 lanes suffice.
__device__ __forceinline__ void mel_plan_setup(int n_mel, int n_out, const int* __restrict__ mel_range,
                                               const double* __restrict__ fbank, const double* __restrict__ dctC,
                                               double* taps, double* CtP, int* asg, int* asg_cmax, int tid, int NT,
                                               int lane, int w, int nwaves) {
    constexpr int NBIN = 257, MAXTAP = 1024;
    // The lane plan, by wave 0 with lane m holding filter m's width: the widths never go through the LDS and nothing
    // is divided.  (As a single-lane loop over LDS-resident widths this cost every block ~25k cycles, 8 % of the kernel:
    // tools/mfcc_stamps.py.)  C >= ceil(widest / 3), so a filter takes 1, 2 or 3 lanes; the lanes of one filter never
    // straddle a 16-lane row -- its partial sums then meet through row-shift DPP moves -- and the lanes skipped for
    // that are idle.
    if (w == 0) {
        const int c = lane < n_mel ? mel_range[2 * lane + 1] - mel_range[2 * lane] : 0;
        int cmx = c;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const int o = __shfl_xor(cmx, sft, 64);
            cmx = o > cmx ? o : cmx;
        }
        int C = (cmx + 2) / 3;
        if (C < 1) C = 1;
        int parts, start;
        for (;; ++C) {
            parts = 1 + (c > C ? 1 : 0) + (c > 2 * C ? 1 : 0);
            start = 0;
            int ln = 0;                                            // wave-uniform scan over the filters
            for (int m = 0; m < n_mel; ++m) {
                const int pm = __builtin_amdgcn_readlane(parts, m);
                if ((ln & 15) + pm > 16) ln = (ln + 15) & ~15;
                if (lane == m) start = ln;
                ln += pm;
            }
            if (ln <= 64) break;
        }
        asg[4 * lane] = -1, asg[4 * lane + 1] = 0, asg[4 * lane + 2] = 0, asg[4 * lane + 3] = 0;
        __builtin_amdgcn_wave_barrier();                           // one wave's LDS stores land in program order
        if (lane < n_mel)
            for (int q = 0; q < parts; ++q) {
                const int i0 = q * C, cnt = (c - i0) < C ? (c - i0) : C, ln = start + q;
                asg[4 * ln] = lane;
                asg[4 * ln + 1] = i0;
                asg[4 * ln + 2] = cnt > 0 ? cnt : 0;
                asg[4 * ln + 3] = (q == 0 ? 1 : 0) | ((q == 0 && parts > 1) ? 2 : 0) | ((q == 0 && parts > 2) ? 4 : 0);
            }
        if (lane == 0) *asg_cmax = C;
    }
    __syncthreads();
    // lane-major padded tap table [cp][64]: a lane's window is the cp = 4 ceil(C / 4) bins from lo2 = min(lo, 260 - cp)
    // (so that the padding never reaches past the three unused slots behind the 257 bins), its taps sit at their bins'
    // positions in that window and everything else is 0.0
    const int C = *asg_cmax, cp = (C + 3) / 4 * 4;
    if (cp * 64 <= MAXTAP) {
        for (int e = tid; e < cp * 64; e += NT) {
            const int i = e >> 6, ln = e & 63;
            const int m = asg[4 * ln], i0 = asg[4 * ln + 1], cnt = asg[4 * ln + 2];
            double v = 0.0;
            if (m >= 0) {
                const int lo = mel_range[2 * m] + i0;
                const int lo2 = lo < 260 - cp ? lo : 260 - cp;
                const int t = i - (lo - lo2);
                if (t >= 0 && t < cnt) v = fbank[(size_t)m * NBIN + lo + t];
            }
            taps[e] = v;
        }
    }
    // lane-major padded DCT table [jp][64] for the split layout (lane = 4 k + part): coefficient of input m0 + j
    {
        const int dq = (n_mel + 3) / 4, jp = (dq + 3) / 4 * 4;
        if (n_out * 4 <= 64)
            for (int e = tid; e < jp * 64; e += NT) {
                const int j = e >> 6, ln = e & 63, k = ln >> 2, part = ln & 3;
                const int m = part * dq + j, m1 = (part * dq + dq) < n_mel ? (part * dq + dq) : n_mel;
                CtP[e] = (k < n_out && j < dq && m < m1) ? dctC[(size_t)k * n_mel + m] : 0.0;
            }
    }
    __syncthreads();
}

__device__ __forceinline__ Mel512Lane mel_plan_lane(int n_mel, int n_out, const int* __restrict__ mel_range,
                                                    const double* __restrict__ fbank, const double* taps,
                                                    const double* __restrict__ dctC, const double* CtP,
                                                    const int* asg, int cmax, int lane) {
    constexpr int NBIN = 257, MAXTAP = 1024;
    Mel512Lane ml;
    ml.m = asg[4 * lane];
    const int i0 = asg[4 * lane + 1];
    ml.cnt = asg[4 * lane + 2];
    ml.flags = asg[4 * lane + 3];
    ml.cmax = cmax;
    ml.n_mel = n_mel;
    ml.n_out = n_out;
    ml.dq = (n_mel + 3) / 4;
    ml.lo = ml.m >= 0 ? mel_range[2 * ml.m] + i0 : 0;
    ml.fb = fbank + (size_t)(ml.m >= 0 ? ml.m : 0) * NBIN + ml.lo;
    ml.C = dctC;
    ml.cp = (cmax + 3) / 4 * 4;
    ml.taps_in_lds = ml.cp * 64 <= MAXTAP;
    ml.lo2 = ml.lo < 260 - ml.cp ? ml.lo : 260 - ml.cp;
    ml.tapT = taps + lane;
    ml.dct_split = n_out * 4 <= 64;            // n_out <= 16 outputs x 4 quarter-sums fill the wave
    ml.jp = (ml.dq + 3) / 4 * 4;
    ml.m0 = (lane & 3) * ml.dq;
    ml.ctT = CtP + lane;
    return ml;
}

// (The mel -> log -> DCT phase itself is mel_dct_frames4 below.  The reference subtracts the utterance mean before
// the DCT (mfcc.py:151-152); the DCT is linear, so that subtraction is applied to the transformed row by
// feat_kernel and nothing in the frame kernel needs an utterance-level quantity.)

// tw512[k] = (cos, -sin)(2 pi k / 512), k < 256; the 256-point twiddles are tw512[2m].
struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx mul_neg_i(cplx a) { return {a.im, -a.re}; }          // a * (-i)

__device__ __forceinline__ void radix4(cplx v[4]) {
    const cplx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    const cplx t2 = cadd(v[1], v[3]), t3 = mul_neg_i(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}

// find_utt by the whole wave (f the same in every lane): 64 probes per round, so two dependent loads find the
// utterance among 4096 where the binary search walks twelve
__device__ __forceinline__ int find_utt_wave(const int64_t* __restrict__ foff, int U, int64_t f, int lane) {
    int lo = 0, hi = U;                 // foff[lo] <= f < foff[hi]
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) / 64;
        const int pos = lo + step * (lane + 1);              // increasing with the lane; probes at or past hi say "greater"
        const bool le = pos < hi && foff[pos] <= f;
        const int nle = __popcll(__ballot(le));              // the probes that are <= f form a prefix of the lanes
        const int nlo = nle ? lo + step * nle : lo;
        const int nhi = lo + step * (nle + 1) < hi ? lo + step * (nle + 1) : hi;
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const T*)(((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------------------------------
// fft_n = 512, FOUR FRAMES PER WAVE: the 256-point complex FFT as 16 x 16 with both 16-point transforms in
// registers.  A 16-lane row owns one frame.
//   pass 1  lane (row, n1) holds z[n1 + 16 n2], n2 = 0..15 (16 strided loads), transforms over n2 -> Y[n1][k2],
//           multiplies by W256^(n1 k2);
//   exchange (the only trip of the transform through the LDS, real parts then imaginary parts so a wave needs
//           8 bytes per point): lane (row, k2) collects Y[n1][k2], n1 = 0..15;
//   pass 2  transforms over n1 -> Z[16 k1 + k2], k1 = 0..15, in registers.
// Real-input split: X[k] and X[256-k] come from the same pair (Z[k], Z[256-k]):
//     s = Z[k] + conj Z[256-k],  t = -i (Z[k] - conj Z[256-k]) W512^k,  |X[k]|^2 = |s + t|^2 / 4,
//     |X[256-k]|^2 = |s - t|^2 / 4
// and Z[256-k] sits in lane 16 - k2 of the same row (register 15 - k1), so every lane handles its k1 < 8 and
// fetches eight values from its mirror lane.  Lane 0 is its own mirror (k = 16 k1 <-> 16 (16 - k1)) and also
// owns k = 0 / 256.
// Against a one-frame-per-wave radix-4 Stockham (the first version of this kernel, 0.60 ms per 256k frames against
// 0.35): one LDS round trip of the data instead of four, the split at half the arithmetic, and 16 independent
// butterflies per lane to hide latencies behind.
constexpr int kX4Waves = 8;            // waves per block (two per SIMD)
constexpr int kX4FS = 308;             // doubles per frame in a wave's LDS slice: 16 x 17 exchange tile, later
                                       // 257 power bins, log-mel row at 260

// radix4 with a fourth input known to be zero (frame_len = 400 of 512: the last three input rows of the first pass)
__device__ __forceinline__ void radix4_z3(cplx v[4]) {
    const cplx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    const cplx t2 = v[1], t3 = mul_neg_i(v[1]);
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}

// NZ: inputs v[n], n >= NZ, are known to be zero (their butterflies lose the additions that would only add 0)
template <int NZ = 16>
__device__ __forceinline__ void fft16(cplx v[16]) {
    // index n = a + 4 a', output k = b + 4 c:  W16^(nk) = W4^(a'b) W16^(ab) W4^(ac)
    constexpr double C1 = 0.92387953251128674, S1 = 0.38268343236508977, R = 0.70710678118654752;
    static_assert(NZ > 12, "only the last input of a butterfly may be a known zero");
    cplx u[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        cplx t[4] = {v[a], v[a + 4], v[a + 8], v[a + 12]};
        if (a + 12 >= NZ) radix4_z3(t);
        else radix4(t);
#pragma unroll
        for (int b = 0; b < 4; ++b) u[a][b] = t[b];
    }
    auto w2 = [&](cplx z) -> cplx { return {R * (z.re + z.im), R * (z.im - z.re)}; };       // * W16^2
    auto w6 = [&](cplx z) -> cplx { return {R * (z.im - z.re), -R * (z.re + z.im)}; };      // * W16^6
    u[1][1] = cmul(u[1][1], {C1, -S1});
    u[1][2] = w2(u[1][2]);
    u[1][3] = cmul(u[1][3], {S1, -C1});
    u[2][1] = w2(u[2][1]);
    u[2][2] = mul_neg_i(u[2][2]);
    u[2][3] = w6(u[2][3]);
    u[3][1] = cmul(u[3][1], {S1, -C1});
    u[3][2] = w6(u[3][2]);
    u[3][3] = cmul(u[3][3], {-C1, S1});
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cplx t[4] = {u[0][b], u[1][b], u[2][b], u[3][b]};
        radix4(t);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[b + 4 * c] = t[c];
    }
}

// One row's raw samples through buffer loads (frames at the edges of an utterance).  Called with only that
// row's lanes active: base, n and the utterance pointer are the same in all of them, so the descriptors are
// uniform (say so with readfirstlane, or the loads are wrapped in waterfall loops) and reads beyond the utterance
// return 0 without per-lane clamping.  x[i-1] uses a second descriptor starting one sample earlier; at the
// utterance start, where there is no earlier sample, that descriptor starts at x[0] and the offsets move back
// by one sample -- the one sample that does not exist (x[-1], never used: y[0] = x[0]) is clamped onto x[0] so
// no offset is negative.
template <typename SAMPLE>
__device__ __noinline__ void fetch_row(const SAMPLE* __restrict__ x, int64_t n, int64_t base, int n1, double alpha,
                                       cplx v[16]) {
    constexpr int SZ = sizeof(SAMPLE);
    int64_t left = (n - base) * SZ;
    left = left < 0 ? 0 : (left > 0x7FFFFFF0L ? 0x7FFFFFF0L : left);
    const int64_t back = base > 0 ? 1 : 0;
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(x + base), 0, __builtin_amdgcn_readfirstlane((int)left), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uniform_ptr(x + base - back), 0, __builtin_amdgcn_readfirstlane((int)(left + back * SZ)), 0x00020000);
    const int vo = n1 * 2 * SZ;
    const int adj = back ? 0 : SZ;
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) {
        int vb = vo + (n2 * 32 * SZ - adj);
        if (n2 == 0) vb = vb < 0 ? 0 : vb;
        double x0, x1, xm;
        if constexpr (SZ == 8) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 q = __builtin_amdgcn_raw_buffer_load_b128(ra, vo + n2 * 32 * SZ, 0, 0);
            x0 = __hiloint2double((int)q[1], (int)q[0]);
            x1 = __hiloint2double((int)q[3], (int)q[2]);
            xm = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rb, vb, 0, 0));
        } else {
            x0 = (double)(short)__builtin_amdgcn_raw_buffer_load_b16(ra, vo + n2 * 32 * SZ, 0, 0);
            x1 = (double)(short)__builtin_amdgcn_raw_buffer_load_b16(ra, vo + n2 * 32 * SZ + SZ, 0, 0);
            xm = (double)(short)__builtin_amdgcn_raw_buffer_load_b16(rb, vb, 0, 0);
        }
        const int64_t i0 = base + 2 * (n1 + 16 * n2);
        double a = i0 == 0 ? x0 : __dsub_rn(x0, __dmul_rn(alpha, xm));   // y[0] = x[0] (mfcc.py:63)
        double c = __dsub_rn(x1, __dmul_rn(alpha, x0));
        if (i0 >= n) a = 0.0;                                             // utils.py:121-125
        if (i0 + 1 >= n) c = 0.0;
        v[n2] = {a, c};
    }
}

// log(x) for the mel and energy sums: the table-driven log of devmath.h (a dozen instructions, error below
// 2e-16 of the result) for normal x; the library log for the rest (0 -> -inf, denormals)
__device__ __forceinline__ double log_fast(double x, const double* ltab) {
    if (__builtin_expect(!(x >= 1e-300 && x <= 1e300), 0)) return log(x);
    return log_sum(x, ltab);
}

// Four logs at once: straight-line, so the four table reads travel together (one LDS round trip where four
// calls of log_fast, each with its own range branch, took four).  The wave-uniform test sends everything to
// log_fast if ANY active lane holds a value the table scheme does not take (0, denormal, huge, NaN).
__device__ __forceinline__ void log_fast4(double x[4], const double* ltab) {
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) ok = ok && (x[r] >= 1e-300 && x[r] <= 1e300);
    if (__builtin_expect(!__all(ok), 0)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = log_fast(x[r], ltab);
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = log_sum(x[r], ltab);
}

// the padded loops of the mel / DCT phase with a compile-time length: every LDS read is issued before the first
// product, one round trip instead of one per group of four taps (the phase is bound by LDS latency, not by its FMAs:
// tools/mfcc_stamps.py)
template <int CP>
__device__ __forceinline__ void padded_dot4(const double* tt, const double* pp, double s[4]) {
    constexpr int FS = kX4FS;
    double tap[CP], pv[4][CP];
#pragma unroll
    for (int i = 0; i < CP; ++i) {
        tap[i] = tt[i * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r][i] = pp[r * FS + i];
    }
#pragma unroll
    for (int i = 0; i < CP; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = fma(pv[r][i], tap[i], s[r]);
}
// WIDE: whole length in one go (60 doubles in flight at cp = 12); otherwise four taps at a time -- the f64-sample kernel
// carries the next group's prefetched samples through this phase (54 registers) and would spill
template <bool WIDE>
__device__ __forceinline__ void padded_dot4_n(int cp, const double* tt, const double* pp, double s[4]) {
    constexpr int FS = kX4FS;
    if (WIDE) {
        if (cp == 12) return padded_dot4<12>(tt, pp, s);
        if (cp == 8) return padded_dot4<8>(tt, pp, s);
        if (cp == 16) return padded_dot4<16>(tt, pp, s);
    }
    for (int i = 0; i < cp; i += 4) {
        double tap[4], pv[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tap[q] = tt[(i + q) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r][q] = pp[r * FS + i + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = fma(pv[r][q], tap[q], s[r]);
    }
}

// mel -> log -> DCT of the four frames of a wave's slice in one walk: a lane keeps its filter chunk (and its
// DCT output) for all four, so taps and DCT coefficients are read once and the four chains overlap.  Both loops run
// over padded, lane-major tables (mel_plan_setup): straight-line groups of four taps, every address a constant offset
// from one base, no clamped index and no select -- the LDS reads pair up (ds_read2_b64 / ds_read2st64_b64) and a tap
// costs one FMA per frame plus its share of the reads.  (Round 2's loops clamped every index and selected every tap:
// ~15 vector instructions per tap against the 4 FMAs that do the work.)
template <bool WIDE>
__device__ __forceinline__ void mel_dct_frames4(const Mel512Lane& ml, double* slice, const double* ltab, int lane,
                                                double* __restrict__ cep_rows, int nlive) {
    constexpr int FS = kX4FS;
    LN_STAMP_DECL;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (ml.taps_in_lds) {
        padded_dot4_n<WIDE>(ml.cp, ml.tapT, slice + ml.lo2, s);
    } else {                                       // filter bank too dense for the LDS table: taps from global memory
        const double* pp = slice + ml.lo;
        for (int i = 0; i < ml.cmax; i += 4) {
            double tap[4], pv[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = i + q < ml.cnt;     // past the lane's count: index 0, tap 0.0 (adds nothing)
                const int ii = in ? i + q : 0;
                const double t = ml.fb[ii];
                tap[q] = in ? t : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r][q] = pp[r * FS + ii];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] = fma(pv[r][q], tap[q], s[r]);
        }
    }
    LN_STAMP(8);
    // a filter's lanes share a 16-lane row (mel_plan_setup): its partial sums meet through row-shift DPP moves
    // (row_shl:1 / row_shl:2: lane l <- lane l + 1 / l + 2 of the row, 0 from beyond it), no LDS round trip
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double s1 = dpp_f64<0x101>(s[r]), s2 = dpp_f64<0x102>(s[r]);
        if (ml.flags & 2) s[r] += s1;
        if (ml.flags & 4) s[r] += s2;
    }
    if (ml.flags & 1) {
        double x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = s[r] == 0.0 ? 2.220446049250313e-16 : s[r];   // mfcc.py:149
        log_fast4(x, ltab);
#pragma unroll
        for (int r = 0; r < 4; ++r) slice[r * FS + 260 + ml.m] = (20.0 / 2.302585092994045684) * x[r];   // 20 log10: mfcc.py:150
    }
    __builtin_amdgcn_wave_barrier();
    LN_STAMP(9);
    const int n_mel = ml.n_mel, dq = ml.dq;
    const double* lm = slice + 260;
    if (ml.dct_split) {
        // lane = 4 k + part: output k, quarter `part` of the inputs; the slots behind the n_mel log-mel values are
        // never written (zero since the kernel's start) and their coefficients are zero
        const int k = lane >> 2, part = lane & 3;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        padded_dot4_n<WIDE>(ml.jp, ml.ctT, lm + ml.m0, acc);
        LN_STAMP(10);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] += dpp_f64<0xB1>(acc[r]);       // (q0 + q1) + (q2 + q3)
            acc[r] += dpp_f64<0x4E>(acc[r]);
            if (part == 0 && k < ml.n_out && r < nlive) cep_rows[(size_t)r * ml.n_out + k] = acc[r];
        }
    } else {
        for (int k = lane; k < ml.n_out; k += 64) {
            double q[4][4];
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const int m0 = part * dq, m1 = (m0 + dq) < n_mel ? (m0 + dq) : n_mel;
#pragma unroll
                for (int r = 0; r < 4; ++r) q[r][part] = 0.0;
                for (int m = m0; m < m1; ++m) {
                    const double c = ml.C[(size_t)k * n_mel + m];       // 13 KB at n_mel = 40: L1-resident
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r][part] = fma(lm[r * FS + m], c, q[r][part]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nlive) cep_rows[(size_t)r * ml.n_out + k] = (q[r][0] + q[r][1]) + (q[r][2] + q[r][3]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    LN_STAMP(11);
}

// raw samples of a lane for one group, as loaded: pairs (x[2n], x[2n+1]), n = n1 + 16 n2, and the one sample
// before the frame.  x[2n-1] is the neighbour lane's x[2(n-1)+1] and is fetched with a row rotate.
template <typename SAMPLE>
struct RawRow;
template <>
struct RawRow<double> {
    double x0[16], x1[16], prev;
};
template <>
struct RawRow<int16_t> {
    int q[16];                 // x[2n] | x[2n+1] << 16
    int prev;
};
typedef int __attribute__((aligned(2))) int_a2;                 // a sample pair starts on any even byte
struct __attribute__((aligned(8))) double_pair {
    double a, b;
};

// NR: rows n2 < NR of the 16 hold samples of the frame (2 (n1 + 16 n2) < frame_len for some n1); the others are zero
template <int NR>
__device__ __forceinline__ void load_row(const double* p, RawRow<double>& rw) {
#pragma unroll
    for (int n2 = 0; n2 < NR; ++n2) {
        const double_pair d = *reinterpret_cast<const double_pair*>(p + 32 * n2);
        rw.x0[n2] = d.a;
        rw.x1[n2] = d.b;
    }
}
template <int NR>
__device__ __forceinline__ void load_row(const int16_t* p, RawRow<int16_t>& rw) {
#pragma unroll
    for (int n2 = 0; n2 < NR; ++n2) rw.q[n2] = *reinterpret_cast<const int_a2*>(p + 32 * n2);
}

// y[2n], y[2n+1] of an interior frame: y[i] = x[i] - alpha x[i-1], multiply and subtract rounded separately
// (mfcc.py:63)
// at_start: the frame begins the utterance, so its first sample has no predecessor: y[0] = x[0] (mfcc.py:63)
template <int NR>
__device__ __forceinline__ void preemph_row(const RawRow<double>& rw, int n1, double alpha, bool at_start, cplx v[16]) {
    double t[16];
#pragma unroll
    for (int n2 = 0; n2 < NR; ++n2) t[n2] = dpp_f64<0x121>(rw.x1[n2]);      // row_ror:1: lane l <- lane l - 1 (mod 16)
#pragma unroll
    for (int n2 = NR; n2 < 16; ++n2) v[n2] = {0.0, 0.0};
#pragma unroll
    for (int n2 = 0; n2 < NR; ++n2) {
        // lane 0 received lane 15's pair n2, which precedes its own pair n2 + 1
        const double wrap = n2 == 0 ? rw.prev : t[n2 - 1];
        const double xm = n1 == 0 ? wrap : t[n2];
        v[n2] = {__dsub_rn(rw.x0[n2], __dmul_rn(alpha, xm)), __dsub_rn(rw.x1[n2], __dmul_rn(alpha, rw.x0[n2]))};
    }
    if (at_start && n1 == 0) v[0].re = rw.x0[0];
}
template <int NR>
__device__ __forceinline__ void preemph_row(const RawRow<int16_t>& rw, int n1, double alpha, bool at_start, cplx v[16]) {
    int t[16];
#pragma unroll
    for (int n2 = 0; n2 < NR; ++n2) t[n2] = __builtin_amdgcn_mov_dpp(rw.q[n2], 0x121, 0xf, 0xf, true);
#pragma unroll
    for (int n2 = NR; n2 < 16; ++n2) v[n2] = {0.0, 0.0};
#pragma unroll
    for (int n2 = 0; n2 < NR; ++n2) {
        const int wrap = n2 == 0 ? rw.prev : (t[n2 - 1] >> 16);
        const double xm = (double)(n1 == 0 ? wrap : (t[n2] >> 16));
        const double x0 = (double)(short)(rw.q[n2] & 0xffff), x1 = (double)(rw.q[n2] >> 16);
        v[n2] = {__dsub_rn(x0, __dmul_rn(alpha, xm)), __dsub_rn(x1, __dmul_rn(alpha, x0))};
    }
    if (at_start && n1 == 0) v[0].re = (double)(short)(rw.q[0] & 0xffff);
}

// NR = 13 when frame_len <= 416 (the reference's 400): input rows n2 >= 13 of the first pass are all zero, so they are
// neither loaded nor windowed and their butterflies shrink (fft16<NR>); NR = 16 otherwise.
template <typename SAMPLE, bool POWER, int NR, int NW>
__global__ __launch_bounds__(64 * NW) void mfcc_frame512x4_kernel(
    const SAMPLE* __restrict__ signal, const int64_t* __restrict__ soff, const int64_t* __restrict__ foff, int U,
    int64_t F, int64_t NS, int L, int S, int n_mel, int n_out, double alpha, const double* __restrict__ window,
    const double* __restrict__ fbank, const int* __restrict__ mel_range, const double* __restrict__ tw512,
    const double* __restrict__ dctC, double* __restrict__ power_out, double* __restrict__ cepraw,
    double* __restrict__ energy) {
    constexpr int NC = 256, NBIN = 257, MAXTAP = 1024, MAXMEL = 48, FS = kX4FS;
    __shared__ double tws[2 * NC];                 // tw512
    __shared__ double wins[2 * NC];                // window, zero beyond L
    __shared__ double T16[2 * 256];                // W256^(n1 k2) at [k2][n1]
    __shared__ double ltab[2 * kLogTab];           // table-driven log (devmath.h)
    __shared__ double taps[MAXTAP];
    __shared__ double CtP[12 * 64];                // padded lane-major DCT table of the split layout (MAXMEL / 4 rows)
    __shared__ int asg[4 * 64];
    __shared__ int asg_cmax;
    __shared__ int ticket;                         // next undrawn group of the block's range (relative)
    __shared__ double bufs[NW][4 * FS];
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef LNASR_MFCC_STAMPS
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long st_rbegin = __builtin_amdgcn_s_memrealtime();     // 100 MHz
#endif
    for (int e = tid; e < 2 * NC; e += NT) {
        tws[e] = tw512[e];
        wins[e] = e < L ? window[e] : 0.0;
    }
    for (int e = tid; e < 256; e += NT) {
        const int m2 = 2 * (e & 15) * (e >> 4);    // W256^m = tw512[2m], tw512[i + 256] = -tw512[i]; 2m <= 450
        const int idx = m2 & (NC - 1);
        const double sg = m2 >= NC ? -1.0 : 1.0;
        T16[2 * e] = sg * tw512[2 * idx];
        T16[2 * e + 1] = sg * tw512[2 * idx + 1];
    }
    log_table_init(ltab, tid, NT);
    // the slices start out zero: the padded loops of the mel / DCT phase read slots that are never written
    for (int e = tid; e < NW * 4 * FS; e += NT) (&bufs[0][0])[e] = 0.0;
    mel_plan_setup(n_mel, n_out, mel_range, fbank, dctC, taps, CtP, asg, &asg_cmax, tid, NT, lane, w, NW);
    const Mel512Lane ml = mel_plan_lane(n_mel, n_out, mel_range, fbank, taps, dctC, CtP, asg, asg_cmax, lane);

    const int row = lane >> 4, n1 = lane & 15;     // pass 1: (frame row, n1); pass 2 and after: k2 = n1
    // Work distribution.  A block owns a contiguous range of groups (its rows change utterance rarely) and its waves
    // draw chunks of kChunk consecutive groups from a ticket counter in the LDS (the 240 samples successive frames share
    // are still in the L1 within a chunk).  A static split -- 1/8 of the range per wave, rounds 1-2 -- loses a fifth of
    // the kernel to its own tail: the two waves of a SIMD are not served alike (the older one issues whenever it can,
    // the younger fills the gaps), so waves 0-3 of every block finished after 194 us and waves 4-7 after 252 us, alone
    // on their SIMDs for the difference (per-wave lifetimes: tools/mfcc_stamps.py); and ceil(G / waves) groups per wave
    // left the last six blocks without work.  With tickets the fast waves simply take more chunks.
    // (chunks of 1 / 2 / 4 / 8 / 16 groups with the phase priorities in place, ms per 256 k frames f64 / int16: 0.263 / 0.256,
    // 0.268 / 0.258, 0.259 / 0.249, 0.260 / 0.258, 0.261 / 0.259: round 4)
    constexpr int kChunk = 4;
    const int64_t G = (F + 3) / 4, per_block = (G + gridDim.x - 1) / gridDim.x;
    const int64_t bbeg = (int64_t)blockIdx.x * per_block, bend = bbeg + per_block < G ? bbeg + per_block : G;
    if (tid == 0) ticket = 0;
    __syncthreads();
    auto draw = [&]() -> int64_t {                 // first group of the next chunk (wave-uniform); >= bend: none left
        int t = 0;
        if (lane == 0) t = atomicAdd(&ticket, kChunk);
        return bbeg + __builtin_amdgcn_readfirstlane(t);
    };
    double* slice = &bufs[w][0];
    double* X = slice + row * FS;
    int64_t g = draw();
    int64_t cend = g + kChunk;                     // end of the chunk in hand
#ifdef LNASR_MFCC_STAMPS
    if (g >= bend && lane == 0 && blockIdx.x * NW + w < 4096) {
        g_mfcc_wave_times[2 * (blockIdx.x * NW + w)] = st_rbegin;
        g_mfcc_wave_times[2 * (blockIdx.x * NW + w) + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (g >= bend) return;
    // The utterance of this row's frame: frames visited by a row only increase, so the bounds of the current
    // utterance stay in registers and memory is touched again only when a row crosses into the next one.
    int u;
    int64_t fbeg, fend, so, n;
    auto locate = [&](int64_t fc, bool first) {
        if (first) u = find_utt_wave(foff, U, 4 * g < F ? 4 * g : F - 1, lane);    // row 0's frame; the rows walk on from there
        while (foff[u + 1] <= fc) ++u;
        fbeg = foff[u];
        fend = foff[u + 1];
        so = soff[u];
        n = soff[u + 1] - so;
    };
    // state of the group about to be transformed
    bool live, inner;                              // inner: all four frames lie wholly inside their utterances
    int64_t f, base;
    const SAMPLE* xu;
    auto place = [&](int64_t gg, bool first) {
        f = 4 * gg + row;
        live = f < F;
        const int64_t fc = live ? f : F - 1;       // a row past the end recomputes the last frame and stores nothing
        if (first || fc >= fend) locate(fc, first);
        base = (fc - fbeg) * S;
        xu = signal + so;
        // The fast path takes every frame whose frame_len samples all exist.  It reads 32 NR samples (16 past a 400-sample
        // frame): what lies behind the utterance -- the next one's first samples -- is multiplied by the window's zeros
        // (and selected away, so that not even a NaN there can leak), it only has to be readable: inside the batch's
        // buffer.  The first frame of an utterance is in too (its first sample has no predecessor: at_start).  Until
        // round 3 both the first and the last frame of EVERY utterance took the row-by-row edge path below -- ~4x a
        // group's time, and the waves that met one set the kernel's duration (tools/mfcc_stamps.py: wave lifetimes
        // 232 us median, 293 us maximum).
        inner = __all(base + L <= n && so + base + 32 * NR <= NS);
    };
    RawRow<SAMPLE> rw;
    auto request = [&]() {                         // plain loads at constant offsets from one pointer
        const SAMPLE* p = xu + base + 2 * n1;
        load_row<NR>(p, rw);
        rw.prev = (decltype(rw.prev))signal[so + base > 0 ? so + base - 1 : 0];    // unused when base == 0 (at_start)
    };
    place(g, true);
    if (inner) request();
    // (Starting the waves w/8 of an iteration apart, so that they are not all in the LDS-heavy phases at once, was
    // measured: no gain -- 0.315 against 0.313 ms; three waves per SIMD at 168 registers: 0.325, it spills.)
    LN_STAMP_DECL;
#ifdef LNASR_MFCC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_mfcc_stamps[12] += st_prev - st_begin;    // tables + plan + first loads
#endif
    for (;;) {
        LN_STAMP(0);
#ifdef LNASR_MFCC_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_mfcc_stamps[15] += 1;     // groups this lane's wave walked
#endif
        cplx v[16];
        if (inner) {
            preemph_row<NR>(rw, n1, alpha, base == 0, v);
        } else {
            cplx vs[16];                           // through the stack (out-of-line call): v itself stays in registers
#pragma unroll 1
            for (int r = 0; r < 4; ++r)
                if (row == r) fetch_row(xu, n, base, n1, alpha, vs);
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) v[n2] = n2 < NR ? vs[n2] : cplx{0.0, 0.0};   // beyond NR: the window is zero
        }
        const int64_t fcur = f;
        const bool lcur = live;
        const int64_t gcur = g;
        ++g;
        if (g == cend) {                           // chunk exhausted: the next one (a wave's groups only ever increase)
            g = draw();
            cend = g + kChunk;
        }
        const bool more = g < bend;
        if (more) place(g, false);                 // the next group's whereabouts, ahead of its loads
        LN_STAMP(1);
        // Wave priority by PHASE (round 4).  The two waves of a SIMD are in different phases most of the time; the arbiter
        // favours the older one whatever it is doing.  Everything except the two register FFTs is a chain of LDS / DPP / memory
        // round trips with few instructions between them: those phases run at priority 1, so their instructions issue the
        // moment their operands arrive and the partner's dense FFT work fills the gaps, instead of the reverse.  Measured on
        // one box, ms per 256 k frames (f64 / int16): 0.291 / 0.281 without, 0.279 / 0.267 with only the exchange and the mel / DCT
        // tail raised, 0.270 / 0.265 as here; priority 3 instead of 1: the same (profiles/r04c_prio_experiment.log, r04d).
        __builtin_amdgcn_s_setprio(0);                       // dense vector work: window, first FFT, twiddles
#pragma unroll
        for (int n2 = 0; n2 < NR; ++n2) {
            const int k0 = 2 * (n1 + 16 * n2);
            v[n2].re *= wins[k0];
            v[n2].im *= wins[k0 + 1];
            if (n2 >= 12) {                        // rows that can reach past frame_len: zero whatever was read there
                if (k0 >= L) v[n2].re = 0.0;
                if (k0 + 1 >= L) v[n2].im = 0.0;
            }
        }
        fft16<NR>(v);                              // over n2
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) {
            const cplx tw = {T16[2 * (16 * k2 + n1)], T16[2 * (16 * k2 + n1) + 1]};
            v[k2] = cmul(v[k2], tw);
        }
        LN_STAMP(2);
        __builtin_amdgcn_s_setprio(1);
        // exchange: tile [k2][n1] with rows of 17 (both sides free of bank conflicts)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) X[k2 * 17 + n1] = v[k2].re;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j].re = X[n1 * 17 + j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) X[k2 * 17 + n1] = v[k2].im;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j].im = X[n1 * 17 + j];
        __builtin_amdgcn_wave_barrier();
        LN_STAMP(3);
        __builtin_amdgcn_s_setprio(0);
        fft16(v);                                  // over n1: v[k1] = Z[16 k1 + k2]
        LN_STAMP(4);
        __builtin_amdgcn_s_setprio(1);                       // split, power, the next group's requests, mel / log / DCT, pre-emphasis
        // mirror lane's upper half
        // lane k2 needs lane (16 - k2) mod 16 of its row: row_mirror (l <- 15 - l) then row_ror:1 (l <- l - 1), two DPP
        // moves per word and no trip through the LDS crossbar (ds_bpermute: one more latency-bound step, tools/mfcc_stamps.py)
        const int k2 = n1;
        auto mirror = [](double x) { return dpp_f64<0x121>(dpp_f64<0x140>(x)); };
        cplx mz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mz[j] = {mirror(v[15 - j].re), mirror(v[15 - j].im)};
        const bool lane0 = k2 == 0;
        double esum = 0.0;
        double* prow = POWER && lcur ? power_out + fcur * NBIN : nullptr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cplx mine = lane0 ? v[j + 1] : v[j];
            const int k = lane0 ? 16 * (j + 1) : 16 * j + k2;
            const cplx tw = {tws[2 * k], tws[2 * k + 1]};
            const cplx sm = {mine.re + mz[j].re, mine.im - mz[j].im};
            const cplx d = {mine.re - mz[j].re, mine.im + mz[j].im};
            const cplx t = cmul({d.im, -d.re}, tw);
            const double ar = sm.re + t.re, ai = sm.im + t.im, br = sm.re - t.re, bi = sm.im - t.im;
            // |X|^2 / 512 (mfcc.py:141-142; the reference squares np.abs: same value to an ulp)
            const double pa = (1.0 / 2048.0) * fma(ar, ar, ai * ai), pb = (1.0 / 2048.0) * fma(br, br, bi * bi);
            X[NC - k] = pb;                        // k = 128 (lane 0, j = 7) is its own mirror: pa is the one kept
            X[k] = pa;
            esum += (lane0 && j == 7) ? pa : pa + pb;
            if (POWER && prow) {
                prow[NC - k] = pb;
                prow[k] = pa;
            }
        }
        if (lane0) {                               // k = 0 and the Nyquist bin from Z[0]
            const double sp = v[0].re + v[0].im, sq = v[0].re - v[0].im;
            const double p0 = (1.0 / 512.0) * (sp * sp), pn = (1.0 / 512.0) * (sq * sq);
            X[0] = p0;
            X[NC] = pn;
            esum += p0 + pn;
            if (POWER && prow) {
                prow[0] = p0;
                prow[NC] = pn;
            }
        }
        esum += dpp_f64<0xB1>(esum);               // sum over the row
        esum += dpp_f64<0x4E>(esum);
        esum += dpp_f64<0x141>(esum);
        esum += dpp_f64<0x140>(esum);
        if (n1 == 0 && lcur) energy[fcur] = log_fast(esum, ltab);            // mfcc.py:158-160
        __builtin_amdgcn_wave_barrier();
        LN_STAMP(5);
        // the next group's samples travel while the mel / DCT phase (few registers) runs
        if (more && inner) request();
        LN_STAMP(6);
        const int64_t left = F - 4 * gcur;
        mel_dct_frames4<sizeof(SAMPLE) == 2>(ml, slice, ltab, lane, cepraw + 4 * gcur * n_out, left < 4 ? (int)left : 4);
        LN_STAMP(7);
        if (!more) break;
    }
#ifdef LNASR_MFCC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {     // the wave's lifetime in shader cycles and in 100 MHz ticks: the clock it ran at
        g_mfcc_stamps[13] += __builtin_amdgcn_s_memtime() - st_begin;
        g_mfcc_stamps[14] += __builtin_amdgcn_s_memrealtime() - st_rbegin;
    }
    if (lane == 0) {
        const int wi = blockIdx.x * NW + w;
        if (wi < 4096) {
            g_mfcc_wave_times[2 * wi] = st_rbegin;
            g_mfcc_wave_times[2 * wi + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

// Column sums over the frames of each utterance of the first nc columns of X [F, ld].  grid = (U, P): an
// utterance's frames are cut into P contiguous parts (a function of its own frame count only, so a batch and a
// single utterance add in the same order); feat_kernel adds the P partial sums in part order.  (No "last block
// finishes" here: a device-scope fence writes back the whole L2 on this multi-die part.)
constexpr int kMeanParts = 8;
__global__ void col_sum_kernel(const double* __restrict__ X, const int64_t* __restrict__ foff, int ld, int nc,
                               double* __restrict__ partial) {
    extern __shared__ double sm[];      // [slices][nc]
    const int u = blockIdx.x, p = blockIdx.y, P = gridDim.y;
    const int64_t f0 = foff[u], f1 = foff[u + 1], chunk = (f1 - f0 + P - 1) / P;
    const int64_t a = f0 + p * chunk, b = a + chunk < f1 ? a + chunk : f1;
    const int slices = blockDim.x / nc;
    const int c = threadIdx.x % nc, sl = threadIdx.x / nc;
    double s = 0.0;
    if (sl < slices)
        for (int64_t f = a + sl; f < b; f += slices) s += X[f * ld + c];
    if (sl < slices) sm[sl * nc + c] = s;
    __syncthreads();
    if (threadIdx.x < nc) {
        double tot = 0.0;
        for (int k = 0; k < slices; ++k) tot += sm[k * nc + threadIdx.x];
        partial[((size_t)u * P + p) * nc + threadIdx.x] = tot;
    }
}

// cep_raw[f,k] = sum_m logmel[f,m] C[k,m];  C = DCT-II ortho matrix (generic-FFT path; mfcc.py:152)
__global__ void dct_kernel(const double* __restrict__ logmel, const int64_t* __restrict__ foff, int U, int64_t F,
                           int n_mel, const double* __restrict__ C, double* __restrict__ cep) {
    extern __shared__ double row[];     // [rows][n_mel]
    const int rows = blockDim.x / n_mel;
    const int r = threadIdx.x / n_mel, k = threadIdx.x % n_mel;
    const int64_t f = (int64_t)blockIdx.x * rows + r;
    const bool ok = r < rows && f < F;
    if (ok) row[r * n_mel + k] = logmel[f * n_mel + k];
    __syncthreads();
    if (ok) {
        double s = 0.0;
        for (int m = 0; m < n_mel; ++m) s += row[r * n_mel + m] * C[(size_t)k * n_mel + m];
        cep[f * n_mel + k] = s;
    }
}

// Returned cepstrum and features from cep_raw = DCT(log-mel) and its utterance column means Mk:
//   first normalisation  (mfcc.py:151-152): cep[f,k] = cep_raw[f,k] - Mk - [k == 0] 1e-8 sqrt(n_mel)
//       (DCT-II ortho of the constant 1e-8 is 1e-8 sqrt(n_mel) in bin 0 and zero elsewhere)
//   second normalisation (mfcc.py:155-156), applied through a view so the RETURNED cepstrum has it
//       too: columns 1..nc lose mean_f(cep[:,k]) + 1e-8, and mean_f(cep[:,k]) is zero by construction
//   features: [cep[:,1..nc], log-energy], delta (row 0 = features[1]), delta-delta (mfcc.py:158-172)
__global__ void feat_kernel(const double* __restrict__ cepraw, const double* __restrict__ M,
                            const double* __restrict__ energy, const int64_t* __restrict__ foff, int n_mel, int nc,
                            int ncols, int ld, double* __restrict__ cep_out, double* __restrict__ feat) {
    // grid.y = utterance (no search for the frame's utterance); ncols = n_mel when the cepstrum is returned,
    // nc + 1 (the feature columns) otherwise
    const int W = nc + 1;
    const int u = blockIdx.y;
    const int64_t f0 = foff[u], f1 = foff[u + 1];
    // utterance means of the columns in use: the kMeanParts partial sums of col_sum_kernel, in part order
    __shared__ double mean_s[256];
    const double* Pu = M + (size_t)u * kMeanParts * ncols;
    for (int kk = threadIdx.x; kk < ncols; kk += blockDim.x) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < kMeanParts; ++q) tot += Pu[q * ncols + kk];
        mean_s[kk] = tot / (double)(f1 - f0);
    }
    __syncthreads();
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;      // < 2^32: checked by the host
    const double shift0 = 1e-8 * sqrt((double)n_mel);
    auto cepv = [&](int64_t ff, int kk) -> double {
        double v = cepraw[ff * ld + kk] - mean_s[kk];
        if (kk == 0) v -= shift0;
        if (kk >= 1 && kk <= nc) v -= 1e-8;
        return v;
    };
    // (One thread per OUTPUT element -- 39 per frame, a wave writing 512 contiguous bytes -- was measured in round 3:
    // 0.115 ms against 0.047: every element then re-reads and re-centres up to five cepstrum values.)
    const unsigned fi = idx / (unsigned)ncols;
    const int64_t f = f0 + fi;
    const int k = (int)(idx - fi * (unsigned)ncols);
    if (f >= f1) return;
    if (cep_out) cep_out[f * n_mel + k] = cepv(f, k);
    if (k >= W) return;
    const int c = k;                            // feature column
    auto base = [&](int64_t ff) -> double { return c < nc ? cepv(ff, 1 + c) : energy[ff]; };
    auto delta = [&](int64_t ff) -> double { return ff == f0 ? base(f0 + 1) : base(ff) - base(ff - 1); };
    double* out = feat + f * 3 * W;
    out[c] = base(f);
    out[W + c] = delta(f);
    out[2 * W + c] = (f == f0) ? delta(f0 + 1) : delta(f) - delta(f - 1);
}

// split_frames (utils.py:106-130)
__global__ void split_frames_kernel(const double* __restrict__ x, int64_t n, int L, int S, int64_t nf,
                                    double* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nf * L) return;
    int64_t f = idx / L;
    int k = (int)(idx % L);
    int64_t i = f * S + k;
    out[idx] = i < n ? x[i] : 0.0;
}

__global__ void preemph_kernel(const double* __restrict__ x, int64_t n, double a, double* __restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = preemph(x, n, i, a);
}

// rows of utterance u (foff[u] .. foff[u+1]) of a packed [F, W] block -> rows u * pad .. of a [U, pad, W] block;
// the padding rows are zero
__global__ void pad_rows_kernel(const double* __restrict__ src, const int64_t* __restrict__ foff, int W, int64_t pad,
                                double* __restrict__ dst) {
    const int u = blockIdx.y;
    const int64_t f0 = foff[u], nf = foff[u + 1] - f0;
    const int64_t n = pad * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / W;
        dst[(size_t)u * n + e] = r < nf ? src[(size_t)f0 * W + e] : 0.0;
    }
}

static int64_t num_frames_host(int64_t n, int L, int S) {
    // int(np.ceil(np.abs(n - (L - S)) / S))  (utils.py:119)
    int64_t a = n - (int64_t)(L - S);
    if (a < 0) a = -a;
    return (a + S - 1) / S;
}

}  // namespace lnasr

using namespace lnasr;

extern "C" {

int lnasr_num_frames(int64_t n_samples, int frame_len, int frame_step, int64_t* out) {
    LN_REQUIRE(out, "null out");
    LN_REQUIRE(frame_len >= 1 && frame_step >= 1 && n_samples >= 0, "bad framing parameters");
    *out = num_frames_host(n_samples, frame_len, frame_step);
    return 0;
}

int lnasr_preemphasis(lnasr_ctx* ctx, const double* x, int64_t n, double alpha, double* y) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (n <= 0) return 0;
    LN_REQUIRE(x && y, "null argument");
    const void* d_x;
    void* d_y;
    LN_TRY(stage_in(ctx, x, (size_t)n * 8, ctx->ws_in, &d_x));
    LN_TRY(stage_out(ctx, y, (size_t)n * 8, ctx->ws_out, &d_y));
    LN_LAUNCH(ctx, "preemphasis", preemph_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const double*)d_x,
              n, alpha, (double*)d_y);
    return finish_out(ctx, y, d_y, (size_t)n * 8);
}

int lnasr_split_frames(lnasr_ctx* ctx, const double* signal, int64_t n, int L, int S, double* frames) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    LN_REQUIRE(L >= 1 && S >= 1 && n >= 0, "bad framing parameters");
    const int64_t nf = num_frames_host(n, L, S);
    if (nf == 0) return 0;
    LN_REQUIRE(frames, "null frames");
    const void* d_x;
    void* d_o;
    LN_TRY(stage_in(ctx, signal, (size_t)n * 8, ctx->ws_in, &d_x));
    LN_TRY(stage_out(ctx, frames, (size_t)nf * L * 8, ctx->ws_out, &d_o));
    LN_LAUNCH(ctx, "split_frames", split_frames_kernel, dim3((unsigned)((nf * L + 255) / 256)), dim3(256), 0,
              (const double*)d_x, n, L, S, nf, (double*)d_o);
    return finish_out(ctx, frames, d_o, (size_t)nf * L * 8);
}

int lnasr_mfcc(lnasr_ctx* ctx, const void* signal, int sample_kind, const int64_t* offsets, int U,
               const lnasr_mfcc_cfg* cfg, const double* window, const double* fbank, double* power_or_null,
               double* cepstrum_or_null, double* features) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && cfg && offsets && window && fbank && features, "null argument");
    LN_REQUIRE(U >= 1, "U must be >= 1");
    LN_REQUIRE(sample_kind == 0 || sample_kind == 1, "sample_kind must be 0 (f64) or 1 (int16)");
    const int L = cfg->frame_len, S = cfg->frame_step, NFFT = cfg->fft_n, n_mel = cfg->n_mel, nc = cfg->n_cep;
    LN_REQUIRE(L >= 1 && S >= 1, "bad frame_len / frame_step");
    int log2n = 0;
    while ((1 << log2n) < NFFT) ++log2n;
    LN_REQUIRE(NFFT >= 8 && NFFT <= kMaxFft && (1 << log2n) == NFFT, "fft_n=%d must be a power of two in [8,%d]",
               NFFT, kMaxFft);
    LN_REQUIRE(n_mel >= 1 && n_mel <= 256 && nc >= 1 && nc + 1 <= n_mel, "bad n_mel=%d / n_cep=%d", n_mel, nc);
    LN_REQUIRE(!is_device_ptr(offsets), "offsets must be a host array");
    const int nbin = NFFT / 2 + 1, W = nc + 1;
    std::vector<int64_t>& foff = ctx->mfcc_foff;
    foff.assign(U + 1, 0);
    for (int u = 0; u < U; ++u) {
        const int64_t n = offsets[u + 1] - offsets[u];
        LN_REQUIRE(n >= 1, "utterance %d is empty", u);
        const int64_t nf = num_frames_host(n, L, S);
        // the reference indexes features[1] for the delta of row 0 (mfcc.py:164): one frame is an IndexError there
        LN_REQUIRE(nf >= 2, "utterance %d yields %lld frame(s); MFCC deltas need at least 2", u, (long long)nf);
        foff[u + 1] = foff[u] + nf;
    }
    const int64_t F = foff[U], NS = offsets[U] - offsets[0];
    LN_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
    const size_t ssz = sample_kind == 1 ? 2 : 8;
    const void *d_sig, *d_win, *d_fb;
    LN_TRY(stage_in(ctx, signal, (size_t)NS * ssz, ctx->ws_in, &d_sig));
    LN_TRY(stage_in(ctx, window, (size_t)L * 8, ctx->ws_in2, &d_win));
    LN_TRY(stage_in(ctx, fbank, (size_t)n_mel * nbin * 8, ctx->ws_in3, &d_fb));
    // constants that depend on (fft_n, n_mel) only: twiddles and the DCT matrix, built once and kept on the device
    const size_t nC = (size_t)n_mel * n_mel;
    if (ctx->mfcc_nfft != NFFT || ctx->mfcc_nmel != n_mel) {
        std::vector<double> cst(NFFT + nC);
        for (int k = 0; k < NFFT / 2; ++k) {
            const double a = 2.0 * M_PI * (double)k / (double)NFFT;
            cst[2 * k] = cos(a);
            cst[2 * k + 1] = -sin(a);
        }
        for (int k = 0; k < n_mel; ++k)
            for (int m = 0; m < n_mel; ++m) {
                // scipy.fftpack.dct(type=2, norm='ortho'): y_k = f_k * 2 sum_m x_m cos(pi k (2m+1) / (2N)),
                // f_0 = sqrt(1/(4N)), f_k = sqrt(1/(2N))
                const double f = k == 0 ? sqrt(1.0 / (4.0 * n_mel)) : sqrt(1.0 / (2.0 * n_mel));
                cst[NFFT + (size_t)k * n_mel + m] = 2.0 * f * cos(M_PI * (double)k * (2.0 * m + 1.0) / (2.0 * n_mel));
            }
        LN_TRY(ctx->mfcc_const.reserve(cst.size() * 8));
        LN_HIP(hipMemcpyAsync(ctx->mfcc_const.p, cst.data(), cst.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        LN_HIP(hipStreamSynchronize(ctx->stream));           // cst goes out of scope
        ctx->mfcc_nfft = NFFT;
        ctx->mfcc_nmel = n_mel;
    }
    double* d_tw = ctx->mfcc_const.as<double>();
    double* d_C = d_tw + NFFT;
    // per call: sample offsets, frame offsets, mel ranges
    const size_t cbytes = (size_t)(2 * (U + 1)) * 8 + (size_t)2 * n_mel * 4;
    LN_TRY(ctx->ws_in4.reserve(cbytes));
    int64_t* d_soff = ctx->ws_in4.as<int64_t>();
    int64_t* d_foff = d_soff + (U + 1);
    int* d_range = reinterpret_cast<int*>(d_foff + (U + 1));
    // Both offset arrays go up from a pinned buffer of the context, so the call does not wait for the copy: the
    // event tells the next call when the buffer may be rewritten (normally long before it asks).
    const size_t nmeta = (size_t)2 * (U + 1);
    if (ctx->mfcc_meta_ev) LN_HIP(hipEventSynchronize(ctx->mfcc_meta_ev));
    else LN_HIP(hipEventCreateWithFlags(&ctx->mfcc_meta_ev, hipEventDisableTiming));
    if (ctx->mfcc_meta_cap < nmeta) {
        if (ctx->mfcc_meta) LN_HIP(hipHostFree(ctx->mfcc_meta));
        ctx->mfcc_meta = nullptr;
        ctx->mfcc_meta_cap = 0;
        LN_HIP(hipHostMalloc((void**)&ctx->mfcc_meta, nmeta * 2 * 8, hipHostMallocDefault));
        ctx->mfcc_meta_cap = nmeta * 2;
    }
    memcpy(ctx->mfcc_meta, offsets, (size_t)(U + 1) * 8);
    memcpy(ctx->mfcc_meta + (U + 1), foff.data(), (size_t)(U + 1) * 8);
    LN_HIP(hipMemcpyAsync(d_soff, ctx->mfcc_meta, nmeta * 8, hipMemcpyHostToDevice, ctx->stream));   // d_foff follows d_soff
    LN_HIP(hipEventRecord(ctx->mfcc_meta_ev, ctx->stream));
    // workspaces: logmel [F,n_mel], cep [F,n_mel], energy [F], partial column sums [U,kMeanParts,ncols]
    LN_TRY(ctx->ws_misc.reserve(((size_t)F * n_mel * 2 + F + (size_t)U * n_mel * kMeanParts) * 8));
    double* logmel = ctx->ws_misc.as<double>();
    double* cep = logmel + (size_t)F * n_mel;
    double* energy = cep + (size_t)F * n_mel;
    double* mean_part = energy + F;                                     // [U, kMeanParts, ncols]
    void *d_pow = nullptr, *d_cep = nullptr, *d_feat;
    if (power_or_null) LN_TRY(stage_out(ctx, power_or_null, (size_t)F * nbin * 8, ctx->ws_out, &d_pow));
    if (cepstrum_or_null) LN_TRY(stage_out(ctx, cepstrum_or_null, (size_t)F * n_mel * 8, ctx->ws_out2, &d_cep));
    LN_TRY(stage_out(ctx, features, (size_t)F * 3 * W * 8, ctx->ws_out3, &d_feat));

    LN_LAUNCH(ctx, "mfcc_mel_range", mel_range_kernel, dim3(n_mel), dim3(64), 0, (const double*)d_fb, n_mel,
              nbin, d_range);
    const size_t shm = (size_t)(2 * NFFT + nbin + 32) * 8;
    const int nt = NFFT / 2 < 256 ? (NFFT / 2 < 64 ? 64 : NFFT / 2) : 256;
    bool fused_dct = false;
    const int n_out = cepstrum_or_null ? n_mel : W;   // cepstral columns the frame kernel has to produce
    if (NFFT == 512 && L <= 512 && n_mel <= 48 && !sw().no_fft512) {
        // four frames per wave, persistent waves walking groups of four frames with a grid stride; the LDS
        // (tables + a slice per wave) allows one block per CU
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        int64_t want = ((F + 3) / 4 + kX4Waves - 1) / kX4Waves;
        if (want > cus) want = cus;
        const unsigned grid = (unsigned)want;
#define LN_X4R(name, SAMPLE, POWER, NR, NW)                                                                           \
    LN_LAUNCH(ctx, name, (mfcc_frame512x4_kernel<SAMPLE, POWER, NR, NW>), dim3(grid), dim3(64 * NW), 0,              \
              (const SAMPLE*)d_sig, d_soff, d_foff, U, F, NS, L, S, n_mel, n_out, cfg->hpf_alpha, (const double*)d_win, \
              (const double*)d_fb, d_range, d_tw, d_C, (double*)d_pow, cep, energy)
#define LN_X4(name, SAMPLE, POWER)                                  \
    do {                                                            \
        if (L <= 416) LN_X4R(name, SAMPLE, POWER, 13, kX4Waves);    \
        else LN_X4R(name, SAMPLE, POWER, 16, kX4Waves);             \
    } while (0)
        if (sample_kind == 1 && d_pow) LN_X4("mfcc_frame512x4_i16", int16_t, true);
        else if (sample_kind == 1) LN_X4("mfcc_frame512x4_i16", int16_t, false);
        else if (d_pow) LN_X4("mfcc_frame512x4", double, true);
        else LN_X4("mfcc_frame512x4", double, false);
#undef LN_X4
#undef LN_X4R
        fused_dct = true;
    } else if (sample_kind == 1)
        LN_LAUNCH(ctx, "mfcc_frame_i16", mfcc_frame_kernel<int16_t>, dim3((unsigned)F), dim3(nt), shm,
                  (const int16_t*)d_sig, d_soff, d_foff, U, L, S, NFFT, log2n, n_mel, cfg->hpf_alpha,
                  (const double*)d_win, (const double*)d_fb, d_range, d_tw, (double*)d_pow, logmel, energy);
    else
        LN_LAUNCH(ctx, "mfcc_frame", mfcc_frame_kernel<double>, dim3((unsigned)F), dim3(nt), shm,
                  (const double*)d_sig, d_soff, d_foff, U, L, S, NFFT, log2n, n_mel, cfg->hpf_alpha,
                  (const double*)d_win, (const double*)d_fb, d_range, d_tw, (double*)d_pow, logmel, energy);
    {
        // cep_raw = DCT(log-mel) per frame; its per-utterance column means carry both of the reference's
        // utterance-level normalisations (mfcc.py:151,156) because the DCT is linear
        int threads = 256 - 256 % n_mel;
        if (threads < n_mel) threads = n_mel;
        const int rows = threads / n_mel;
        if (!fused_dct)
            LN_LAUNCH(ctx, "mfcc_dct", dct_kernel, dim3((unsigned)((F + rows - 1) / rows)), dim3(threads),
                      (size_t)threads * 8, logmel, d_foff, U, F, n_mel, d_C, cep);
    }
    // the fused kernels write rows of n_out columns (the ones in use), the DCT kernel rows of n_mel
    const int ncols = n_out, ldc = fused_dct ? n_out : n_mel;
    LN_REQUIRE(U <= 65535, "U=%d utterances is too many for one MFCC call", U);
    {
        // four row slices whatever the column count: the order of the sums (and so every bit of the features) does
        // not depend on whether the cepstrum was asked for
        const int threads = 4 * ncols;
        LN_LAUNCH(ctx, "mfcc_col_sum", col_sum_kernel, dim3(U, kMeanParts), dim3(threads), (size_t)threads * 8, cep, d_foff,
                  ldc, ncols, mean_part);
    }
    {
        int64_t maxnf = 0;
        for (int u = 0; u < U; ++u) maxnf = std::max(maxnf, foff[u + 1] - foff[u]);
        LN_REQUIRE((size_t)maxnf * ncols + 256 < 0xFFFFFFFFull, "utterance of %lld frames is too long for one MFCC call",
                   (long long)maxnf);
        LN_LAUNCH(ctx, "mfcc_feat", feat_kernel, dim3((unsigned)(((size_t)maxnf * ncols + 255) / 256), U), dim3(256), 0, cep,
                  mean_part, energy, d_foff, n_mel, nc, ncols, ldc, (double*)d_cep, (double*)d_feat);
    }
    if (power_or_null) LN_TRY(finish_out(ctx, power_or_null, d_pow, (size_t)F * nbin * 8));
    if (cepstrum_or_null) LN_TRY(finish_out(ctx, cepstrum_or_null, d_cep, (size_t)F * n_mel * 8));
    LN_TRY(finish_out(ctx, features, d_feat, (size_t)F * 3 * W * 8));
    return 0;
}

#ifdef LNASR_MFCC_STAMPS
// developer diagnostic only (not in the header): read and clear the phase stamps
int lnasr_dbg_mfcc_stamps(lnasr_ctx* ctx, unsigned long long* out16) {
    LN_ENTER(ctx);
    LN_HIP(hipDeviceSynchronize());
    LN_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mfcc_stamps), 16 * 8));
    unsigned long long z[16] = {0};
    LN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mfcc_stamps), z, 16 * 8));
    return 0;
}
int lnasr_dbg_mfcc_wave_times(lnasr_ctx* ctx, unsigned long long* out8192) {
    LN_ENTER(ctx);
    LN_HIP(hipDeviceSynchronize());
    LN_HIP(hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_mfcc_wave_times), 8192 * 8));
    return 0;
}
#endif

int lnasr_pad_rows(lnasr_ctx* ctx, const double* packed, const int64_t* row_offsets, int U, int W, int64_t pad_rows,
                   double* padded) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && packed && row_offsets && padded, "null argument");
    LN_REQUIRE(U >= 1 && U <= 65535 && W >= 1 && pad_rows >= 1, "bad U=%d / W=%d / pad_rows=%lld", U, W, (long long)pad_rows);
    LN_REQUIRE(!is_device_ptr(row_offsets), "row_offsets must be a host array");
    LN_REQUIRE(row_offsets[0] == 0, "row_offsets[0] must be 0");
    for (int u = 0; u < U; ++u) {
        const int64_t nf = row_offsets[u + 1] - row_offsets[u];
        LN_REQUIRE(nf >= 0 && nf <= pad_rows, "utterance %d has %lld rows, pad_rows is %lld", u, (long long)nf, (long long)pad_rows);
    }
    const int64_t F = row_offsets[U];
    const void* d_src;
    void* d_dst;
    LN_TRY(stage_in(ctx, packed, (size_t)F * W * 8, ctx->ws_in, &d_src));
    LN_TRY(stage_out(ctx, padded, (size_t)U * pad_rows * W * 8, ctx->ws_out, &d_dst));
    LN_TRY(ctx->ws_in4.reserve((size_t)(U + 1) * 8));
    LN_HIP(hipMemcpyAsync(ctx->ws_in4.p, row_offsets, (size_t)(U + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    LN_HIP(hipStreamSynchronize(ctx->stream));           // row_offsets belongs to the caller
    const unsigned gx = (unsigned)std::min<int64_t>((pad_rows * W + 255) / 256, 1024);
    LN_LAUNCH(ctx, "pad_rows", pad_rows_kernel, dim3(gx, U), dim3(256), 0, (const double*)d_src, ctx->ws_in4.as<int64_t>(), W,
              pad_rows, (double*)d_dst);
    return finish_out(ctx, padded, d_dst, (size_t)U * pad_rows * W * 8);
}

}  // extern "C"
