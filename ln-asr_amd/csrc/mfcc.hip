// MFCC front end (mfcc.py:108-175, utils.py:106-138): pre-emphasis -> framing -> Hamming ->
// rFFT -> power -> mel filter bank -> 20 log10 -> mean norm -> DCT-II (ortho) -> cepstral mean
// norm -> log-energy -> delta / delta-delta.  fp64 throughout, utterances batched by offsets.
//
// Kernels
//   mfcc_frame_kernel   one block per frame: samples -> power [nbin], log-mel [n_mel], log-energy
//   col_mean_kernel     per-utterance column means (the reference's two utterance-level means)
//   dct_kernel          per frame: DCT-II of the mean-normalised log-mel row
//   feat_kernel         per frame: second mean norm (applied to the returned cepstrum as well, as the
//                       reference does through a view, mfcc.py:155-156), energy, delta, delta-delta
#include <algorithm>

#include "common.h"
#include "devmath.h"

namespace lnasr {

constexpr int kMaxFft = 4096;

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

// fft_n = 512 fast path: ONE WAVE PER FRAME, no block-level barriers.
//   real 512-point FFT = 256-point complex FFT of z[n] = y[2n] + i y[2n+1] (Stockham radix-4, four
//   stages, one radix-4 butterfly per lane per stage, ping-pong buffers in the wave's private LDS
//   slice) followed by the even/odd split  X[k] = E[k] + W512^k O[k].
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

constexpr int kMfccWaves = 12;           // waves per block of the fft-512 frame kernel (they share ~35 KB of tables)

template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const T*)(((uint64_t)hi << 32) | lo);
}

// raw samples x[i-1], x[i], x[i+1] for the four point pairs of a lane
struct RawFrame {
    double xm[4], x0[4], x1[4];
    int64_t base, n;
};

// Buffer loads: the frame's base address and the bytes left in the utterance are wave-uniform (SGPRs), one
// VGPR holds the lane offset, the four pair offsets are immediates, and reads past the end of the utterance
// return 0 without per-lane clamping.  x[i-1] uses a second descriptor starting one sample earlier (at the
// utterance start, where there is no earlier sample, the first read goes out of range and returns 0).
template <typename SAMPLE>
__device__ __forceinline__ void fetch_frame(const SAMPLE* __restrict__ x, int64_t n, int64_t base, int lane,
                                            RawFrame& rw) {
    constexpr int SZ = sizeof(SAMPLE);
    rw.base = base;
    rw.n = n;
    int64_t left = (n - base) * SZ;
    left = left < 0 ? 0 : (left > 0x7FFFFFF0L ? 0x7FFFFFF0L : left);
    const int64_t back = base > 0 ? 1 : 0;
    // everything in the descriptors is wave-uniform; say so, or the loads are wrapped in waterfall loops
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(x + base), 0, __builtin_amdgcn_readfirstlane((int)left), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)uniform_ptr(x + base - back), 0, __builtin_amdgcn_readfirstlane((int)(left + back * SZ)), 0x00020000);
    const int vo = lane * 2 * SZ;
    // x[i-1]: at base == 0 descriptor rb starts at x[0] and the offsets move back by one sample; the one
    // sample that does not exist (x[-1], never used: y[0] = x[0]) is clamped onto x[0] so no offset is negative
    const int adj = back ? 0 : SZ;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int vb = vo + (r * 128 * SZ - adj);
        if (r == 0) vb = vb < 0 ? 0 : vb;
        if constexpr (SZ == 8) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 q = __builtin_amdgcn_raw_buffer_load_b128(ra, vo + r * 128 * SZ, 0, 0);
            rw.x0[r] = __hiloint2double((int)q[1], (int)q[0]);
            rw.x1[r] = __hiloint2double((int)q[3], (int)q[2]);
            rw.xm[r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rb, vb, 0, 0));
        } else {
            // two 16-bit loads, not one dword: the range check is per load, and a dword holding the last sample
            // of an odd-length utterance would be rejected whole
            rw.x0[r] = (double)(short)__builtin_amdgcn_raw_buffer_load_b16(ra, vo + r * 128 * SZ, 0, 0);
            rw.x1[r] = (double)(short)__builtin_amdgcn_raw_buffer_load_b16(ra, vo + r * 128 * SZ + SZ, 0, 0);
            rw.xm[r] = (double)(short)__builtin_amdgcn_raw_buffer_load_b16(rb, vb, 0, 0);
        }
    }
}

template <typename SAMPLE>
__global__ __launch_bounds__(64 * kMfccWaves) __attribute__((amdgpu_waves_per_eu(3, 3))) void mfcc_frame512_kernel(
    const SAMPLE* __restrict__ signal, const int64_t* __restrict__ soff, const int64_t* __restrict__ foff, int U,
    int64_t F, int L, int S, int n_mel, int n_out, double alpha, const double* __restrict__ window,
    const double* __restrict__ fbank, const int* __restrict__ mel_range, const double* __restrict__ tw512,
    const double* __restrict__ dctC, double* __restrict__ power_out, double* __restrict__ cepraw,
    double* __restrict__ energy) {
    constexpr int NC = 256, NBIN = 257, MAXTAP = 1024, MAXMEL = 48;
    __shared__ double tws[2 * NC];                 // tw512, shared by the block
    __shared__ double wins[2 * NC];                // window (L <= 512)
    __shared__ double taps[MAXTAP];                // non-zero filter taps, filter after filter
    __shared__ double Ct[MAXMEL * MAXMEL];         // DCT matrix, transposed: Ct[m][k] = C[k][m]
    __shared__ int tap_off[MAXMEL + 1];
    __shared__ int asg[4 * 64];                    // mel work split: per lane (filter, first tap, count, chain)
    __shared__ int asg_cmax;
    // per wave: FFT buffer (re, im interleaved; padding against bank conflicts was measured: no gain), then
    // power | log-mel
    __shared__ double bufs[kMfccWaves][2 * NC];
    constexpr int NT = 64 * kMfccWaves;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 2 * NC; e += NT) {
        tws[e] = tw512[e];
        wins[e] = e < L ? window[e] : 0.0;
    }
    for (int e = tid; e < n_mel * n_mel; e += NT) Ct[(e % n_mel) * n_mel + e / n_mel] = dctC[e];
    if (tid < n_mel) tap_off[tid + 1] = mel_range[2 * tid + 1] - mel_range[2 * tid];
    __syncthreads();
    if (tid == 0) {
        // exclusive prefix of the tap counts, then the lane assignment of the mel phase: a filter of cnt taps
        // gets ceil(cnt / C) consecutive lanes (at most 3), C the smallest chunk for which 64 lanes suffice.
        // C = 0 (keep one filter per lane) when that is impossible.
        int o = 0, cmx = 0;
        for (int m = 0; m < n_mel; ++m) {
            const int c = tap_off[m + 1];
            tap_off[m] = o;
            o += c;
            cmx = c > cmx ? c : cmx;
        }
        tap_off[n_mel] = o;
        int C = (cmx + 2) / 3;
        if (C < 1) C = 1;
        for (;; ++C) {
            int need = 0;
            for (int m = 0; m < n_mel; ++m) {
                const int c = tap_off[m + 1] - tap_off[m];
                need += c > 0 ? (c + C - 1) / C : 1;
            }
            if (need <= 64) break;
        }
        int ln = 0;
        for (int m = 0; m < n_mel; ++m) {
            const int c = tap_off[m + 1] - tap_off[m];
            const int parts = c > 0 ? (c + C - 1) / C : 1;
            for (int q = 0; q < parts; ++q, ++ln) {
                const int i0 = q * C, cnt = (c - i0) < C ? (c - i0) : C;
                asg[4 * ln] = m;
                asg[4 * ln + 1] = i0;
                asg[4 * ln + 2] = cnt > 0 ? cnt : 0;
                asg[4 * ln + 3] = (q == 0 ? 1 : 0) | ((q == 0 && parts > 1) ? 2 : 0) | ((q == 0 && parts > 2) ? 4 : 0);
            }
        }
        for (; ln < 64; ++ln) asg[4 * ln] = -1, asg[4 * ln + 1] = 0, asg[4 * ln + 2] = 0, asg[4 * ln + 3] = 0;
        asg_cmax = C;
    }
    __syncthreads();
    const bool taps_in_lds = tap_off[n_mel] <= MAXTAP;
    if (taps_in_lds)
        for (int m = w; m < n_mel; m += kMfccWaves) {
            const int lo = mel_range[2 * m], cnt = mel_range[2 * m + 1] - lo;
            for (int e = lane; e < cnt; e += 64) taps[tap_off[m] + e] = fbank[(size_t)m * NBIN + lo + e];
        }
    __syncthreads();                               // last block-level barrier: waves are independent from here
    // this lane's share of the mel phase (constant over frames)
    const int my_m = asg[4 * lane], my_i0 = asg[4 * lane + 1], my_cnt = asg[4 * lane + 2], my_flags = asg[4 * lane + 3];
    const int cmax = asg_cmax;
    const int my_lo = my_m >= 0 ? mel_range[2 * my_m] + my_i0 : 0;
    const double* my_tp = taps + (my_m >= 0 ? tap_off[my_m] + my_i0 : 0);
    const double* my_fb = fbank + (size_t)(my_m >= 0 ? my_m : 0) * NBIN + my_lo;
    // DCT split: n_out <= 16 outputs x 4 quarter-sums fill the wave; otherwise one output per lane
    const bool dct_split = n_out * 4 <= 64;
    const int dq = (n_mel + 3) / 4;

    const int64_t stride = (int64_t)gridDim.x * kMfccWaves;
    int64_t f = (int64_t)blockIdx.x * kMfccWaves + w;
    if (f >= F) return;
    int u = find_utt(foff, U, f);
    RawFrame cur;
    auto fetch = [&](int64_t ff) {
        while (foff[u + 1] <= ff) ++u;             // frames visited by a wave are increasing
        fetch_frame(signal + soff[u], soff[u + 1] - soff[u], (ff - foff[u]) * S, lane, cur);
    };
    fetch(f);
    double* A = &bufs[w][0];
    double* lm = A + 320;                          // log-mel row: beyond the 257 power bins
    for (; f < F; f += stride) {
    // stage 0 (Ns = 1): v[r] = z[lane + 64 r] = (y[2n], y[2n+1]) * window, y = pre-emphasised signal
    cplx v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k0 = 2 * (lane + 64 * r);
        const int64_t i0 = cur.base + k0;
        // y[i] = x[i] - alpha x[i-1] (mfcc.py:63), y[0] = x[0]; zero beyond the utterance (utils.py:121-125)
        double a = i0 == 0 ? cur.x0[r] : __dsub_rn(cur.x0[r], __dmul_rn(alpha, cur.xm[r]));
        double c = __dsub_rn(cur.x1[r], __dmul_rn(alpha, cur.x0[r]));
        if (i0 >= cur.n) a = 0.0;
        if (i0 + 1 >= cur.n) c = 0.0;
        v[r] = {a * wins[k0], c * wins[k0 + 1]};   // wins is zero beyond the frame length
    }
    radix4(v);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        A[2 * (4 * lane + r)] = v[r].re;
        A[2 * (4 * lane + r) + 1] = v[r].im;
    }
    __builtin_amdgcn_wave_barrier();
    // stages Ns = 4, 16, 64: all reads of a stage happen before its writes, so one buffer is enough
#pragma unroll
    for (int Ns = 4; Ns < NC; Ns *= 4) {
        const int k = lane & (Ns - 1);
        const int step = NC / (Ns * 4);            // twiddle index stride in units of 2 pi / 256
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = lane + 64 * r;
            cplx z = {A[2 * src], A[2 * src + 1]};
            if (r > 0) {
                const int m2 = 2 * k * r * step;   // W256^m = tw512[2m]; the table holds half a turn:
                const int idx = m2 & (NC - 1);     // tw512[i + 256] = -tw512[i]
                const double sg = m2 >= NC ? -1.0 : 1.0;
                const cplx tw = {sg * tws[2 * idx], sg * tws[2 * idx + 1]};
                z = cmul(z, tw);
            }
            v[r] = z;
        }
        radix4(v);
        __builtin_amdgcn_wave_barrier();
        const int j0 = (lane / Ns) * Ns * 4 + k;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dst = j0 + r * Ns;
            A[2 * dst] = v[r].re;
            A[2 * dst + 1] = v[r].im;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // Z = A[0..255]; X[k] for k = 0..256; power |X|^2 / 512 (mfcc.py:141-142)
    double pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        const int kb = (NC - k) & (NC - 1);
        const cplx zk = {A[2 * k], A[2 * k + 1]};
        const cplx zc = {A[2 * kb], -A[2 * kb + 1]};
        const cplx e = {0.5 * (zk.re + zc.re), 0.5 * (zk.im + zc.im)};
        const cplx d = csub(zk, zc);
        const cplx o = {0.5 * d.im, -0.5 * d.re};                    // -(i/2) (zk - zc)
        const cplx tw = {tws[2 * k], tws[2 * k + 1]};
        const cplx xk = cadd(e, cmul(o, tw));
        // |X|^2 directly (the reference squares np.abs: same value to an ulp, a fraction of the instructions)
        pk[q] = (1.0 / 512.0) * fma(xk.re, xk.re, xk.im * xk.im);
    }
    // bin 256 (Nyquist): X = Re Z0 - Im Z0, the value the general expression gives with tw = -1
    const double xn = A[0] - A[1];
    __builtin_amdgcn_wave_barrier();
    double* pw = A;                                // the FFT buffer is dead: reuse it for the power spectrum
    double esum = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        pw[k] = pk[q];
        esum += pk[q];
        if (power_out) power_out[f * NBIN + k] = pk[q];
    }
    if (lane == 0) {
        const double pn = (1.0 / 512.0) * (xn * xn);
        pw[NC] = pn;
        esum += pn;
        if (power_out) power_out[f * NBIN + NC] = pn;
    }
    esum = wave_sum_dpp(esum);
    if (lane == 0) energy[f] = log(esum);                               // mfcc.py:158-160
    __builtin_amdgcn_wave_barrier();
    // Request the wave's next frame here, not right after stage 0: the FFT is the register-hungry phase, and
    // values prefetched across it get spilled -- which forces the loads to complete on the spot.  The mel and
    // DCT phases are long enough to cover the latency and need few registers.
    if (f + stride < F) fetch(f + stride);
    if (cmax > 0) {
        // mel filter bank (mfcc.py:148): every lane sums its chunk of one filter's taps; a filter's two or
        // three partial sums are added in the lane of its first chunk
        double s = 0.0;
        const double* pp = pw + my_lo;
        // branch-free: past the lane's count the index falls back to 0 and the tap to 0.0
        if (taps_in_lds) {
#pragma unroll 4
            for (int i = 0; i < cmax; ++i) {
                const bool in = i < my_cnt;
                const int ii = in ? i : 0;
                s = fma(pp[ii], in ? my_tp[ii] : 0.0, s);
            }
        } else {
            for (int i = 0; i < cmax; ++i) {
                const bool in = i < my_cnt;
                const int ii = in ? i : 0;
                s = fma(pp[ii], in ? my_fb[ii] : 0.0, s);
            }
        }
        const double s1 = shfl_f64(s, (lane + 1) & 63), s2 = shfl_f64(s, (lane + 2) & 63);
        if (my_flags & 2) s += s1;
        if (my_flags & 4) s += s2;
        if (my_flags & 1) {
            if (s == 0.0) s = 2.220446049250313e-16;                     // mfcc.py:149
            lm[my_m] = 20.0 * log10(s);                                  // mfcc.py:150
        }
    }
    __builtin_amdgcn_wave_barrier();
    // DCT-II (ortho) of the RAW log-mel row.  The reference subtracts the utterance mean first
    // (mfcc.py:151-152); the DCT is linear, so that subtraction is applied to the transformed
    // row by feat_kernel and the frame kernel needs no utterance-level quantity.
    if (dct_split) {
        const int k = lane >> 2, part = lane & 3;
        const int m0 = part * dq, m1 = (m0 + dq) < n_mel ? (m0 + dq) : n_mel;
        double s = 0.0;
        if (k < n_out)
            for (int m = m0; m < m1; ++m) s = fma(lm[m], Ct[m * n_mel + k], s);
        s += dpp_f64<0xB1>(s);                     // quad_perm [1,0,3,2], then [2,3,0,1]: (q0 + q1) + (q2 + q3)
        s += dpp_f64<0x4E>(s);
        if (part == 0 && k < n_out) cepraw[f * n_mel + k] = s;
    } else {
        // same association as the split form (four quarter sums, pairwise) so both give identical bits
        for (int k = lane; k < n_out; k += 64) {
            double q[4];
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const int m0 = part * dq, m1 = (m0 + dq) < n_mel ? (m0 + dq) : n_mel;
                q[part] = 0.0;
                for (int m = m0; m < m1; ++m) q[part] = fma(lm[m], Ct[m * n_mel + k], q[part]);
            }
            cepraw[f * n_mel + k] = (q[0] + q[1]) + (q[2] + q[3]);
        }
    }
    __builtin_amdgcn_wave_barrier();               // the next frame overwrites this wave's buffer
    }
}

// mean over the frames of each utterance of columns [c0, c0+nc) of X [F, ld]
__global__ void col_mean_kernel(const double* __restrict__ X, const int64_t* __restrict__ foff, int ld, int c0,
                                int nc, double* __restrict__ out) {
    extern __shared__ double sm[];      // [slices][nc]
    const int u = blockIdx.x;
    const int64_t f0 = foff[u], f1 = foff[u + 1];
    const int slices = blockDim.x / nc;
    const int c = threadIdx.x % nc, sl = threadIdx.x / nc;
    double s = 0.0;
    if (sl < slices)
        for (int64_t f = f0 + sl; f < f1; f += slices) s += X[f * ld + c0 + c];
    if (sl < slices) sm[sl * nc + c] = s;
    __syncthreads();
    if (threadIdx.x < nc) {
        double tot = 0.0;
        for (int k = 0; k < slices; ++k) tot += sm[k * nc + threadIdx.x];
        out[(size_t)u * nc + threadIdx.x] = tot / (double)(f1 - f0);
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
                            int ncols, double* __restrict__ cep_out, double* __restrict__ feat) {
    // grid.y = utterance (no search for the frame's utterance); ncols = n_mel when the cepstrum is returned,
    // nc + 1 (the feature columns) otherwise
    const int W = nc + 1;
    const int u = blockIdx.y;
    const int64_t f0 = foff[u], f1 = foff[u + 1];
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t f = f0 + idx / ncols;
    const int k = (int)(idx % ncols);
    if (f >= f1) return;
    const double* Mu = M + (size_t)u * n_mel;
    const double shift0 = 1e-8 * sqrt((double)n_mel);
    auto cepv = [&](int64_t ff, int kk) -> double {
        double v = cepraw[ff * n_mel + kk] - Mu[kk];
        if (kk == 0) v -= shift0;
        if (kk >= 1 && kk <= nc) v -= 1e-8;
        return v;
    };
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
    LN_HIP(hipMemcpyAsync(d_soff, offsets, (size_t)(U + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    LN_HIP(hipMemcpyAsync(d_foff, foff.data(), (size_t)(U + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    // `offsets` belongs to the caller and `foff` is rewritten by the next call: both uploads are over before any
    // kernel is queued (the kernels themselves stay asynchronous, like every other entry point)
    LN_HIP(hipStreamSynchronize(ctx->stream));
    // workspaces: logmel [F,n_mel], cep [F,n_mel], energy [F], mean1 [U,n_mel], mean2 [U,nc]
    LN_TRY(ctx->ws_misc.reserve(((size_t)F * n_mel * 2 + F + (size_t)U * n_mel + (size_t)U * nc) * 8));
    double* logmel = ctx->ws_misc.as<double>();
    double* cep = logmel + (size_t)F * n_mel;
    double* energy = cep + (size_t)F * n_mel;
    double* mean1 = energy + F;
    double* mean2 = mean1 + (size_t)U * n_mel;
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
    if (NFFT == 512 && L <= 512 && n_mel <= 48 && !getenv("LNASR_NO_FFT512")) {
        // persistent waves, each walking frames with a grid stride
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        int64_t want = (F + kMfccWaves - 1) / kMfccWaves;
        static int bpc_cache[2] = {0, 0};             // resident blocks per CU: the whole grid runs in one round
        int& bpc = bpc_cache[sample_kind];
        if (bpc == 0) {                               // (the occupancy query costs ~2 ms: once per process)
            if (sample_kind == 1)
                LN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, mfcc_frame512_kernel<int16_t>, 64 * kMfccWaves, 0));
            else
                LN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, mfcc_frame512_kernel<double>, 64 * kMfccWaves, 0));
        }
        if (bpc < 1) bpc = 1;
        if (want > (int64_t)cus * bpc) want = (int64_t)cus * bpc;
        const unsigned grid = (unsigned)want;
        if (sample_kind == 1)
            LN_LAUNCH(ctx, "mfcc_frame512_i16", mfcc_frame512_kernel<int16_t>, dim3(grid), dim3(64 * kMfccWaves), 0,
                      (const int16_t*)d_sig, d_soff, d_foff, U, F, L, S, n_mel, n_out, cfg->hpf_alpha, (const double*)d_win,
                      (const double*)d_fb, d_range, d_tw, d_C, (double*)d_pow, cep, energy);
        else
            LN_LAUNCH(ctx, "mfcc_frame512", mfcc_frame512_kernel<double>, dim3(grid), dim3(64 * kMfccWaves), 0,
                      (const double*)d_sig, d_soff, d_foff, U, F, L, S, n_mel, n_out, cfg->hpf_alpha, (const double*)d_win,
                      (const double*)d_fb, d_range, d_tw, d_C, (double*)d_pow, cep, energy);
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
        LN_LAUNCH(ctx, "mfcc_col_mean", col_mean_kernel, dim3(U), dim3(threads), (size_t)threads * 8, cep, d_foff,
                  n_mel, 0, n_mel, mean1);
    }
    {
        int64_t maxnf = 0;
        for (int u = 0; u < U; ++u) maxnf = std::max(maxnf, foff[u + 1] - foff[u]);
        const int ncols = d_cep ? n_mel : W;
        LN_REQUIRE(U <= 65535, "U=%d utterances is too many for one MFCC call", U);
        LN_LAUNCH(ctx, "mfcc_feat", feat_kernel, dim3((unsigned)(((size_t)maxnf * ncols + 255) / 256), U), dim3(256), 0, cep,
                  mean1, energy, d_foff, n_mel, nc, ncols, (double*)d_cep, (double*)d_feat);
    }
    if (power_or_null) LN_TRY(finish_out(ctx, power_or_null, d_pow, (size_t)F * nbin * 8));
    if (cepstrum_or_null) LN_TRY(finish_out(ctx, cepstrum_or_null, d_cep, (size_t)F * n_mel * 8));
    LN_TRY(finish_out(ctx, features, d_feat, (size_t)F * 3 * W * 8));
    return 0;
}

}  // extern "C"
