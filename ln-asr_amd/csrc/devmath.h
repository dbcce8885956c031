// Device-side helpers: wave64 reductions, log-sum-exp pieces, storage conversion.
#pragma once
#include <hip/hip_runtime.h>

namespace lnasr {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;   // ln(2 pi)

#define LN_NEG_INF (-__builtin_huge_val())

template <typename TS>
__device__ __forceinline__ double ld(const TS* p, size_t i) { return (double)p[i]; }
template <typename TS>
__device__ __forceinline__ void st(TS* p, size_t i, double v) { p[i] = (TS)v; }
// value as it reads back after being stored in TS
template <typename TS>
__device__ __forceinline__ double rnd(double v) { return (double)(TS)v; }

// one-instruction max: the library fmax first canonicalises operands the compiler cannot prove quiet
// (MFMA results, LDS / global loads), which doubles the instruction count of max chains
__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, shfl_xor_f64(v, m));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

// (value, index) arg-max with numpy.argmax semantics: first index among equal maxima.
__device__ __forceinline__ void argmax_combine(double& v, int& i, double ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}
__device__ __forceinline__ void wave_argmax(double& v, int& i) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double ov = shfl_xor_f64(v, m);
        int oi = __shfl_xor(i, m, 64);
        argmax_combine(v, i, ov, oi);
    }
}

// Block-wide reductions through a small LDS scratch (>= 2 * nwaves doubles). All threads must call.
__device__ __forceinline__ double block_max(double v, double* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    if (nw == 1) return v;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = scratch[0];
    for (int k = 1; k < nw; ++k) r = fmax(r, scratch[k]);
    return r;
}
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (nw == 1) return v;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = scratch[0];
    for (int k = 1; k < nw; ++k) r += scratch[k];
    return r;
}

// Online log-sum-exp accumulator (max-shifted; an all -inf input yields -inf like the
// reference's logaddexp fold, utils.py:176-179).
struct Lse {
    double m = LN_NEG_INF, s = 0.0;
    __device__ __forceinline__ void add(double z) {
        if (z == LN_NEG_INF) return;
        if (z > m) {
            s = s * exp(m - z) + 1.0;   // m = -inf: s is 0, exp(-inf) = 0
            m = z;
        } else {
            s += exp(z - m);
        }
    }
    __device__ __forceinline__ void merge(double om, double os) {
        if (om == LN_NEG_INF) return;
        if (om > m) {
            s = s * exp(m - om) + os;
            m = om;
        } else {
            s += os * exp(om - m);
        }
    }
    __device__ __forceinline__ double value() const { return (m == LN_NEG_INF) ? m : m + log(s); }
};

}  // namespace lnasr
