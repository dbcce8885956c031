// Does fp64 VALU work overlap with fp64 MFMA on gfx950?  (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only; 1: VALU only; 2: both interleaved in one wave (1 MFMA : NV VALU)
template <int MODE, int NV, bool F32>
__global__ void k(double* out, int iters, double a, double b) {
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double x = a + threadIdx.x, y = b;
    double v[8];
    float vf[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; vf[i] = (float)(a + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE != 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    if (F32) vf[j % 8] = fmaf(vf[j % 8], (float)b, (float)a);
                    else v[j % 8] = fma(v[j % 8], b, a);
                }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + vf[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// two waves per SIMD: waves 0-3 do MFMA, waves 4-7 do VALU
template <int NV>
__global__ void k_split(double* out, int iters, double a, double b) {
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double x = a + threadIdx.x, y = b;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const bool mf = (threadIdx.x >> 8) == 0;
    if (mf) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j % 8] = fma(v[j % 8], b, a);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// two waves per SIMD: waves 0-3 do MFMA, waves 4-7 do KIND 1: f32 FMA, 2: 32-bit integer add/xor, 3: nothing (exit)
template <int NV, int KIND, bool MF = true>
__global__ void k_split2(double* out, int iters, double a, double b) {
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double x = a + threadIdx.x, y = b;
    float vf[8];
    unsigned vi[8];
    for (int i = 0; i < 8; ++i) { vf[i] = (float)(a + i); vi[i] = threadIdx.x * 7 + i; }
    const bool mf = (threadIdx.x >> 8) == 0;
    const unsigned ia = (unsigned)iters | 1u, ib = (unsigned)(a * 3.0) + 5u;
    if (mf) {
        if (MF)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    } else if (KIND != 3) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    if (KIND == 1) vf[j % 8] = fmaf(vf[j % 8], (float)b, (float)a);
                    else {
                        unsigned t;
                        asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(vi[j % 8]), "v"(ia));
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(vi[j % 8]) : "v"(t), "v"(ib));
                    }
                }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += vf[i] + (double)vi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
#define RUN(label, kern, threads)                                                                    \
    printf("%-48s %.3f ms\n", label, timeit([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0, 0.999); }))

int main() {
    double* out;
    (void)hipMalloc(&out, 8 << 20);
    const int iters = 50000;
    RUN("1 wave/SIMD  MFMA only", (k<0, 8, false>), 256);
    RUN("1 wave/SIMD  VALU f64 only (8 per slot)", (k<1, 8, false>), 256);
    RUN("1 wave/SIMD  MFMA + 8 f64 VALU interleaved", (k<2, 8, false>), 256);
    RUN("1 wave/SIMD  MFMA + 4 f64 VALU interleaved", (k<2, 4, false>), 256);
    RUN("1 wave/SIMD  VALU f32 only (8 per slot)", (k<1, 8, true>), 256);
    RUN("1 wave/SIMD  MFMA + 8 f32 VALU interleaved", (k<2, 8, true>), 256);
    RUN("2 waves/SIMD MFMA wave || f64 VALU wave (8)", (k_split<8>), 512);
    RUN("2 waves/SIMD MFMA wave || f64 VALU wave (4)", (k_split<4>), 512);
    RUN("2 waves/SIMD MFMA wave || idle wave", (k_split2<8, 3>), 512);
    RUN("2 waves/SIMD MFMA wave || f32 FMA wave (8)", (k_split2<8, 1>), 512);
    RUN("2 waves/SIMD MFMA wave || int add+xor wave (4x2)", (k_split2<4, 2>), 512);
    RUN("2 waves/SIMD MFMA wave || int add+xor wave (8x2)", (k_split2<8, 2>), 512);
    RUN("int add+xor wave alone (8x2)", (k_split2<8, 2, false>), 512);
    RUN("f32 FMA wave alone (8)", (k_split2<8, 1, false>), 512);
    return 0;
}
