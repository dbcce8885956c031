// How fast do i8 MFMAs (v_mfma_i32_16x16x64_i8) issue on gfx950 with an LDS operand read per four of them, and how many
// fp64 vector instructions fit beside them?  (development aid for the sliced-integer emission kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4_t __attribute__((ext_vector_type(4)));

template <int NV, int WAVES, bool LDSREAD, bool MF = true, int KIND = 0>
__global__ __launch_bounds__(64 * WAVES) void k(double* out, unsigned long long* cyc, int iters, double a, double b) {
    __shared__ int4_t F[2048];
    for (int i = threadIdx.x; i < 2048; i += 64 * WAVES) F[i] = int4_t{i * 3 + 1, i * 5 + 2, i * 7 + 3, i * 11 + 4};
    __syncthreads();
    int4_t W[16];
    for (int i = 0; i < 16; ++i) W[i] = int4_t{(int)threadIdx.x * 13 + i, i * 17 + 5, (int)threadIdx.x + 7 * i, 3 * i + 1};
    int4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = int4_t{0, 0, 0, 0};
    double v[8];
    float vf[8];
    unsigned vi[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i + threadIdx.x; vf[i] = (float)v[i]; vi[i] = threadIdx.x * 3 + i; }
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    int4_t f = F[lane];
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int U = 0; U < 4; ++U) {
        int4_t fn = f;
        if (LDSREAD) fn = F[((it + U + 1) * 64 + lane) & 2047];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MF) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f, W[t + 4 * U], acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = (t * NV + j) % 8;
                if (KIND == 0) v[c] = fma(v[c], b, a);
                if (KIND == 1) vf[c] = fmaf(vf[c], (float)b, (float)a);
                if (KIND == 2) vi[c] = (vi[c] << 7) + (unsigned)iters;
            }
        }
        f = fn;
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + vf[i] + vi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

#define RUN(label, kern, threads)                                                                            \
    do {                                                                                                     \
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, 1000, 1.0, 0.5);                  \
        hipDeviceSynchronize();                                                                              \
        hipEventRecord(e0);                                                                                  \
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0, 0.5); \
        hipEventRecord(e1);                                                                                  \
        hipEventSynchronize(e1);                                                                             \
        float ms;                                                                                            \
        hipEventElapsedTime(&ms, e0, e1);                                                                    \
        unsigned long long h[512];                                                                           \
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);                                                  \
        double c = 0, rt = 0;                                                                                \
        for (int i = 0; i < 256; ++i) { c += h[2 * i]; rt += h[2 * i + 1]; }                                 \
        c /= 256; rt /= 256;                                                                                 \
        printf("%-58s %8.3f ms/launch  %7.2f cyc per 4-MFMA step  clock %.2f GHz\n", label, ms / 20, c / iters, c / rt * 0.1); \
    } while (0)

int main() {
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8);
    hipMalloc(&cyc, 512 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200000;
    RUN("1 wave/SIMD  i8 MFMA x4 per step, operands in registers", (k<0, 4, false>), 256);
    RUN("1 wave/SIMD  + one ds_read_b128 per step", (k<0, 4, true>), 256);
    RUN("1 wave/SIMD  + LDS + 1 f64 fma per MFMA", (k<1, 4, true>), 256);
    RUN("1 wave/SIMD  + LDS + 2 f64 fma per MFMA", (k<2, 4, true>), 256);
    RUN("1 wave/SIMD  + LDS + 3 f64 fma per MFMA", (k<3, 4, true>), 256);
    RUN("1 wave/SIMD  + LDS + 4 f64 fma per MFMA", (k<4, 4, true>), 256);
    RUN("1 wave/SIMD  f64 fma only, 2 per slot", (k<2, 4, false, false>), 256);
    RUN("2 waves/SIMD i8 MFMA x4 per step + LDS", (k<0, 8, true>), 512);
    RUN("2 waves/SIMD + LDS + 1 f64 fma per MFMA", (k<1, 8, true>), 512);
    RUN("2 waves/SIMD + LDS + 2 f64 fma per MFMA", (k<2, 8, true>), 512);
    RUN("2 waves/SIMD + LDS + 3 f64 fma per MFMA", (k<3, 8, true>), 512);
    RUN("2 waves/SIMD + LDS + 4 f64 fma per MFMA", (k<4, 8, true>), 512);
    RUN("2 waves/SIMD + LDS + 2 f32 fma per MFMA", (k<2, 8, true, true, 1>), 512);
    RUN("2 waves/SIMD + LDS + 4 f32 fma per MFMA", (k<4, 8, true, true, 1>), 512);
    RUN("2 waves/SIMD + LDS + 2 int shift-add per MFMA", (k<2, 8, true, true, 2>), 512);
    RUN("2 waves/SIMD + LDS + 4 int shift-add per MFMA", (k<4, 8, true, true, 2>), 512);
    return 0;
}
