// v_mfma_i32_16x16x64_i8 with the operand pattern of emit_diag_i8: 16 stationary A quads, 8 B quads, two accumulators
// alternating, diagonals d = 7..0 (36 plane pairs x 2 tiles = 72 MFMAs per "tile").  (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4_t __attribute__((ext_vector_type(4)));

template <int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(const int4_t* src, int* out, int iters) {
    int4_t W[2][8], f[8];
    for (int q = 0; q < 2; ++q)
        for (int j = 0; j < 8; ++j) W[q][j] = src[(q * 8 + j) * 64 + (threadIdx.x & 63)];
    for (int j = 0; j < 8; ++j) f[j] = src[(16 + j) * 64 + (threadIdx.x & 63)];
    int s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 7; d >= 0; --d) {
            int4_t acc[2] = {int4_t{0, 0, 0, 0}, int4_t{0, 0, 0, 0}};
#pragma unroll
            for (int i = 0; i <= d; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (MODE == 0) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[q][d - i], f[i], acc[q], 0, 0, 0);
                    if (MODE == 1) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[q][0], f[0], acc[q], 0, 0, 0);
                    if (MODE == 2) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[q][d - i], f[0], acc[q], 0, 0, 0);
                    if (MODE == 3) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[q][0], f[i], acc[q], 0, 0, 0);
                }
            s += acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
        }
        f[it & 7][0] += s;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define RUN(label, kern, threads, waves)                                                                        \
    do {                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, out, 100);                                \
        hipDeviceSynchronize();                                                                                 \
        hipEventRecord(e0);                                                                                     \
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, out, iters); \
        hipEventRecord(e1);                                                                                     \
        hipEventSynchronize(e1);                                                                                \
        float ms;                                                                                               \
        hipEventElapsedTime(&ms, e0, e1);                                                                       \
        const double per = (double)ms / 10 * 1e-3 * 2.4e9 / ((double)iters * 72 * waves);                       \
        printf("%-60s %8.3f ms/launch  %6.2f cycles per MFMA and SIMD at 2.4 GHz\n", label, ms / 10, per);      \
    } while (0)

int main() {
    int* out;
    int4_t* src;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&src, 24 * 64 * 16);
    hipMemset(src, 1, 24 * 64 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    RUN("2 waves/SIMD, kernel pattern (16 A quads, 8 B quads)", (k<8, 0>), 512, 2);
    RUN("2 waves/SIMD, one A quad per tile, one B quad", (k<8, 1>), 512, 2);
    RUN("2 waves/SIMD, 16 A quads, one B quad", (k<8, 2>), 512, 2);
    RUN("2 waves/SIMD, one A quad per tile, 8 B quads", (k<8, 3>), 512, 2);
    RUN("1 wave/SIMD, kernel pattern", (k<4, 0>), 256, 1);
    return 0;
}
