// Issue rate of v_mfma_i32_16x16x64_i8 on gfx950: independent accumulators vs chains of distance NACC.  (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4_t __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(int* out, int iters) {
    int4_t A[4], B[2];
    for (int i = 0; i < 4; ++i) A[i] = int4_t{(int)threadIdx.x * 13 + i, i * 17 + 5, (int)threadIdx.x + 7 * i, 3 * i + 1};
    for (int i = 0; i < 2; ++i) B[i] = int4_t{(int)threadIdx.x * 3 + i, i * 19 + 5, (int)threadIdx.x + 5 * i, 7 * i + 1};
    int4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = int4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[u & 3], B[u & 1], acc[u % NACC], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define RUN(label, kern, threads, waves)                                                                        \
    do {                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 100);                                     \
        hipDeviceSynchronize();                                                                                 \
        hipEventRecord(e0);                                                                                     \
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);      \
        hipEventRecord(e1);                                                                                     \
        hipEventSynchronize(e1);                                                                                \
        float ms;                                                                                               \
        hipEventElapsedTime(&ms, e0, e1);                                                                       \
        const double per = (double)ms / 10 * 1e-3 * 2.4e9 / ((double)iters * 16 * waves);                       \
        printf("%-50s %8.3f ms/launch  %6.2f cycles per MFMA and SIMD at 2.4 GHz\n", label, ms / 10, per);      \
    } while (0)

int main() {
    int* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 100000;
    RUN("1 wave/SIMD, 16 independent accumulators", (k<16, 4>), 256, 1);
    RUN("1 wave/SIMD, 4 accumulators (distance 4)", (k<4, 4>), 256, 1);
    RUN("1 wave/SIMD, 2 accumulators (distance 2)", (k<2, 4>), 256, 1);
    RUN("1 wave/SIMD, 1 accumulator (back to back)", (k<1, 4>), 256, 1);
    RUN("2 waves/SIMD, 16 independent accumulators", (k<16, 8>), 512, 2);
    RUN("2 waves/SIMD, 2 accumulators (distance 2)", (k<2, 8>), 512, 2);
    RUN("2 waves/SIMD, 1 accumulator", (k<1, 8>), 512, 2);
    RUN("4 waves/SIMD, 2 accumulators", (k<2, 16>), 1024, 4);
    return 0;
}
