// MFMA f64 issue rate with the emission kernel's operand pattern (development aid):
// 80 stationary A registers, B fragments re-read from LDS per tile, 4 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: B from LDS each tile, no barrier; 1: + __syncthreads per tile; 2: B constant in regs
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(const double* wp, double* out, int tiles) {
    __shared__ double Fl[2][80 * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 2 * 80 * 16; e += 512) (&Fl[0][0])[e] = 1e-3 * e;
    double wr[4][20];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ks = 0; ks < 20; ++ks) wr[q][ks] = wp[(q * 20 + ks) * 64 + lane];
    __syncthreads();
    const int g = lane >> 4, fr = lane & 15;
    double tot = 0;
    for (int t = 0; t < tiles; ++t) {
        double4_t acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = double4_t{0, 0, 0, 0};
        const double* Fc = &Fl[t & 1][0];
#pragma unroll
        for (int ks = 0; ks < 20; ++ks) {
            const double bf = MODE == 2 ? wr[0][ks] : Fc[(4 * ks + g) * 16 + fr];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q][ks], bf, acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) tot += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        if (MODE == 1) __syncthreads();
    }
    out[blockIdx.x * 512 + tid] = tot;
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

int main() {
    double *wp, *out;
    (void)hipMalloc(&wp, 80 * 64 * 8);
    (void)hipMemset(wp, 0, 80 * 64 * 8);
    (void)hipMalloc(&out, 256 * 512 * 8);
    const int tiles = 2000;   // per wave: 2000 tiles x 80 MFMA, 2 waves/SIMD -> same MFMA count per SIMD as the real launch
    const double ideal = 2.0 * tiles * 80 * 64.5 / 2.4e9 * 1e3;
    printf("ideal (64.5 cyc/MFMA @2.4GHz): %.3f ms\n", ideal);
    printf("B from LDS, no barrier : %.3f ms\n", timeit([&] { hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, wp, out, tiles); }));
    printf("B from LDS, barrier/tile: %.3f ms\n", timeit([&] { hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, wp, out, tiles); }));
    printf("B in registers          : %.3f ms\n", timeit([&] { hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, wp, out, tiles); }));
    return 0;
}
