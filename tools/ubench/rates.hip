// Micro-benchmarks of gfx950 fp64 issue rates (development aid; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o rates rates.hip && ./rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(double* out, int iters, double a, double b) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
__global__ void k_valu(double* out, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) x[i] = fma(x[i], b, a);
                if (OP == 1) x[i] = x[i] + b;
                if (OP == 2) x[i] = fmax(x[i] * 1.0, b + i);   // mul+max pair? keep simple below
                if (OP == 3) x[i] = fmax(x[i], b);
                if (OP == 4) x[i] = x[i] * b;
            }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* out;
    hipMalloc(&out, 8 << 20);
    const int iters = 20000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int wps = 1; wps <= 2; ++wps) {
        int threads = 256 * wps, blocks = p.multiProcessorCount;
        double ms4 = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); });
        double ms1 = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); });
        double ms16 = timeit([&] { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); });
        double n = (double)iters * 16;   // mfma per wave
        double flop = n * 2048.0 * blocks * (threads / 64);
        printf("MFMA f64 16x16x4, %d wave/SIMD: 4 acc %.3f ms (%.1f TF, %.1f ns/mfma/SIMD) | 1 acc %.3f ms (%.1f TF) | 16 acc %.3f ms (%.1f TF)\n", wps,
               ms4, flop / ms4 / 1e9, ms4 * 1e6 / (n * wps), ms1, flop / ms1 / 1e9, ms16, flop / ms16 / 1e9);
    }
    const char* names[] = {"fma", "add", "mul+max", "max", "mul"};
    for (int op : {0, 1, 3, 4}) {
        for (int wps = 1; wps <= 4; wps *= 2) {
            int threads = 256 * wps, blocks = p.multiProcessorCount;
            double ms = 0;
            if (op == 0) ms = timeit([&] { hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.999); });
            if (op == 1) ms = timeit([&] { hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.999); });
            if (op == 3) ms = timeit([&] { hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.999); });
            if (op == 4) ms = timeit([&] { hipLaunchKernelGGL(k_valu<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.999); });
            double ninst = (double)iters * 16;
            printf("VALU f64 %-4s %d wave/SIMD: %.3f ms  %.2f ns per wave-instr per SIMD  (%.1f Tinstr-lane/s chip)\n", names[op], wps, ms,
                   ms * 1e6 / (ninst * wps), ninst * 64 * blocks * (threads / 64) / ms / 1e9);
        }
    }
    return 0;
}
