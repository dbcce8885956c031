// Issue rate of v_fmac_f64_dpp (row_newbcast) against plain v_fma_f64 on gfx950 (development aid).
//   hipcc --offload-arch=gfx950 -O3 -o dpprate dpprate.hip && ./dpprate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void k(double* out, int iters, double a, double b) {
    double x[8], e = a + (threadIdx.x & 15), y = b + threadIdx.x;
    for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) x[i] = fma(e, y, x[i]);
                if (OP == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(e), "v"(y));
                if (OP == 2) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(e));
                if (OP == 3) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[i]) : "v"(e), "v"(y));
            }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
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

int main() {
    double* out;
    (void)hipMalloc(&out, 8 << 20);
    const int iters = 20000;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const char* names[] = {"v_fma_f64 (compiler)", "v_fmac_f64_dpp row_newbcast", "v_mov_b64_dpp row_newbcast", "v_fmac_f64 (asm)"};
    for (int op = 0; op < 4; ++op)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps, blocks = p.multiProcessorCount;
            double ms = 0;
            if (op == 0) ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.5); });
            if (op == 1) ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.5); });
            if (op == 2) ms = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.5); });
            if (op == 3) ms = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.5); });
            const double n = (double)iters * 32 * wps;          // instructions per SIMD
            printf("%-30s %d wave/SIMD: %.3f ms  %.2f ns/instr/SIMD (%.2f cycles at 2.4 GHz)\n", names[op], wps, ms,
                   ms * 1e6 / n, ms * 1e6 / n * 2.4);
        }
    return 0;
}
