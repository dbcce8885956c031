// Achievable HBM bandwidth on the box ("measure achievable HBM with a copy kernel"): streaming read
// (sum), copy and write of a 4 GiB buffer, b128 accesses, grid-stride.  Development aid; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -o hbm hbm.hip && ./hbm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));

__global__ void k_read(const double2_t* __restrict__ a, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2_t v = a[i];
        s += v[0] + v[1];
    }
    if (s == 1.2345) out[0] = s;
}
__global__ void k_copy(const double2_t* __restrict__ a, double2_t* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_write(double2_t* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = double2_t{1.0, 2.0};
}

int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    double2_t *a, *b;
    double* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 8);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int which = 0; which < 3; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (which == 0) k_read<<<blocks, 256>>>(a, n, o);
                if (which == 1) k_copy<<<blocks, 256>>>(a, b, n);
                if (which == 2) k_write<<<blocks, 256>>>(b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double moved = which == 1 ? 2.0 * bytes : (double)bytes;
            printf("blocks=%6d %-5s %7.3f ms  %7.1f GB/s\n", blocks, which == 0 ? "read" : which == 1 ? "copy" : "write", best,
                   moved / best / 1e6);
        }
    }
    return 0;
}
