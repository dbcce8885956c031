// HBM write rate by store pattern: what does the bm store of the emission kernels (32-byte pieces, four per 128-byte row, from
// four store instructions of one wave) cost against whole rows per instruction?  (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// MODE 0: the emission kernel's pattern -- lane (frame = lane & 15, state = lane >> 4) owns 128 contiguous bytes (16 mixtures)
//         of row [frame][state]; store instruction q writes bytes 32 q .. 32 q + 31 of every lane's row.  Rows of one frame:
//         4 states x 128 B = 512 contiguous bytes; frames 64 KB apart.
// MODE 1: the same bytes, transposed through nothing (synthetic): instruction q writes, per lane, 32 bytes such that the
//         wave's 64 lanes cover 2 KB contiguous (lane-major): whole 128-byte lines per instruction.
// MODE 2: dwordx4 (16 B) per lane, 1 KB contiguous per instruction.
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, long nframes, int N) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t row_bytes = (size_t)N * 16 * 8;                    // one frame of bm
    const double4_t val = double4_t{1.0 * lane, 2.0, 3.0, 4.0 + w};
    // block: 16 frames x 32 states (8 waves x 4 states); grid covers frames and state slabs
    const int slabs = N / 32;
    const long tile = blockIdx.x / slabs;
    const int y = blockIdx.x % slabs;
    for (long t = tile; t < nframes / 16; t += gridDim.x / slabs) {
        char* base = reinterpret_cast<char*>(out) + (size_t)t * 16 * row_bytes + (size_t)(32 * y + 4 * w) * 128;
        if (MODE == 0) {
            char* p = base + (size_t)(lane & 15) * row_bytes + (size_t)(lane >> 4) * 128;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<double4_t*>(p + 32 * q) = val;
        } else if (MODE == 1) {
            // instruction q covers frames 4 q .. 4 q + 3: lane -> (frame 4 q + (lane >> 4), 32-byte piece lane & 15 of the 512 B)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                char* p = base + (size_t)(4 * q + (lane >> 4)) * row_bytes + (size_t)(lane & 15) * 32;
                *reinterpret_cast<double4_t*>(p) = val;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                char* p = base + (size_t)(2 * q + (lane >> 5)) * row_bytes + (size_t)(lane & 31) * 16;
                *reinterpret_cast<double2_t*>(p) = double2_t{val[0], val[1]};
            }
        }
    }
}

#define RUN(label, kern)                                                                          \
    do {                                                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, nframes, N);                   \
        hipDeviceSynchronize();                                                                   \
        hipEventRecord(e0);                                                                       \
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, nframes, N); \
        hipEventRecord(e1);                                                                       \
        hipEventSynchronize(e1);                                                                  \
        float ms;                                                                                 \
        hipEventElapsedTime(&ms, e0, e1);                                                         \
        printf("%-70s %7.3f ms  %6.2f TB/s\n", label, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);   \
    } while (0)

int main() {
    const int N = 512;                       // config 5: 256 states x 32 mixtures = 512 virtual 16-mixture states
    const long nframes = 128000;
    const double bytes = (double)nframes * N * 16 * 8;
    double* out;
    hipMalloc(&out, (size_t)bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * (N / 32) / (N / 32) * (N / 32);   // 256 frame ranges x slabs
    RUN("32-byte pieces, four instructions per 128-byte row (the kernels' pattern)", (k<0>));
    RUN("32 bytes per lane, 512 contiguous bytes per 16 lanes", (k<1>));
    RUN("16 bytes per lane, 512 contiguous bytes per 32 lanes", (k<2>));
    return 0;
}
