// write_bench2.hip -- the draw writer's store patterns in the writer's OWN occupancy: one 512-thread workgroup per CU (100 KB of LDS
// requested), a wave owns two 16-draw groups and walks down the rows block by block with `spin` dependent fp64 FMAs per block and group
// in between (the generator / MFMA work of the real kernel).  Patterns per 16 x 16 block:
//   N  natural MFMA layout: lane (q, c) = rows 4q..4q+3 of draw c, two 16-byte stores (16 columns x 4 x 16 B per instruction)
//   T  transposed product: lane (q, c) = row c of draws q, q+4, q+8, q+12, four 8-byte stores (4 columns x 128 contiguous B each)
// build: hipcc -O3 --offload-arch=gfx950 write_bench2.hip -o write_bench2 ; run: ./write_bench2 [fits] [N] [d]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(double *x, int d, int N, int spin, int fits) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15;
    const int ngroups = N / 16, nblk = d / 16;
    if (tid == 0) lds[0] = 1.0;
    for (int fit = blockIdx.x; fit < fits; fit += gridDim.x) {
        double *xf = x + (size_t)fit * N * d;
        for (int g0 = wv * 2; g0 < ngroups; g0 += 16) {
            double v[2] = {(double)(g0 + lane), (double)(g0 + lane) + 0.5};
            if (MODE == 2) {        // phased: a store-free walk, then a walk with the stores of both groups (the writer's two passes)
                for (int pass = 0; pass < 2; ++pass)
                    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            double a = v[g];
                            for (int s = 0; s < spin; ++s) a = fma(a, 1.0000001, 1e-9);
                            v[g] = a;
                            if (pass == 1 && g0 + g < ngroups) {
                                double *o = xf + (size_t)(g0 + g) * 16 * d + (size_t)c * d + blk * 16 + 4 * q;
                                d2 w = {a, a};
                                *reinterpret_cast<d2 *>(o) = w;
                                *reinterpret_cast<d2 *>(o + 2) = w;
                            }
                        }
                    }
            }
            if (MODE == 3) {        // fused: two walks, each with the stores of ONE group (same work, stores spread evenly)
                for (int pass = 0; pass < 2; ++pass)
                    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            double a = v[g];
                            for (int s = 0; s < spin; ++s) a = fma(a, 1.0000001, 1e-9);
                            v[g] = a;
                            if (pass == g && g0 + g < ngroups) {
                                double *o = xf + (size_t)(g0 + g) * 16 * d + (size_t)c * d + blk * 16 + 4 * q;
                                d2 w = {a, a};
                                *reinterpret_cast<d2 *>(o) = w;
                                *reinterpret_cast<d2 *>(o + 2) = w;
                            }
                        }
                    }
            }
            for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if (g0 + g >= ngroups) continue;
                    double *xg = xf + (size_t)(g0 + g) * 16 * d;
                    double a = v[g];
                    for (int s = 0; s < spin; ++s) a = fma(a, 1.0000001, 1e-9);
                    v[g] = a;
                    if (MODE >= 2) continue;
                    if (MODE == 0) {
                        double *o = xg + (size_t)c * d + blk * 16 + 4 * q;
                        d2 w = {a, a};
                        *reinterpret_cast<d2 *>(o) = w;
                        *reinterpret_cast<d2 *>(o + 2) = w;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xg[(size_t)(4 * r + q) * d + blk * 16 + c] = a;
                    }
                }
            }
        }
    }
}

int main(int argc, char **argv) {
    const int fits = argc > 1 ? atoi(argv[1]) : 1280, N = argc > 2 ? atoi(argv[2]) : 1008, d = argc > 3 ? atoi(argv[3]) : 1008;
    double *x;
    const size_t bytes = sizeof(double) * (size_t)fits * N * d;
    if (hipMalloc(&x, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int spins[] = {0, 10, 20, 30, 50};
    for (int mode = 0; mode < 4; ++mode)
        for (int si = 0; si < 5; ++si) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 100 * 1024, 0, x, d, N, spins[si], fits);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 100 * 1024, 0, x, d, N, spins[si], fits);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 100 * 1024, 0, x, d, N, spins[si], fits);
                else hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 100 * 1024, 0, x, d, N, spins[si], fits);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s spin %3d (%4d FMA cycles per block and group): %8.3f ms  %7.1f GB/s\n", mode == 0 ? "N natural 2 x 16 B" : mode == 1 ? "T transposed 4 x 8 B" : mode == 2 ? "P phased (store-free walk + store walk)" : "F fused (each walk stores one group)", spins[si],
                   spins[si] * 8, best, bytes / (best * 1e6));
        }
    return 0;
}
