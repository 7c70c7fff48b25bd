// write_bench.hip -- what does the draw writer's STORE PATTERN cost by itself?  The writer's waves each own 16 draws (columns of d
// contiguous doubles, 8 KB apart at d = 1000) and walk down the rows: every store instruction touches 16 columns.  This microbenchmark
// writes the same total (fits x N x d doubles) with nothing but the stores, in the writer's geometry (256 x fits workgroups of 16 waves,
// a wave owns 16 columns), for several run lengths per column and instruction shapes:
//   A  16 columns x 128 B per step, 8-byte lanes (the LDS-transposed tile stores: 4 instructions of 4 x 128 B)
//   B  16 columns x 128 B per step, 16-byte lanes in the natural MFMA layout (lane (q, c): rows 4q..4q+3 of column c; 2 instructions)
//   C  16 columns x 512 B per step (4 blocks buffered), 16-byte lanes: 4 instructions of 16 x 64 B ... see body
//   D  ONE column at a time, 1 KB per instruction fully contiguous (upper bound: a wave streams 8 KB columns one after another)
// build: hipcc -O3 --offload-arch=gfx950 write_bench.hip -o write_bench ; run: ./write_bench [fits] [N] [d]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(double *x, int d, int N) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15;
    const int ngroups = N / 16, nblk = d / 16;
    double *xf = x + (size_t)blockIdx.x * N * d;
    for (int grp = wv; grp < ngroups; grp += 16) {
        double *xg = xf + (size_t)grp * 16 * d;
        const double v = (double)(grp + lane);
        if (MODE == 0) {
            for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                for (int it = 0; it < 4; ++it) xg[(size_t)(4 * it + q) * d + blk * 16 + c] = v;       // 4 columns x 128 B per instruction
            }
        } else if (MODE == 1) {
            for (int blk = 0; blk < nblk; ++blk) {
                double *o = xg + (size_t)c * d + blk * 16 + 4 * q;
                d2 a = {v, v};
                *reinterpret_cast<d2 *>(o) = a;
                *reinterpret_cast<d2 *>(o + 2) = a;
            }
        } else if (MODE == 2) {
            // 4 blocks (64 rows = 512 B per column) per step: instruction `it` writes, for column c, rows 64 s + 16 it + 4q .. +1 (16 B)
            // i.e. the same pieces as B but 8 instructions back to back per 4 blocks
            for (int blk = 0; blk + 3 < nblk; blk += 4) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    double *o = xg + (size_t)c * d + (blk + b) * 16 + 4 * q;
                    d2 a = {v, v};
                    *reinterpret_cast<d2 *>(o) = a;
                    *reinterpret_cast<d2 *>(o + 2) = a;
                }
            }
        } else if (MODE == 3) {
            // one column at a time: 64 lanes x 16 B = 1 KB contiguous per instruction
            for (int col = 0; col < 16; ++col) {
                double *o = xg + (size_t)col * d;
                d2 a = {v, v};
                for (int r = lane * 2; r + 1 < d; r += 128) *reinterpret_cast<d2 *>(o + r) = a;
            }
        } else if (MODE == 4) {
            // 4 columns x 256 B per instruction (16-byte lanes, 16 lanes per column): what a 32-row transposed tile would store
            for (int blk = 0; blk + 1 < nblk; blk += 2) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    d2 a = {v, v};
                    *reinterpret_cast<d2 *>(xg + (size_t)(4 * it + q) * d + blk * 16 + 2 * c) = a;
                }
            }
        }
    }
}

int main(int argc, char **argv) {
    const int fits = argc > 1 ? atoi(argv[1]) : 1400, N = argc > 2 ? atoi(argv[2]) : 1008, d = argc > 3 ? atoi(argv[3]) : 1008;
    double *x;
    const size_t bytes = sizeof(double) * (size_t)fits * N * d;
    if (hipMalloc(&x, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"A 4 cols x 128 B / instr (8-byte lanes)", "B 16 cols x 4 x 16 B / instr (natural layout)",
                           "C as B, 4 blocks back to back", "D one column, 1 KB contiguous / instr", "E 4 cols x 256 B / instr (16-byte lanes)"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(fits), dim3(1024), 0, 0, x, d, N);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(fits), dim3(1024), 0, 0, x, d, N);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(fits), dim3(1024), 0, 0, x, d, N);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(fits), dim3(1024), 0, 0, x, d, N);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(fits), dim3(1024), 0, 0, x, d, N);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-48s %8.3f ms  %7.1f GB/s\n", names[mode], ms, bytes / (ms * 1e6));
        }
    }
    return 0;
}
