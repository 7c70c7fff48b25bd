// gather_bench.hip -- where should the inverse-CDF table of the generator live?  Per normal the generator reads 32 bytes at a
// data-dependent index (two 16-byte reads); in the ELBO scan that is ~350 GB per step through LDS with ~3x bank-conflict
// serialisation (random 16-lane groups over 16 bank quads), and the draw writer pays it twice.  This microbenchmark times, with
// the scan's occupancy (8 waves per CU) and the scan's index distribution (geometric over binades, uniform inside):
//   A  both reads from LDS                         (round 2)
//   B  both reads from global memory (L1 / TCP)    (table = 19 KB, L2 resident)
//   C  (c0, c1) from LDS, (c2, c3) from global     (two pipes in parallel)
//   D  one 16-byte LDS read only                   (lower bound of a 16-byte-per-normal scheme)
// build: hipcc -O3 --offload-arch=gfx950 gather_bench.hip -o gather_bench ; run: ./gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define NENT 608
#define ITER 4096

template <int MODE>
__global__ __launch_bounds__(512) void k(const double2 *__restrict__ gtab, double *out, int iters) {
    __shared__ double2 tab[2 * NENT];
    for (int i = threadIdx.x; i < 2 * NENT; i += 512) tab[i] = gtab[i];
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    double acc0 = 0.0, acc1 = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = s * 1664525u + 1013904223u;                       // LCG: cheap, keeps the VALU nearly idle
            const uint32_t w = s >> 1;
            // geometric binade (leading zeros), uniform sub-interval: the distribution of the real generator's index
            const int b = min(__clz((int)(w | 1u)) - 1, 18);
            const int idx = b * 32 + ((s >> 3) & 31);
            double2 c01, c23;
            if (MODE == 0) { c01 = tab[idx]; c23 = tab[NENT + idx]; }
            else if (MODE == 1) { c01 = gtab[idx]; c23 = gtab[NENT + idx]; }
            else if (MODE == 2) { c01 = tab[idx]; c23 = gtab[NENT + idx]; }
            else { c01 = tab[idx]; c23 = make_double2(1.0, 2.0); }
            acc0 += c01.x + c23.x; acc1 += c01.y + c23.y;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0 + acc1;
}

int main() {
    std::vector<double2> h(2 * NENT);
    for (int i = 0; i < 2 * NENT; ++i) h[i] = make_double2(i * 1e-3, i * 2e-3);
    double2 *g; double *o;
    hipMalloc(&g, sizeof(double2) * 2 * NENT); hipMalloc(&o, 8 * 512 * 256);
    hipMemcpy(g, h.data(), sizeof(double2) * 2 * NENT, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"A both LDS", "B both global (TCP)", "C LDS + global", "D one LDS read"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, g, o, ITER);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, g, o, ITER);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, g, o, ITER);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, g, o, ITER);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) {
                const double normals = 256.0 * 512 * ITER * 4;
                printf("%-22s %8.3f ms  %6.2f ns per wave-normal-set (64 lanes)  => %5.1f ms per 1.1e10 normals\n", names[mode], ms,
                       ms * 1e6 / (normals / 64.0) * 256 / 1.0 / 256, ms * 1.1e10 / normals);
            }
        }
    }
    return 0;
}
