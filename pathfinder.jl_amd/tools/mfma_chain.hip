// mfma_chain.hip -- cost of DEPENDENT f64 MFMA chains at the draw writer's occupancy (512-thread workgroup per CU, 2 waves per SIMD).
//   A  12 quarter-size MFMAs (4x4x4) into 3 accumulators (pass 1 of the writer: 4 dependent steps x 3 independent chains), per group
//   B  3 dependent 16x16x4 MFMAs on one accumulator, result consumed by 4 FMAs (pass 2 of the writer), per group
//   C  as B but the three products go to three accumulators and are summed by VALU adds (no MFMA -> MFMA dependency)
//   D  as B for two groups, the MFMAs of the two groups interleaved (g0 s0, g1 s0, g0 s1, ...)
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_chain.hip -o build/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int iters) {
    extern __shared__ double lds[];
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, b0 = 1.0 + threadIdx.x * 1e-4, b1 = b0 + 0.3, b2 = b0 + 0.7;
    double s = 0.0, z0 = 0.1 * threadIdx.x, z1 = z0 + 1.0;
    double w[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    w[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0 + r, w[g][0], 0, 0, 0);
                    w[g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b0 + r, w[g][1], 0, 0, 0);
                    w[g][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b0 + r, w[g][2], 0, 0, 0);
                }
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const double z = g ? z1 : z0;
                d4 x = {z, z + 1, z + 2, z + 3};
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, x, 0, 0, 0);
                s = fma(x[0], 1.01, s); s = fma(x[1], 1.02, s); s = fma(x[2], 1.03, s); s = fma(x[3], 1.04, s);
            }
            z0 += 1e-3; z1 += 1e-3;
        } else if (MODE == 2) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const double z = g ? z1 : z0;
                d4 x = {z, z + 1, z + 2, z + 3}, y = {0, 0, 0, 0}, u = {0, 0, 0, 0};
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, x, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, y, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, u, 0, 0, 0);
                x += y + u;
                s = fma(x[0], 1.01, s); s = fma(x[1], 1.02, s); s = fma(x[2], 1.03, s); s = fma(x[3], 1.04, s);
            }
            z0 += 1e-3; z1 += 1e-3;
        } else {
            d4 x = {z0, z0 + 1, z0 + 2, z0 + 3}, y = {z1, z1 + 1, z1 + 2, z1 + 3};
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, x, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0 + 1, y, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, x, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1 + 1, y, 0, 0, 0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, x, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2 + 1, y, 0, 0, 0);
            s = fma(x[0], 1.01, s); s = fma(x[1], 1.02, s); s = fma(x[2], 1.03, s); s = fma(x[3], 1.04, s);
            s = fma(y[0], 1.01, s); s = fma(y[1], 1.02, s); s = fma(y[2], 1.03, s); s = fma(y[3], 1.04, s);
            z0 += 1e-3; z1 += 1e-3;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + w[0][0] + w[0][1] + w[0][2] + w[1][0] + w[1][1] + w[1][2] + lds[threadIdx.x & 7];
}

int main() {
    const int blocks = 256 * 4, iters = 20000;
    const size_t lds = 100 * 1024;
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 512));
    const char *names[] = {"A 2 x 12 MFMA 4x4x4 (3 chains of 4)", "B 2 x 3 dependent MFMA 16x16x4 + 4 FMA", "C 2 x 3 independent MFMA 16x16x4 + adds", "D 2 groups interleaved, dependent within a group"};
    for (int m = 0; m < 4; ++m) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            auto launch = [&](auto kern) {
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, out, iters);
            };
            CHECK(hipEventRecord(e0));
            if (m == 0) launch(k<0>); else if (m == 1) launch(k<1>); else if (m == 2) launch(k<2>); else launch(k<3>);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // per SIMD: 4 rounds of workgroups x 2 waves x iters iterations
        const double cyc = best * 1e-3 * 2.3e9 / (4.0 * 2.0 * iters);
        printf("%-52s %8.3f ms  -> %6.0f cycles @2.3 GHz per wave-iteration (2 groups)\n", names[m], best, cyc);
    }
    return 0;
}
