// microbench.hip -- gfx950 micro-benchmarks that size the ELBO kernel design (not part of libpfmi.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o build/microbench && build/microbench
// Measures: fp64 VALU FMA rate, v_mfma_f64_16x16x4 rate, Philox4x32-10 rate, the round-2 normal generator (Philox4x32-7 +
// table inverse CDF) and MFMA/VALU co-issue (see also coissue.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../csrc/pfmi_common.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_fma(double *out, int iters) {
    double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
        a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_mfma(double *out, int iters) {
    d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    d4 s = acc0 + acc1 + acc2 + acc3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// MFMA and VALU work interleaved in one wave: does the VALU FMA stream hide under the MFMA pipe?
__global__ void k_mfma_fma(double *out, int iters) {
    d4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    const double bb = 1.0000001, cc = 1e-9;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        v0 = fma(v0, bb, cc); v1 = fma(v1, bb, cc); v2 = fma(v2, bb, cc); v3 = fma(v3, bb, cc);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        v4 = fma(v4, bb, cc); v5 = fma(v5, bb, cc); v6 = fma(v6, bb, cc); v7 = fma(v7, bb, cc);
    }
    d4 s = acc0 + acc1;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

__global__ void k_philox(double *out, int iters) {
    uint32_t acc = 0;
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        uint32_t x[4];
        pf_philox4x32_10((uint32_t)i, n, 0u, 0u, 12345u, 678u, x);
        acc ^= x[0] ^ x[1] ^ x[2] ^ x[3];
    }
    out[n] = (double)acc;
}

// round-2 generator: Philox4x32-7 + table inverse CDF (LDS copy of the common-case table), 4 normals per call
__global__ void k_randn_icdf(double *out, int iters) {
    __shared__ double2 tab[2 * PF_ICDF_LDS_ENTRIES];
    pf_icdf_load(tab);
    __syncthreads();
    double acc = 0;
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        double z[4];
        pf_randn4(0x123456789abcdefull, (uint32_t)i, n, 0u, tab, z);
        acc += z[0] + z[1] + z[2] + z[3];
    }
    out[n] = acc;
}

// table inverse CDF only (no Philox): look-up + cubic cost in isolation
__global__ void k_icdf_only(double *out, int iters) {
    __shared__ double2 tab[2 * PF_ICDF_LDS_ENTRIES];
    pf_icdf_load(tab);
    __syncthreads();
    double acc = 0;
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = n * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t x[4] = {s | 0x00100000u, (s ^ 0x9E3779B9u) | 0x00100000u, (s * 3u + 1u) | 0x00100000u, (~s) | 0x00100000u};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double dp;
            double2 c01, c23;
            pf_icdf_issue(x[t], tab, dp, c01, c23);
            acc += pf_icdf_finish(x[t], dp, c01, c23);
        }
    }
    out[n] = acc;
}

template <typename F>
static double time_kernel(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int blocks = 256 * 8, threads = 256;
    const size_t nthr = (size_t)blocks * threads;
    double *out;
    CHECK(hipMalloc(&out, nthr * sizeof(double) + 64));
    const int iters = 4096;
    double ms;
    ms = time_kernel([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, iters); }, 3);
    printf("fp64 VALU FMA      : %8.3f ms  -> %7.2f TFLOP/s\n", ms, 2.0 * 8 * iters * nthr / ms / 1e9);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, iters); }, 3);
    {
        double nm = 4.0 * iters * (nthr / 64);   // MFMAs
        double cyc = ms * 1e-3 * 2.4e9 * 1024 / nm;
        printf("f64 MFMA 16x16x4   : %8.3f ms  -> %7.2f TFLOP/s  (~%.1f cycles/MFMA/SIMD @2.4GHz)\n", ms, nm * 2048 / ms / 1e9, cyc);
    }
    ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_fma, dim3(blocks), dim3(threads), 0, 0, out, iters); }, 3);
    printf("MFMA + 4 FMA each  : %8.3f ms  -> MFMA %7.2f TF + VALU %7.2f TF\n", ms, 2.0 * iters * (nthr / 64) * 2048 / ms / 1e9,
           2.0 * 8 * iters * nthr / ms / 1e9);
    const int it2 = 1024;
    ms = time_kernel([&] { hipLaunchKernelGGL(k_philox, dim3(blocks), dim3(threads), 0, 0, out, it2); }, 3);
    printf("Philox4x32-10      : %8.3f ms  -> %7.2f G calls/s (%.2f G u32/s)\n", ms, (double)it2 * nthr / ms / 1e6, 4.0 * it2 * nthr / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_randn_icdf, dim3(blocks), dim3(threads), 0, 0, out, it2); }, 3);
    printf("randn4 (Philox-7 + table inverse CDF): %8.3f ms  -> %7.2f G normals/s\n", ms, 4.0 * it2 * nthr / ms / 1e6);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_icdf_only, dim3(blocks), dim3(threads), 0, 0, out, it2); }, 3);
    printf("table inverse CDF only               : %8.3f ms  -> %7.2f G normals/s\n", ms, 4.0 * it2 * nthr / ms / 1e6);
    return 0;
}
