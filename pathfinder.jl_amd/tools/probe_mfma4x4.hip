// probe_mfma4x4.hip -- discovers the lane layout of v_mfma_f64_4x4x4_4b_f64 and measures its issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_probe(double *out) {   // out[(la*64+lb)*64 + lane]
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[((size_t)la * 64 + lb) * 64 + lane] = d;
        }
}
__global__ void k_rate(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}
int main() {
    double *d; size_t n = 64 * 64 * 64;
    hipMalloc(&d, n * 8 + (size_t)2048 * 256 * 8);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d);
    double *h = (double *)malloc(n * 8);
    hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    // for each A lane: list of (B lane -> D lane) hits
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" B%02d->D%02d", lb, l);
        printf("\n");
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 2048, threads = 256;
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nm = 4.0 * iters * ((double)blocks * threads / 64);
    printf("4x4x4_4b f64: %.3f ms, %.2f TFLOP/s, ~%.1f cycles/MFMA/SIMD @2.3GHz\n", ms, nm * 512 / ms / 1e9, ms * 1e-3 * 2.3e9 * 1024 / nm);
    return 0;
}
