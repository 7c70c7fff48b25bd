// cvt_bench.hip -- issue cost of the integer -> double conversion of the inverse-CDF look-up against the exponent trick
//   v = (double) mag            (v_cvt_f64_u32)
//   v = as_double(0x4330000000000000 | mag) - 2^52   (one v_add_f64; exact for mag < 2^32)
// and, for scale, v_fma_f64 and v_mad_u64_u32.  512-thread workgroups, 2 waves per SIMD.  build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int iters, uint32_t seed) {
    extern __shared__ double lds[];
    uint32_t m[8];
    double acc[8];
    for (int j = 0; j < 8; ++j) { m[j] = seed * (threadIdx.x + 1) + j * 0x9E3779B9u; acc[j] = 0.0; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) { double v; asm volatile("v_cvt_f64_u32_e32 %0, %1" : "=v"(v) : "v"(m[j])); acc[j] = v; }
            else if (MODE == 1) { double v = __hiloint2double(0x43300000, (int)m[j]); asm volatile("v_add_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(-4503599627370496.0)); acc[j] = v; }
            else if (MODE == 2) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(1.0000001), "v"(1e-9)); }
            else { uint64_t p; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(m[j]), "v"(0xD2511F53u) : "vcc"); m[j] = (uint32_t)(p >> 32) ^ (uint32_t)p; }
        }
    }
    double s = 0.0;
    for (int j = 0; j < 8; ++j) s += acc[j] + (double)m[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 7];
}
int main() {
    const int blocks = 1024, iters = 20000;
    double *out; hipMalloc(&out, sizeof(double) * blocks * 512);
    const char *names[] = {"v_cvt_f64_u32", "exponent trick (v_add_f64)", "v_fma_f64", "v_mad_u64_u32 (+ xor)"};
    for (int m = 0; m < 4; ++m) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](auto kern) { hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 100 * 1024, 0, out, iters, 12345u); };
            hipEventRecord(e0);
            if (m == 0) launch(k<0>); else if (m == 1) launch(k<1>); else if (m == 2) launch(k<2>); else launch(k<3>);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-30s %8.3f ms -> %5.1f cycles per instruction and SIMD @ 2.3 GHz\n", names[m], best, best * 1e-3 * 2.3e9 / (4.0 * 2.0 * iters * 8));
    }
    return 0;
}
