// coissue.hip -- which instruction classes overlap with the f64 matrix pipe on gfx950?  (not part of libpfmi.so)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/coissue.hip -o build/coissue && build/coissue
// One workgroup of 512 threads per CU (forced by 100 KB of LDS), i.e. exactly 2 waves per SIMD like the ELBO scan.
// Modes: every wave runs the MFMA loop / the VALU loop / even waves MFMA + odd waves VALU (inter-wave overlap) /
// every wave alternates MFMA and VALU instructions (intra-wave interleave).  VALU flavours: Philox (integer), fp64 FMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
}

template <int MODE, int FLAVOUR>   // MODE 0: mfma only, 1: valu only, 2: even waves mfma / odd waves valu, 3: interleaved in every wave
__global__ __launch_bounds__(512) void k(double *out, int iters, uint32_t key) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    uint32_t c0 = threadIdx.x, c1 = blockIdx.x, c2 = 1, c3 = 2;
    double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
    const bool do_m = (MODE == 0) || (MODE == 3) || (MODE == 2 && (wave & 1) == 0);
    const bool do_v = (MODE == 1) || (MODE == 3) || (MODE == 2 && (wave & 1) == 1);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
                if (FLAVOUR == 0) { philox_round(c0, c1, c2, c3, key + j, key ^ j); }
                else { f0 = fma(f0, 1.0000001, 1e-9); f1 = fma(f1, 1.0000001, 1e-9); f2 = fma(f2, 1.0000001, 1e-9); f3 = fma(f3, 1.0000001, 1e-9); }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (FLAVOUR == 0) { philox_round(c0, c1, c2, c3, key + j, key ^ j); }
                    else { f0 = fma(f0, 1.0000001, 1e-9); f1 = fma(f1, 1.0000001, 1e-9); f2 = fma(f2, 1.0000001, 1e-9); f3 = fma(f3, 1.0000001, 1e-9); }
                }
            }
        }
    }
    double s = f0 + f1 + f2 + f3 + (double)(c0 ^ c1 ^ c2 ^ c3);
    for (int j = 0; j < 8; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 7];
}

template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 4, iters = 20000;
    const size_t lds = 100 * 1024;
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 512));
#define RUN(M, F, label)                                                                                                  \
    do {                                                                                                                    \
        auto kern = k<M, F>;                                                                                                \
        CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, out, iters, 12345u); });          \
        printf("%-58s %8.3f ms\n", label, ms);                                                                              \
    } while (0)
    RUN(0, 0, "all waves: 8 MFMA f64 4x4x4 per iteration");
    RUN(1, 0, "all waves: 8 Philox rounds (integer VALU) per iteration");
    RUN(1, 1, "all waves: 32 fp64 FMA per iteration");
    RUN(2, 0, "even waves MFMA, odd waves Philox (inter-wave)");
    RUN(2, 1, "even waves MFMA, odd waves fp64 FMA (inter-wave)");
    RUN(3, 0, "every wave: MFMA / Philox round interleaved");
    RUN(3, 1, "every wave: MFMA / 4 fp64 FMA interleaved");
    return 0;
}
