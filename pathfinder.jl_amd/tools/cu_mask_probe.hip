// cu_mask_probe.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask use on MI355X (8 XCDs x 32 CUs)?
// Build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe [reserved_bits]
// Each workgroup (1 per CU: 64 KB LDS + long spin) records (XCC_ID, SE, SH, CU) of the CU it ran on.  Prints how many distinct CUs the
// masked stream used, how many per XCD, and whether a second, unmasked launch running at the same time got the excluded ones.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void where_kernel(unsigned *out, long long spin) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        lds[0] = (char)hw;
        const long long t0 = clock64();
        while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
}

static void report(const char *what, const std::vector<unsigned> &v, int n, std::set<unsigned> *keep) {
    std::set<unsigned> cus; std::map<unsigned, int> per_xcc;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 0xF;
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        const unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (cus.insert(id).second) per_xcc[xcc]++;
    }
    printf("%s: %d workgroups on %zu distinct CUs; per XCD:", what, n, cus.size());
    for (auto &p : per_xcc) printf(" %u:%d", p.first, p.second);
    printf("\n");
    if (keep) *keep = cus;
}

int main(int argc, char **argv) {
    const int reserve = argc > 1 ? atoi(argv[1]) : 32;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    printf("%s: %d CUs, reserving the lowest %d mask bits\n", pr.name, ncu, reserve);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0xFFFFFFFFu);
    for (int b = 0; b < reserve; ++b) mask[b / 32] &= ~(1u << (b % 32));
    if (ncu % 32) mask.back() &= (1u << (ncu % 32)) - 1;
    hipStream_t sm, su;
    CK(hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data()));
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    const int n = 1024;
    unsigned *dm, *du; CK(hipMalloc(&dm, 8 * n)); CK(hipMalloc(&du, 8 * n));
    CK(hipFuncSetAttribute((const void *)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    std::vector<unsigned> hm(2 * n), hu(2 * n);
    // masked launch alone
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 100 * 1024, sm, dm, 200000LL);
    CK(hipStreamSynchronize(sm));
    CK(hipMemcpy(hm.data(), dm, 8 * n, hipMemcpyDeviceToHost));
    std::set<unsigned> masked;
    report("masked stream alone", hm, n, &masked);
    // masked launch saturating its CUs, an unmasked launch of `reserve` workgroups beside it: does it start at once (on the reserved CUs)?
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 100 * 1024, sm, dm, 2000000LL);
    CK(hipEventRecord(e0, su));
    hipLaunchKernelGGL(where_kernel, dim3(reserve > 0 ? reserve : 1), dim3(64), 100 * 1024, su, du, 1000LL);
    CK(hipEventRecord(e1, su));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hu.data(), du, 8 * n, hipMemcpyDeviceToHost));
    std::set<unsigned> un;
    report("unmasked launch beside a saturating masked one", hu, reserve > 0 ? reserve : 1, &un);
    int overlap = 0; for (unsigned c : un) overlap += masked.count(c);
    printf("  it took %.3f ms (a masked round is ~%.1f ms); %d of its %zu CUs are CUs the masked stream also uses\n", ms, 2000000.0 / 100e3, overlap, un.size());
    return 0;
}
