// pfmi_fastmath.h -- one Philox4x32 round as a separate step, for kernels that spread a Philox call over several scheduling
// phases (elbo_mfma_kernel.hip).  (Round 1 kept its table-driven fp64 Box-Muller here; round 2 replaced it by the table
// inverse CDF of pfmi_common.h, which needs no transcendental function at all.)
#pragma once
#ifdef __HIPCC__
__device__ __forceinline__ void pf_philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                                uint32_t &k0, uint32_t &k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
}
#endif
