// elbo_xw_kernel.hip -- the draw WRITER: rand_and_logpdf (reference src/mvnormal.jl:24-39, unwhiten! src/woodbury.jl:401-406,
// 136-143) for launches that MATERIALISE x = mu + U'(z - Vh T Vh'z) in HBM: the pool of the winning fits (src/multipath.jl:217),
// pfmi_draws, and every fit of an ELBO scan whose target is a device-resident closure (PFMI_TARGET_DEVICE_CALLBACK: the closure reads
// the draws where this kernel wrote them).  Any d, any history length; in-kernel generator only.
//
// Mapping (that of the single-pass scan, elbo_qf_kernel.hip): a WAVE owns 16 draws end to end -- lane (q, c) generates rows
// 16 blk + 4q + {0..3} of draw c with one Philox4x32-7 call per block -- the 8 waves of a workgroup own 8 different groups, Vh is
// only read (LDS, in MFMA operand order, resident or streamed in 256-row chunks): no cross-wave reduction, no barrier in the
// steady state.  The compact-WY form needs w = Vh'z BEFORE the first row of x can be formed, so the rows are walked twice:
//   pass 1   normals, |u|^2, head transform z_head = V'u_head, w += Vh'z           (NT quarter-size MFMAs per k-step)
//            tv = T w on v_mfma_f64_16x16x4 (lane (q, c) holds entries 4 s + q of draw c = B operand and result layout at once)
//   pass 2   the SAME normals again (counter-based generator: nothing was kept), x~ = z - Vh tv as 16x16x4 MFMAs with z as the
//            C operand, x = mu + sqrt(alpha) x~
// The two-pass kernel of round 1 (elbo_mfma_kernel.hip) keeps z in registers instead and pays for it with the rows of ONE group
// split over the 8 waves: two barriers and an LDS reduction per 16 draws, 16 draws in flight per CU, 8-byte stores scattered over
// 16 columns.  Here the second Philox + table pass buys 128 draws in flight per CU and no synchronisation.
//
// Stores (round 4): a draw is a column of d contiguous doubles and lane (q, c) ends up with rows 4q .. 4q+3 of column c -- 32 contiguous
// bytes -- which leave as two 16-byte stores per lane (every instruction touches 16 columns, 64 of each 128-byte line; L2 merges the
// halves).  Round 3 sent each block through a wave-private, XOR-swizzled LDS tile to form whole lines first: measured alone the two
// store patterns sustain the same 4.7 - 5.5 TB/s (tools/write_bench.hip), and without the tile the kernel has 32 KB of LDS back:
// sqrt(alpha) / mu are staged there with the factor block, so the block loop contains no vector-memory LOAD at all (on gfx9 stores and
// loads share vmcnt: a load issued behind a block's stores cannot be waited for without waiting for their write acknowledgements).
// Two 16-draw groups per wave, 8 waves: config-3 sample 5.9 -> 5.5 ms, config-5 shape 75 -> 53 ms (profiles/r04_experiments.md).
// logq is accumulated exactly as the scan does (same lane ownership, same order), so it is bit-identical to the scan's.
#include <type_traits>
#include "pfmi_common.h"
#include "elbo_args.h"

#ifndef XW_WAVES
#define XW_WAVES 8                     // waves per workgroup (2 per SIMD, 256 VGPRs each)
#endif
#define XW_THREADS (XW_WAVES * 64)
#ifndef XW_NG
#define XW_NG 2                        // 16-draw groups per wave: they share the operand fetches of a block
#endif
#ifndef XW_PRIO_FAIR
#define XW_PRIO_FAIR 8                 // the waves of a SIMD alternate at s_setprio 1 every XW_PRIO_FAIR blocks (0: off)
#endif
#ifndef XW_ABLATE
#define XW_ABLATE 0                    // timing experiments only, bit mask: 1 no global stores, 2 pass 2 without generator, 4 no pass 1,
                                       // 8 pass 1 without generator, 16 generator without table look-ups, 32 generator without Philox, 64 pass 2 without MFMAs
#endif
#define XW_MIN_FRONT 1024              // doubles in front of the inverse-CDF table (>= (32 - 19) * 32 * 2 = 832)
#define XW_NB PF_ICDF_NB_LDS           // all 19 binades of the inverse-CDF table in LDS
// blocks per streamed chunk; KC = 32 halves it to stay inside 160 KB of LDS
template <int KC> struct xw_chb { static constexpr int v = (KC > 20) ? 8 : 16; };

// fused operations are explicit fma(); nothing else may be contracted (results must not depend on the launch geometry)
#pragma clang fp contract(off)

typedef double xw_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ xw_d4 xw_mfma16(double a, double b, xw_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double xw_mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// LDS layout of one staged chunk: [(bl*4 + r)*NT + T][q*4 + i] = Vh[row 16 bl + 4 q + r][4 T + i]   (16 contiguous doubles = one
// A operand of the quarter-size MFMA; the same layout as the scan's)
template <int KC>
__device__ __forceinline__ int xw_vh_pos(int lrow, int col) {
    const int bl = lrow >> 4, rr = lrow & 15, q = rr >> 2, r = rr & 3, T = col >> 2, i = col & 3;
    return (((bl * 4 + r) * (KC / 4) + T) << 4) + q * 4 + i;
}

template <int KC>
__global__ __launch_bounds__(XW_THREADS) void pf_elbo_xw_kernel(ElboArgs A, int ch_blocks, int nchunks, int groups_per_wg, int ngroups) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NT = KC / 4, NG = XW_NG;
    constexpr int XW_CHB = xw_chb<KC>::v;
    constexpr int PRE = (XW_CHB * 16 * KC + XW_THREADS - 1) / XW_THREADS;      // prefetch registers per thread (streaming)
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15, l3 = lane & 3;
    const int d = A.d, nblk = (d + 15) >> 4;
    const int slot = blockIdx.y;
    const int p = A.points[slot];
    const size_t blkidx = A.by_point ? (size_t)p : (size_t)slot;
    double *out_lp = A.logp + blkidx * A.log_stride, *out_lq = A.logq + blkidx * A.log_stride;
    const int g_begin = blockIdx.x * groups_per_wg;
    const int g_end = (g_begin + groups_per_wg < ngroups) ? g_begin + groups_per_wg : ngroups;
    const int nlb = (g_end - g_begin + XW_WAVES * NG - 1) / (XW_WAVES * NG);    // batches of XW_WAVES * NG groups
    if (A.status[p] != PFMI_FIT_OK) {          // failed fit: no draws (x stays unwritten), NaN log densities
        for (int64_t n = (int64_t)g_begin * 16 + tid; n < (int64_t)g_end * 16 && n < A.N; n += XW_THREADS) {
            out_lp[n] = NAN; out_lq[n] = NAN;
        }
        return;
    }
    // ---- LDS carve-up (offsets in doubles)
    // a staged chunk = the Householder block (MFMA operand order) + per block of 16 rows: sqrt(alpha)[16], mu[16]
    const int vh_sz = ch_blocks * 16 * KC, sm_sz = ch_blocks * 32;
    const int buf_stride = (nchunks > 1) ? vh_sz + sm_sz : 0;
    const int stage_sz = (nchunks > 1 ? 2 : 1) * (vh_sz + sm_sz);
    // the index-clamped fast look-up (pf_icdf_issue_adj) may read up to (32 - XW_NB) binades x 32 x 16 B = 6.5 KB in FRONT of the table:
    // keep at least that much staged data before it (only matters for d < 64)
    double *t_s = lds + (stage_sz > XW_MIN_FRONT ? stage_sz : XW_MIN_FRONT);   // [KC][KC]
    double2 *icdf = reinterpret_cast<double2 *>(t_s + KC * KC);                 // [2 * 32 XW_NB]

    const double *Vh = A.vh + (size_t)p * d * KC, *mu = A.mu + (size_t)p * d, *sqa = A.sqrt_alpha + (size_t)p * d;
    {
        const double *T = A.tmat + (size_t)p * KC * KC;
        for (int i = tid; i < KC * KC; i += XW_THREADS) t_s[i] = T[i];
        pf_icdf_load<XW_NB>(icdf);
        {   // chunk 0 (the whole block when resident): 16-byte loads of column pairs, six in flight per thread (as a plain loop with a
            // guarded load per element this was one global round trip per element: load, s_waitcnt vmcnt(0), ds_write; round 4)
            const int npair = ch_blocks * 8 * KC, lim = d * KC;
            const double2 *src = reinterpret_cast<const double2 *>(Vh);
            constexpr int U = 6;
            for (int j0 = tid; j0 < npair; j0 += XW_THREADS * U) {
                double2 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * XW_THREADS, jc = (2 * j + 1 < lim) ? j : (lim >> 1) - 1;
                    v[u] = src[jc];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * XW_THREADS, idx = 2 * j;
                    if (j < npair) {
                        const int lrow = idx / KC, col = idx - lrow * KC;
                        *reinterpret_cast<double2 *>(lds + xw_vh_pos<KC>(lrow, col)) = (idx < lim) ? v[u] : make_double2(0.0, 0.0);
                    }
                }
            }
        }
        for (int l0 = tid; l0 < ch_blocks * 16; l0 += XW_THREADS * 2) {
            double sv[2], mv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int lrow = l0 + u * XW_THREADS, rc = lrow < d ? lrow : d - 1; sv[u] = sqa[rc]; mv[u] = mu[rc]; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int lrow = l0 + u * XW_THREADS;
                if (lrow < ch_blocks * 16) {
                    double *o = lds + vh_sz + (lrow >> 4) * 32 + (lrow & 15);
                    o[0] = (lrow < d) ? sv[u] : 0.0; o[16] = (lrow < d) ? mv[u] : 0.0;
                }
            }
        }
    }
    // head transform z_head = V'u_head on 16x16x4 MFMAs: lane (q, c) supplies A_r[i' = c][k = q] = H[rho(c)][4 q + r], H = V'
    // identity padded (rho: the row permutation that makes result register r of lane (q, c) row 4 q + r); block 1 when KC > 16
    const int rho = 4 * (c & 3) + (c >> 2);
    // KC <= 16: the four operands of H00 live in registers; KC > 16 (two groups per wave leave no room for 24 more registers) fetches
    // the operands of H00 / H10 / H11 from the Cholesky factor when one of the two head blocks of a walk comes up
    constexpr bool HREG = KC <= 16;
    const double *Vc = A.vchol + (size_t)p * KC * KC;
    auto head_op = [&](const int which, const int r) -> double {          // which: 0 = H00, 1 = H10, 2 = H11
        const int b = 4 * q + r;
        if (which == 0) { double v = (rho == b) ? 1.0 : 0.0; if (rho < KC && b < KC) v = Vc[b * KC + rho]; return v; }
        const int i1 = 16 + rho, b1 = 16 + b;
        if (which == 1) return (i1 < KC) ? Vc[b * KC + i1] : 0.0;
        double v1 = (i1 == b1) ? 1.0 : 0.0;
        if (i1 < KC && b1 < KC) v1 = Vc[b1 * KC + i1];
        return v1;
    };
    double a_h00[HREG ? 4 : 1];
    if (HREG) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a_h00[HREG ? r : 0] = head_op(0, r);
    }
    auto hop = [&](const int which, const int r) -> double {
        if (HREG && which == 0) return a_h00[HREG ? r : 0];
        return head_op(which, r);
    };
    const uint64_t seed = A.seeds[slot];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const double logdet = A.logdet[p];
    const uint32_t icdf_adj = pf_icdf_adj<XW_NB>(icdf);
    // every draw of this launch starts on a 32-byte boundary: the 16-byte stores are aligned
    const bool x_aligned = ((d & 3) == 0) && ((A.x_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(A.x) & 31) == 0);
    // the priority hand-over between the two waves of a SIMD pays when the factor block is resident (config 3: 4.93 -> 4.71 ms); with a streamed
    // block every chunk ends in a barrier that re-aligns the waves anyway, and the hand-over only costs (config-5 shape: 34.1 -> 33.0 ms
    // without it; profiles/r06_experiments.md)
#ifndef XW_FAIR_MODE
#define XW_FAIR_MODE 0                    // experiments: 1 = always, 2 = never
#endif
    const bool fair = XW_FAIR_MODE == 1 ? true : XW_FAIR_MODE == 2 ? false : nchunks == 1;
    __syncthreads();

    int cur = 0;
    for (int lb = 0; lb < nlb; ++lb) {
        // a wave owns NG 16-draw groups of the batch: they share every operand fetch of a block and give the scheduler NG independent
        // generator / MFMA chains to interleave (one wave's Philox -> table -> cubic -> MFMA chain alone leaves the SIMD half idle)
        int64_t nl0[NG];
        uint32_t n[NG];
        bool act[NG], fullc[NG];
        double *xg[NG];
        double accw[NG][NT], ntv[NG][NT], u0[NG][4], usq[NG];
        uint32_t xnext[NG][4];                                 // Philox words of the next block (software pipeline, see `normals`)
        int nact = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int grp = g_begin + (lb * XW_WAVES + wv) * NG + g;           // wave-uniform
            act[g] = grp < g_end;
            nact += act[g] ? 1 : 0;
            nl0[g] = (int64_t)grp * 16;
            n[g] = (uint32_t)(A.n0 + nl0[g] + c);
            fullc[g] = act[g] && nl0[g] + 16 <= A.N;
            xg[g] = act[g] ? A.x + (size_t)slot * A.x_stride + (size_t)nl0[g] * d : A.x;   // draw nl0 + col starts at xg + col * d
            usq[g] = 0.0;
#pragma unroll
            for (int T = 0; T < NT; ++T) { accw[g][T] = 0.0; ntv[g][T] = 0.0; }
#pragma unroll
            for (int r = 0; r < 4; ++r) u0[g][r] = 0.0;
        }
        // normals of rows 16 blk + 4q + {0..3} of the draws of group g, head transform included.  FIRST: accumulate |u|^2 (pass 1).
        // SPECIAL: the first / second / last block of the walk (head transform, rows >= d); interior blocks are straight-line code
        auto normals = [&](const int g, const int blk, double (&z)[4], auto first_tag, auto special_tag) {
            constexpr bool FIRST = decltype(first_tag)::value, SPECIAL = decltype(special_tag)::value;
            // software pipeline over the blocks of a walk: the Philox words of THIS block were computed during the previous block
            // (xnext), its table reads are issued first, and the Philox call of the NEXT block (7 dependent rounds of pure VALU work)
            // runs while they are in flight -- a wave's chain Philox -> table -> cubic -> MFMA otherwise pays every latency in turn
            uint32_t x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = xnext[g][r];
            {   // the scan's 9-instruction look-up: all four table reads in flight before the first cubic is evaluated
                double dp[4];
                double2 c01[4], c23[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#if XW_ABLATE & 16
                    dp[r] = (double)(x[r] & 0x7FFFFFFFu); c01[r] = make_double2(0.5, 1e-9); c23[r] = make_double2(1e-19, 1e-29);
#else
                    pf_icdf_issue_adj<XW_NB>(x[r], icdf_adj, dp[r], c01[r], c23[r]);
#endif
                }
#if XW_ABLATE & 32
#pragma unroll
                for (int r = 0; r < 4; ++r) xnext[g][r] = (x[r] * 2654435761u + (uint32_t)blk) ^ (n[g] + r * 0x9E3779B9u);
#else
                pf_philox_normals(n[g], (uint32_t)((blk + 1) * 4 + q), 0u, 0u, k0, k1, xnext[g]);   // (one call past the last block: discarded)
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = pf_icdf_finish(x[r], dp[r], c01[r], c23[r]);
                if (__builtin_expect(__any(pf_icdf_miss4<XW_NB>(x)), 0))         // probability 2^-XW_NB per normal
                    pf_icdf4_fix<XW_NB>(x, n[g], (uint32_t)(blk * 4 + q), 0u, k0, k1, z);
            }
            if (SPECIAL && blk == nblk - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = (blk * 16 + 4 * q + r < d) ? z[r] : 0.0;
            }
            if (FIRST) {
#pragma unroll
                for (int r = 0; r < 4; ++r) usq[g] = fma(z[r], z[r], usq[g]);  // |u|^2 before the transform (src/mvnormal.jl:31)
            }
            if (SPECIAL) {
                if (blk == 0) {                                                // z[1:k] = V'u[1:k] (src/woodbury.jl:139)
                    xw_d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) { u0[g][r] = z[r]; h = xw_mfma16(hop(0, r), z[r], h); }
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = h[r];
                } else if (KC > 16 && blk == 1) {
                    xw_d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) h = xw_mfma16(hop(1, r), u0[g][r], h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h = xw_mfma16(hop(2, r), z[r], h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = h[r];
                }
            }
        };
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int g = 0; g < NG; ++g) pf_philox_normals(n[g], (uint32_t)q, 0u, 0u, k0, k1, xnext[g]);      // block 0 of this walk
            for (int ck = 0; ck < nchunks; ++ck) {
                // ---- streaming: fetch the next chunk of the walk (pass 1 -> pass 2 -> next batch) into registers while this one is used
                double pre[PRE], pr_s = 0.0, pr_m = 0.0;
                const int nck = (ck + 1 < nchunks) ? ck + 1 : 0;
                const bool do_pre = (nchunks > 1) && (ck + 1 < nchunks || pass == 0 || lb + 1 < nlb);
                if (do_pre) {
                    const int row0 = nck * XW_CHB * 16;
                    int tid_p = tid;                               // opaque: keeps the per-thread offsets of the chunk prefetch out of the prologue (spills)
                    asm volatile("" : "+v"(tid_p));
#pragma unroll
                    for (int e = 0; e < PRE; ++e) {
                        const int idx = tid_p + e * XW_THREADS;
                        const int lrow = idx / KC, row = row0 + lrow;
                        pre[e] = (idx < XW_CHB * 16 * KC && row < d) ? Vh[(size_t)row0 * KC + idx] : 0.0;
                    }
                    if (tid_p < XW_CHB * 16 && row0 + tid_p < d) { pr_s = sqa[row0 + tid_p]; pr_m = mu[row0 + tid_p]; }
                }
                if (nact > 0) {
                    const double *vs = lds + cur * buf_stride;
                    const int blk0 = ck * ch_blocks;
                    const int nb = (nblk - blk0 < ch_blocks) ? nblk - blk0 : ch_blocks;
                    if (pass == 0) {
                        // ---- pass 1: w += Vh'z   (NGA = groups of this wave that exist: the last batch may have fewer)
                        auto body1 = [&](const int bl, auto special_tag, auto nga_tag) {
                            constexpr int NGA = decltype(nga_tag)::value;
                            const double *ap = vs + ((bl * 4) * NT << 4) + q * 4 + l3;
                            if (KC <= 16) {                      // the A tiles of the whole block are fetched before the generator runs
                                double av[4][NT];
#pragma unroll
                                for (int r = 0; r < 4; ++r)
#pragma unroll
                                    for (int T = 0; T < NT; ++T) av[r][T] = ap[(r * NT + T) << 4];
                                double z[NGA][4];
#pragma unroll
                                for (int g = 0; g < NGA; ++g) {
#if XW_ABLATE & 8
                                    z[g][0] = av[0][0]; z[g][1] = av[1][0]; z[g][2] = av[2][0]; z[g][3] = av[3][0];
#else
                                    normals(g, blk0 + bl, z[g], std::true_type{}, special_tag);
#endif
                                }
#pragma unroll
                                for (int g = 0; g < NGA; ++g)
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
#pragma unroll
                                        for (int T = 0; T < NT; ++T) accw[g][T] = xw_mfma4(av[r][T], z[g][r], accw[g][T]);
                            } else {                             // KC = 20, 32: the A tiles of ONE k-step live at a time, shared by the groups
                                double z[NGA][4];
#pragma unroll
                                for (int g = 0; g < NGA; ++g) normals(g, blk0 + bl, z[g], std::true_type{}, special_tag);
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    double a1[NT];
#pragma unroll
                                    for (int T = 0; T < NT; ++T) a1[T] = ap[(r * NT + T) << 4];
#pragma unroll
                                    for (int g = 0; g < NGA; ++g)
#pragma unroll
                                        for (int T = 0; T < NT; ++T) accw[g][T] = xw_mfma4(a1[T], z[g][r], accw[g][T]);
                                }
                            }
                        };
                        for (int bl = 0; bl < ((XW_ABLATE & 4) ? 1 : nb); ++bl) {
                            const int blk = blk0 + bl;
#if XW_PRIO_FAIR
                            if (fair && (blk & (XW_PRIO_FAIR - 1)) == 0) {   // the two waves of a SIMD take turns at the higher issue priority (see elbo_qf_kernel.hip)
                                if (((blk / XW_PRIO_FAIR) ^ (wv >> 2) ^ lb ^ pass) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                            }
#endif
                            const bool special = (blk == 0) | (blk == nblk - 1) | (KC > 16 && blk == 1);
                            // (the groups of a wave that lie beyond the last one are computed too -- their stores are predicated off)
                            if (__builtin_expect(special, 0)) body1(bl, std::true_type{}, std::integral_constant<int, NG>{});
                            else body1(bl, std::false_type{}, std::integral_constant<int, NG>{});
                        }
                    } else {
                        // ---- pass 2: the same normals again, x~ = z - Vh tv, x = mu + sqrt(alpha) x~, full-line stores
                        // FAST (round 6): every group of this wave is a full, 32-byte-aligned block of 16 draws (decided once per batch, wave-uniform):
                        // the interior body then holds no store predicate and no branch at all; the MFMA chains of the groups are issued
                        // INTERLEAVED (s outer, g inner: a group's chain order is unchanged -- same bits), so that a chain's result latency is
                        // covered by the other group's MFMAs instead of s_nop, and the fma + store tails follow the last MFMA
                        auto body2 = [&](const int bl, auto special_tag, auto nga_tag, auto fast_tag) {
                            constexpr bool SPECIAL = decltype(special_tag)::value;
                            constexpr int NGA = decltype(nga_tag)::value;
                            constexpr bool FAST = decltype(fast_tag)::value;
                            const int blk = blk0 + bl;
                            // A[i' = c][k = q] = Vh[16 blk + rho(c)][4 s + q]
                            const double *a2p = vs + ((bl * 4 + (c >> 2)) * NT << 4) + (c & 3) * 4 + q;
                            double a2v[NT], s4[4], m4[4];
#pragma unroll
                            for (int s = 0; s < NT; ++s) a2v[s] = a2p[s << 4];
                            {
                                const double *smp = vs + vh_sz + bl * 32 + 4 * q;        // rows >= d hold zeros: x = 0 there, never stored
#pragma unroll
                                for (int r = 0; r < 4; ++r) { s4[r] = smp[r]; m4[r] = smp[16 + r]; }
                            }
                            double z[NGA][4];
#pragma unroll
                            for (int g = 0; g < NGA; ++g) {
#if XW_ABLATE & 2
                                z[g][0] = s4[0]; z[g][1] = s4[1]; z[g][2] = m4[2]; z[g][3] = m4[3];
#else
                                normals(g, blk, z[g], std::false_type{}, special_tag);
#endif
                            }
                            xw_d4 xa[NGA];
#pragma unroll
                            for (int g = 0; g < NGA; ++g) xa[g] = xw_d4{z[g][0], z[g][1], z[g][2], z[g][3]};
#pragma unroll
                            for (int s = 0; s < ((XW_ABLATE & 64) ? 0 : NT); ++s)
#pragma unroll
                                for (int g = 0; g < NGA; ++g) xa[g] = xw_mfma16(a2v[s], ntv[g][s], xa[g]);
#pragma unroll
                            for (int g = 0; g < NGA; ++g) {
                                double xv[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) xv[r] = fma(s4[r], xa[g][r], m4[r]);    // x = mu + sqrt(alpha) x~
                                double *xo = xg[g] + (size_t)c * d + blk * 16 + 4 * q;              // rows 4q .. 4q+3 of draw c: 32 contiguous bytes
                                if ((XW_ABLATE & 1) && xv[0] != 1.2345e301) continue;
                                if (FAST || (!SPECIAL && fullc[g] && x_aligned)) {
                                    typedef double xw_d2 __attribute__((ext_vector_type(2)));
                                    xw_d2 lo = {xv[0], xv[1]}, hi = {xv[2], xv[3]};
                                    *reinterpret_cast<xw_d2 *>(xo) = lo;
                                    *reinterpret_cast<xw_d2 *>(xo + 2) = hi;
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        if (act[g] && blk * 16 + 4 * q + r < d && nl0[g] + c < A.N) xo[r] = xv[r];
                                }
                            }
                        };
                        bool fast_all = x_aligned;
#pragma unroll
                        for (int g = 0; g < NG; ++g) fast_all = fast_all && fullc[g];
                        fast_all = __builtin_amdgcn_readfirstlane((int)fast_all) != 0;      // wave-uniform (groups are per wave): a scalar branch, no exec masking
                        for (int bl = 0; bl < nb; ++bl) {
                            const int blk = blk0 + bl;
#if XW_PRIO_FAIR
                            if (fair && (blk & (XW_PRIO_FAIR - 1)) == 0) {   // the two waves of a SIMD take turns at the higher issue priority (see elbo_qf_kernel.hip)
                                if (((blk / XW_PRIO_FAIR) ^ (wv >> 2) ^ lb ^ pass) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                            }
#endif
                            const bool special = (blk == 0) | (blk == nblk - 1) | (KC > 16 && blk == 1);
                            // (the groups of a wave that lie beyond the last one are computed too -- their stores are predicated off)
                            if (__builtin_expect(special, 0)) body2(bl, std::true_type{}, std::integral_constant<int, NG>{}, std::false_type{});
                            else if (fast_all) body2(bl, std::false_type{}, std::integral_constant<int, NG>{}, std::true_type{});
                            else body2(bl, std::false_type{}, std::integral_constant<int, NG>{}, std::false_type{});
                        }
                    }
                }
                if (nchunks > 1) {
                    if (do_pre) {
                        double *vs = lds + (cur ^ 1) * buf_stride;
                        int tid_e = tid;
                        asm volatile("" : "+v"(tid_e));
#pragma unroll
                        for (int e = 0; e < PRE; ++e) {
                            const int idx = tid_e + e * XW_THREADS;
                            if (idx < XW_CHB * 16 * KC) { const int lrow = idx / KC; vs[xw_vh_pos<KC>(lrow, idx - lrow * KC)] = pre[e]; }
                        }
                        if (tid_e < XW_CHB * 16) {
                            double *o = vs + vh_sz + (tid_e >> 4) * 32 + (tid_e & 15);
                            o[0] = pr_s; o[16] = pr_m;
                        }
                    }
                    __syncthreads();
                    cur ^= 1;
                }
            }
            if (pass == 0) {
                int lane_t = lane;                                 // opaque lane coordinates: the LDS offsets of T are not kept alive across the walks
                asm volatile("" : "+v"(lane_t));
                const int qt = lane_t >> 4, ct = lane_t & 15;
                // tv = T w on v_mfma_f64_16x16x4: lane (q, c) holds entries 4 s + q of draw c -- B operand (k = q, column = draw) and
                // result (rows q + 4 reg) layout at once; A = T[row 16 rt + c][4 s + q] from LDS
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int rt = 0; rt < (NT + 3) / 4; ++rt) {
                        xw_d4 acc = {0.0, 0.0, 0.0, 0.0};
                        const int row = 16 * rt + ct;
#pragma unroll
                        for (int st = 0; st < NT; ++st) {
                            const int col = 4 * st + qt;
                            const double av = (row < KC) ? t_s[row * KC + col] : 0.0;
                            acc = xw_mfma16(av, accw[g][st], acc);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (4 * rt + r < NT) ntv[g][4 * rt + r] = -acc[r];
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (!act[g]) continue;
            const double us = pf_sum_q(usq[g]);
            const int64_t nl = nl0[g] + c;
            if (q == 0 && nl < A.N) {
                out_lq[nl] = ((double)d * PF_LOG2PI + logdet + us) / -2.0;      // src/mvnormal.jl:36
                out_lp[nl] = NAN;                                                // the target is evaluated by the caller's next launch
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
static size_t xw_lds_bytes(int ch_blocks, int nchunks, int kc) {
    size_t stage = ((size_t)ch_blocks * 16 * kc + (size_t)ch_blocks * 32) * (nchunks > 1 ? 2 : 1);
    if (stage < XW_MIN_FRONT) stage = XW_MIN_FRONT;
    return sizeof(double) * (stage + (size_t)kc * kc + 4 * (XW_NB << PF_ICDF_B));
}

template <int KC>
static int32_t launch_xw(pfmi_ctx *c, const ElboArgs &a, int64_t nfits) {
    const int nblk = (a.d + 15) / 16;
    int ch_blocks = nblk, nchunks = 1;
    constexpr int XW_CHB = xw_chb<KC>::v;
    if (xw_lds_bytes(nblk, 1, KC) > 160 * 1024) { ch_blocks = XW_CHB; nchunks = (nblk + XW_CHB - 1) / XW_CHB; }
    const size_t lds_bytes = xw_lds_bytes(ch_blocks, nchunks, KC);
    PF_CHECK(lds_bytes <= 160 * 1024, PFMI_ERR_UNSUPPORTED, "xw kernel LDS %zu too large", lds_bytes);
    auto kern = pf_elbo_xw_kernel<KC>;
    PF_TRY(pf_raise_lds_limit(c, reinterpret_cast<const void *>(kern), 160 * 1024));
    const int ngroups = (int)((a.N + 15) / 16);
    // one workgroup per fit; a fit's groups are cut into several workgroups (whole batches of 8 groups) while there are fewer
    // workgroups than CUs can hold
    int split = 1;
    constexpr int SLOTS = XW_WAVES * XW_NG;
    // (a streamed Vh is re-read per batch whatever the cut, so pieces may shrink to a few groups; a resident one is staged once per
    // piece: whole batches only)
    const int min_piece = nchunks > 1 ? 4 : SLOTS;
    while ((int64_t)split * nfits < 2 * (c->ncu > 0 ? c->ncu : 256) && ngroups / (split * 2) >= min_piece) split *= 2;
    int gpw = (ngroups + split - 1) / split;
    if (nchunks == 1) gpw = (gpw + SLOTS - 1) / SLOTS * SLOTS;
    const int gx = (ngroups + gpw - 1) / gpw;
    for (int64_t s0 = 0; s0 < nfits; s0 += 32768) {
        const int64_t ns = (nfits - s0 < 32768) ? (nfits - s0) : 32768;
        ElboArgs b = a;
        b.points = a.points + s0; b.seeds = a.seeds + s0;
        b.x = a.x + s0 * a.x_stride;
        if (!a.by_point) { b.logp += s0 * a.log_stride; b.logq += s0 * a.log_stride; }
        hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ns), dim3(XW_THREADS), lds_bytes, c->stream, b, ch_blocks, nchunks, gpw, ngroups);
    }
    return PFMI_OK;
}

// draws written, in-kernel generator, no target; kpad in {4, 8, 12, 16, 20, 32}
int32_t pf_launch_elbo_xw(pfmi_ctx *c, const ElboArgs &a, int64_t nfits, bool *handled) {
    *handled = false;
    if (a.u != nullptr || a.x == nullptr) return PFMI_OK;
    *handled = true;
    switch (c->kpad) {
        case 4: return launch_xw<4>(c, a, nfits);
        case 8: return launch_xw<8>(c, a, nfits);
        case 12: return launch_xw<12>(c, a, nfits);
        case 16: return launch_xw<16>(c, a, nfits);
        case 20: return launch_xw<20>(c, a, nfits);
        case 32: return launch_xw<32>(c, a, nfits);
        default: *handled = false; return PFMI_OK;
    }
}
