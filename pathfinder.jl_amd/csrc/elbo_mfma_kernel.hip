// elbo_mfma_kernel.hip -- the production ELBO draw kernel for d <= 1024: rand_and_logpdf + target
// (reference src/mvnormal.jl:24-39, src/elbo.jl:12-16) with the standard normals held in REGISTERS and the
// Householder contractions on the f64 matrix cores.
//
// Work unit: one 512-thread workgroup (8 waves, one per CU) owns one fit; its factor (Householder block
// Vh [d][KC], mu, sqrt(alpha), T, target diagonal) is staged once into LDS (~150 KB at d = 1000, KC = 12)
// and the workgroup then walks over groups of 16 draws.  Within a group the d rows are split over the 8
// waves in 16-row blocks; a lane is (q = lane>>4, c = lane&15): it owns rows 16*blk + 4q + {0,1,2,3} of
// draw c, i.e. exactly one Philox4x32 call (4 normals) per block.  That ownership IS the C/D fragment
// layout of v_mfma_f64_16x16x4_f64 (row = (lane>>4) + 4*reg) up to a row permutation applied to the A operand,
// so z never leaves registers between the two contractions:
//   pass 1  W[j][c]   += sum_rows Vh[row][j] z[row][c]        A = Vh^T (4 MFMAs / block), B = z
//   (LDS)   tv = T (sum over waves of W)                      (Q = I - Vh T Vh', compact WY)
//   pass 2  x~[row][c] = z[row][c] - sum_j Vh[row][j] tv[j][c] A = Vh (KC/4 MFMAs / block), B = -tv, C = z
//   target  t[j'][c]  += sum_rows Wd[row][j'] e[row][c]        A = Wd^T (4 MFMAs / block), B = e = x - mean
// The VALU only generates the normals (Philox + table inverse CDF) and does the elementwise epilogue; MFMA is used
// only for these true contractions over the d x 2m history block (BASELINE.json north_star).  Nothing but the
// per-draw logp/logq (and the draws of a winning fit, when asked) is written to HBM.
//
// Results are identical (to fp64 roundoff) to the lane-per-draw kernel in elbo_kernels.hip, which remains the
// general path (d > 1024, history_length > 8, parity mode with host-supplied normals).
#include "pfmi_common.h"
#include "elbo_args.h"

#define MF_THREADS 512
#define MF_WAVES 8
// binades of the inverse-CDF table kept in LDS: 12 (12 KB) instead of 19 -- at d = 1000 this kernel uses 148 KB of LDS for the
// factor block and its reduction buffers; words below 2^19 (probability 2^-12 per normal) take the global-table path
#define MF_ICDF_NB 12

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 pf_mfma(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products, 16 cycles.  Lane layout measured on gfx950
// (tools/probe_mfma4x4.hip, profiles/r01_probe_mfma_f64_4x4x4.txt):
//   A: lane = 16 k + 4 blk + i     B: lane = 16 k + 4 blk + j     D: lane = 16 i + 4 blk + j
// With blk = draw quartet (c >> 2), j = c & 3, k = q this is exactly the (q, c) row ownership of this kernel, and a
// result row i lands in lane (q' = i, c): the same place as register `reg` of the 16x16x4 C/D layout when the
// instruction index I plays the role of reg (row = q' + 4 I).  Contractions whose output has fewer than 16 rows
// (pass 1: KC = 12 history columns; target: RPAD = 8) therefore cost KC/4 resp. RPAD/4 quarter-size MFMAs instead
// of one full 16-row MFMA: 48 resp. 32 cycles per k-step instead of 64.
__device__ __forceinline__ double pf_mfma4(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// TGT: 0 none, 1 Gaussian family (RPAD = 0 or 8 or 16 low-rank columns), 2 funnel
template <int KC, int NBW, int TGT, int RPAD, bool WX>
__global__ __launch_bounds__(MF_THREADS) void pf_elbo_mfma_kernel(ElboArgs A, int groups_per_block, int ngroups) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr bool FOLD = (TGT == 1) && !WX;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15;
    const int d = A.d;
    const int nblk = (d + 15) >> 4;            // 16-row blocks that contain real rows
    const int rows = nblk * 16;
    const int slot = blockIdx.y;
    const int p = A.points[slot];
    const size_t blkidx = A.by_point ? (size_t)p : (size_t)slot;
    double *out_lp = A.logp + blkidx * A.log_stride;
    double *out_lq = A.logq + blkidx * A.log_stride;
    const int g_begin = blockIdx.x * groups_per_block;
    const int g_end = (g_begin + groups_per_block < ngroups) ? g_begin + groups_per_block : ngroups;

    if (A.status[p] != PFMI_FIT_OK) {          // failed fit: NaN, like an exception in the reference
        for (int64_t n = (int64_t)g_begin * 16 + tid; n < (int64_t)g_end * 16 && n < A.N; n += MF_THREADS) {
            out_lp[n] = NAN;
            out_lq[n] = NAN;
        }
        return;
    }

    // ---- LDS carve-up (doubles)
    double *vh_s = lds;                        // [rows][KC]
    double *mu_s = vh_s + (size_t)rows * KC;   // [rows]
    double *sqa_s = mu_s + rows;               // [rows]
    double *tm_s = sqa_s + rows;               // [rows] target mean       (TGT == 1)
    double *t_s = tm_s + rows;                 // [KC][KC] compact-WY T
    double *red = t_s + KC * KC;               // [8][64 * KC/4] per-wave partial W tiles (rows j < KC only)
    double *redb = red + MF_WAVES * 64 * (KC / 4);            // [8][64 * ceil(RPAD/4)] per-wave partial target tiles
    double *wsum = redb + MF_WAVES * 64 * ((RPAD + 3) / 4);   // [256] summed W tile
    double *tsum = wsum + 256;                 // [64 * ceil(RPAD/4)] summed target tile
    double *ssum = tsum + 64 * ((RPAD + 3) / 4);              // [2][16] summed |u|^2 and diagonal quadratic form
    double *gs = ssum + 32;                    // [RPAD][RPAD] target capacitance factor
    double *red2 = gs + RPAD * RPAD;           // [8][16][4] per-wave per-draw scalars
    double2 *icdf = reinterpret_cast<double2 *>(red2 + MF_WAVES * 64);     // [2 * 608] inverse-CDF table of the generator
    double *zero_s = red2 + MF_WAVES * 64 + 4 * (MF_ICDF_NB << PF_ICDF_B);       // [2] zeros (masked A-operand lanes)

    {
        const double *Vh = A.vh + (size_t)p * d * KC;
        {   // six 16-byte loads in flight per thread (a guarded load per element was a global round trip per element; round 4)
            const int npair = rows * KC / 2, lim = d * KC;
            const double2 *src = reinterpret_cast<const double2 *>(Vh);
            for (int j0 = tid; j0 < npair; j0 += MF_THREADS * 6) {
                double2 v[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) { const int j = j0 + u * MF_THREADS; v[u] = src[(2 * j + 1 < lim) ? j : (lim >> 1) - 1]; }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int j = j0 + u * MF_THREADS;
                    if (j < npair) *reinterpret_cast<double2 *>(vh_s + 2 * j) = (2 * j < lim) ? v[u] : make_double2(0.0, 0.0);
                }
            }
        }
        const double *mu = A.mu + (size_t)p * d, *sqa = A.sqrt_alpha + (size_t)p * d;
        // ELBO scan of a Gaussian-family target (no draws written): only e = x - m is needed, so the three per-row LDS
        // vectors hold (mu - m, sqrt(alpha), a); otherwise (mu, sqrt(alpha), m) and the diagonal precision comes from L2
        for (int i = tid; i < rows; i += MF_THREADS) {
            const double mi = (i < d) ? mu[i] : 0.0;
            sqa_s[i] = (i < d) ? sqa[i] : 0.0;
            if (FOLD) {
                mu_s[i] = (i < d) ? mi - A.t_mean[i] : 0.0;
                tm_s[i] = (i < d) ? A.t_a[i] : 0.0;
            } else {
                mu_s[i] = mi;
                if (TGT == 1) tm_s[i] = (i < d) ? A.t_mean[i] : 0.0;
            }
        }
        const double *T = A.tmat + (size_t)p * KC * KC;
        for (int i = tid; i < KC * KC; i += MF_THREADS) t_s[i] = T[i];
        if (TGT == 1 && RPAD > 0) for (int i = tid; i < RPAD * RPAD; i += MF_THREADS) gs[i] = A.t_g[i];
        pf_icdf_load<MF_ICDF_NB>(icdf);
        if (tid < 2) zero_s[tid] = 0.0;
    }
    // head transform z_head = V' u_head as 4 MFMAs: A_r[i'][k] = M[rho(i')][4k + r], M = V' (identity padded)
    const int rho = 4 * (c & 3) + (c >> 2);    // row permutation of the A operand (lane&15 = c here = i')
    double a_head[4];
    {
        const double *Vc = A.vchol + (size_t)p * KC * KC;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = rho, b = 4 * q + r;  // M[i][b]
            double v = (i == b) ? 1.0 : 0.0;
            if (i < KC && b < KC) v = Vc[b * KC + i];
            a_head[r] = v;
        }
    }
    const uint64_t seed = A.seeds[slot];
    const double logdet = A.logdet[p];
    __syncthreads();

    // ---- software pipeline over draw groups: while group g is finished (pass 2 + target, MFMA heavy), the
    //      normals of group g+1 are generated (VALU heavy) into the SAME registers block by block, and its pass-1
    //      contraction is issued.  Every block body is straight-line code mixing ~11 MFMAs with the RNG VALU
    //      stream so that one wave keeps both pipes busy.
    constexpr int WR = KC / 4;                                   // registers of the W tile that carry rows j < KC
    constexpr int TR = (RPAD + 3) / 4;                           // registers of the target tile that carry rows j' < RPAD
    double z[NBW][4];
    double usq_next = 0.0;
    double accw[KC / 4];                                         // W tile: accw[I] = W[4 I + q][c]
#pragma unroll
    for (int I = 0; I < KC / 4; ++I) accw[I] = 0.0;
    double ntv[KC / 4];

    // normals of rows 16 blk + 4q + {0..3} of draw n (+ head transform for block 0) and their pass-1 MFMAs
    auto pass1_block = [&](const int blk, const uint32_t n, const bool head, double (&zz)[4]) {
        uint32_t x[4];
        pf_philox_normals(n, (uint32_t)(blk * 4 + q), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
        pf_icdf4<MF_ICDF_NB>(x, n, (uint32_t)(blk * 4 + q), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), icdf, zz);
        const int rowbase = blk * 16 + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zz[r] = (rowbase + r < d) ? zz[r] : 0.0;                // rows >= d do not exist
            usq_next += zz[r] * zz[r];                              // |u|^2 before the transform (src/mvnormal.jl:31)
        }
        if (head) {                                                 // z[1:k] = V' u[1:k]  (src/woodbury.jl:139)
            d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) h = pf_mfma(a_head[r], zz[r], h);
#pragma unroll
            for (int r = 0; r < 4; ++r) zz[r] = h[r];
        }
    };
    const int l3 = lane & 3;
    auto pass1_mfma_r = [&](const int blk, const int r, const double zr) {         // one k-step (4 rows) of pass 1
        const double *ap = vh_s + (blk * 16 + 4 * q + r) * KC + l3;                // A[i][k = q] = Vh[row 4q + r][4 I + i]
#pragma unroll
        for (int I = 0; I < KC / 4; ++I) accw[I] = pf_mfma4(ap[4 * I], zr, accw[I]);
    };
    auto pass1_mfma = [&](const int blk, const double (&zz)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pass1_mfma_r(blk, r, zz[r]);
    };
    // W: sum the 8 per-wave tiles, tv = T W  (lane (q,c) needs tv[4s + q][c]); two barriers
    auto reduce_w = [&]() {
        if (tid < 64 * WR) {
            double s = red[tid];
#pragma unroll
            for (int w = 1; w < MF_WAVES; ++w) s += red[w * 64 * WR + tid];
            wsum[tid] = s;
        }
    };
    auto compute_ntv = [&]() {
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const int jp = 4 * s + q;
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < KC; ++j)       // W[j][c] lives at entry (reg = j >> 2, lane = (j & 3) * 16 + c)
                v += t_s[jp * KC + j] * wsum[(j >> 2) * 64 + (j & 3) * 16 + c];
            ntv[s] = -v;
        }
    };

    int wvs = __builtin_amdgcn_readfirstlane(wv);
    asm volatile("" : "+s"(wvs));
    uint32_t xc[4];                      // Philox output of the NEXT (group, block) pair to be transformed
    {   // prologue: pass 1 of the first group
        const uint32_t n = (uint32_t)(A.n0 + (int64_t)g_begin * 16 + c);
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int blk = wvs * NBW + b;
            if (blk < nblk) {
                double zz[4];
                pass1_block(blk, n, blk == 0, zz);
                pass1_mfma(blk, zz);
#pragma unroll
                for (int r = 0; r < 4; ++r) z[b][r] = zz[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) z[b][r] = 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pf_philox_normals(n + 16u, (uint32_t)(wvs * NBW * 4 + q), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), xc);
#pragma unroll
        for (int reg = 0; reg < WR; ++reg) red[wv * 64 * WR + reg * 64 + lane] = accw[reg];   // entry (j = q + 4 reg, c)
        __syncthreads();
        reduce_w();
        __syncthreads();
        compute_ntv();
    }

    for (int grp = g_begin; grp < g_end; ++grp) {
        // the wave index is made opaque once per group so that nothing derived from the block index is treated as
        // loop-invariant (LLVM would hoist per-block Philox partial products / addresses and spill them)
        wvs = __builtin_amdgcn_readfirstlane(wv);
        asm volatile("" : "+s"(wvs));
        const int64_t nl = (int64_t)grp * 16 + c;                 // local draw index of this lane's column
        const bool valid = nl < A.N;
        const uint32_t n_next = (uint32_t)(A.n0 + nl + 16);       // the group whose normals are generated in this iteration
        const double usq_cur = usq_next;
        usq_next = 0.0;
#pragma unroll
        for (int I = 0; I < KC / 4; ++I) accw[I] = 0.0;
        double qd = 0.0, tau = 0.0;
        double acc3[TR > 0 ? TR : 1];                              // target tile: acc3[I] = t[4 I + q][c]
#pragma unroll
        for (int I = 0; I < (TR > 0 ? TR : 1); ++I) acc3[I] = 0.0;
        double *X = (WX && valid) ? (A.x + (size_t)slot * A.x_stride + (size_t)nl * d) : nullptr;
        // Block body.  Independent dependency chains are advanced side by side in each of 11 phases -- the table look-ups of
        // (next group, this block: issued in P0, finished in P6) and the Philox call of the following block -- and one MFMA is
        // issued per phase (3 pass-2, 4 pass-1 of the previous block, 4 target), with a scheduling barrier after each
        // phase: the wave keeps the matrix pipe (64 cycles per f64 MFMA) and the VALU busy at once, and the VALU
        // always has an independent instruction to issue while a dependent fp64 result is in flight.
#define PF_PHASE_END() __builtin_amdgcn_sched_barrier(0)
        // pin a value to the phase that produced it (IR-level sinking would otherwise move the work to its first use
        // and unbalance the phases); costs no instruction
#define PF_PIN(x) asm volatile("" : "+v"(x))
#define PF_PIN_RNG() do { PF_PIN(c0); PF_PIN(c1); PF_PIN(c2); PF_PIN(c3); } while (0)
        auto body = [&](const int b, const int blk, const bool head) {
            const int rowbase = blk * 16 + 4 * q;
            // ---- operand fetch (LDS / L2) for this block, issued up front
            double a3[4][TR > 0 ? TR : 1], ta4[4] = {0.0, 0.0, 0.0, 0.0};
            if (TGT == 1) {
                if (RPAD > 0) {
                    const double *w16 = A.t_wd16 + ((size_t)blk * 16 + 4 * q) * 16 + l3;  // A[i][k = q] = Wd[row 4q + r][4 I + i]
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int I = 0; I < TR; ++I) a3[r][I] = w16[r * 16 + 4 * I];
                }
                if (!FOLD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ta4[r] = A.t_a[(rowbase + r < d) ? rowbase + r : d - 1];
                }
            }
            const double *a2p = vh_s + (blk * 16 + rho) * KC + q;                       // A[i'][k = q] = Vh[16 blk + rho(i')][4s + q]
            double a2v[KC / 4];
#pragma unroll
            for (int s2 = 0; s2 < KC / 4; ++s2) a2v[s2] = a2p[4 * s2];
            const double mu4[4] = {mu_s[rowbase], mu_s[rowbase + 1], mu_s[rowbase + 2], mu_s[rowbase + 3]};
            const double sq4[4] = {sqa_s[rowbase], sqa_s[rowbase + 1], sqa_s[rowbase + 2], sqa_s[rowbase + 3]};
            const double tm4[4] = {tm_s[rowbase], tm_s[rowbase + 1], tm_s[rowbase + 2], tm_s[rowbase + 3]};
            d4 xa = {z[b][0], z[b][1], z[b][2], z[b][3]};
            // Philox state of the FOLLOWING pair: next block of this wave, or block 0 of the group after
            const bool nbv = (b + 1 < NBW) && (blk + 1 < nblk);
            uint32_t c0 = nbv ? n_next : n_next + 16u;
            uint32_t c1 = (uint32_t)((nbv ? blk + 1 : wvs * NBW) * 4 + q), c2 = 0u, c3 = 0u;
            uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
            double zz[4], e4[4], xi4[4];
            // P0: the normals of (next group, this block) come from the Philox words xc = philox(n_next, 4 blk + q): start the
            // interval look-ups now, evaluate the cubics in P6 (the table reads land behind five phases of other work)
            xa = pf_mfma(a2v[0], ntv[0], xa);
            double dpv[4];
            double2 c01v[4], c23v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pf_icdf_issue<MF_ICDF_NB>(xc[r], icdf, dpv[r], c01v[r], c23v[r]);
            const uint32_t xw[4] = {xc[0], xc[1], xc[2], xc[3]};
            pf_philox_round(c0, c1, c2, c3, k0, k1);
            PF_PIN_RNG();
            PF_PHASE_END();
            // P1
            if (KC >= 8) xa = pf_mfma(a2v[KC >= 8 ? 1 : 0], ntv[KC >= 8 ? 1 : 0], xa);
            pf_philox_round(c0, c1, c2, c3, k0, k1);
            PF_PIN_RNG();
            PF_PHASE_END();
            // P2
            if (KC >= 12) xa = pf_mfma(a2v[KC >= 12 ? 2 : 0], ntv[KC >= 12 ? 2 : 0], xa);
            if (KC >= 16) xa = pf_mfma(a2v[KC >= 16 ? 3 : 0], ntv[KC >= 16 ? 3 : 0], xa);
            pf_philox_round(c0, c1, c2, c3, k0, k1);
            PF_PIN_RNG();
            PF_PHASE_END();
            // P3
            if (b > 0) pass1_mfma_r(blk - 1, 0, z[b > 0 ? b - 1 : 0][0]);
            pf_philox_round(c0, c1, c2, c3, k0, k1);
            PF_PIN_RNG();
            PF_PHASE_END();
            // P4: first half of the epilogue of the current group: x = mu + sqrt(alpha) x~
            if (b > 0) pass1_mfma_r(blk - 1, 1, z[b > 0 ? b - 1 : 0][1]);
            pf_philox_round(c0, c1, c2, c3, k0, k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xi4[r] = mu4[r] + sq4[r] * xa[r];                                       // FOLD: this is already e = x - m
                e4[r] = FOLD ? xi4[r] : xi4[r] - tm4[r];
            }
            PF_PIN_RNG();
#pragma unroll
            for (int r = 0; r < 4; ++r) PF_PIN(e4[r]);
            PF_PHASE_END();
            // P5: second half: target accumulation, optional store
            if (b > 0) pass1_mfma_r(blk - 1, 2, z[b > 0 ? b - 1 : 0][2]);
            pf_philox_round(c0, c1, c2, c3, k0, k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowbase + r;
                if (TGT == 1) qd += (FOLD ? tm4[r] : ta4[r]) * e4[r] * e4[r];           // padded rows: e = 0
                else if (TGT == 2) { if (row == 0) tau = xi4[r]; else qd += xi4[r] * xi4[r]; }   // padded rows: xi = 0
                if (WX) { if (X && row < d) X[row] = xi4[r]; }
            }
            PF_PIN_RNG(); PF_PIN(qd);
            PF_PHASE_END();
            // P6
            if (TGT == 1 && RPAD > 0) {
#pragma unroll
                for (int I = 0; I < TR; ++I) acc3[I] = pf_mfma4(a3[0][I], e4[0], acc3[I]);
            }
            pf_philox_round(c0, c1, c2, c3, k0, k1);
#pragma unroll
            for (int r = 0; r < 4; ++r) zz[r] = pf_icdf_finish(xw[r], dpv[r], c01v[r], c23v[r]);
            if (__builtin_expect(__any(pf_icdf_miss4<MF_ICDF_NB>(xw)), 0))
                pf_icdf4_fix<MF_ICDF_NB>(xw, n_next, (uint32_t)(blk * 4 + q), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), zz);
            PF_PIN_RNG(); PF_PIN(zz[0]); PF_PIN(zz[1]); PF_PIN(zz[2]); PF_PIN(zz[3]);
            PF_PHASE_END();
            // P7
            if (TGT == 1 && RPAD > 0) {
#pragma unroll
                for (int I = 0; I < TR; ++I) acc3[I] = pf_mfma4(a3[1][I], e4[1], acc3[I]);
            }
            if (PF_NORMAL_ROUNDS > 7) pf_philox_round(c0, c1, c2, c3, k0, k1);
            if (blk == nblk - 1) {                                                      // rows >= d exist only in the last block
#pragma unroll
                for (int r = 0; r < 4; ++r) zz[r] = (rowbase + r < d) ? zz[r] : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) usq_next += zz[r] * zz[r];                      // |u|^2 before the transform
            PF_PIN_RNG(); PF_PIN(usq_next);
            PF_PHASE_END();
            // P8
            if (TGT == 1 && RPAD > 0) {
#pragma unroll
                for (int I = 0; I < TR; ++I) acc3[I] = pf_mfma4(a3[2][I], e4[2], acc3[I]);
            }
            if (PF_NORMAL_ROUNDS > 8) pf_philox_round(c0, c1, c2, c3, k0, k1);
            PF_PIN_RNG();
            PF_PHASE_END();
            // P9
            if (TGT == 1 && RPAD > 0) {
#pragma unroll
                for (int I = 0; I < TR; ++I) acc3[I] = pf_mfma4(a3[3][I], e4[3], acc3[I]);
            }
            if (PF_NORMAL_ROUNDS > 9) pf_philox_round(c0, c1, c2, c3, k0, k1);
            if (head) {                                                                 // z[1:k] = V' u[1:k]  (src/woodbury.jl:139)
                d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) h = pf_mfma(a_head[r], zz[r], h);
#pragma unroll
                for (int r = 0; r < 4; ++r) zz[r] = h[r];
            }
            PF_PHASE_END();
            // P10
            if (b > 0) pass1_mfma_r(blk - 1, 3, z[b > 0 ? b - 1 : 0][3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[b][r] = zz[r];
            xc[0] = c0; xc[1] = c1; xc[2] = c2; xc[3] = c3;
            PF_PHASE_END();
        };
#undef PF_PHASE_END
#undef PF_PIN
#undef PF_PIN_RNG
#pragma unroll
        for (int b = 0; b <= NBW; ++b) {
            const int blk = wvs * NBW + b;
            if (b < NBW && blk < nblk) {
                if (b == 0 && blk == 0) body(b, blk, true);
                else body(b, blk, false);
            } else if (b > 0 && blk - 1 < nblk) {
                pass1_mfma(blk - 1, z[b > 0 ? b - 1 : 0]);             // tail: pass 1 of the last block of this wave
            }
        }
        // ---------------- per-draw reductions over q (lanes) and waves (LDS), W tile of the next group
        double usq = usq_cur;
        usq += __shfl_xor(usq, 16, 64); usq += __shfl_xor(usq, 32, 64);
        qd += __shfl_xor(qd, 16, 64);   qd += __shfl_xor(qd, 32, 64);
        if (q == 0) {
            red2[(wv * 16 + c) * 4 + 0] = usq;
            red2[(wv * 16 + c) * 4 + 1] = qd;
            if (TGT == 2 && wv == 0) red2[(wv * 16 + c) * 4 + 2] = tau;
        }
        if (TGT == 1 && RPAD > 0) {
#pragma unroll
            for (int reg = 0; reg < TR; ++reg) redb[wv * 64 * (TR > 0 ? TR : 1) + reg * 64 + lane] = acc3[reg];
        }
#pragma unroll
        for (int reg = 0; reg < WR; ++reg) red[wv * 64 * WR + reg * 64 + lane] = accw[reg];
        __syncthreads();
        // stage 1 (parallel, between the barriers): sums over the 8 waves
        reduce_w();                                                                    // threads 0 .. 64 WR - 1
        if (TGT == 1 && RPAD > 0 && tid >= 256 && tid < 256 + 64 * TR) {
            const int e = tid - 256;
            double sacc = redb[e];
#pragma unroll
            for (int w = 1; w < MF_WAVES; ++w) sacc += redb[w * 64 * (TR > 0 ? TR : 1) + e];
            tsum[e] = sacc;
        }
        if (tid >= 448 && tid < 480) {
            const int cc = (tid - 448) & 15, which = (tid - 448) >> 4;
            double sacc = 0.0;
#pragma unroll
            for (int w = 0; w < MF_WAVES; ++w) sacc += red2[(w * 16 + cc) * 4 + which];
            ssum[which * 16 + cc] = sacc;
        }
        __syncthreads();
        compute_ntv();
        if (tid >= 256 && tid < 272) {                             // finalise the current group (16 lanes of wave 4)
            const int cc = tid - 256;
            const double us = ssum[cc], qq = ssum[16 + cc];
            double lp = NAN;
            if (TGT == 1) {
                double corr = 0.0;
                if (RPAD > 0) {
                    double t[RPAD > 0 ? RPAD : 1];
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) t[j] = tsum[(j >> 2) * 64 + (j & 3) * 16 + cc];
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) {
                        double g = 0.0;
#pragma unroll
                        for (int l = 0; l <= j; ++l) g += gs[j * RPAD + l] * t[l];
                        corr += g * g;
                    }
                }
                lp = A.t_offset - 0.5 * (qq - corr);
            } else if (TGT == 2) {
                const double ta = red2[(0 * 16 + cc) * 4 + 2], t3 = ta / 3.0;
                lp = (t3 * t3 + (double)(d - 1) * ta + qq * exp(-ta)) / -2.0;
            }
            const int64_t no = (int64_t)grp * 16 + cc;
            if (no < A.N) {
                out_lq[no] = ((double)d * PF_LOG2PI + logdet + us) / -2.0;       // src/mvnormal.jl:36
                out_lp[no] = lp;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
static size_t mf_lds_bytes(int d, int kc, int rpad) {
    const size_t rows = (size_t)((d + 15) / 16) * 16;
    return sizeof(double) * (rows * kc + 3 * rows + (size_t)kc * kc + MF_WAVES * 64 * (kc / 4) +
                             MF_WAVES * 64 * ((rpad + 3) / 4) + 256 + 64 * ((rpad + 3) / 4) + 32 + (size_t)rpad * rpad +
                             MF_WAVES * 64 + 4 * (MF_ICDF_NB << PF_ICDF_B) + 2);
}

template <int KC, int NBW, int TGT, int RPAD, bool WX>
static int32_t launch_mf(pfmi_ctx *c, const ElboArgs &a, int64_t nfits) {
    const size_t lds_bytes = mf_lds_bytes(a.d, KC, RPAD);
    PF_CHECK(lds_bytes <= 160 * 1024, PFMI_ERR_UNSUPPORTED, "mfma kernel LDS %zu too large", lds_bytes);
    auto kern = pf_elbo_mfma_kernel<KC, NBW, TGT, RPAD, WX>;
    PF_TRY(pf_raise_lds_limit(c, reinterpret_cast<const void *>(kern), 160 * 1024));
    const int ngroups = (int)((a.N + 15) / 16);
    // one fit per workgroup; split a fit's draw groups over several workgroups only when there are few fits -- until every CU has a
    // workgroup, not further: each piece stages the whole factor (97 KB at d = 1000) into LDS again (pool_build of config 3, 64 fits:
    // 16 pieces per fit 0.33 ms, 8 pieces 0.28 ms, 4 pieces 0.26 ms)
    int ncu = 0;
    PF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device));
    int split = 1;
    while ((int64_t)split * nfits < ncu && split * 2 <= ngroups) split *= 2;
    const int gpb = (ngroups + split - 1) / split;
    const int gx = (ngroups + gpb - 1) / gpb;
    for (int64_t s0 = 0; s0 < nfits; s0 += 32768) {
        const int64_t ns = (nfits - s0 < 32768) ? (nfits - s0) : 32768;
        ElboArgs b = a;
        b.points = a.points + s0; b.seeds = a.seeds + s0;
        if (b.x) b.x += s0 * a.x_stride;
        if (!a.by_point) { b.logp += s0 * a.log_stride; b.logq += s0 * a.log_stride; }
        hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ns), dim3(MF_THREADS), lds_bytes, c->stream, b, gpb, ngroups);
    }
    return PFMI_OK;
}

template <int KC, int NBW, int TGT, int RPAD>
static int32_t launch_mf_w(pfmi_ctx *c, const ElboArgs &a, int64_t nfits) {
    if (a.x) return launch_mf<KC, NBW, TGT, RPAD, true>(c, a, nfits);
    return launch_mf<KC, NBW, TGT, RPAD, false>(c, a, nfits);
}

template <int KC, int NBW>
static int32_t launch_mf_t(pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad) {
    if (tgt == 0) return launch_mf_w<KC, NBW, 0, 0>(c, a, nfits);
    if (tgt == 2) return launch_mf_w<KC, NBW, 2, 0>(c, a, nfits);
    if (rpad == 0) return launch_mf_w<KC, NBW, 1, 0>(c, a, nfits);
    if (rpad == 8) return launch_mf_w<KC, NBW, 1, 8>(c, a, nfits);
    return launch_mf_w<KC, NBW, 1, 16>(c, a, nfits);
}

template <int KC>
static int32_t launch_mf_k(pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad) {
    const int nblk = (a.d + 15) / 16;
    const int nbw = (nblk + MF_WAVES - 1) / MF_WAVES;
    if (nbw <= 1) return launch_mf_t<KC, 1>(c, a, nfits, tgt, rpad);
    if (nbw <= 2) return launch_mf_t<KC, 2>(c, a, nfits, tgt, rpad);
    if (nbw <= 4) return launch_mf_t<KC, 4>(c, a, nfits, tgt, rpad);
    return launch_mf_t<KC, 8>(c, a, nfits, tgt, rpad);
}

int32_t pf_launch_elbo_mfma(pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad, bool *handled) {
    *handled = false;
    const int kc = c->kpad;
    if (a.u != nullptr) return PFMI_OK;                       // parity mode -> lane kernel
    if (!(kc == 4 || kc == 8 || kc == 12 || kc == 16)) return PFMI_OK;
    if (a.d > 1024) return PFMI_OK;
    if (mf_lds_bytes(a.d, kc, tgt == 1 ? rpad : 0) > 160 * 1024) return PFMI_OK;
    *handled = true;
    switch (kc) {
        case 4: return launch_mf_k<4>(c, a, nfits, tgt, rpad);
        case 8: return launch_mf_k<8>(c, a, nfits, tgt, rpad);
        case 12: return launch_mf_k<12>(c, a, nfits, tgt, rpad);
        default: return launch_mf_k<16>(c, a, nfits, tgt, rpad);
    }
}
