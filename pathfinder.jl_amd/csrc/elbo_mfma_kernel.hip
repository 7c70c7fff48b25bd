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
// The VALU only generates the normals (Philox + Box-Muller) and does the elementwise epilogue; MFMA is used
// only for these true contractions over the d x 2m history block (BASELINE.json north_star).  Nothing but the
// per-draw logp/logq (and the draws of a winning fit, when asked) is written to HBM.
//
// Results are identical (to fp64 roundoff) to the lane-per-draw kernel in elbo_kernels.hip, which remains the
// general path (d > 1024, history_length > 8, parity mode with host-supplied normals).
#include "pfmi_common.h"
#include "pfmi_fastmath.h"
#include "elbo_args.h"

#define MF_THREADS 512
#define MF_WAVES 8

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 pf_mfma(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// TGT: 0 none, 1 Gaussian family (RPAD = 0 or 8 or 16 low-rank columns), 2 funnel
template <int KC, int NBW, int TGT, int RPAD>
__global__ __launch_bounds__(MF_THREADS) void pf_elbo_mfma_kernel(ElboArgs A, int groups_per_block, int ngroups) {
    extern __shared__ double lds[];
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
    double *ta_s = tm_s + rows;                // [rows] target diag prec  (TGT == 1)
    double *t_s = ta_s + rows;                 // [KC][KC] compact-WY T
    double *red = t_s + KC * KC;               // [8][256] per-wave partial tiles
    double *wsum = red + MF_WAVES * 256;       // [256]
    double *red2 = wsum + 256;                 // [8][16][4] per-wave per-draw scalars
    double2 *logtab = reinterpret_cast<double2 *>(red2 + MF_WAVES * 64);   // [128]
    double *zero_s = red2 + MF_WAVES * 64 + 256;                           // [2] zeros (masked A-operand lanes)

    {
        const double *Vh = A.vh + (size_t)p * d * KC;
        for (int i = tid; i < rows * KC; i += MF_THREADS) vh_s[i] = (i < d * KC) ? Vh[i] : 0.0;
        const double *mu = A.mu + (size_t)p * d, *sqa = A.sqrt_alpha + (size_t)p * d;
        for (int i = tid; i < rows; i += MF_THREADS) {
            mu_s[i] = (i < d) ? mu[i] : 0.0;
            sqa_s[i] = (i < d) ? sqa[i] : 0.0;
            if (TGT == 1) {
                tm_s[i] = (i < d) ? A.t_mean[i] : 0.0;
                ta_s[i] = (i < d) ? A.t_a[i] : 0.0;
            }
        }
        const double *T = A.tmat + (size_t)p * KC * KC;
        for (int i = tid; i < KC * KC; i += MF_THREADS) t_s[i] = T[i];
        pf_logtab_load(logtab);
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

    for (int grp = g_begin; grp < g_end; ++grp) {
        // the wave index is made opaque once per group so that nothing derived from the block index is treated
        // as loop-invariant: otherwise LLVM hoists per-block Philox partial products and addresses out of the
        // group loop and spills them (the kernel is register-bound: z alone is 8*NBW VGPRs)
        int wvs = __builtin_amdgcn_readfirstlane(wv);
        asm volatile("" : "+s"(wvs));
        const int64_t nl = (int64_t)grp * 16 + c;             // local draw index of this lane's column
        const bool valid = nl < A.N;
        const uint32_t n = (uint32_t)(A.n0 + nl);
        double z[NBW][4];
        double usq = 0.0;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        // ---------------- pass 1: normals -> registers, W = Vh' z on the matrix cores
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int blk = wvs * NBW + b;
            if (blk < nblk) {
                double zz[4];
                pf_randn4_fast(seed, (uint32_t)(blk * 4 + q), n, 0u, logtab, zz);
                const int rowbase = blk * 16 + 4 * q;
                if (blk == nblk - 1) {                                      // rows >= d of the last block do not exist
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (rowbase + r >= d) zz[r] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) usq += zz[r] * zz[r];           // |u|^2 before the transform
                if (blk == 0) {                                             // z[1:k] = V' u[1:k]  (src/woodbury.jl:139)
                    d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) h = pf_mfma(a_head[r], zz[r], h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) zz[r] = h[r];
                }
                const double *a1p = (c < KC) ? (vh_s + rowbase * KC + c) : zero_s;     // A[i = c][k = q] = Vh[row][c]
                const int a1s = (c < KC) ? KC : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc = pf_mfma(a1p[r * a1s], zz[r], acc);
                    z[b][r] = zz[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) z[b][r] = 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one block's RNG temporaries live at a time
        }
        // ---------------- W: sum the 8 per-wave tiles, tv = T W  (lane (q,c) needs tv[4s + q][c])
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wv * 256 + reg * 64 + lane] = acc[reg];   // entry (j = q + 4 reg, c)
        __syncthreads();
        if (tid < 256) {
            double s = red[tid];
#pragma unroll
            for (int w = 1; w < MF_WAVES; ++w) s += red[w * 256 + tid];
            wsum[tid] = s;
        }
        __syncthreads();
        double ntv[KC / 4];
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const int jp = 4 * s + q;
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                // W[j][c] lives at entry (reg = j >> 2, lane = (j & 3) * 16 + c)
                v += t_s[jp * KC + j] * wsum[(j >> 2) * 64 + (j & 3) * 16 + c];
            }
            ntv[s] = -v;
        }
        // ---------------- pass 2: x = mu + sqrt(alpha) (z - Vh tv), target, optional store
        double qd = 0.0, tau = 0.0;
        d4 acc3 = {0.0, 0.0, 0.0, 0.0};
        double *X = (A.x && valid) ? (A.x + (size_t)slot * A.x_stride + (size_t)nl * d) : nullptr;
        double a3n[4] = {0.0, 0.0, 0.0, 0.0};                  // software-prefetched A operands of the target contraction
        if (TGT == 1 && RPAD > 0 && wvs * NBW < nblk) {
            const double *w16 = A.t_wd16 + ((size_t)(wvs * NBW) * 16 + 4 * q) * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) a3n[r] = w16[r * 16];
        }
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int blk = wvs * NBW + b;
            if (blk < nblk) {
                double a3[4];
                if (TGT == 1 && RPAD > 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a3[r] = a3n[r];
                    if (b + 1 < NBW && blk + 1 < nblk) {
                        const double *w16 = A.t_wd16 + ((size_t)(blk + 1) * 16 + 4 * q) * 16 + c;   // A[j' = c][k = q]
#pragma unroll
                        for (int r = 0; r < 4; ++r) a3n[r] = w16[r * 16];
                    }
                }
                d4 xa = {z[b][0], z[b][1], z[b][2], z[b][3]};
                const double *a2p = vh_s + (blk * 16 + rho) * KC + q;                    // A[i'][k = q] = Vh[16 blk + rho(i')][4s + q]
#pragma unroll
                for (int s = 0; s < KC / 4; ++s) xa = pf_mfma(a2p[4 * s], ntv[s], xa);
                const int rowbase = blk * 16 + 4 * q;
                double e4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rowbase + r;
                    const double xi = mu_s[row] + sqa_s[row] * xa[r];
                    if (TGT == 1) {
                        const double e = xi - tm_s[row];
                        qd += ta_s[row] * e * e;
                        e4[r] = e;
                    } else if (TGT == 2) {
                        if (row == 0) tau = xi; else qd += xi * xi;           // padded rows give xi = 0
                    }
                    if (X && row < d) X[row] = xi;
                }
                if (TGT == 1 && RPAD > 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc3 = pf_mfma(a3[r], e4[r], acc3);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---------------- per-draw reductions over q (lanes) and waves (LDS)
        usq += __shfl_xor(usq, 16, 64); usq += __shfl_xor(usq, 32, 64);
        qd += __shfl_xor(qd, 16, 64);   qd += __shfl_xor(qd, 32, 64);
        if (q == 0) {
            red2[(wv * 16 + c) * 4 + 0] = usq;
            red2[(wv * 16 + c) * 4 + 1] = qd;
            if (TGT == 2 && wv == 0) red2[(wv * 16 + c) * 4 + 2] = tau;
        }
        if (TGT == 1 && RPAD > 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) red[wv * 256 + reg * 64 + lane] = acc3[reg];
        }
        __syncthreads();
        if (tid < 16) {
            const int cc = tid;
            double us = 0.0, qq = 0.0;
#pragma unroll
            for (int w = 0; w < MF_WAVES; ++w) { us += red2[(w * 16 + cc) * 4 + 0]; qq += red2[(w * 16 + cc) * 4 + 1]; }
            double lp = NAN;
            if (TGT == 1) {
                double corr = 0.0;
                if (RPAD > 0) {
                    double t[RPAD > 0 ? RPAD : 1];
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) {
                        double s = 0.0;
#pragma unroll
                        for (int w = 0; w < MF_WAVES; ++w) s += red[w * 256 + (j >> 2) * 64 + (j & 3) * 16 + cc];
                        t[j] = s;
                    }
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) {
                        double g = 0.0;
#pragma unroll
                        for (int l = 0; l <= j; ++l) g += A.t_g[j * RPAD + l] * t[l];
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
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
template <int KC, int NBW, int TGT, int RPAD>
static int32_t launch_mf(pfmi_ctx *c, const ElboArgs &a, int64_t nfits) {
    const int d = a.d;
    const int nblk = (d + 15) / 16, rows = nblk * 16;
    const size_t lds_bytes = sizeof(double) * ((size_t)rows * KC + 4 * (size_t)rows + KC * KC + MF_WAVES * 256 + 256 +
                                               MF_WAVES * 64 + 256 + 2);
    PF_CHECK(lds_bytes <= 160 * 1024, PFMI_ERR_UNSUPPORTED, "mfma kernel LDS %zu too large", lds_bytes);
    auto kern = pf_elbo_mfma_kernel<KC, NBW, TGT, RPAD>;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_set = true;
    }
    const int ngroups = (int)((a.N + 15) / 16);
    // one fit per workgroup; split a fit's draw groups over several workgroups only when there are few fits
    int split = 1;
    while ((int64_t)split * nfits < 1024 && split * 2 <= ngroups) split *= 2;
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

template <int KC, int NBW>
static int32_t launch_mf_t(pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad) {
    if (tgt == 0) return launch_mf<KC, NBW, 0, 0>(c, a, nfits);
    if (tgt == 2) return launch_mf<KC, NBW, 2, 0>(c, a, nfits);
    if (rpad == 0) return launch_mf<KC, NBW, 1, 0>(c, a, nfits);
    if (rpad == 8) return launch_mf<KC, NBW, 1, 8>(c, a, nfits);
    return launch_mf<KC, NBW, 1, 16>(c, a, nfits);
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
    const int rows = ((a.d + 15) / 16) * 16;
    const size_t lds_bytes = sizeof(double) * ((size_t)rows * kc + 4 * (size_t)rows + kc * kc + MF_WAVES * 256 + 256 +
                                               MF_WAVES * 64 + 256 + 2);
    if (lds_bytes > 160 * 1024) return PFMI_OK;
    *handled = true;
    switch (kc) {
        case 4: return launch_mf_k<4>(c, a, nfits, tgt, rpad);
        case 8: return launch_mf_k<8>(c, a, nfits, tgt, rpad);
        case 12: return launch_mf_k<12>(c, a, nfits, tgt, rpad);
        default: return launch_mf_k<16>(c, a, nfits, tgt, rpad);
    }
}
