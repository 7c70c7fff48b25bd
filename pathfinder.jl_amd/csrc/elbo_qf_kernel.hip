// elbo_qf_kernel.hip -- single-pass ELBO scan: rand_and_logpdf + target of every draw of a fit (reference
// src/mvnormal.jl:24-39, src/elbo.jl:12-16) WITHOUT ever forming the draw x.
//
// For the built-in targets logp(x) is a quadratic form in x (Gaussian family) or needs only x_1 and sum x_i^2 (funnel),
// and x = mu + U'(z - Vh tv), tv = T Vh'z is affine in the transformed normals z (z = u except z_head = V'u_head,
// src/woodbury.jl:136-143).  Expanding, with s = sqrt(alpha), c = mu - m, a = target diagonal precision:
//   sum_i a_i e_i^2 = C0 + sum a s^2 z^2 + 2 sum a c s z  +  tv'M tv - 2 tv'(v + A3),      A3 = Vh'(a s^2 . z)
//   Wd'e            = t0 + A4 - Nn tv,                                                        A4 = Wd'(s . z)
// with per-fit constants C0 = sum a c^2, M = Vh' diag(a s^2) Vh, v = Vh'(a c s), t0 = Wd'c, Nn = Wd' diag(s) Vh
// (pf_qf_prep_kernel).  One pass over the rows therefore accumulates, per draw, |u|^2, two scalars and the three skinny
// contractions w = Vh'z, A3, A4 -- all with the SAME A operand tiles -- and the draw is finished with O(KC^2) flops.
// (Verified against the direct evaluation in extended precision: same 1e-14 relative error, no cancellation, because
// every term is a sum of squares or a projection of one.)
//
// Consequences for the hardware mapping:
//   * no second pass => the normals never have to be kept: a wave owns 16 draws end to end (lane (q, c): rows
//     16 blk + 4q + {0..3} of draw c = one Philox4x32 call per block), the 8 waves of a workgroup own 8 different
//     16-draw groups, and there is NO cross-wave reduction and no barrier in the steady state;
//   * the factor is only read: Vh is staged through LDS in 256-row chunks (double buffered) for any d, or kept fully
//     resident when it fits (d <= ~1100 at J = 6), in the exact lane order of the v_mfma_f64_4x4x4 A operand so that every
//     fetch is a conflict-free broadcast read;
//   * per 16-row block a wave issues 4 k-steps x (2 KC/4 + RPAD/4) quarter-size MFMAs (16 cycles each) -- the same
//     matrix work as the two-pass kernel -- plus the RNG.
#include "pfmi_common.h"
#include "pfmi_fastmath.h"
#include "elbo_args.h"

#ifndef QF_WAVES
#define QF_WAVES 8                     // waves per workgroup = 16-draw groups in flight per fit
#endif
#define QF_THREADS (QF_WAVES * 64)
#ifndef QF_PF2
#define QF_PF2 0                       // 1: operands fetched one block ahead into a second register set
#endif
#define QF_CHB 16                      // blocks (of 16 rows) per streamed chunk

typedef double qf_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ qf_d4 qf_mfma16(double a, double b, qf_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double qf_mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// per-fit constant block (doubles): [0] C0  [1] mu_0  [2] s_0  [3] -   then v[KC], M[KC][KC], t0[RPAD], Nn[RPAD][KC], vh0[KC]
__host__ __device__ constexpr int qf_nconst(int KC, int RPAD) { return 4 + KC + KC * KC + RPAD + RPAD * KC + KC; }

// a_i (target diagonal weight), c_i = mu_i - m_i for row i
template <int TGT>
__device__ __forceinline__ void qf_row_ac(const ElboArgs &A, const double *mu, int i, double &a, double &c) {
    if (TGT == 1) { a = A.t_a[i]; c = mu[i] - A.t_mean[i]; }
    else if (TGT == 2) { a = (i >= 1) ? 1.0 : 0.0; c = mu[i]; }
    else { a = 0.0; c = 0.0; }
}

// ---------------------------------------------------------------------------------------------------
// per-fit constants.  One 256-thread workgroup per fit.  Every output strip of 4 adjacent entries is the same "job"
//   acc[j] += scal_i * row_i[el] * row_i[vec + j]   over the rows i,   row_i = [ Vh_i (KC) | Wd_i (RPAD) | 1 0 0 0 ]
// (M strips (a, 4Tb..4Tb+3), 4Tb+3 >= a: scal = a s^2, el = a;  Nn strips: scal = s, el = KC + jr;  v strips: scal = a c s,
// el = the 1;  t0 strips: scal = c, el = the 1, vec in the Wd part;  C0: scal = a c^2, el = vec = the 1) so all lanes run
// one divergence-free loop.  Lane l of every wave owns jobs l, l + 64, ...; the 4 waves split the rows and their partial
// strips are summed through LDS at the end.
template <int KC, int TGT, int RPAD>
__global__ __launch_bounds__(256) void pf_qf_prep_kernel(ElboArgs A, double *__restrict__ qfc) {
    constexpr int NT = KC / 4, TR = RPAD / 4;
    constexpr int NMS = 4 * (NT * (NT + 1) / 2);                   // M strips with 4 Tb + 3 >= a
    constexpr int NJ = NMS + RPAD * NT + NT + TR + 1;
    constexpr int JPL = (NJ + 63) / 64;
    constexpr int NC = qf_nconst(KC, RPAD);
    constexpr int CR = 128, RS = KC + RPAD + 4, ONE = KC + RPAD;
    __shared__ double row_s[CR * RS], sc_s[5 * CR];                 // sc: a s^2 | a c s | a c^2 | s | c
    __shared__ double red_s[4 * 64 * JPL * 4];
    const int slot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, d = A.d;
    const int p = A.points[slot];
    double *out = qfc + (size_t)p * NC;
    if (A.status[p] != PFMI_FIT_OK) return;
    const double *Vh = A.vh + (size_t)p * d * KC, *mu = A.mu + (size_t)p * d, *sqa = A.sqrt_alpha + (size_t)p * d;
    int j_sc[JPL], j_el[JPL], j_vec[JPL], j_kind[JPL], j_a[JPL], j_t[JPL];   // kind 0 M, 1 Nn, 2 v, 3 t0, 4 C0, -1 none
    double acc[JPL][4];
#pragma unroll
    for (int e = 0; e < JPL; ++e) {
        int idx = lane + e * 64;
        j_kind[e] = -1; j_sc[e] = 0; j_el[e] = ONE + 1; j_vec[e] = 0; j_a[e] = 0; j_t[e] = 0;     // el = a zero -> no-op job
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[e][j] = 0.0;
        if (idx < NMS) {
            int g = 0;                                             // a in group g = a / 4 has NT - g strips
            while (idx >= 4 * (NT - g)) { idx -= 4 * (NT - g); ++g; }
            const int a = 4 * g + idx / (NT - g), tb = g + idx % (NT - g);
            j_kind[e] = 0; j_sc[e] = 0; j_el[e] = a; j_vec[e] = 4 * tb; j_a[e] = a; j_t[e] = tb;
        } else if (idx < NMS + RPAD * NT) {
            const int k = idx - NMS;
            j_kind[e] = 1; j_sc[e] = 3 * CR; j_el[e] = KC + k / NT; j_vec[e] = 4 * (k % NT); j_a[e] = k / NT; j_t[e] = k % NT;
        } else if (idx < NMS + RPAD * NT + NT) {
            const int k = idx - NMS - RPAD * NT;
            j_kind[e] = 2; j_sc[e] = CR; j_el[e] = ONE; j_vec[e] = 4 * k; j_t[e] = k;
        } else if (idx < NMS + RPAD * NT + NT + TR) {
            const int k = idx - NMS - RPAD * NT - NT;
            j_kind[e] = 3; j_sc[e] = 4 * CR; j_el[e] = ONE; j_vec[e] = KC + 4 * k; j_t[e] = k;
        } else if (idx < NJ) { j_kind[e] = 4; j_sc[e] = 2 * CR; j_el[e] = ONE; j_vec[e] = ONE; }
    }
    for (int r0 = 0; r0 < d; r0 += CR) {
        const int nr = (d - r0 < CR) ? d - r0 : CR;
        __syncthreads();
        for (int i = tid; i < CR * KC; i += 256) { const int r = i / KC, cc = i - r * KC; row_s[r * RS + cc] = (r < nr) ? Vh[(size_t)r0 * KC + i] : 0.0; }
        if (RPAD > 0)
            for (int i = tid; i < CR * RPAD; i += 256) { const int r = i / RPAD, cc = i - r * RPAD; row_s[r * RS + KC + cc] = (r < nr) ? A.t_wd[(size_t)r0 * RPAD + i] : 0.0; }
        if (tid < CR) {
            double a = 0.0, c = 0.0, s = 0.0;
            if (tid < nr) { qf_row_ac<TGT>(A, mu, r0 + tid, a, c); s = sqa[r0 + tid]; }
            sc_s[tid] = a * s * s; sc_s[CR + tid] = a * c * s; sc_s[2 * CR + tid] = a * c * c; sc_s[3 * CR + tid] = s; sc_s[4 * CR + tid] = c;
            row_s[tid * RS + ONE] = 1.0; row_s[tid * RS + ONE + 1] = 0.0; row_s[tid * RS + ONE + 2] = 0.0; row_s[tid * RS + ONE + 3] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < JPL; ++e) {
            const double *scp = sc_s + j_sc[e], *elp = row_s + j_el[e], *vcp = row_s + j_vec[e];
#pragma unroll 8
            for (int i = wv; i < CR; i += 4) {
                const double coef = scp[i] * elp[i * RS];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[e][j] = fma(coef, vcp[i * RS + j], acc[e][j]);
            }
        }
    }
    // sum the 4 row-partitions and scatter to the constant block
#pragma unroll
    for (int e = 0; e < JPL; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) red_s[((wv * JPL + e) * 64 + lane) * 4 + j] = acc[e][j];
    __syncthreads();
    if (wv == 0) {
        double *vv = out + 4, *Mm = vv + KC, *t0 = Mm + KC * KC, *Nn = t0 + RPAD;
#pragma unroll
        for (int e = 0; e < JPL; ++e) {
            double r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) r[j] += red_s[((w * JPL + e) * 64 + lane) * 4 + j];
            }
            const int kind = j_kind[e], a = j_a[e], t = j_t[e];
            if (kind == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int b = 4 * t + j; if (b >= a) { Mm[a * KC + b] = r[j]; Mm[b * KC + a] = r[j]; } }
            } else if (kind == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Nn[a * KC + 4 * t + j] = r[j];
            } else if (kind == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) vv[4 * t + j] = r[j];
            } else if (kind == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) t0[4 * t + j] = r[j];
            } else if (kind == 4) out[0] = r[0];
        }
    }
    if (tid < KC) out[4 + KC + KC * KC + RPAD + RPAD * KC + tid] = Vh[tid];     // row 0 of Vh
    if (tid == 0) { out[1] = mu[0]; out[2] = sqa[0]; out[3] = 0.0; }
}

// ---------------------------------------------------------------------------------------------------
// LDS layout of one staged chunk (nb blocks of 16 rows):
//   vh:  [(bl*4 + r)*NT + T][q*4 + i]  = Vh[row 16 bl + 4 q + r][4 T + i]      (16 contiguous doubles = one A operand)
//   rs:  [bl][arr][4 q + r], arr = 0: a s^2, 1: 2 a c s, 2: s
template <int KC>
__device__ __forceinline__ int qf_vh_pos(int lrow, int col) {
    const int bl = lrow >> 4, rr = lrow & 15, q = rr >> 2, r = rr & 3, T = col >> 2, i = col & 3;
    return (((bl * 4 + r) * (KC / 4) + T) << 4) + q * 4 + i;
}

template <int KC, int TGT, int RPAD>
__global__ __launch_bounds__(QF_THREADS) void pf_elbo_qf_kernel(ElboArgs A, const double *__restrict__ qfc, int ch_blocks, int nchunks,
                                                                int batches_per_wg, int nbatches, int ngroups) {
    extern __shared__ double lds[];
    constexpr int NT = KC / 4, TR = RPAD / 4, NC = qf_nconst(KC, RPAD);
    constexpr int PRE = (QF_CHB * 16 * KC + QF_THREADS - 1) / QF_THREADS;      // prefetch registers per thread (streaming)
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15, l3 = lane & 3;
    const int d = A.d, nblk = (d + 15) >> 4;
    const int slot = blockIdx.y;
    const int p = A.points[slot];
    const size_t blkidx = A.by_point ? (size_t)p : (size_t)slot;
    double *out_lp = A.logp + blkidx * A.log_stride, *out_lq = A.logq + blkidx * A.log_stride;
    const int b_begin = blockIdx.x * batches_per_wg;
    const int b_end = (b_begin + batches_per_wg < nbatches) ? b_begin + batches_per_wg : nbatches;
    if (A.status[p] != PFMI_FIT_OK) {
        for (int64_t n = (int64_t)b_begin * QF_WAVES * 16 + tid; n < (int64_t)b_end * QF_WAVES * 16 && n < A.N; n += QF_THREADS) {
            out_lp[n] = NAN; out_lq[n] = NAN;
        }
        return;
    }
    // ---- LDS carve-up (offsets in doubles from `lds`; plain offsets keep every access a ds_ instruction)
    const int vh_sz = ch_blocks * 16 * KC, rs_sz = ch_blocks * 48;
    const int buf_stride = (nchunks > 1) ? vh_sz + rs_sz : 0;       // second staging buffer only when streaming
    const int fix_off = (nchunks > 1 ? 2 : 1) * (vh_sz + rs_sz);
    double *t_s = lds + fix_off;                   // [KC][KC]
    double *cn_s = t_s + KC * KC;                  // [NC]
    double *g_s = cn_s + NC;                       // [RPAD][RPAD]
    double2 *logtab = reinterpret_cast<double2 *>(g_s + RPAD * RPAD + ((KC * KC + NC + RPAD * RPAD) & 1));
    double2 *sctab = logtab + 128;

    const double *Vh = A.vh + (size_t)p * d * KC, *mu = A.mu + (size_t)p * d, *sqa = A.sqrt_alpha + (size_t)p * d;
    auto stage_direct = [&](int ck, int buf) {
        const int row0 = ck * ch_blocks * 16;
        double *vs = lds + buf * buf_stride, *rs = vs + vh_sz;
        for (int idx = tid; idx < ch_blocks * 16 * KC; idx += QF_THREADS) {
            const int lrow = idx / KC, col = idx - lrow * KC, row = row0 + lrow;
            vs[qf_vh_pos<KC>(lrow, col)] = (row < d) ? Vh[(size_t)row * KC + col] : 0.0;
        }
        for (int lrow = tid; lrow < ch_blocks * 16; lrow += QF_THREADS) {
            const int row = row0 + lrow;
            double a = 0.0, cc = 0.0, s = 0.0;
            if (row < d) { qf_row_ac<TGT>(A, mu, row, a, cc); s = sqa[row]; }
            double *o = rs + (lrow >> 4) * 48 + (lrow & 15);
            o[0] = a * s * s; o[16] = 2.0 * a * cc * s; o[32] = s;
        }
    };
    {
        const double *T = A.tmat + (size_t)p * KC * KC;
        for (int i = tid; i < KC * KC; i += QF_THREADS) t_s[i] = T[i];
        for (int i = tid; i < NC; i += QF_THREADS) cn_s[i] = qfc[(size_t)p * NC + i];
        if (RPAD > 0) for (int i = tid; i < RPAD * RPAD; i += QF_THREADS) g_s[i] = A.t_g[i];
        pf_logtab_load(logtab);
        pf_sctab_load(sctab);
        stage_direct(0, 0);
    }
    // head transform z_head = V'u_head as 16x16x4 MFMAs (same operand trick as the two-pass kernel): lane (q, c) supplies
    // A_r[i' = c][k = q] = H[rho(c)][4 q + r], H = V' identity padded; block 1 needs H10, H11 when KC > 16
    const int rho = 4 * (c & 3) + (c >> 2);
    double a_h00[4], a_h10[4], a_h11[4];
    {
        const double *Vc = A.vchol + (size_t)p * KC * KC;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = 4 * q + r;
            double v = (rho == b) ? 1.0 : 0.0;
            if (rho < KC && b < KC) v = Vc[b * KC + rho];
            a_h00[r] = v;
            if (KC > 16) {
                const int i1 = 16 + rho, b1 = 16 + b;
                a_h10[r] = (i1 < KC) ? Vc[b * KC + i1] : 0.0;
                double v1 = (i1 == b1) ? 1.0 : 0.0;
                if (i1 < KC && b1 < KC) v1 = Vc[b1 * KC + i1];
                a_h11[r] = v1;
            } else { a_h10[r] = 0.0; a_h11[r] = 0.0; }
        }
    }
    const uint64_t seed = A.seeds[slot];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const double logdet = A.logdet[p];
    __syncthreads();

    int cur = 0;
    for (int batch = b_begin; batch < b_end; ++batch) {
        const int grp = batch * QF_WAVES + wv;
        const bool active = grp < ngroups;                          // wave-uniform
        const int64_t nl = (int64_t)grp * 16 + c;
        const uint32_t n = (uint32_t)(A.n0 + nl);
        double accw[NT], acc3[NT], acc4[TR > 0 ? TR : 1];
#pragma unroll
        for (int T = 0; T < NT; ++T) { accw[T] = 0.0; acc3[T] = 0.0; }
#pragma unroll
        for (int T = 0; T < (TR > 0 ? TR : 1); ++T) acc4[T] = 0.0;
        double usq = 0.0, q12 = 0.0, z00 = 0.0;
        double u0[4] = {0.0, 0.0, 0.0, 0.0};
        for (int ck = 0; ck < nchunks; ++ck) {
            // ---- streaming: fetch the next chunk (or chunk 0 for the next batch) into registers while this one is consumed
            double pre[PRE], pr_s = 0.0, pr_a = 0.0, pr_c = 0.0;
            const int nck = (ck + 1 < nchunks) ? ck + 1 : 0;
            const bool do_pre = (nchunks > 1) && (ck + 1 < nchunks || batch + 1 < b_end);
            if (do_pre) {
                const int row0 = nck * QF_CHB * 16;
#pragma unroll
                for (int e = 0; e < PRE; ++e) {
                    const int idx = tid + e * QF_THREADS;
                    const int lrow = idx / KC, row = row0 + lrow;
                    pre[e] = (idx < QF_CHB * 16 * KC && row < d) ? Vh[(size_t)row0 * KC + idx] : 0.0;
                }
                if (tid < QF_CHB * 16) {
                    const int row = row0 + tid;
                    if (row < d) { qf_row_ac<TGT>(A, mu, row, pr_a, pr_c); pr_s = sqa[row]; }
                }
            }
            if (active) {
                const double *vs = lds + cur * buf_stride, *rs = vs + vh_sz;
                const int blk0 = ck * ch_blocks;
                const int nb = (nblk - blk0 < ch_blocks) ? nblk - blk0 : ch_blocks;
                // operands of one block: A tiles of Vh (LDS), A tiles of Wd (L2), row scalars (LDS).  They are fetched one block
                // AHEAD into a second register set (the loop is unrolled by two so that the sets swap roles without moves):
                // the fetch latency hides behind the RNG + MFMA work of the current block.
                struct Ops { double av[4][NT]; double wd[4][TR > 0 ? TR : 1]; double rs[12]; };
                auto load_ops = [&](const int bl, Ops &o) {
                    const double *rp = rs + bl * 48 + 4 * q;
                    const double *ap = vs + ((bl * 4) * NT << 4) + q * 4 + l3;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int T = 0; T < NT; ++T) o.av[r][T] = ap[(r * NT + T) << 4];
                        o.rs[r] = rp[r]; o.rs[4 + r] = rp[16 + r]; o.rs[8 + r] = rp[32 + r];
                    }
                    if (TGT == 1 && RPAD > 0) {
                        const double *wp = A.t_wd16 + ((size_t)(blk0 + bl) * 16 + 4 * q) * 16 + l3;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int T = 0; T < TR; ++T) o.wd[r][T] = wp[r * 16 + 4 * T];
                    }
                };
                auto compute = [&](const int blk, const Ops &o) {
                    // ---- normals of rows 16 blk + 4q + {0..3} of draw n
                    uint32_t x[4];
                    pf_philox4x32_10(n, (uint32_t)(blk * 4 + q), 0u, 0u, k0, k1, x);
                    PfPair p1, p2;
                    p1.s0(x[0], x[1]); p2.s0(x[2], x[3]);
                    p1.s1(logtab, sctab); p2.s1(logtab, sctab);
                    p1.s2(); p2.s2(); p1.s3(); p2.s3(); p1.s4(); p2.s4(); p1.s5(); p2.s5();
                    double z[4];
                    p1.s6(z[0], z[1]); p2.s6(z[2], z[3]);
                    if (blk == nblk - 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) z[r] = (blk * 16 + 4 * q + r < d) ? z[r] : 0.0;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) usq = fma(z[r], z[r], usq);            // |u|^2 before the transform (src/mvnormal.jl:31)
                    if (blk == 0) {                                                    // z[1:k] = V'u[1:k] (src/woodbury.jl:139)
                        qf_d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; ++r) { u0[r] = z[r]; h = qf_mfma16(a_h00[r], z[r], h); }
#pragma unroll
                        for (int r = 0; r < 4; ++r) z[r] = h[r];
                        z00 = z[0];
                    } else if (KC > 16 && blk == 1) {
                        qf_d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; ++r) h = qf_mfma16(a_h10[r], u0[r], h);
#pragma unroll
                        for (int r = 0; r < 4; ++r) h = qf_mfma16(a_h11[r], z[r], h);
#pragma unroll
                        for (int r = 0; r < 4; ++r) z[r] = h[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double zr = z[r];
                        double bp = 0.0;
                        if (TGT != 0) {
                            bp = o.rs[r] * zr;
                            q12 = fma(bp + o.rs[4 + r], zr, q12);
                        }
#pragma unroll
                        for (int T = 0; T < NT; ++T) {
                            accw[T] = qf_mfma4(o.av[r][T], zr, accw[T]);
                            if (TGT != 0) acc3[T] = qf_mfma4(o.av[r][T], bp, acc3[T]);
                        }
                        if (TGT == 1 && RPAD > 0) {
                            const double bs = o.rs[8 + r] * zr;
#pragma unroll
                            for (int T = 0; T < TR; ++T) acc4[T] = qf_mfma4(o.wd[r][T], bs, acc4[T]);
                        }
                    }
                };
#if QF_PF2
                Ops oa, ob;
                load_ops(0, oa);
                for (int bl = 0; bl < nb; bl += 2) {
                    if (bl + 1 < nb) load_ops(bl + 1, ob);
                    compute(blk0 + bl, oa);
                    if (bl + 1 < nb) {
                        if (bl + 2 < nb) load_ops(bl + 2, oa);
                        compute(blk0 + bl + 1, ob);
                    }
                }
#else
                for (int bl = 0; bl < nb; ++bl) {
                    Ops oa;
                    load_ops(bl, oa);                          // issued first; the RNG below hides the LDS / L2 latency
                    __builtin_amdgcn_sched_barrier(0);
                    compute(blk0 + bl, oa);
                }
#endif
            }
            if (nchunks > 1) {
                if (do_pre) {
                    double *vs = lds + (cur ^ 1) * buf_stride, *rs = vs + vh_sz;
#pragma unroll
                    for (int e = 0; e < PRE; ++e) {
                        const int idx = tid + e * QF_THREADS;
                        if (idx < QF_CHB * 16 * KC) { const int lrow = idx / KC; vs[qf_vh_pos<KC>(lrow, idx - lrow * KC)] = pre[e]; }
                    }
                    if (tid < QF_CHB * 16) {
                        double *o = rs + (tid >> 4) * 48 + (tid & 15);
                        o[0] = pr_a * pr_s * pr_s; o[16] = 2.0 * pr_a * pr_c * pr_s; o[32] = pr_s;
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
        }
        if (!active) continue;
        // ---- finish the 16 draws of this wave in registers: lane (q, c) holds entries 4T + q of w, A3, A4 of draw c
        usq += __shfl_xor(usq, 16, 64); usq += __shfl_xor(usq, 32, 64);
        double lp = NAN;
        if (TGT != 0) {
            q12 += __shfl_xor(q12, 16, 64); q12 += __shfl_xor(q12, 32, 64);
            double tv[KC];
#pragma unroll
            for (int a = 0; a < KC; ++a) {
                double s = 0.0;
#pragma unroll
                for (int T = 0; T < NT; ++T) s = fma(t_s[a * KC + 4 * T + q], accw[T], s);
                s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                tv[a] = s;
            }
            const double *vv = cn_s + 4, *Mm = cn_s + 4 + KC, *t0 = Mm + KC * KC, *Nn = t0 + RPAD, *vh0 = Nn + RPAD * KC;
            double qa = 0.0;
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                const int a = 4 * T + q;
                double mt = 0.0, tva = 0.0;
#pragma unroll
                for (int b = 0; b < KC; ++b) mt = fma(Mm[a * KC + b], tv[b], mt);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) tva = (q == qq) ? tv[4 * T + qq] : tva;
                qa = fma(tva, mt - 2.0 * (vv[a] + acc3[T]), qa);
            }
            qa += __shfl_xor(qa, 16, 64); qa += __shfl_xor(qa, 32, 64);
            const double q1 = cn_s[0] + q12 + qa;
            if (TGT == 1) {
                double corr = 0.0;
                if (RPAD > 0) {
                    double tt[TR > 0 ? TR : 1];
#pragma unroll
                    for (int T = 0; T < TR; ++T) {
                        const int j = 4 * T + q;
                        double s = t0[j] + acc4[T];
#pragma unroll
                        for (int b = 0; b < KC; ++b) s = fma(-Nn[j * KC + b], tv[b], s);
                        tt[T] = s;
                    }
                    double tall[RPAD > 0 ? RPAD : 1];
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) tall[j] = __shfl(tt[j >> 2], (j & 3) * 16 + c, 64);
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) {
                        double g = 0.0;
#pragma unroll
                        for (int l = 0; l <= j; ++l) g = fma(g_s[j * RPAD + l], tall[l], g);
                        corr = fma(g, g, corr);
                    }
                }
                lp = A.t_offset - 0.5 * (q1 - corr);
            } else {                                                   // funnel: tau = x_1, ss = sum_{i>=2} x_i^2
                const double zh = __shfl(z00, c, 64);
                double pr = 0.0;
#pragma unroll
                for (int b = 0; b < KC; ++b) pr = fma(vh0[b], tv[b], pr);
                const double ta = cn_s[1] + cn_s[2] * (zh - pr), t3 = ta / 3.0;
                lp = (t3 * t3 + (double)(d - 1) * ta + q1 * exp(-ta)) / -2.0;
            }
        }
        if (q == 0 && nl < A.N) {
            out_lq[nl] = ((double)d * PF_LOG2PI + logdet + usq) / -2.0;        // src/mvnormal.jl:36
            out_lp[nl] = lp;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
static size_t qf_lds_bytes(int ch_blocks, int nchunks, int kc, int rpad) {
    const size_t per = (size_t)ch_blocks * 16 * kc + (size_t)ch_blocks * 48;
    return sizeof(double) * (per * (nchunks > 1 ? 2 : 1) + (size_t)kc * kc + qf_nconst(kc, rpad) + (size_t)rpad * rpad + 1 + 256 + 512);
}

template <int KC, int TGT, int RPAD>
static int32_t launch_qf(pfmi_ctx *c, const ElboArgs &a, int64_t nfits) {
    const int nblk = (a.d + 15) / 16;
    int ch_blocks = nblk, nchunks = 1;
    if (qf_lds_bytes(nblk, 1, KC, RPAD) > 156 * 1024) { ch_blocks = QF_CHB; nchunks = (nblk + QF_CHB - 1) / QF_CHB; }
    const size_t lds_bytes = qf_lds_bytes(ch_blocks, nchunks, KC, RPAD);
    PF_CHECK(lds_bytes <= 160 * 1024, PFMI_ERR_UNSUPPORTED, "qf kernel LDS %zu too large", lds_bytes);
    PF_TRY(c->qfc.ensure(sizeof(double) * (size_t)c->P * qf_nconst(KC, RPAD)));
    auto kern = pf_elbo_qf_kernel<KC, TGT, RPAD>;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int ngroups = (int)((a.N + 15) / 16);
    const int nbatches = (ngroups + QF_WAVES - 1) / QF_WAVES;
    int split = 1;                              // split a fit's batches over several workgroups only when there are few fits
    while ((int64_t)split * nfits < 1024 && split * 2 <= nbatches) split *= 2;
    const int bpw = (nbatches + split - 1) / split;
    const int gx = (nbatches + bpw - 1) / bpw;
    for (int64_t s0 = 0; s0 < nfits; s0 += 32768) {
        const int64_t ns = (nfits - s0 < 32768) ? (nfits - s0) : 32768;
        ElboArgs b = a;
        b.points = a.points + s0; b.seeds = a.seeds + s0;
        if (!a.by_point) { b.logp += s0 * a.log_stride; b.logq += s0 * a.log_stride; }
        if (TGT != 0)
            hipLaunchKernelGGL((pf_qf_prep_kernel<KC, TGT, RPAD>), dim3((unsigned)ns), dim3(256), 0, c->stream, b, c->qfc.as<double>());
        hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)ns), dim3(QF_THREADS), lds_bytes, c->stream, b,
                           (const double *)c->qfc.as<double>(), ch_blocks, nchunks, bpw, nbatches, ngroups);
    }
    return PFMI_OK;
}

template <int KC>
static int32_t launch_qf_t(pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad) {
    if (tgt == 0) return launch_qf<KC, 0, 0>(c, a, nfits);
    if (tgt == 2) return launch_qf<KC, 2, 0>(c, a, nfits);
    if (rpad == 0) return launch_qf<KC, 1, 0>(c, a, nfits);
    if (rpad == 8) return launch_qf<KC, 1, 8>(c, a, nfits);
    return launch_qf<KC, 1, 16>(c, a, nfits);
}

// single-pass scan: in-kernel RNG, no draws written, any d; kpad in {4, 8, 12, 16, 20, 32}
int32_t pf_launch_elbo_qf(pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad, bool *handled) {
    *handled = false;
    if (a.u != nullptr || a.x != nullptr) return PFMI_OK;
    *handled = true;
    switch (c->kpad) {
        case 4: return launch_qf_t<4>(c, a, nfits, tgt, rpad);
        case 8: return launch_qf_t<8>(c, a, nfits, tgt, rpad);
        case 12: return launch_qf_t<12>(c, a, nfits, tgt, rpad);
        case 16: return launch_qf_t<16>(c, a, nfits, tgt, rpad);
        case 20: return launch_qf_t<20>(c, a, nfits, tgt, rpad);
        case 32: return launch_qf_t<32>(c, a, nfits, tgt, rpad);
        default: *handled = false; return PFMI_OK;
    }
}
