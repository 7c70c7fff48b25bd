// fit_tsqr_kernel.hip -- the Woodbury fit for large d (1024 < d <= 16384), gfx950, organised so that the d x 2J block crosses HBM ONCE
// in each direction (round 6; the panel kernel of fit_panel_kernel.hip sweeps its scratch block ~12 times: 19 MB per fit at d = 10^4,
// J = 10 against 3.6 MB of inputs + outputs).
//
// Same outputs and the same LAPACK reflector convention as pf_fit_kernel (fit_kernels.hip; reference src/inverse_hessian.jl:98-133,
// src/woodbury.jl:201-207, src/mvnormal.jl:14-21): Vh = the Householder vectors of qr(U' \ [alpha.Y  S]) with explicit unit diagonal,
// the compact-WY T (dlarft), R, D, V = chol(I + R D R'), log det, mu.  What differs is HOW the reflectors are found.  A CU cannot
// hold the block (1.6 MB at d = 10^4, J = 10, against 512 KB of registers), and a left-looking factorisation has to stream every finished column past
// every later panel twice (dot products, then the update).  Here:
//
//   1  TSQR.  The rows are cut into chunks of CH = 512 RC rows; a chunk's m = 2j columns live in REGISTERS (thread t owns chunk rows
//      t + 512 i).  Each chunk is factored by Householder QR on the spot (dgeqr2; ONE block reduction per column: the norm, the dots with the
//      later columns, the dots with the EARLIER reflectors -- the chunk's own compact-WY T -- and the transformed right-hand side U g ride
//      in the same reduction), its reflectors V_i go to a scratch block (the only intermediate that touches HBM: written once, read
//      once), its R_i (m x m) and the head of Q_i'(U g) go onto a stack.  The stack (n_chunks m rows) is factored the same way:
//      B~ = Q_in R_in with Q_in = diag(Q_i) Q_top.
//   2  Householder reconstruction (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik 2014; LAPACK dorhr_col): the reflectors LAPACK's
//      dgeqr2 would have produced for B~ are the unit-lower-trapezoidal factor of the LU decomposition WITHOUT pivoting of Q_in - S,
//      S = diag(+-1) chosen on the fly (pivots >= 1 in magnitude), once diag(R_in) >= 0:   Q_in - S = V U,   T = -U S V_1^-T,   R = S R_in.
//      Only the top m x m block needs elimination; every other row is a row-local triangular solve.  With Q_in(rows of chunk i) =
//      [W_i; 0] - V_i K_i W_i  (W_i = the chunk's m x m block of the first m columns of Q_top, K_i = T_i V_i[0:m,:]') this is, per row,
//               Vh[r, :] = V_i[r, :] N_i   (+ W_i[r', :] U^-1 for the first m rows r' of a chunk),       N_i = -K_i W_i U^-1,
//      an m x m matrix per chunk -- so the second pass over the scratch block is row-local too and writes Vh, row-major, directly.
//   3  The mean needs no sweep either: with b = Q'(U g) (its head comes out of the factorisation as the transformed extra column),
//      mu = theta + U'(U g + Q_in S (V_c'V_c - I) head): one more row-local dot product in the same pass.
//
// HBM traffic per fit at d = 10^4, J = 10: the 4 j input rows (shared with the neighbouring fits, XCD-aware order as in the panel kernel),
// 1.6 MB of reflectors out and in, 1.6 MB of Vh out, a few vectors: ~5.5 MB.  tools/tsqr_hr_check.py is the NumPy statement of the data
// flow (the exact formulas of this kernel against LAPACK's dgeqrf / dlarft: agreement to 1e-15).
#include <type_traits>
#include "pfmi_common.h"
#include <stdlib.h>
#include "fit_args.h"

#define TS_NT 512
#define TS_NW (TS_NT / 64)
#define TS_OPAQUE(x) asm volatile("" : "+v"(x))
#ifndef TS_PROF
#define TS_PROF 0                      // 1: workgroup 0 accumulates cycle counts per section and prints them (experiments only)
#endif
#if TS_PROF
#define TS_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); prof[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define TS_STAMP(k) do { } while (0)
#endif

// ---- block sum of NV values (NV % 4 == 0, NV <= 64): the multi-value butterfly of pf_block_sum_mv inside a wave, then lane l of EVERY
// wave adds the waves' partials of value l in wave order and the totals are broadcast with v_readlane -- 8 LDS reads per thread instead
// of 8 NV, and the totals are wave-uniform (scalar registers).  One barrier per call (ping-pong halves of `red`, 2 x TS_NW x NV doubles).
template <int NV>
__device__ __forceinline__ void ts_block_sum(double (&v)[NV], double *red, int &flip) {
    static_assert(NV % 4 == 0 && NV <= 64, "ts_block_sum: NV must be a multiple of 4, at most 64");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *buf = red + flip * (TS_NW * NV);
    flip ^= 1;
    double r[NV / 2], q[NV / 4];
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) r[j] = pf_swap32_add(v[j], v[j + NV / 2]);
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) q[j] = pf_swap16_add(r[j], r[j + NV / 4]);
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
        q[j] = pf_dpp_add<0x111, 0xf>(q[j]);
        q[j] = pf_dpp_add<0x112, 0xf>(q[j]);
        q[j] = pf_dpp_add<0x114, 0xf>(q[j]);
        q[j] = pf_dpp_add<0x118, 0xf>(q[j]);   // lane 15 of row rho: the wave totals of values rho NV/4 + j
    }
    if ((lane & 15) == 15) {
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) buf[wave * NV + (lane >> 4) * (NV / 4) + j] = q[j];
    }
    __syncthreads();
    double s = 0.0;
    {
        const int l = lane < NV ? lane : NV - 1;
        double t[TS_NW];
#pragma unroll
        for (int w = 0; w < TS_NW; ++w) t[w] = buf[w * NV + l];
        s = t[0];
#pragma unroll
        for (int w = 1; w < TS_NW; ++w) s += t[w];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = pf_readlane_f64(s, i);
}

// ---- Householder QR (dgeqr2 / dlarfg) of a row block held in registers: thread t owns rows t + 512 i, i < RC; column c's diagonal row
// is row c (thread c, i = 0).  U is an extra column that is transformed but not factored.  On return P holds the explicit reflectors
// (unit diagonal, zeros above); row c of R (columns c .. m-1) and the transformed U[c] went to rdst[c * rstride + ..] / hdst[c]; the
// block's compact-WY T (dlarft) sits in sT (row a written and read by thread a only while the loop runs).
// (column index as a TEMPLATE parameter: `#pragma unroll` gives up on the 20- and 32-column loops -- "unrolled size is too large" -- and a
// rolled loop would index the register block dynamically, i.e. put it into scratch memory)
template <int c, int RC, int MC>
__device__ __forceinline__ void ts_qr_col(double (&P)[RC][MC], double (&U)[RC], const int m, double *sT, double *srow2, double *red, int &flip,
                                          double *rdst, const int rstride, double *hdst) {
    constexpr int NV = MC + 4;
    const int tid = threadIdx.x;
    double *srow = srow2 + (c & 1) * NV;
    if (tid == c) {
#pragma unroll
        for (int cc = 0; cc < MC; ++cc) srow[cc] = P[0][cc];
        srow[MC] = U[0];
    }
    double dots[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) dots[v] = 0.0;
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const double xc = (i > 0 || tid > c) ? P[i][c] : 0.0;            // rows below the diagonal
#pragma unroll
        for (int v = 0; v < MC; ++v) dots[v] = fma(xc, P[i][v], dots[v]);
        dots[MC] = fma(xc, U[i], dots[MC]);
    }
    ts_block_sum<NV>(dots, red, flip);                                   // its barrier also publishes srow (double buffered)
    const double xn2 = dots[c], alpha_c = srow[c];
    const double xnorm = sqrt(xn2);
    double tau, scal, beta;
    if (xnorm == 0.0) { tau = 0.0; scal = 0.0; beta = alpha_c; }
    else {
        beta = -copysign(sqrt(fma(alpha_c, alpha_c, xn2)), alpha_c);
        tau = (beta - alpha_c) / beta;
        scal = 1.0 / (alpha_c - beta);
    }
    double vs[RC];
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const bool below = (i > 0 || tid > c);
        vs[i] = below ? P[i][c] * scal : 0.0;                            // the reflector below the diagonal (v_c = 1 on it)
        P[i][c] = below ? vs[i] : P[i][c];
    }
#pragma unroll
    for (int cc = c + 1; cc < MC; ++cc) {                                // A <- A - v (tau v'A): one column at a time, no array of coefficients
        const double w = tau * (srow[cc] + scal * dots[cc]);
#pragma unroll
        for (int i = 0; i < RC; ++i) P[i][cc] = fma(-w, vs[i], P[i][cc]);
        if (tid == c) {                                                  // row c: the R entry out, an explicit zero in
            if (cc < m) rdst[c * rstride + cc] = P[0][cc] - w;
            P[0][cc] = 0.0;
        }
    }
    {
        const double w = tau * (srow[MC] + scal * dots[MC]);             // the extra column
#pragma unroll
        for (int i = 0; i < RC; ++i) U[i] = fma(-w, vs[i], U[i]);
        if (tid == c) {
            hdst[c] = U[0] - w;
            rdst[c * rstride + c] = beta;
            P[0][c] = 1.0;                                               // explicit unit diagonal
            sT[c * MC + c] = tau;
        }
    }
    if (tid < c) {                                                       // dlarft: T[0:c, c] = -tau T[0:c, 0:c] (V[:, 0:c]' v_c)
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < c; ++b) {
            const double g = srow[b] + scal * dots[b];                   // V[c, b] . 1 + sum over the rows below
            acc += (b >= tid) ? sT[tid * MC + b] * g : 0.0;
        }
        sT[tid * MC + c] = -tau * acc;
    }
}
template <int c, int RC, int MC>
__device__ __forceinline__ void ts_qr_from(double (&P)[RC][MC], double (&U)[RC], const int m, double *sT, double *srow2, double *red, int &flip,
                                           double *rdst, const int rstride, double *hdst) {
    if constexpr (c < MC) {
        if (c < m) ts_qr_col<c, RC, MC>(P, U, m, sT, srow2, red, flip, rdst, rstride, hdst);
        ts_qr_from<c + 1, RC, MC>(P, U, m, sT, srow2, red, flip, rdst, rstride, hdst);
    }
}
template <int RC, int MC>
__device__ __forceinline__ void ts_qr_cols(double (&P)[RC][MC], double (&U)[RC], const int m, double *sT, double *srow2, double *red, int &flip,
                                           double *rdst, const int rstride, double *hdst) {
    ts_qr_from<0, RC, MC>(P, U, m, sT, srow2, red, flip, rdst, rstride, hdst);
}

static int ts_lds_doubles(int MC) {
    const int NV = MC + 4;
    return 2 * TS_NW * NV + 2 * NV + 12 * MC * MC + 12 * MC + 32;
}

template <int MC, int RC>
__global__ __launch_bounds__(TS_NT, 1) void pf_fit_tsqr_kernel(FitArgs A, const int DP, const int nch_max, double *scratch_all, int *counter) {
    constexpr int NV = MC + 4, CH = TS_NT * RC, SW = MC + 4;
    const int tid = threadIdx.x;
    const int d = A.d, J = A.J;
    // per-workgroup scratch: reflectors column-major [MC][DP] | stack / W rows [512][SW] | K_i [nch_max][MC][MC]
    const size_t wg_doubles = (size_t)MC * DP + (size_t)TS_NT * SW + (size_t)nch_max * MC * MC;
    double *scr = scratch_all + (size_t)blockIdx.x * wg_doubles;
    double *stk = scr + (size_t)MC * DP, *Kg = stk + (size_t)TS_NT * SW;

    extern __shared__ double lds[];
    double *red = lds;
    double *srow2 = red + 2 * TS_NW * NV;
    double *sT = srow2 + 2 * NV;
    double *sR = sT + MC * MC, *sD = sR + MC * MC, *sV = sD + MC * MC, *sG = sV + MC * MC;
    double *X1 = sG + MC * MC, *X2 = X1 + MC * MC, *X3 = X2 + MC * MC;
    double *sL = X3 + MC * MC, *sU = sL + MC * MC, *sM = sU + MC * MC, *sN = sM + MC * MC;     // 12 MC^2 in all
    double *sS = sN + MC * MC, *sHead = sS + MC, *sSd = sHead + MC, *sY = sSd + MC, *sWd = sY + MC, *sTmp = sWd + MC, *sDg = sTmp + MC;
    double *sWu = X1;                                            // the small algebra's scratch matrices are free again in the last pass
    __shared__ int sNext, sStatus;
    __shared__ double sLogdetV;
    int flip = 0;
#if TS_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    int nfit = 0;
#endif

    for (;;) {
        __syncthreads();
        if (tid == 0) {
            // XCD-aware work order (see fit_panel_kernel.hip): XCD x walks its own contiguous eighth of the points, then helps the next ones
            const int P_ = (int)A.P, q8 = P_ >> 3, r8 = P_ & 7;
            int next = -1;
            for (int t = 0; t < 8 && next < 0; ++t) {
                const int x = ((int)blockIdx.x + t) & 7, cnt = q8 + (x < r8 ? 1 : 0);
                if (cnt == 0) continue;
                const int l = atomicAdd(counter + x, 1);
                if (l < cnt) next = x * q8 + (x < r8 ? x : r8) + l;
            }
            sNext = next;
        }
        __syncthreads();
        if (sNext < 0) break;
        int64_t p;
        if (!pf_fit_point(A, sNext, p)) { if (tid == 0) A.status[p] = PFMI_FIT_ABSENT; continue; }
#if TS_PROF
        ++nfit;
#endif
        TS_STAMP(0);

        const int path = A.path_of[p];
        const int64_t p0 = A.off[path];
        const int j = A.hist_len[p], m = 2 * j, k = m;           // d > 1024 >= 2 J: k = min(d, m) = m
        const int nch = (d + CH - 1) / CH;
        const double *alpha = A.alpha_all + (size_t)p * d;
        double *Vh = A.vh + (size_t)p * d * MC;
        double *sqa = A.sqrt_alpha + (size_t)p * d;
        double *mu = A.mu + (size_t)p * d;
        const double *theta_p = A.theta + (size_t)p * d, *grad_p = A.grad + (size_t)p * d;
        const size_t sm = (size_t)p * MC * MC;

        for (int t = tid; t < 12 * MC * MC; t += TS_NT) sT[t] = 0.0;          // the twelve small matrices are contiguous
        for (int t = tid; t < 7 * MC; t += TS_NT) sS[t] = 0.0;
        __syncthreads();

        // ---- pass 1: chunk by chunk -- scaled rows B~ = U' \ [alpha.Y  S] (src/inverse_hessian.jl:117-118) into registers, Householder QR,
        //      reflectors -> scratch, R_i and the head of Q_i'(U g) -> stack, K_i = T_i V_i[0:m, :]' -> Kg
        double bad = 0.0, ldu = 0.0;
        for (int ci = 0; ci < nch; ++ci) {
            const int base = ci * CH;
            double P[RC][MC], U[RC];
            {
                int tb = tid;
                TS_OPAQUE(tb);
                double al[RC], isa[RC];
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    const int row = base + tb + TS_NT * i;
                    double a_ = 1.0, g_ = 0.0;
                    if (row < d) { a_ = alpha[row]; g_ = grad_p[row]; }
                    if (!(a_ > 0.0) || !isfinite(a_)) bad = 1.0;
                    const double s = sqrt(a_);
                    al[i] = a_; isa[i] = 1.0 / s;
                    if (row < d) { sqa[row] = s; ldu += log(s); }           // U = sqrt(alpha) (src/woodbury.jl:202-203)
                    U[i] = (row < d) ? s * g_ : 0.0;
                }
#pragma unroll
                for (int cc = 0; cc < MC; ++cc) {
                    if (cc < m) {                                              // column cc: alpha.y of pair cc (cc < j) or s of pair cc - j
                        const bool isy = cc < j;
                        const int src = A.hist_src[(size_t)p * J + (isy ? cc : cc - j)];
                        const double *g0 = A.grad + (size_t)(p0 + src) * d, *t0 = A.theta + (size_t)(p0 + src) * d;
                        const double *pa = isy ? g0 : t0 + d, *pb = isy ? g0 + d : t0;     // y = grad_l - grad_{l+1} :46,  s = theta_{l+1} - theta_l :45
#pragma unroll
                        for (int i = 0; i < RC; ++i) {
                            const int row = base + tb + TS_NT * i, rl = row < d ? row : d - 1;
                            const double df = pa[rl] - pb[rl];
                            const double v = isy ? (al[i] * df) * isa[i] : df * isa[i];
                            P[i][cc] = (row < d) ? v : 0.0;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < RC; ++i) P[i][cc] = 0.0;
                    }
                }
            }
            TS_STAMP(1);                                                       // inputs
            ts_qr_cols<RC, MC>(P, U, m, sT, srow2, red, flip, stk + (size_t)ci * m * SW, SW, sTmp);
            __syncthreads();                                                   // T_i complete, sTmp (the chunk's head) written
            TS_STAMP(2);                                                       // chunk QR
            if (tid < m) {
                stk[(size_t)(ci * m + tid) * SW + MC] = sTmp[tid];
                // column tid of K_i = T_i V_top':  K_i[a][tid] = sum_{b = a .. tid} T_i[a][b] V_top[tid][b]   (V_top = this thread's row 0)
                double kcol[MC];
#pragma unroll
                for (int a = 0; a < MC; ++a) kcol[a] = 0.0;
#pragma unroll
                for (int b = 0; b < MC; ++b) {
                    const double vb = P[0][b];                                 // zero above the diagonal, one on it
#pragma unroll
                    for (int a = 0; a <= b; ++a) kcol[a] = fma(sT[a * MC + b], vb, kcol[a]);
                }
#pragma unroll
                for (int a = 0; a < MC; ++a) Kg[(size_t)ci * MC * MC + a * MC + tid] = kcol[a];
            }
            {
                int td = tid;
                TS_OPAQUE(td);
#pragma unroll
                for (int cc = 0; cc < MC; ++cc) {
                    if (cc < m) {
#pragma unroll
                        for (int i = 0; i < RC; ++i) scr[(size_t)cc * DP + base + td + TS_NT * i] = P[i][cc];
                    }
                }
            }
            __syncthreads();                                                   // sT is rewritten by the next chunk
            TS_STAMP(3);                                                       // reflectors out
        }
        {
            double v[4] = {bad, ldu, 0.0, 0.0};
            ts_block_sum<4>(v, red, flip);
            bad = v[0]; ldu = v[1];
        }
        if (bad > 0.0) {                                                       // A not positive definite (src/woodbury.jl:202)
            for (int row = tid; row < d; row += TS_NT) {
                mu[row] = NAN;
                for (int c = 0; c < MC; ++c) Vh[(size_t)row * MC + c] = 0.0;
            }
            for (int t = tid; t < MC * MC; t += TS_NT) { A.tmat[sm + t] = 0.0; A.vchol[sm + t] = 0.0; A.rq[sm + t] = 0.0; A.dmat[sm + t] = 0.0; }
            if (tid == 0) { A.status[p] = PFMI_FIT_A_NOT_PD; A.logdet[p] = NAN; }
            continue;
        }

        // ---- the stack: QR of [R_0; R_1; ...] with the heads as the extra column; W = D-signed first m columns of Q_top (row-local)
        const int nst = nch * m;                                               // <= 512 by the choice of RC
        {
            double Pt[1][MC], Ut[1];
#pragma unroll
            for (int cc = 0; cc < MC; ++cc) Pt[0][cc] = 0.0;
            Ut[0] = 0.0;
            if (tid < nst) {
                const int cr = tid % (m > 0 ? m : 1);                          // row cr of its R_i: entries left of the diagonal were never written
#pragma unroll
                for (int cc = 0; cc < MC; ++cc) Pt[0][cc] = (cc >= cr && cc < m) ? stk[(size_t)tid * SW + cc] : 0.0;
                Ut[0] = stk[(size_t)tid * SW + MC];
            }
            ts_qr_cols<1, MC>(Pt, Ut, m, sT, srow2, red, flip, sR, MC, sHead);
            __syncthreads();
            if (tid < m) {                                                     // K_t = T_t V_t[0:m, :]'  -> sM (column tid), sign D of R_in's diagonal
                double kcol[MC];
#pragma unroll
                for (int a = 0; a < MC; ++a) kcol[a] = 0.0;
#pragma unroll
                for (int b = 0; b < MC; ++b) {
                    const double vb = Pt[0][b];
#pragma unroll
                    for (int a = 0; a <= b; ++a) kcol[a] = fma(sT[a * MC + b], vb, kcol[a]);
                }
#pragma unroll
                for (int a = 0; a < MC; ++a) sM[a * MC + tid] = kcol[a];
                sDg[tid] = (sR[tid * MC + tid] < 0.0) ? -1.0 : 1.0;
            }
            __syncthreads();
            if (tid < nst) {                                                   // W[r, c] = D_c (delta_rc - V_t[r, :] K_t[:, c])
                double w[MC];
#pragma unroll
                for (int c = 0; c < MC; ++c) w[c] = (tid == c) ? 1.0 : 0.0;
#pragma unroll
                for (int a = 0; a < MC; ++a) {
                    const double va = Pt[0][a];
#pragma unroll
                    for (int c = 0; c < MC; ++c) w[c] = fma(-va, sM[a * MC + c], w[c]);
                }
#pragma unroll
                for (int c = 0; c < MC; ++c) stk[(size_t)tid * SW + c] = (c < m) ? w[c] * sDg[c] : 0.0;
            }
            if (tid < m) {                                                     // R_in <- D R_in, head <- D head
                const double dg = sDg[tid];
#pragma unroll
                for (int c = 0; c < MC; ++c) sR[tid * MC + c] *= dg;
                sHead[tid] *= dg;
            }
            __syncthreads();
        }
        TS_STAMP(4);                                                           // stack QR + W
        // ---- the top m x m block of Q_in = W_0 - V_0[0:m, :] (K_0 W_0), modified LU (S chosen on the fly), T = -U S V_1^-T, R = S R_in
        for (int t = tid; t < m * m; t += TS_NT) {
            const int a = t / m, c = t % m;
            double v = 0.0;
            for (int b = 0; b < m; ++b) v += Kg[a * MC + b] * stk[(size_t)b * SW + c];
            sN[a * MC + c] = v;                                                // M_0
        }
        __syncthreads();
        for (int t = tid; t < m * m; t += TS_NT) {
            const int r = t / m, c = t % m;
            double v = stk[(size_t)r * SW + c];
            for (int a = 0; a <= r; ++a) v -= scr[(size_t)a * DP + r] * sN[a * MC + c];     // V_0[r, a]: zero for a > r
            sL[r * MC + c] = v;
        }
        __syncthreads();
        if (tid < 64) {                                                        // wave 0, lane r = row r of the block
            const int r = tid;
            double arow[MC];
#pragma unroll
            for (int c = 0; c < MC; ++c) arow[c] = (r < m && c < m) ? sL[r * MC + c] : 0.0;
            double sgn_mine = 1.0;
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                if (c < m) {
                const double pcc = pf_readlane_f64(arow[c], c);
                const double sg = (pcc >= 0.0) ? -1.0 : 1.0;                   // S_c = -sign(q_cc): |q_cc - S_c| >= 1
                const double ucc = pcc - sg;
                if (r == c) { arow[c] = ucc; sgn_mine = sg; }
                const double l = arow[c] / ucc;
                if (r > c) arow[c] = l;
#pragma unroll
                for (int cc = c + 1; cc < MC; ++cc) {
                    const double pv = pf_readlane_f64(arow[cc], c);
                    if (r > c) arow[cc] = fma(-l, pv, arow[cc]);
                }
                }
            }
            if (r < m) {
                sS[r] = sgn_mine;
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    sL[r * MC + c] = (c < r) ? arow[c] : (c == r ? 1.0 : 0.0);
                    sU[r * MC + c] = (c >= r && c < m) ? arow[c] : 0.0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (r < m) {                                                       // row r of T: T V_1' = -U S  (V_1 = L, unit lower)
                double trow[MC];
#pragma unroll
                for (int c = 0; c < MC; ++c) trow[c] = 0.0;
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    if (c < m && c >= r) {
                        double v = -arow[c] * sS[c];
#pragma unroll
                        for (int b = 0; b < c; ++b) v = fma(-trow[b], sL[c * MC + b], v);   // trow[b] = 0 for b < r
                        trow[c] = v;
                    }
                }
#pragma unroll
                for (int c = 0; c < MC; ++c) sT[r * MC + c] = trow[c];
                const double sg = sgn_mine;
#pragma unroll
                for (int c = 0; c < MC; ++c) sR[r * MC + c] *= sg;
                sHead[r] *= sg;                                                // head of Q_out'(U g)
            } else if (r < MC) {
#pragma unroll
                for (int c = 0; c < MC; ++c) sT[r * MC + c] = 0.0;
            }
        }
        __syncthreads();
        TS_STAMP(5);                                                           // reconstruction of the top block

        // ---- small algebra (as in fit_panel_kernel.hip).  G = B~'B~ = R'R:  G[c][b] (c, b < j) = Y'alpha Y,  G[j + a][b] = S'Y
        for (int t = tid; t < m * m; t += TS_NT) {
            const int a = t / m, b = t % m, u1 = a < b ? a : b;
            double v = 0.0;
            for (int u = 0; u <= u1; ++u) v += sR[u * MC + a] * sR[u * MC + b];
            sG[a * MC + b] = v;
        }
        __syncthreads();
        for (int t = tid; t < j * j; t += TS_NT) {       // D (m x m)   (src/inverse_hessian.jl:119-130)
            const int aa = t / j, b = t % j;
            X1[aa * MC + b] = (b >= aa) ? sG[(j + aa) * MC + b] : 0.0;       // R_ = triu(S'Y)   :119-121
            X2[aa * MC + b] = 0.0;
        }
        __syncthreads();
        if (tid < j) {                                   // -R_^{-1}: thread c solves column c by back substitution :122-124
            const int c = tid;
            for (int r = c; r >= 0; --r) {
                double rhs = (r == c) ? -1.0 : 0.0;
                for (int t = r + 1; t <= c; ++t) rhs -= X1[r * MC + t] * X2[t * MC + c];
                X2[r * MC + c] = rhs / X1[r * MC + r];
            }
        }
        __syncthreads();
        for (int t = tid; t < j * j; t += TS_NT) {       // M = Y'alpha Y + diag(R_); D12, D21
            const int aa = t / j, b = t % j;
            sD[aa * MC + (j + b)] = X2[aa * MC + b];
            sD[(j + aa) * MC + b] = X2[b * MC + aa];
            double v = (aa <= b) ? sG[aa * MC + b] : sG[b * MC + aa];
            if (aa == b) v += X1[aa * MC + aa];
            X3[aa * MC + b] = v;
        }
        __syncthreads();
        for (int t = tid; t < j * j; t += TS_NT) {       // M nRinv -> sG (G is no longer needed)
            const int aa = t / j, b = t % j;
            double v = 0.0;
            for (int u = 0; u <= b; ++u) v += X3[aa * MC + u] * X2[u * MC + b];
            sG[aa * MC + b] = v;
        }
        __syncthreads();
        for (int t = tid; t < j * j; t += TS_NT) {       // D22 = nRinv' (M nRinv)
            const int aa = t / j, b = t % j;
            double v = 0.0;
            for (int u = 0; u <= aa; ++u) v += X2[u * MC + aa] * sG[u * MC + b];
            sD[(j + aa) * MC + (j + b)] = v;
        }
        __syncthreads();
        for (int t = tid; t < k * m; t += TS_NT) {       // C = I + R D R' (k x k), V = chol(C).U     (src/woodbury.jl:205)
            const int aa = t / m, b = t % m;
            double v = 0.0;
            for (int u = aa; u < m; ++u) v += sR[aa * MC + u] * sD[u * MC + b];
            sG[aa * MC + b] = v;
        }
        __syncthreads();
        for (int t = tid; t < k * k; t += TS_NT) {
            const int aa = t / k, b = t % k;
            if (b >= aa) {
                double v = (aa == b) ? 1.0 : 0.0;
                for (int u = b; u < m; ++u) v += sG[aa * MC + u] * sR[b * MC + u];
                sV[aa * MC + b] = v;
            }
        }
        __syncthreads();
        if (tid < 64) {                          // wave 0: left-looking Cholesky, lane b owns column b
            const int b = tid;
            volatile double *Vv = sV;
            volatile int *vst = &sStatus;
            volatile double *vld = &sLogdetV;
            if (b == 0) { *vst = PFMI_FIT_OK; *vld = 0.0; }
            __builtin_amdgcn_wave_barrier();
            for (int c = 0; c < k; ++c) {
                if (*vst != PFMI_FIT_OK) break;
                if (b == c) {
                    double diag = Vv[c * MC + c];
                    for (int t = 0; t < c; ++t) { const double x = Vv[t * MC + c]; diag -= x * x; }
                    if (!(diag > 0.0) || !isfinite(diag)) *vst = PFMI_FIT_C_NOT_PD;
                    else { diag = sqrt(diag); Vv[c * MC + c] = diag; *vld = *vld + log(diag); }
                }
                __builtin_amdgcn_wave_barrier();
                if (*vst != PFMI_FIT_OK) break;
                if (b > c && b < k) {
                    double v = Vv[c * MC + b];
                    for (int t = 0; t < c; ++t) v -= Vv[t * MC + c] * Vv[t * MC + b];
                    Vv[c * MC + b] = v / Vv[c * MC + c];
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (b >= k && b < MC) Vv[b * MC + b] = 1.0;                             // identity padding
        }
        __syncthreads();
        for (int t = tid; t < MC * MC; t += TS_NT) {
            A.tmat[sm + t] = sT[t]; A.vchol[sm + t] = sV[t]; A.rq[sm + t] = sR[t]; A.dmat[sm + t] = sD[t];
        }
        const bool ok = (sStatus == PFMI_FIT_OK);
        // ---- S (V_c'V_c - I) head: what the mean adds to U g, in the basis of Q_in's columns
        if (tid < 64) {
            const int a = tid;
            volatile double *tmp = sTmp, *sd = sSd;
            if (a < k) {
                double v = 0.0;
                for (int b = a; b < k; ++b) v += sV[a * MC + b] * sHead[b];
                tmp[a] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (a < k) {
                double v = 0.0;
                for (int b = 0; b <= a; ++b) v += sV[b * MC + a] * tmp[b];
                sd[a] = ok ? sS[a] * (v - sHead[a]) : 0.0;
            }
        }
        __syncthreads();
        TS_STAMP(6);                                                           // small algebra + Cholesky

        // ---- pass 2: chunk by chunk -- N_i = -K_i W_i U^-1 (and, for the chunk's first m rows, W_i U^-1), then row-local: Vh and mu
        for (int ci = 0; ci < nch; ++ci) {
            const int base = ci * CH;
            for (int t = tid; t < m * m; t += TS_NT) {                         // M_i = K_i W_i
                const int a = t / m, c = t % m;
                double v = 0.0;
                for (int b = 0; b < m; ++b) v += Kg[(size_t)ci * MC * MC + a * MC + b] * stk[(size_t)(ci * m + b) * SW + c];
                sM[a * MC + c] = v;
            }
            __syncthreads();
            if (tid < 2 * m) {                                                 // X U = Y by rows: Y = -M_i (rows 0 .. m-1) and W_i (rows m .. 2m-1)
                const bool isw = tid >= m;
                const int a = isw ? tid - m : tid;
                double x[MC];
                double dot = 0.0;
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    x[c] = 0.0;
                    if (c < m) {
                        double v = isw ? stk[(size_t)(ci * m + a) * SW + c] : -sM[a * MC + c];
                        dot = fma(v, sSd[c], dot);                             // W_i[a, :] . Sd   resp.   -M_i[a, :] . Sd
#pragma unroll
                        for (int b = 0; b < c; ++b) v = fma(-x[b], sU[b * MC + c], v);
                        x[c] = v / sU[c * MC + c];
                    }
                }
                double *dst = isw ? sWu : sN;
#pragma unroll
                for (int c = 0; c < MC; ++c) dst[a * MC + c] = x[c];
                (isw ? sWd : sY)[a] = dot;
            }
            __syncthreads();
            {
                int tb = tid;
                TS_OPAQUE(tb);
                double out[RC][MC], yv[RC];
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    yv[i] = 0.0;
#pragma unroll
                    for (int c = 0; c < MC; ++c) out[i][c] = 0.0;
                }
#pragma unroll
                for (int a = 0; a < MC; ++a) {
                    if (a < m) {
                        double va[RC];
#pragma unroll
                        for (int i = 0; i < RC; ++i) va[i] = scr[(size_t)a * DP + base + tb + TS_NT * i];
                        const double ya = sY[a];
#pragma unroll
                        for (int c = 0; c < MC; ++c) {
                            const double n = sN[a * MC + c];
#pragma unroll
                            for (int i = 0; i < RC; ++i) out[i][c] = fma(va[i], n, out[i][c]);
                        }
#pragma unroll
                        for (int i = 0; i < RC; ++i) yv[i] = fma(va[i], ya, yv[i]);
                    }
                }
                if (tb < m) {                                                  // the chunk's first m rows carry the W_i term; the very first m rows ARE L
#pragma unroll
                    for (int c = 0; c < MC; ++c) out[0][c] = (ci == 0) ? sL[tb * MC + c] : out[0][c] + sWu[tb * MC + c];
                    yv[0] += sWd[tb];
                }
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    const int row = base + tb + TS_NT * i;
                    if (row < d) {
                        const double sq = sqa[row];
                        mu[row] = ok ? theta_p[row] + sq * (sq * grad_p[row] + yv[i]) : NAN;
                        double2 *o = reinterpret_cast<double2 *>(Vh + (size_t)row * MC);
#pragma unroll
                        for (int c = 0; c < MC; c += 2) o[c >> 1] = make_double2(out[i][c], out[i][c + 1]);
                    }
                }
            }
            __syncthreads();                                                   // sM / sN / sWu are rewritten by the next chunk
        }
        TS_STAMP(7);                                                           // pass 2
        if (tid == 0) {
            A.status[p] = ok ? PFMI_FIT_OK : sStatus;
            A.logdet[p] = ok ? 2.0 * (ldu + sLogdetV) : NAN;
        }
    }
#if TS_PROF
    if (blockIdx.x == 0 && tid == 0)
        printf("TS_PROF fits %d (100 MHz ticks per fit): fetch %lld  inputs %lld  chunkQR %lld  reflectors-out %lld  stack %lld  top-block %lld  small+chol %lld  pass2 %lld\n",
               nfit, prof[0] / nfit, prof[1] / nfit, prof[2] / nfit, prof[3] / nfit, prof[4] / nfit, prof[5] / nfit, prof[6] / nfit, prof[7] / nfit);
#endif
}

// ---------------------------------------------------------------------------------------------------
template <int MC, int RC>
static int32_t launch_tsqr_t(pfmi_ctx *c, const FitArgs &a, int ncu) {
    constexpr int CH = TS_NT * RC;
    const int lds = ts_lds_doubles(MC) * (int)sizeof(double);
    auto kern = pf_fit_tsqr_kernel<MC, RC>;
    PF_TRY(pf_raise_lds_limit(c, reinterpret_cast<const void *>(kern), lds));
    const int nch = (a.d + CH - 1) / CH, DP = nch * CH;
    PF_CHECK(nch * MC <= TS_NT, PFMI_ERR_UNSUPPORTED, "tsqr fit: %d chunks x %d columns exceed the stack", nch, MC);
    int64_t slots = ncu;
    if (const char *g = pf_debug_get("PFMI_FIT_PANEL_GRID")) { const int v = atoi(g); if (v > 0) slots = v; }   // experiment hook: resident workgroups
    const int grid = slots < a.P ? (int)slots : (int)a.P;
    const size_t wg_doubles = (size_t)MC * DP + (size_t)TS_NT * (MC + 4) + (size_t)nch * MC * MC;
    const size_t scr_bytes = (size_t)grid * wg_doubles * sizeof(double);
    PF_TRY(c->fit_scratch.ensure(scr_bytes + 256));
    int *counter = reinterpret_cast<int *>(c->fit_scratch.as<char>() + scr_bytes);
    PF_HIP(hipMemsetAsync(counter, 0, 8 * sizeof(int), c->stream));               // one work counter per XCD
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TS_NT), lds, c->stream, a, DP, nch, c->fit_scratch.as<double>(), counter);
    return PFMI_OK;
}

// returns PFMI_OK and sets *handled when the TSQR kernel took the launch (1024 < d <= 16384, 4 <= J <= 16)
int32_t pf_launch_fit_tsqr(pfmi_ctx *c, const FitArgs &a, bool *handled) {
    *handled = false;
    if (a.d <= 1024 || a.d > 16384) return PFMI_OK;
    int ncu = 0;
    PF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device));
    switch (c->kpad) {                       // rows per thread: the chunk (RC x KPAD doubles per thread) stays inside ~130 of the 256 registers
        case 8: PF_TRY((launch_tsqr_t<8, 8>(c, a, ncu))); break;
        case 12: PF_TRY((launch_tsqr_t<12, 5>(c, a, ncu))); break;
        case 16: PF_TRY((launch_tsqr_t<16, 4>(c, a, ncu))); break;
        case 20: PF_TRY((launch_tsqr_t<20, 3>(c, a, ncu))); break;
        case 32: PF_TRY((launch_tsqr_t<32, 2>(c, a, ncu))); break;
        default: return PFMI_OK;
    }
    *handled = true;
    return PFMI_OK;
}
