// fit_tsqr_kernel.hip -- the Woodbury fit for large d (1024 < d <= 32768 while chunks x columns <= 512: KPAD <= 20 throughout, KPAD = 32 to 16384), gfx950, organised so that the d x 2J block crosses HBM ONCE
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

#ifndef TS_NT
#define TS_NT 512                      // threads per workgroup: 512 = one workgroup per CU (2 waves x 256 registers per SIMD).  256 puts TWO workgroups on a
                                       // CU so that one fit's memory phases run under the other's reductions -- measured equal at config 5's shape (5.86 against
                                       // 5.92 ms: the SIMDs are issue-bound in the column loop either way) and slower for a handful of fits (twice the chunks per
                                       // fit); profiles/r06_experiments.md
#endif
#define TS_NW (TS_NT / 64)
#ifndef TS_TWAVE
#define TS_TWAVE 1                     // wave that owns the rows of a block's compact-WY T during its factorisation: the SECOND wave (another SIMD), beside wave 0's row-c work (0: behind it; config-5 shape 5.72 against 6.05 ms)
#endif
#define TS_STACK 512                   // rows of the stack of R factors (n_chunks x m <= 512): TS_STACK / TS_NT rows per thread in the stack stage
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
// of 8 NV; a total is fetched into scalar registers where it is used (all NV at once spill the scalar file).  One barrier per call (ping-pong halves of `red`, 2 x TS_NW x NV doubles).
template <int NV>
__device__ __forceinline__ double ts_block_sum(double (&v)[NV], double *red, int &flip) {
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
    const int l = lane < NV ? lane : NV - 1;
    double t[TS_NW];
#pragma unroll
    for (int w = 0; w < TS_NW; ++w) t[w] = buf[w * NV + l];
    double s = t[0];
#pragma unroll
    for (int w = 1; w < TS_NW; ++w) s += t[w];
    return s;                                  // lane l < NV of every wave: the total of value l (pf_readlane_f64(s, l) where it is used)
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
    int tid = threadIdx.x;
    TS_OPAQUE(tid);                                                      // per column: the compiler otherwise precomputes the masks tid == c, tid > c,
                                                                         // tid < c of ALL columns (and the T row addresses) in front of the fit loop and spills them
    double *srow = srow2 + (c & 1) * NV;
    if (tid == c) {
#pragma unroll
        for (int cc = 0; cc < MC; ++cc) srow[cc] = P[0][cc];
        srow[MC] = U[0];
    }
    double dots[NV];
    {
        const double x0 = (tid > c) ? P[0][c] : 0.0;                     // rows below the diagonal (row 0 of a thread may be the diagonal row or above it)
#pragma unroll
        for (int v = 0; v < MC; ++v) dots[v] = x0 * P[0][v];
        dots[MC] = x0 * U[0];
#pragma unroll
        for (int v = MC + 1; v < NV; ++v) dots[v] = 0.0;
    }
#pragma unroll
    for (int i = 1; i < RC; ++i) {
        const double xc = P[i][c];
#pragma unroll
        for (int v = 0; v < MC; ++v) dots[v] = fma(xc, P[i][v], dots[v]);
        dots[MC] = fma(xc, U[i], dots[MC]);
    }
    const double tot = ts_block_sum<NV>(dots, red, flip);                // its barrier also publishes srow (double buffered)
    const double xn2 = pf_readlane_f64(tot, c), alpha_c = srow[c];
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
        const double w = tau * (srow[cc] + scal * pf_readlane_f64(tot, cc));
#pragma unroll
        for (int i = 0; i < RC; ++i) P[i][cc] = fma(-w, vs[i], P[i][cc]);
    }
    {
        const double w = tau * (srow[MC] + scal * pf_readlane_f64(tot, MC));   // the extra column
#pragma unroll
        for (int i = 0; i < RC; ++i) U[i] = fma(-w, vs[i], U[i]);
    }
    if (tid == c) {                                                      // row c (its entries were not touched above: vs = 0 there): the R entries and the
                                                                         // transformed U out, explicit unit diagonal / zeros in -- ONE divergent region per column
#pragma unroll
        for (int cc = c + 1; cc < MC; ++cc) {
            const double w = tau * (srow[cc] + scal * pf_readlane_f64(tot, cc));
            if (cc < m) rdst[c * rstride + cc] = P[0][cc] - w;
            P[0][cc] = 0.0;
        }
        hdst[c] = U[0] - tau * (srow[MC] + scal * pf_readlane_f64(tot, MC));
        rdst[c * rstride + c] = beta;
        P[0][c] = 1.0;
    }
    // dlarft: T[0:c, c] = -tau T[0:c, 0:c] (V[:, 0:c]' v_c), row a by thread TS_TWAVE * 64 + a (a T row is written and read by that one thread only
    // while the loop runs).  TS_TWAVE = 1 puts the rows on the SECOND wave (another SIMD), beside wave 0's row-c work instead of behind it
    {
        const int ta = tid - 64 * TS_TWAVE;
        if (ta == c) sT[c * MC + c] = tau;
        if (ta >= 0 && ta < c) {
            double acc = 0.0;
#pragma unroll
            for (int b = 0; b < c; ++b) {
                const double g = srow[b] + scal * pf_readlane_f64(tot, b);   // V[c, b] . 1 + sum over the rows below
                acc += (b >= ta) ? sT[ta * MC + b] * g : 0.0;
            }
            sT[ta * MC + c] = -tau * acc;
        }
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

// LDS carve-up of the kernel.  Every function derives it from the dynamic-LDS symbol itself (not from pointers handed down): inside a
// non-inlined function a plain `double *` argument is a generic pointer, and every access through it a flat load
struct TsLds {
    double *red, *srow2, *sT, *sR, *sD, *sV, *sG, *X1, *X2, *X3, *sL, *sU, *sM, *sN, *sS, *sHead, *sSd, *sY, *sWd, *sTmp, *sDg;
    int *status, *hist;
    double *logdetV;
};
template <int MC>
__device__ __forceinline__ TsLds ts_lds_layout() {
    extern __shared__ double lds[];
    constexpr int NV = MC + 4, Q = MC * MC;
    TsLds L;
    L.red = lds;
    L.srow2 = L.red + 2 * TS_NW * NV;
    L.sT = L.srow2 + 2 * NV;
    L.sR = L.sT + Q; L.sD = L.sR + Q; L.sV = L.sD + Q; L.sG = L.sV + Q;
    L.X1 = L.sG + Q; L.X2 = L.X1 + Q; L.X3 = L.X2 + Q;
    L.sL = L.X3 + Q; L.sU = L.sL + Q; L.sM = L.sU + Q; L.sN = L.sM + Q;                      // 12 MC^2 in all
    L.sS = L.sN + Q; L.sHead = L.sS + MC; L.sSd = L.sHead + MC; L.sY = L.sSd + MC; L.sWd = L.sY + MC; L.sTmp = L.sWd + MC; L.sDg = L.sTmp + MC;
    L.logdetV = L.sDg + MC;
    L.status = reinterpret_cast<int *>(L.logdetV + 1);
    L.hist = reinterpret_cast<int *>(L.logdetV + 2);                                          // [16]: this fit's row of hist_src
    return L;
}

template <int MC>
__device__ __noinline__ int ts_stack_stage(double *stk, const int m_, const int nst_, int flip_) {
    const TsLds L = ts_lds_layout<MC>();
    constexpr int SW = MC + 4;
    // (arguments of a non-inlined function arrive in vector registers: tell the compiler they are wave-uniform)
    const int m = __builtin_amdgcn_readfirstlane(m_), nst = __builtin_amdgcn_readfirstlane(nst_);
    int flip = __builtin_amdgcn_readfirstlane(flip_);
    double *red = L.red, *srow2 = L.srow2, *sT = L.sT, *sR = L.sR, *sHead = L.sHead, *sM = L.sM, *sDg = L.sDg;
    int tid = threadIdx.x;
    TS_OPAQUE(tid);
    {
        constexpr int RCT = TS_STACK / TS_NT;                              // stack row r = tid + TS_NT i
        double Pt[RCT][MC], Ut[RCT];
#pragma unroll
        for (int i = 0; i < RCT; ++i) {
            const int r = tid + TS_NT * i;
            const int cr = r % (m > 0 ? m : 1);                            // row cr of its R_i: entries left of the diagonal were never written
#pragma unroll
            for (int cc = 0; cc < MC; ++cc) Pt[i][cc] = (r < nst && cc >= cr && cc < m) ? stk[(size_t)(r < nst ? r : 0) * SW + cc] : 0.0;
            Ut[i] = (r < nst) ? stk[(size_t)r * SW + MC] : 0.0;
        }
        ts_qr_cols<RCT, MC>(Pt, Ut, m, sT, srow2, red, flip, sR, MC, sHead);
        __syncthreads();
        TS_OPAQUE(tid);
        if (tid < m) {                                                     // K_t = T_t V_t[0:m, :]'  -> sM (column tid), sign D of R_in's diagonal
#pragma unroll
            for (int a = 0; a < MC; ++a) {                                 // one entry at a time, stored at once: no arrays, no wall of hoisted LDS reads
                double kv = 0.0;
#pragma unroll
                for (int b = a; b < MC; ++b) kv = fma(sT[a * MC + b], Pt[0][b], kv);
                sM[a * MC + tid] = kv;
                __builtin_amdgcn_sched_barrier(0);
            }
            sDg[tid] = (sR[tid * MC + tid] < 0.0) ? -1.0 : 1.0;
        }
        __syncthreads();
        TS_OPAQUE(tid);
#pragma unroll
        for (int i = 0; i < RCT; ++i) {
            const int r = tid + TS_NT * i;
            if (r < nst) {                                                 // W[r, c] = D_c (delta_rc - V_t[r, :] K_t[:, c])
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    double wv = (r == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int a = 0; a < MC; ++a) wv = fma(-Pt[i][a], sM[a * MC + c], wv);
                    stk[(size_t)r * SW + c] = (c < m) ? wv * sDg[c] : 0.0;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (tid < m) {                                                     // R_in <- D R_in, head <- D head
            const double dg = sDg[tid];
#pragma unroll
            for (int c = 0; c < MC; ++c) sR[tid * MC + c] *= dg;
            sHead[tid] *= dg;
        }
        __syncthreads();
    }
    return flip;
}

template <int MC>
__device__ __noinline__ void ts_small_stage(const FitArgs &A, const int64_t p, const int j_, const int nch_, const int DP_, double *scr, double *stk, double *Kg,
                                            double *Mg, double *Ng, double *Wug, double *Yg, double *Wdg) {
    const TsLds L = ts_lds_layout<MC>();
    constexpr int SW = MC + 4;
    const int j = __builtin_amdgcn_readfirstlane(j_), nch = __builtin_amdgcn_readfirstlane(nch_), DP = __builtin_amdgcn_readfirstlane(DP_);
    const int m = 2 * j, k = m;
    const size_t sm = (size_t)p * MC * MC;
    double *sT = L.sT, *sR = L.sR, *sD = L.sD, *sV = L.sV, *sG = L.sG, *X1 = L.X1, *X2 = L.X2, *X3 = L.X3, *sL = L.sL, *sU = L.sU, *sN = L.sN,
           *sS = L.sS, *sHead = L.sHead, *sSd = L.sSd, *sTmp = L.sTmp;
    int &sStatus = *L.status;
    double &sLogdetV = *L.logdetV;
    int tid = threadIdx.x;
    // ---- the top m x m block of Q_in = W_0 - V_0[0:m, :] (K_0 W_0), modified LU (S chosen on the fly), T = -U S V_1^-T, R = S R_in
    TS_OPAQUE(tid);
    // (unconditional loads from clamped addresses in fully unrolled loops: all in flight at once -- as run-time loops these two were 2 m dependent
    //  global round trips per fit)
    for (int t = tid; t < m * m; t += TS_NT) {
        const int a = t / m, c = t % m;
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < MC; ++b) {
            const int bb = b < m ? b : m - 1;
            const double kb = Kg[a * MC + bb], wb = stk[(size_t)bb * SW + c];
            v = fma(b < m ? kb : 0.0, wb, v);
        }
        sN[a * MC + c] = v;                                                // M_0
    }
    __syncthreads();
    for (int t = tid; t < m * m; t += TS_NT) {
        const int r = t / m, c = t % m;
        double v = stk[(size_t)r * SW + c];
#pragma unroll
        for (int a = 0; a < MC; ++a) {
            const int aa = a <= r ? a : r;
            const double va = scr[(size_t)aa * DP + r];                        // V_0[r, a]: zero for a > r
            v = fma(a <= r ? -va : 0.0, sN[aa * MC + c], v);
        }
        sL[r * MC + c] = v;
    }
    __syncthreads();
    TS_OPAQUE(tid);
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

    // ---- small algebra (as in fit_panel_kernel.hip).  G = B~'B~ = R'R:  G[c][b] (c, b < j) = Y'alpha Y,  G[j + a][b] = S'Y
    TS_OPAQUE(tid);
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


    // ---- the m x m matrices of pass 2 for ALL chunks at once (every thread busy; the chunk loop below then has no small algebra left):
    //      M_i = K_i W_i,   N_i = -M_i U^-1,   y_i = -M_i Sd,   and for the chunks' first m rows  W_i U^-1,  W_i Sd
    TS_OPAQUE(tid);
    for (int t = tid; t < nch * m * m; t += TS_NT) {
        const int ci = t / (m * m), r = t - ci * m * m, a = r / m, c = r - a * m;
        const double *kr = Kg + (size_t)ci * MC * MC + a * MC, *wc = stk + (size_t)(ci * m) * SW + c;
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < MC; ++b) {                                     // unconditional loads from clamped addresses: all in flight at once
            const int bb = b < m ? b : m - 1;
            const double kb = kr[bb], wb = wc[(size_t)bb * SW];
            v = fma(b < m ? kb : 0.0, wb, v);
        }
        Mg[(size_t)ci * MC * MC + a * MC + c] = v;
    }
    if (tid < m) sTmp[tid] = 1.0 / sU[tid * MC + tid];                     // |U_cc| >= 1
    __threadfence_block();
    __syncthreads();
    TS_OPAQUE(tid);
    for (int t = tid; t < nch * 2 * m; t += TS_NT) {                       // X U = Y by rows: Y = -M_i (rows 0 .. m-1) and W_i (rows m .. 2m-1)
        const int ci = t / (2 * m), r = t - ci * 2 * m;
        const bool isw = r >= m;
        const int a = isw ? r - m : r;
        const double *src = isw ? stk + (size_t)(ci * m + a) * SW : Mg + (size_t)ci * MC * MC + a * MC;
        double x[MC], dot = 0.0;
#pragma unroll
        for (int c = 0; c < MC; ++c) { const double v = src[c < m ? c : 0]; x[c] = (c < m) ? (isw ? v : -v) : 0.0; }
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            if (c < m) {
                double v = x[c];
                dot = fma(v, sSd[c], dot);                                 // W_i[a, :] . Sd   resp.   -M_i[a, :] . Sd
#pragma unroll
                for (int b2 = 0; b2 < c; ++b2) v = fma(-x[b2], sU[b2 * MC + c], v);
                x[c] = v * sTmp[c];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double *dst = (isw ? Wug : Ng) + (size_t)ci * MC * MC + a * MC;
#pragma unroll
        for (int c = 0; c < MC; ++c) dst[c] = x[c];
        (isw ? Wdg : Yg)[ci * MC + a] = dot;
    }
    __threadfence_block();
    __syncthreads();

}

static int ts_lds_doubles(int MC) {
    const int NV = MC + 4;
    return 2 * TS_NW * NV + 2 * NV + 12 * MC * MC + 12 * MC + 32;
}

template <int MC, int RC>
__global__ __launch_bounds__(TS_NT, 512 / TS_NT) void pf_fit_tsqr_kernel(FitArgs A, const int DP, const int nch_max, double *scratch_all, int *counter) {
    constexpr int NV = MC + 4, CH = TS_NT * RC, SW = MC + 4;
    int tid = threadIdx.x;                                       // re-defined (opaque) at every phase boundary: whatever is derived from it -- masks tid == c,
                                                                 // LDS row addresses, ... -- is invariant across fits and would be hoisted in front of the fit loop and spilled
    const int d = A.d, J = A.J;
    // per-workgroup scratch: reflectors column-major [MC][DP] | stack / W rows [TS_STACK][SW] | per chunk: K_i, M_i, N_i, W_i U^-1 [MC][MC], y_i, W_i Sd [MC]
    const size_t small = (size_t)nch_max * MC * MC;
    const size_t wg_doubles = (size_t)MC * DP + (size_t)TS_STACK * SW + 4 * small + 2 * (size_t)nch_max * MC;
    double *scr_wg = scratch_all + (size_t)blockIdx.x * wg_doubles;

    const TsLds L = ts_lds_layout<MC>();
    double *red = L.red, *srow2 = L.srow2, *sT = L.sT, *sL = L.sL, *sM = L.sM, *sN = L.sN, *sS = L.sS, *sY = L.sY, *sWd = L.sWd, *sTmp = L.sTmp;
    int &sStatus = *L.status;
    __shared__ int sNext;
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
        // the scratch base is made opaque once per fit: everything derived from it (dozens of column / chunk addresses, all invariant across
        // fits) would otherwise be hoisted in front of the fit loop and live -- spilled -- through every phase
        double *scr = scr_wg;
        asm volatile("" : "+s"(scr));
        double *stk = scr + (size_t)MC * DP, *Kg = stk + (size_t)TS_STACK * SW, *Mg = Kg + small, *Ng = Mg + small, *Wug = Ng + small;
        double *Yg = Wug + small, *Wdg = Yg + (size_t)nch_max * MC;
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

        TS_OPAQUE(tid);
        for (int t = tid; t < 12 * MC * MC; t += TS_NT) sT[t] = 0.0;          // the twelve small matrices are contiguous
        for (int t = tid; t < 7 * MC; t += TS_NT) sS[t] = 0.0;
        if (tid < j) L.hist[tid] = A.hist_src[(size_t)p * J + tid];            // (one load each: as scalar loads in the column loops they were 2 m dependent round trips per chunk)
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
                // column cc: alpha.y of pair cc (cc < j) or s of pair cc - j;  y = grad_l - grad_{l+1} :46,  s = theta_{l+1} - theta_l :45, l = hist_src.
                // Rows l + 1 ("hi") of ALL columns are fetched first, straight into the register block (RC x m loads in flight); row l ("lo") of a
                // column is the hi row of its left neighbour whenever the two pairs are consecutive trace steps (the usual case: every update
                // accepted), so a chunk reads ~(j + 1) rows of each trace array instead of 2 j.
#pragma unroll
                for (int cc = 0; cc < MC; ++cc) {
                    if (cc < m) {
                        const bool isy = cc < j;
                        const int src = __builtin_amdgcn_readfirstlane(L.hist[isy ? cc : cc - j]);
                        const double *hi = (isy ? A.grad : A.theta) + (size_t)(p0 + src + 1) * d;
#pragma unroll
                        for (int i = 0; i < RC; ++i) {
                            const int row = base + tb + TS_NT * i, rl = row < d ? row : d - 1;
                            P[i][cc] = hi[rl];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < RC; ++i) P[i][cc] = 0.0;
                    }
                }
#pragma unroll
                for (int cc = MC - 1; cc >= 0; --cc) {                         // right to left: the left neighbour still holds its hi row
                    if (cc < m) {
                        const bool isy = cc < j;
                        const int pair = isy ? cc : cc - j;
                        const int src = __builtin_amdgcn_readfirstlane(L.hist[pair]);
                        const bool reuse = pair > 0 && __builtin_amdgcn_readfirstlane(L.hist[pair > 0 ? pair - 1 : 0]) + 1 == src;   // wave-uniform
                        const double *lo = (isy ? A.grad : A.theta) + (size_t)(p0 + src) * d;
                        double lov[RC];
                        if (reuse) {
#pragma unroll
                            for (int i = 0; i < RC; ++i) lov[i] = P[i][cc > 0 ? cc - 1 : 0];
                        } else {
#pragma unroll
                            for (int i = 0; i < RC; ++i) {
                                const int row = base + tb + TS_NT * i, rl = row < d ? row : d - 1;
                                lov[i] = lo[rl];
                            }
                        }
#pragma unroll
                        for (int i = 0; i < RC; ++i) {
                            const int row = base + tb + TS_NT * i;
                            const double df = isy ? lov[i] - P[i][cc] : P[i][cc] - lov[i];
                            const double v = isy ? (al[i] * df) * isa[i] : df * isa[i];
                            P[i][cc] = (row < d) ? v : 0.0;
                        }
                    }
                }
            }
            TS_STAMP(1);                                                       // inputs
            ts_qr_cols<RC, MC>(P, U, m, sT, srow2, red, flip, stk + (size_t)ci * m * SW, SW, sTmp);
            __syncthreads();                                                   // T_i complete, sTmp (the chunk's head) written
            TS_STAMP(2);                                                       // chunk QR
            TS_OPAQUE(tid);
            if (tid < m) {
                stk[(size_t)(ci * m + tid) * SW + MC] = sTmp[tid];
                // column tid of K_i = T_i V_top':  K_i[a][tid] = sum_{b = a .. tid} T_i[a][b] V_top[tid][b]   (V_top = this thread's row 0: zero above
                // the diagonal, one on it)
#pragma unroll
                for (int a = 0; a < MC; ++a) {
                    double kv = 0.0;
#pragma unroll
                    for (int b = a; b < MC; ++b) kv = fma(sT[a * MC + b], P[0][b], kv);
                    Kg[(size_t)ci * MC * MC + a * MC + tid] = kv;
                    __builtin_amdgcn_sched_barrier(0);
                }
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
            const double tot = ts_block_sum<4>(v, red, flip);
            bad = pf_readlane_f64(tot, 0); ldu = pf_readlane_f64(tot, 1);
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

        // ---- the stack: QR of [R_0; R_1; ...] with the heads as the extra column; W = D-signed first m columns of Q_top (row-local).
        //      (__noinline__ functions from here to pass 2: inlined, their values shared one register allocation with the chunk loops and the
        //      allocator spilled the stack's row block through all of its 20 columns)
        const int nst = nch * m;                                               // <= 512 by the choice of RC
        flip = ts_stack_stage<MC>(stk, m, nst, flip);
        TS_STAMP(4);                                                           // stack QR + W
        // ---- the top m x m block of Q_in, modified LU, T, R; small algebra, Cholesky; the m x m matrices of pass 2 for all chunks
        ts_small_stage<MC>(A, p, j, nch, DP, scr, stk, Kg, Mg, Ng, Wug, Yg, Wdg);
        const bool ok = (sStatus == PFMI_FIT_OK);
        TS_STAMP(6);                                                           // top block, small algebra, Cholesky, pass-2 matrices
        // ---- pass 2: chunk by chunk, row-local -- Vh[r, :] = V_i[r, :] N_i (+ W_i U^-1 for a chunk's first m rows; the very first m rows ARE L),
        //      mu = theta + U'(U g + V_i[r, :] y_i (+ W_i Sd)).  N_i / y_i are staged in LDS, two buffers in turn: ONE barrier per chunk
        for (int ci = 0; ci < nch; ++ci) {
            const int base = ci * CH;
            double *bN = (ci & 1) ? sN : sM, *bY = (ci & 1) ? sWd : sY;
            for (int t = tid; t < MC * MC; t += TS_NT) bN[t] = (t / MC < m) ? Ng[(size_t)ci * MC * MC + t] : 0.0;
            if (tid < MC) bY[tid] = (tid < m) ? Yg[ci * MC + tid] : 0.0;
            __syncthreads();
            {
                int tb = tid;
                TS_OPAQUE(tb);
                // every load of the chunk is issued before the first use (RC x m reflector entries + 3 RC row scalars per thread: one latency
                // per chunk instead of m / 4); the outputs are then formed in two column halves so that accumulators + reflectors fit
                double va[MC][RC], sqv[RC], thv[RC], grv[RC], yv[RC];
#pragma unroll
                for (int a = 0; a < MC; ++a) {
#pragma unroll
                    for (int i = 0; i < RC; ++i) va[a][i] = (a < m) ? scr[(size_t)a * DP + base + tb + TS_NT * i] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    const int row = base + tb + TS_NT * i, rl = row < d ? row : d - 1;
                    sqv[i] = sqa[rl]; thv[i] = theta_p[rl]; grv[i] = grad_p[rl];
                    yv[i] = 0.0;
                }
#pragma unroll
                for (int a = 0; a < MC; ++a) {
                    const double ya = bY[a];                                   // zero for a >= m
#pragma unroll
                    for (int i = 0; i < RC; ++i) yv[i] = fma(va[a][i], ya, yv[i]);
                }
                if (tb < m) yv[0] += Wdg[ci * MC + tb];
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    const int row = base + tb + TS_NT * i;
                    if (row < d) mu[row] = ok ? thv[i] + sqv[i] * (sqv[i] * grv[i] + yv[i]) : NAN;
                }
                constexpr int HC = MC / 2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double out[RC][HC];
#pragma unroll
                    for (int i = 0; i < RC; ++i)
#pragma unroll
                        for (int c = 0; c < HC; ++c) out[i][c] = 0.0;
#pragma unroll
                    for (int a = 0; a < MC; ++a) {
                        if (a < m) {
#pragma unroll
                            for (int c = 0; c < HC; ++c) {
                                const double n = bN[a * MC + h * HC + c];
#pragma unroll
                                for (int i = 0; i < RC; ++i) out[i][c] = fma(va[a][i], n, out[i][c]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);                     // one row of N at a time
                    }
                    if (tb < m) {                                              // the chunk's first m rows carry the W_i term; the very first m rows ARE L
                        const double *wu = Wug + (size_t)ci * MC * MC + tb * MC + h * HC;
#pragma unroll
                        for (int c = 0; c < HC; ++c) out[0][c] = (ci == 0) ? sL[tb * MC + h * HC + c] : out[0][c] + wu[c];
                    }
#pragma unroll
                    for (int i = 0; i < RC; ++i) {
                        const int row = base + tb + TS_NT * i;
                        if (row < d) {
                            double2 *o = reinterpret_cast<double2 *>(Vh + (size_t)row * MC + h * HC);
#pragma unroll
                            for (int c = 0; c < HC; c += 2) o[c >> 1] = make_double2(out[i][c], out[i][c + 1]);
                        }
                    }
                }
            }
        }
        TS_STAMP(7);                                                           // pass 2
        if (tid == 0) {
            A.status[p] = ok ? PFMI_FIT_OK : sStatus;
            A.logdet[p] = ok ? 2.0 * (ldu + *L.logdetV) : NAN;
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
static int32_t launch_tsqr_t(pfmi_ctx *c, const FitArgs &a, int ncu, bool *handled) {
    constexpr int CH = TS_NT * RC;
    const int lds = ts_lds_doubles(MC) * (int)sizeof(double);
    auto kern = pf_fit_tsqr_kernel<MC, RC>;
    PF_TRY(pf_raise_lds_limit(c, reinterpret_cast<const void *>(kern), lds));
    const int nch = (a.d + CH - 1) / CH, DP = nch * CH;
    if (nch * MC > TS_STACK) { *handled = false; return PFMI_OK; }              // more chunks than the stack holds: the panel kernel takes the launch
    int occ = 1;
    PF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, TS_NT, lds));
    if (occ < 1) occ = 1;
    if (occ > 512 / TS_NT) occ = 512 / TS_NT;
    int64_t slots = (int64_t)ncu * occ;
    if (const char *g = pf_debug_get("PFMI_FIT_PANEL_GRID")) { const int v = atoi(g); if (v > 0) slots = v; }   // experiment hook: resident workgroups
    const int grid = slots < a.P ? (int)slots : (int)a.P;
    const size_t wg_doubles = (size_t)MC * DP + (size_t)TS_STACK * (MC + 4) + 4 * (size_t)nch * MC * MC + 2 * (size_t)nch * MC;
    const size_t scr_bytes = (size_t)grid * wg_doubles * sizeof(double);
    PF_TRY(c->fit_scratch.ensure(scr_bytes + 256));
    int *counter = reinterpret_cast<int *>(c->fit_scratch.as<char>() + scr_bytes);
    PF_HIP(hipMemsetAsync(counter, 0, 8 * sizeof(int), c->stream));               // one work counter per XCD
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TS_NT), lds, c->stream, a, DP, nch, c->fit_scratch.as<double>(), counter);
    return PFMI_OK;
}

// returns PFMI_OK and sets *handled when the TSQR kernel took the launch (1024 < d <= 32768 and the stack of R factors holds chunks x KPAD <= 512 rows:
// a launcher declines otherwise and the panel / column-by-column kernel takes over)
int32_t pf_launch_fit_tsqr(pfmi_ctx *c, const FitArgs &a, bool *handled) {
    *handled = false;
    if (a.d <= 1024 || a.d > 32768) return PFMI_OK;
    *handled = true;                         // (a launcher that declines resets it)
    int ncu = 0;
    PF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device));
    switch (c->kpad) {                       // rows per thread: the chunk (RC x KPAD doubles per thread) stays inside ~130 of the 256 registers
        case 8: PF_TRY((launch_tsqr_t<8, 8>(c, a, ncu, handled))); break;
        case 12: PF_TRY((launch_tsqr_t<12, 5>(c, a, ncu, handled))); break;
        case 16: PF_TRY((launch_tsqr_t<16, 4>(c, a, ncu, handled))); break;
        case 20: PF_TRY((launch_tsqr_t<20, 3>(c, a, ncu, handled))); break;
        // KPAD = 32 (history_length 11 .. 16): wins once the block is large (d = 8000, 968 fits: 7.5 against the panel kernel's 10.6 ms; d = 10^4:
        // 7.7 against 9.2), loses for a short block (d = 2000, 46 fits: 0.66 against 0.54 ms -- 32 columns cost 33 block reductions per chunk):
        // from d = 4096; "tsqr" forces it (tests, probes)
        case 32: {
            const char *f = pf_debug_get("PFMI_FIT_KERNEL");
            if (a.d < 4096 && !(f && f[0] == 't')) { *handled = false; return PFMI_OK; }
            PF_TRY((launch_tsqr_t<32, 2>(c, a, ncu, handled)));
        } break;
        default: *handled = false; return PFMI_OK;
    }
    return PFMI_OK;
}
