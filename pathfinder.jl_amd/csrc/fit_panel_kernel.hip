// fit_panel_kernel.hip -- panel-blocked Woodbury fit for large d (1024 < d <= 16384), gfx950.
//
// Same outputs and the same LAPACK reflector convention as pf_fit_kernel (fit_kernels.hip; reference
// src/inverse_hessian.jl:98-133, src/woodbury.jl:201-207, src/mvnormal.jl:14-21), organised around HBM traffic: at
// d = 10^4, J = 10 the d x 2J block is 1.6 MB per fit, lives in HBM, and the column-by-column kernel sweeps it ~50 times
// (80 MB per fit).  Here a persistent 512-thread workgroup per CU factors the block in panels of PW columns that live in
// REGISTERS (thread t owns rows t + 512 i), left-looking:
//
//   phase A   inputs -> scaled block B~ = U'\[alpha.Y  S], column-major into a per-workgroup scratch (panel 0 stays in
//             registers); sqrt(alpha), log det U
//   panel pi  sweep 1 (MFMA): C = V_prev' [ P_raw | V_panel(pi-1) ] with v_mfma_f64_16x16x4 -- the accumulator tile is
//             spread over the lanes, so 16 x 16 dot products cost 8 VGPRs; lanes read the column-major scratch in operand
//             order.  The second column group completes the compact-WY T (dlarft) of the previous panel for free.
//             sweep 2: P = P_raw - V_prev (T' W) into registers; dgeqr2/dlarfg on the PW register columns (one block
//             reduction per column); R rows to LDS; explicit Householder vectors back to the scratch.
//   final     one more MFMA sweep: V'V (-> last T columns), w1 = V'(U g); G = R'R replaces the Gram sweep of the input
//             block (B~'B~ = R'R); D, C = I + R D R', Cholesky; the mean through the factor needs no further sweep:
//             V'b = w1 - (V'V) t1.  Last pass: scratch -> row-major Vh (what the draw kernels read) + mu.
//
// Traffic per fit at d = 10^4, J = 10: ~14 MB read + ~4 MB written instead of ~80 MB.
#include <type_traits>
#include "pfmi_common.h"
#include <stdlib.h>
#include "fit_args.h"

#ifndef FP_NT
#define FP_NT 512
#endif
#define FP_NW (FP_NT / 64)
#define FP_NVMAX 8
typedef double fp_d4 __attribute__((ext_vector_type(4)));
// Hides a thread index from loop-invariant code motion: without it the compiler precomputes the row addresses of every phase once per
// kernel and keeps ~200 VGPRs of pointers alive next to the register panel.
#define FP_OPAQUE(x) asm volatile("" : "+v"(x))
#ifndef FP_PROF
#define FP_PROF 0                      // 1: workgroup 0 accumulates cycle counts per section and prints them (experiments only)
#endif
#if FP_PROF
#define FP_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = wall_clock64(); prof[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define FP_STAMP(k) do { } while (0)
#endif

static int fp_ntile(int KPAD) { return (KPAD + 15) / 16; }
// dynamic LDS (doubles): red | 6 KPAD^2 matrices (T R D V G Gv) | tile staging [NW][NT16][256] | C [NT16][256] | vectors
static int fp_lds_doubles(int KPAD) {
    return 2 * FP_NW * FP_NVMAX + 6 * KPAD * KPAD + FP_NW * fp_ntile(KPAD) * 256 + fp_ntile(KPAD) * 256 + 12 * KPAD + 16;
}

// KPAD (the column padding of Vh and of the small matrices) is a run-time argument: only the number of 16-column MFMA tiles,
// the rows per thread and the panel width shape the register allocation.
template <int NT16, int RPT, int PW>
__global__ __launch_bounds__(FP_NT, 1) void pf_fit_panel_kernel(FitArgs A, const int KPAD, double *scratch_all, int *counter) {
    constexpr int DPAD = FP_NT * RPT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = A.d, J = A.J;
    double *scr = scratch_all + (size_t)blockIdx.x * (size_t)(KPAD + 4) * DPAD;   // column-major [KPAD + 4][DPAD]; extra columns: U g, 1 / sqrt(alpha), zeros, write-only sink
    double *scr_ug = scr + (size_t)KPAD * DPAD, *scr_isa = scr + (size_t)(KPAD + 1) * DPAD;
    double *scr_zero = scr + (size_t)(KPAD + 2) * DPAD, *scr_sink = scr + (size_t)(KPAD + 3) * DPAD;   // keep every load / store of the sweeps unconditional

    extern __shared__ double lds[];
    double *red = lds;
    double *sT = red + 2 * FP_NW * FP_NVMAX;
    double *sR = sT + KPAD * KPAD, *sD = sR + KPAD * KPAD, *sV = sD + KPAD * KPAD, *sG = sV + KPAD * KPAD, *sGv = sG + KPAD * KPAD;
    double *sTile = sGv + KPAD * KPAD;                       // [NW][NT16][256]; scratch matrices of the small algebra afterwards
    double *sC = sTile + FP_NW * NT16 * 256;                 // [NT16][16][16]  (V column within tile, B column)
    double *sZ = sC + NT16 * 256;                            // [KPAD][PW]
    double *sRowc = sZ + KPAD * 4;                           // [2][4]
    double *sTau = sRowc + 8, *sW1 = sTau + KPAD, *sT1 = sW1 + KPAD, *sT2 = sT1 + KPAD, *sHead = sT2 + KPAD, *sHead2 = sHead + KPAD,
           *sTmp = sHead2 + KPAD;
    __shared__ int sNext, sStatus;
    __shared__ double sLogdetV;
    int flip = 0;
#if FP_PROF
    long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
    int nfit = 0;
#endif

    for (;;) {
        __syncthreads();
        FP_STAMP(11);
#if FP_PROF
        ++nfit;
#endif
        if (tid == 0) {
            // XCD-aware work order (round 4): workgroup b runs on XCD b % 8, each XCD has its own L2, and a fit shares all but two of its
            // 4 J trace rows with its neighbours -- XCD x therefore walks its own contiguous eighth of the points (one counter per
            // XCD) and helps the following XCDs' ranges once its own is exhausted.
            const int P = (int)A.P, q8 = P >> 3, r8 = P & 7;
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
        FP_STAMP(0);

        const int path = A.path_of[p];
        const int64_t p0 = A.off[path];
        const int j = A.hist_len[p], m = 2 * j, k = m;        // d > 1024 >= 2 J: k = min(d, m) = m
        const int npan = (m + PW - 1) / PW;
        const double *alpha = A.alpha_all + (size_t)p * d;
        double *Vh = A.vh + (size_t)p * d * KPAD;
        double *sqa = A.sqrt_alpha + (size_t)p * d;
        double *mu = A.mu + (size_t)p * d;
        const double *theta_p = A.theta + (size_t)p * d, *grad_p = A.grad + (size_t)p * d;
        const size_t sm = (size_t)p * KPAD * KPAD;

        for (int t = tid; t < 6 * KPAD * KPAD; t += FP_NT) sT[t] = 0.0;   // T R D V G Gv are contiguous
        for (int t = tid; t < KPAD; t += FP_NT) { sTau[t] = 0.0; sW1[t] = 0.0; sT1[t] = 0.0; sT2[t] = 0.0; }

        // ---- phase A: U = sqrt(alpha) (src/woodbury.jl:202-203); rows of B~ = U' \ [alpha.Y  S] (src/inverse_hessian.jl:117-118)
        double bad = 0.0, ldu = 0.0;
#pragma unroll 1
        for (int row = tid; row < DPAD; row += FP_NT) {
            double ug = 0.0, isa = 0.0;
            if (row < d) {
                const double al = alpha[row];
                if (!(al > 0.0) || !isfinite(al)) bad = 1.0;
                const double s = sqrt(al);
                isa = 1.0 / s;
                sqa[row] = s;
                ldu += log(s);
                ug = s * grad_p[row];
            }
            scr_ug[row] = ug;
            scr_isa[row] = isa;
            scr_zero[row] = 0.0;
        }
        double P[RPT][PW];
#pragma unroll
        for (int i = 0; i < RPT; ++i)
#pragma unroll
            for (int cc = 0; cc < PW; ++cc) P[i][cc] = 0.0;
        for (int c = 0; c < j; ++c) {
            const int src = A.hist_src[(size_t)p * J + c];
            const double *g0 = A.grad + (size_t)(p0 + src) * d, *g1 = g0 + d, *t0 = A.theta + (size_t)(p0 + src) * d, *t1 = t0 + d;
            double *oy = scr + (size_t)c * DPAD, *os = scr + (size_t)(j + c) * DPAD;
            int tb = tid;
            FP_OPAQUE(tb);
            double *oyc = (c < PW) ? scr_sink : oy, *osc = (j + c < PW) ? scr_sink : os;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                if ((i & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // at most 4 rows (24 loads) in flight next to the register panel
                const int row = tb + FP_NT * i, rl = row < d ? row : d - 1;
                const double al = alpha[rl], isa = scr_isa[row];
                const double y = g0[rl] - g1[rl];                  // y = grad_l - grad_{l+1}   :46
                const double s_ = t1[rl] - t0[rl];                 // s = theta_{l+1} - theta_l :45
                const double by = row < d ? (al * y) * isa : 0.0, bs = row < d ? s_ * isa : 0.0;
#pragma unroll
                for (int cc = 0; cc < PW; ++cc) {
                    P[i][cc] = (cc == c) ? by : P[i][cc];
                    P[i][cc] = (cc == j + c) ? bs : P[i][cc];
                }
                oyc[row] = by;
                osc[row] = bs;
            }
        }
        {
            double v[2] = {bad, ldu};
            pf_block_sum_pp<2, FP_NVMAX>(v, red, flip);
            bad = v[0]; ldu = v[1];
        }
        if (bad > 0.0) {                                           // A not positive definite (src/woodbury.jl:202)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = tid + FP_NT * i;
                if (row < d) {
                    mu[row] = NAN;
                    for (int c = 0; c < KPAD; ++c) Vh[(size_t)row * KPAD + c] = 0.0;
                }
            }
            for (int t = tid; t < KPAD * KPAD; t += FP_NT) { A.tmat[sm + t] = 0.0; A.vchol[sm + t] = 0.0; A.rq[sm + t] = 0.0; A.dmat[sm + t] = 0.0; }
            if (tid == 0) { A.status[p] = PFMI_FIT_A_NOT_PD; A.logdet[p] = NAN; }
            continue;
        }

        FP_STAMP(1);                                               // phase A
        // ---- left-looking panel loop.  The factorisation of a register panel is a lambda called for panel 0 in FRONT of the loop and for
        //      the later panels inside it: with the first iteration peeled, the register panel P is plainly dead across sweep 1 (it is
        //      reloaded by sweep 2), which the compiler could not see through the loop-carried value (36 of its registers were spilled
        //      around every sweep 1; round 4)
        auto factor_panel = [&](const int c0) {
            // ---- dgeqr2 on the register panel: one block reduction per column
            const int ncol = (m - c0 < PW) ? m - c0 : PW;
#pragma unroll
            for (int cl = 0; cl < PW; ++cl) {
                if (cl >= ncol) break;
                const int c = c0 + cl;
                double *srow = sRowc + 4 * (cl & 1);
                double dots[PW];
#pragma unroll
                for (int cc = 0; cc < PW; ++cc) dots[cc] = 0.0;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const double xc = (i > 0 || tid > c) ? P[i][cl] : 0.0;      // rows below the diagonal
#pragma unroll
                    for (int cc = cl; cc < PW; ++cc) dots[cc] += xc * P[i][cc];
                }
                if (tid == c) {
#pragma unroll
                    for (int cc = 0; cc < PW; ++cc) srow[cc] = P[0][cc];
                }
                if constexpr (PW % 4 == 0) pf_block_sum_mv<PW, FP_NVMAX>(dots, red, flip);     // its barrier also publishes srow (double buffered)
                else pf_block_sum_pp<PW, FP_NVMAX>(dots, red, flip);
                const double xn2 = dots[cl];
                const double alpha_c = srow[cl];
                const double xnorm = sqrt(xn2);
                double tau, scal, beta;
                if (xnorm == 0.0) { tau = 0.0; scal = 0.0; beta = alpha_c; }
                else {
                    beta = -copysign(sqrt(fma(alpha_c, alpha_c, xn2)), alpha_c);
                    tau = (beta - alpha_c) / beta;
                    scal = 1.0 / (alpha_c - beta);
                }
                double wv[PW];
#pragma unroll
                for (int cc = 0; cc < PW; ++cc) wv[cc] = tau * (srow[cc] + scal * dots[cc]);
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const bool below = (i > 0 || tid > c);
                    const double v = P[i][cl] * scal;
                    P[i][cl] = below ? v : P[i][cl];
#pragma unroll
                    for (int cc = cl + 1; cc < PW; ++cc) P[i][cc] = below ? P[i][cc] - wv[cc] * v : P[i][cc];
                }
                if (tid == c) {                                    // row c: R entries out, explicit unit diagonal / zeros in
                    sTau[c] = tau;
                    sR[c * KPAD + c] = beta;
                    P[0][cl] = 1.0;
#pragma unroll
                    for (int cc = cl + 1; cc < PW; ++cc) {
                        if (c0 + cc < m) sR[c * KPAD + c0 + cc] = P[0][cc] - wv[cc];
                        P[0][cc] = 0.0;
                    }
                }
            }
            FP_STAMP(5);                                           // panel QR
            // explicit Householder vectors of the panel -> scratch
            int td = tid;
            FP_OPAQUE(td);
            double *oc[PW];
#pragma unroll
            for (int cc = 0; cc < PW; ++cc) oc[cc] = (c0 + cc < m) ? scr + (size_t)(c0 + cc) * DPAD : scr_sink;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = td + FP_NT * i;
#pragma unroll
                for (int cc = 0; cc < PW; ++cc) oc[cc][row] = P[i][cc];
            }
            FP_STAMP(6);                                           // panel write-back
        };
        if (npan > 0) factor_panel(0);
        for (int pi = 1; pi <= npan; ++pi) {                       // iteration npan is the final MFMA sweep only
            const int c0 = pi * PW;                                // first column of this panel = number of finished columns
            {
                __threadfence_block();
                __syncthreads();                                   // scratch columns < c0 written by the whole workgroup
                FP_STAMP(7);
                // sweep 1: C[t] = V[:, 16 t .. 16 t + 15]' B,  B columns: [0, PW) raw panel pi, [PW, 2 PW) V panel pi - 1, 2 PW: U g
                int ln = lane, wv_ = wave;
                FP_OPAQUE(ln); FP_OPAQUE(wv_);
                const int kq = ln >> 4, cl = ln & 15;
                const double *ap[NT16];
                const int nfin = c0 < m ? c0 : m;                  // finished columns
#pragma unroll
                for (int t = 0; t < NT16; ++t) ap[t] = (16 * t + cl < nfin) ? scr + (size_t)(16 * t + cl) * DPAD : scr_zero;
                const double *bp = scr_zero;
                if (cl < PW) { if (pi < npan && c0 + cl < m) bp = scr + (size_t)(c0 + cl) * DPAD; }
                else if (cl < 2 * PW) { if (c0 - 2 * PW + cl < m) bp = scr + (size_t)(c0 - 2 * PW + cl) * DPAD; }
                else if (cl == 2 * PW) bp = scr_ug;
                fp_d4 acc[NT16];
#pragma unroll
                for (int t = 0; t < NT16; ++t) acc[t] = fp_d4{0.0, 0.0, 0.0, 0.0};
                const int roff = 64 * wv_ + 2 * kq;                // this wave's 64-row slab of every 512-row slice, rows r, r + 1 per lane
                // (round 4: sweeping only the column tiles that hold finished columns -- at m = 20 the second tile is all zeros for four
                //  of the five sweeps -- shortens this section by 20 % and makes the kernel 5 % SLOWER as a whole: 9.96 against 9.46 ms
                //  for 1 600 fits on one box, both code versions of the loop resident; profiles/r04_experiments.md)
#pragma unroll 1
                for (int h = 0; h < RPT * 2; ++h) {
                    const int r0 = FP_NT * (h >> 1) + 32 * (h & 1) + roff;
                    double2 b[4], a[NT16][4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        b[s4] = *reinterpret_cast<const double2 *>(bp + r0 + 8 * s4);
#pragma unroll
                        for (int t = 0; t < NT16; ++t) a[t][s4] = *reinterpret_cast<const double2 *>(ap[t] + r0 + 8 * s4);
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int t = 0; t < NT16; ++t) {
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][s4].x, b[s4].x, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][s4].y, b[s4].y, acc[t], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int t = 0; t < NT16; ++t)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) sTile[(wv_ * NT16 + t) * 256 + rg * 64 + ln] = acc[t][rg];
                __syncthreads();
                FP_STAMP(2);                                       // sweep 1
                if (tid < NT16 * 256) {                            // fixed summation order over the 16 waves
                    const int t = tid >> 8, e = tid & 255, rg = e >> 6, l = e & 63;
                    double s = 0.0;
                    for (int w = 0; w < FP_NW; ++w) s += sTile[(w * NT16 + t) * 256 + e];
                    sC[t * 256 + ((l >> 4) + 4 * rg) * 16 + (l & 15)] = s;     // row = V column within the tile, col = B column
                }
                __syncthreads();
                // compact-WY T of panel pi - 1 (dlarft: T[0:c, c] = -tau_c T[0:c,0:c] (V[:,0:c]' v_c)), V'V, w1 -- wave 0, lane a = row a
                if (tid < 64) {
                    volatile double *Tv = sT;
                    const int a = tid;
                    for (int cc = 0; cc < PW; ++cc) {
                        const int c = c0 - PW + cc;
                        if (c < m) {
                            const double tau = sTau[c];
                            if (a <= c) {
                                const double x = sC[(a >> 4) * 256 + (a & 15) * 16 + PW + cc];
                                sGv[a * KPAD + c] = x; sGv[c * KPAD + a] = x;
                            }
                            if (a < c) {
                                double v = 0.0;
                                for (int b = a; b < c; ++b) v += Tv[a * KPAD + b] * sC[(b >> 4) * 256 + (b & 15) * 16 + PW + cc];
                                Tv[a * KPAD + c] = -tau * v;
                            } else if (a == c) Tv[a * KPAD + c] = tau;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (pi == npan && a < m) sW1[a] = sC[(a >> 4) * 256 + (a & 15) * 16 + 2 * PW];
                }
                __syncthreads();
                if (pi == npan) break;
                // Z = T' W  (c0 x PW),  W = V_prev' P_raw
                if (tid < c0 * PW) {
                    const int a = tid / PW, cc = tid % PW;
                    double v = 0.0;
                    for (int b = 0; b <= a; ++b) v += sT[b * KPAD + a] * sC[(b >> 4) * 256 + (b & 15) * 16 + cc];
                    sZ[a * PW + cc] = v;
                }
                __syncthreads();
                FP_STAMP(3);                                       // tile sum, T columns, Z
                // sweep 2: P = P_raw - V_prev Z
                int tc = tid;
                FP_OPAQUE(tc);
                const double *pc[PW];
#pragma unroll
                for (int cc = 0; cc < PW; ++cc) pc[cc] = (c0 + cc < m) ? scr + (size_t)(c0 + cc) * DPAD : scr_zero;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int row = tc + FP_NT * i;
#pragma unroll
                    for (int cc = 0; cc < PW; ++cc) P[i][cc] = pc[cc][row];
                }
#pragma unroll 1
                for (int a = 0; a < c0; ++a) {
                    double z[PW];
#pragma unroll
                    for (int cc = 0; cc < PW; ++cc) z[cc] = sZ[a * PW + cc];
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const double v = scr[(size_t)a * DPAD + tc + FP_NT * i];
#pragma unroll
                        for (int cc = 0; cc < PW; ++cc) P[i][cc] -= v * z[cc];
                    }
                }
                if (tid < c0) {                                    // rows above the panel's diagonal block belong to R
#pragma unroll
                    for (int cc = 0; cc < PW; ++cc) {
                        if (c0 + cc < m) sR[tid * KPAD + c0 + cc] = P[0][cc];
                        P[0][cc] = 0.0;
                    }
                }
                FP_STAMP(4);                                       // sweep 2
            }
            factor_panel(c0);
        }
        __syncthreads();
        FP_STAMP(3);

        // ---- small algebra.  G = B~'B~ = R'R:  G[c][b] (c, b < j) = Y'alpha Y,  G[j + a][b] = S'Y
        double *X1 = sTile, *X2 = sTile + KPAD * KPAD, *X3 = sTile + 2 * KPAD * KPAD;    // tile staging is free now
        for (int t = tid; t < m * m; t += FP_NT) {
            const int a = t / m, b = t % m, u1 = a < b ? a : b;
            double v = 0.0;
            for (int u = 0; u <= u1; ++u) v += sR[u * KPAD + a] * sR[u * KPAD + b];
            sG[a * KPAD + b] = v;
        }
        __syncthreads();
        // D (m x m)   (src/inverse_hessian.jl:119-130)
        for (int t = tid; t < j * j; t += FP_NT) {
            const int aa = t / j, b = t % j;
            X1[aa * KPAD + b] = (b >= aa) ? sG[(j + aa) * KPAD + b] : 0.0;      // R_ = triu(S'Y)   :119-121
            X2[aa * KPAD + b] = 0.0;
        }
        __syncthreads();
        if (tid < j) {                                   // -R_^{-1}: thread c solves column c by back substitution :122-124
            const int c = tid;
            for (int r = c; r >= 0; --r) {
                double rhs = (r == c) ? -1.0 : 0.0;
                for (int t = r + 1; t <= c; ++t) rhs -= X1[r * KPAD + t] * X2[t * KPAD + c];
                X2[r * KPAD + c] = rhs / X1[r * KPAD + r];
            }
        }
        __syncthreads();
        for (int t = tid; t < j * j; t += FP_NT) {       // M = Y'alpha Y + diag(R_); D12, D21
            const int aa = t / j, b = t % j;
            sD[aa * KPAD + (j + b)] = X2[aa * KPAD + b];
            sD[(j + aa) * KPAD + b] = X2[b * KPAD + aa];
            double v = (aa <= b) ? sG[aa * KPAD + b] : sG[b * KPAD + aa];
            if (aa == b) v += X1[aa * KPAD + aa];
            X3[aa * KPAD + b] = v;
        }
        __syncthreads();
        for (int t = tid; t < j * j; t += FP_NT) {       // M nRinv -> sG (G is no longer needed)
            const int aa = t / j, b = t % j;
            double v = 0.0;
            for (int u = 0; u <= b; ++u) v += X3[aa * KPAD + u] * X2[u * KPAD + b];
            sG[aa * KPAD + b] = v;
        }
        __syncthreads();
        for (int t = tid; t < j * j; t += FP_NT) {       // D22 = nRinv' (M nRinv)
            const int aa = t / j, b = t % j;
            double v = 0.0;
            for (int u = 0; u <= aa; ++u) v += X2[u * KPAD + aa] * sG[u * KPAD + b];
            sD[(j + aa) * KPAD + (j + b)] = v;
        }
        __syncthreads();
        // C = I + R D R' (k x k), V = chol(C).U     (src/woodbury.jl:205)
        for (int t = tid; t < k * m; t += FP_NT) {
            const int aa = t / m, b = t % m;
            double v = 0.0;
            for (int u = aa; u < m; ++u) v += sR[aa * KPAD + u] * sD[u * KPAD + b];
            sG[aa * KPAD + b] = v;
        }
        __syncthreads();
        for (int t = tid; t < k * k; t += FP_NT) {
            const int aa = t / k, b = t % k;
            if (b >= aa) {
                double v = (aa == b) ? 1.0 : 0.0;
                for (int u = b; u < m; ++u) v += sG[aa * KPAD + u] * sR[b * KPAD + u];
                sV[aa * KPAD + b] = v;
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
                    double diag = Vv[c * KPAD + c];
                    for (int t = 0; t < c; ++t) { const double x = Vv[t * KPAD + c]; diag -= x * x; }
                    if (!(diag > 0.0) || !isfinite(diag)) *vst = PFMI_FIT_C_NOT_PD;
                    else { diag = sqrt(diag); Vv[c * KPAD + c] = diag; *vld = *vld + log(diag); }
                }
                __builtin_amdgcn_wave_barrier();
                if (*vst != PFMI_FIT_OK) break;
                if (b > c && b < k) {
                    double v = Vv[c * KPAD + b];
                    for (int t = 0; t < c; ++t) v -= Vv[t * KPAD + c] * Vv[t * KPAD + b];
                    Vv[c * KPAD + b] = v / Vv[c * KPAD + c];
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (b >= k && b < KPAD) Vv[b * KPAD + b] = 1.0;                             // identity padding
        }
        __syncthreads();
        for (int t = tid; t < KPAD * KPAD; t += FP_NT) {
            A.tmat[sm + t] = sT[t]; A.vchol[sm + t] = sV[t]; A.rq[sm + t] = sR[t]; A.dmat[sm + t] = sD[t];
        }
        FP_STAMP(8);                                               // small algebra + Cholesky
        const bool ok = (sStatus == PFMI_FIT_OK);
        // ---- mean through the factor: b = Q'Ug = Ug - V t1 (t1 = T'w1), head <- V_c'V_c head, x = b' - V t2, t2 = T V'b',
        //      V'b' = w1 - (V'V) t1 - V[0:k,:]'(head_b - head')   -- no sweep over the block
        if (ok && tid < 64) {
            const int a = tid;
            volatile double *t1 = sT1, *t2 = sT2, *hb = sHead, *h2 = sHead2, *tmp = sTmp;
            if (a < k) {
                double v = 0.0;
                for (int b = 0; b <= a; ++b) v += sT[b * KPAD + a] * sW1[b];
                t1[a] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (a < k) {                                   // head of b: row a of V is [v_a0 .. v_a,a-1, 1, 0 ..]
                double v = scr_ug[a];
                for (int c = 0; c <= a; ++c) v -= scr[(size_t)c * DPAD + a] * t1[c];
                hb[a] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (a < k) {
                double v = 0.0;
                for (int b = a; b < k; ++b) v += sV[a * KPAD + b] * hb[b];
                tmp[a] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (a < k) {
                double v = 0.0;
                for (int b = 0; b <= a; ++b) v += sV[b * KPAD + a] * tmp[b];
                h2[a] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (a < k) {                                   // w2 -> tmp
                double v = sW1[a];
                for (int b = 0; b < k; ++b) v -= sGv[a * KPAD + b] * t1[b];
                for (int i = a; i < k; ++i) v -= scr[(size_t)a * DPAD + i] * (hb[i] - h2[i]);
                tmp[a] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (a < k) {
                double v = 0.0;
                for (int b = a; b < k; ++b) v += sT[a * KPAD + b] * tmp[b];
                t2[a] = v;
            }
        }
        __syncthreads();
        FP_STAMP(9);                                               // mean (wave 0)
        // ---- last pass: scratch (column-major) -> Vh (row-major [d][KPAD], what the draw kernels read) and mu.  A wave takes 16 rows
        //      at a time: lane (r = lane & 15, q = lane >> 4) reads columns q, q + 4, .. of row r (128 contiguous bytes per column),
        //      the 16 x KPAD tile is transposed through LDS and leaves as one contiguous run of full cache lines.
        {
            int ln = lane, wv_ = wave;
            FP_OPAQUE(ln); FP_OPAQUE(wv_);
            const int r16 = ln & 15, cq = ln >> 4, S = KPAD + 1;
            volatile double *stg = sTile + wv_ * 16 * S;           // 8 x 16 x (KPAD + 1) doubles <= tile staging + C
            const unsigned inv = (65536u + KPAD - 1) / KPAD;       // e / KPAD for e < 2048 (exact for the even KPAD <= 32 used here)
            const int nblk = (d + 15) >> 4;
            const double *colp[8];                                 // this lane's columns q, q + 4, .. (KPAD <= 32)
            double t1c[8], t2c[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c = cq + 4 * t;
                colp[t] = (c < m) ? scr + (size_t)c * DPAD : scr_zero;
                t1c[t] = (c < m) ? sT1[c] : 0.0; t2c[t] = (c < m) ? sT2[c] : 0.0;
            }
            const int nct = (KPAD - cq + 3) >> 2;                  // columns of this lane
            double cur[8], th = 0.0, sq = 0.0, ug = 0.0;
            auto fetch = [&](int b, double (&v)[8], double &th_, double &sq_, double &ug_) {
                const int row = 16 * b + r16, rl = row < d ? row : d - 1;
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = colp[t][row];
                th_ = theta_p[rl]; sq_ = sqa[rl]; ug_ = scr_ug[row];
            };
            if (wv_ < nblk) fetch(wv_, cur, th, sq, ug);
#pragma unroll 1
            for (int b = wv_; b < nblk; b += FP_NW) {
                const int row = 16 * b + r16;                      // < DPAD; scratch rows >= d hold zeros
                double nxt[8], th2 = 0.0, sq2 = 0.0, ug2 = 0.0;
                const int bn = (b + FP_NW < nblk) ? b + FP_NW : b; // prefetch the next tile while this one goes through LDS
                fetch(bn, nxt, th2, sq2, ug2);
                double x = 0.0, y = 0.0;                           // V[row,:].t1, V[row,:].t2
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t < nct) stg[r16 * S + cq + 4 * t] = cur[t];
                    x += cur[t] * t1c[t]; y += cur[t] * t2c[t];
                }
                x = pf_sum_q(x); y = pf_sum_q(y);
                if (cq == 0 && row < d) {
                    const double bp = (row < k) ? sHead2[row] : ug - x;
                    mu[row] = ok ? th + sq * (bp - y) : NAN;
                }
                __builtin_amdgcn_wave_barrier();
                const int nrow = (d - 16 * b < 16) ? d - 16 * b : 16, nel = nrow * KPAD;
                double *o = Vh + (size_t)16 * b * KPAD;
                for (int e = ln; e < nel; e += 64) {
                    const int r = (int)(((unsigned)e * inv) >> 16), c = e - r * KPAD;
                    const double v = stg[r * S + c];
                    o[e] = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
                th = th2; sq = sq2; ug = ug2;
            }
        }
        FP_STAMP(10);                                              // last pass
        if (tid == 0) {
            A.status[p] = ok ? PFMI_FIT_OK : sStatus;
            A.logdet[p] = ok ? 2.0 * (ldu + sLogdetV) : NAN;
        }
    }
#if FP_PROF
    if (blockIdx.x == 0 && tid == 0)
        printf("FP_PROF fits %d (100 MHz ticks per fit): fetch %lld  phaseA %lld  sweep1 %lld  tile/T/Z %lld  sweep2 %lld  panelQR %lld  writeback %lld  "
               "barrier-before-sweep1 %lld  small+chol %lld  mean %lld  lastpass %lld  loop %lld\n", nfit - 1, prof[0] / (nfit - 1), prof[1] / (nfit - 1),
               prof[2] / (nfit - 1), prof[3] / (nfit - 1), prof[4] / (nfit - 1), prof[5] / (nfit - 1), prof[6] / (nfit - 1), prof[7] / (nfit - 1),
               prof[8] / (nfit - 1), prof[9] / (nfit - 1), prof[10] / (nfit - 1), prof[11] / (nfit - 1));
#endif
}

// ---------------------------------------------------------------------------------------------------
template <int NT16, int RPT, int PW>
static int32_t launch_panel_t(pfmi_ctx *c, const FitArgs &a, int ncu) {
    const int KPAD = c->kpad;
    const int lds = fp_lds_doubles(KPAD) * (int)sizeof(double);
    auto kern = pf_fit_panel_kernel<NT16, RPT, PW>;
    PF_TRY(pf_raise_lds_limit(c, reinterpret_cast<const void *>(kern), fp_lds_doubles(16 * NT16) * (int)sizeof(double)));   // largest KPAD of this tile count
    int occ = 1;
    PF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, FP_NT, lds));
    if (occ < 1) occ = 1;
    int64_t slots = (int64_t)ncu * occ;
    if (const char *g = pf_debug_get("PFMI_FIT_PANEL_GRID")) { const int v = atoi(g); if (v > 0) slots = v; }   // experiment hook: resident workgroups
    const int grid = slots < a.P ? (int)slots : (int)a.P;
    const size_t scr_bytes = (size_t)grid * (KPAD + 4) * (size_t)(FP_NT * RPT) * sizeof(double);
    PF_TRY(c->fit_scratch.ensure(scr_bytes + 256));
    int *counter = reinterpret_cast<int *>(c->fit_scratch.as<char>() + scr_bytes);
    PF_HIP(hipMemsetAsync(counter, 0, 8 * sizeof(int), c->stream));               // one work counter per XCD
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FP_NT), lds, c->stream, a, KPAD, c->fit_scratch.as<double>(), counter);
    return PFMI_OK;
}

template <int NT16>
static int32_t launch_panel_k(pfmi_ctx *c, const FitArgs &a, int ncu) {
    const int rpt = (a.d + FP_NT - 1) / FP_NT;
    if (rpt <= 5) return launch_panel_t<NT16, 5, 4>(c, a, ncu);
    if (rpt <= 10) return launch_panel_t<NT16, 10, 4>(c, a, ncu);
    if (rpt <= 20) return launch_panel_t<NT16, 20, 4>(c, a, ncu);
    return launch_panel_t<NT16, 32, 2>(c, a, ncu);
}

// returns PFMI_OK and sets *handled when the panel kernel took the launch
int32_t pf_launch_fit_panel(pfmi_ctx *c, const FitArgs &a, bool *handled) {
    *handled = false;
    if (a.d <= 1024 || a.d > 32 * FP_NT || c->kpad < 8 || c->kpad > 32 || (c->kpad & 1)) return PFMI_OK;
    int ncu = 0;
    PF_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device));
    if (c->kpad <= 16) PF_TRY(launch_panel_k<1>(c, a, ncu));
    else PF_TRY(launch_panel_k<2>(c, a, ncu));
    *handled = true;
    return PFMI_OK;
}
