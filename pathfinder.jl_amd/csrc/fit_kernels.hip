// fit_kernels.hip -- batched L-BFGS inverse-Hessian reconstruction + Woodbury factorisation (gfx950).
//
//   pf_history_kernel : one workgroup per PATH.  Sequential walk over the trace
//                       (lbfgs_inverse_hessians, reference src/inverse_hessian.jl:25-66): s, y,
//                       curvature test, ring-buffer bookkeeping, gilbert_init recurrence (:5-10).
//   pf_fit_kernel     : one workgroup per trace POINT (= per fitted MvNormal), all points of all
//                       paths in one launch.  Byrd compact form (src/inverse_hessian.jl:98-133),
//                       pdfactorize (src/woodbury.jl:201-207): U = sqrt(alpha), Householder QR of
//                       U'\B in LAPACK convention (so that draws match the reference's lmul!(Q, .)),
//                       compact-WY T, C = I + R D R', Cholesky V, logdet (:76-80) and
//                       mu = theta + Sigma*grad through the factor (src/mvnormal.jl:14-21).
//
// Data layout: the scaled history block B~ = U'\[alpha.Y  S] lives row-major [d][KPAD] in HBM and is
// factored in place, so one row (KPAD doubles, 96 B at J = 6) is one contiguous coalesced read
// and is what the draw kernel later scalar-loads per row.  Column padding (zeros) sits at the END of
// the column list only (a zero column in the middle would shift later pivots).
#include "pfmi_common.h"
#include "fit_args.h"
#include <limits.h>
#include <stdlib.h>

#ifndef FITREG_ABLATE
#define FITREG_ABLATE 0                // timing experiments only: 1 no Cholesky, 2 no mean
#endif
#ifndef FITREG_XCD
#define FITREG_XCD 1                   // 0: point = workgroup index (A/B of the XCD-aware order)
#endif
#ifndef FITREG_PROF
#define FITREG_PROF 0                  // 1: one workgroup in 512 prints the 100 MHz ticks of its sections (experiments only)
#endif
#if FITREG_PROF
#define FR_STAMP(k) do { if ((blockIdx.x & 511) == 7 && threadIdx.x == 0) { const long long t_ = wall_clock64(); prof[k] = t_ - tlast; tlast = t_; } } while (0)
#else
#define FR_STAMP(k) do { } while (0)
#endif
#ifndef HIST_PROF
#define HIST_PROF 0                     // 1: section cycle counters in pf_history_kernel (experiment builds)
#endif
#define HIST_THREADS 256
#define FIT_THREADS 256

// ---------------------------------------------------------------------------------------------------
// 1024 threads per path; each thread keeps its EPT coordinates of alpha, theta_l, grad_l in registers and prefetches
// point l+1 while the four dot products of step l are block-reduced, so one trace iteration costs one reduction
// instead of a round trip through HBM.
template <int EPT, int HIST_NT>
__global__ __launch_bounds__(HIST_NT) void pf_history_kernel(
    int d, int J, double eps, const int64_t *__restrict__ off, const double *__restrict__ theta,
    const double *__restrict__ grad, double *__restrict__ alpha_all, int *__restrict__ hist_len,
    int *__restrict__ hist_src, int *__restrict__ n_rej, int *__restrict__ acc_list, HistSeg sg) {
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = off[k];
    // the steps of THIS launch: b + 1 .. L (HistSeg above; the packed route: b = 0, L = the path's last point)
    const int Ltot = sg.npts ? sg.npts[k] - 1 : (int)(off[k + 1] - p0 - 1);
    const int L = Ltot < sg.l_end - 1 ? Ltot : sg.l_end - 1;
    const int b = sg.l_begin > 0 ? sg.l_begin - 1 : 0;
    if (sg.l_begin > 0 && b >= L) return;                   // the path ended before this segment
    __shared__ double red[2 * 4 * (HIST_NT / 64)];          // two halves: one barrier per iteration (pf_block_sum_pp)
    int flip = 0;
    // ring bookkeeping (src/inverse_hessian.jl:49-52, 105): every thread sees the same `accept` (the block sums are bit-identical
    // in every lane).  During the walk thread 0 only appends the accepted step to a list and stores the running count (two plain
    // stores, nothing to wait for); the ring contents of every point -- the last min(count, J) accepted steps, oldest first, which
    // is what mod1 / hist_inds produce -- are expanded from that list by all threads after the walk.  (Until round 2 a keeper
    // thread rebuilt the ring row by row inside the loop: up to J dependent LDS reads + stores per step on the critical wave.)
    int n_acc = 0;
    int *const acc = acc_list + p0;                          // acc[k] = source index (l - 1) of the k-th accepted update of this path

    // The walk is sequential in l and every iteration needs two fresh rows (theta_l, grad_l) that nobody has touched before: with
    // the loads issued one iteration ahead the iteration time WAS the DRAM latency (3 us at config 3).  The rows of NS consecutive
    // points live in NS register sets used round-robin (set p % NS holds point p): step l reads sets (l - 1) % NS and l % NS and
    // then refills the older one with point l - 1 + NS, i.e. NS - 2 steps ahead of its first use.  The loop is unrolled NS times so
    // that every set index is static and NO value is ever copied between sets, and its body is branch-free: round 2's first version
    // (predicated loads, a conditional per unrolled step, t0 <- t1 copies) made the compiler wait with vmcnt(0) / copy registers of
    // pending loads at every join, so each step still paid a full memory round trip.
    // (EPT > 6: three sets -- 1024 threads have 128 VGPRs.)
    constexpr int NS = EPT <= 6 ? 4 : 3;
    double al[EPT], ial[EPT], tq[NS][EPT], gq[NS][EPT];
    // unconditional loads from clamped addresses; points beyond L are never consumed, coordinates beyond d are zeroed where s and
    // y are formed (masking here would make the load's first use immediate)
    auto load_point = [&](const int pt, double (&tt)[EPT], double (&gg)[EPT]) {
        const size_t row = (size_t)(p0 + (pt <= L ? pt : L)) * d;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + HIST_NT * e, ic = i < d ? i : d - 1;
            tt[e] = theta[row + ic]; gg[e] = grad[row + ic];
        }
    };
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + HIST_NT * e, ic = i < d ? i : d - 1;
        if (sg.l_begin > 0) {                                                  // resume: alpha of point b as stored, the CARRIED 1 / alpha from the state
            al[e] = alpha_all[(size_t)(p0 + b) * d + ic]; ial[e] = sg.ial_state[(size_t)k * d + ic];
        } else {
            al[e] = 1.0; ial[e] = 1.0;                                         // H0 = I  (:38-39)
            if (i < d) alpha_all[(size_t)p0 * d + i] = 1.0;
        }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) load_point(b + q, tq[q], gq[q]);                // points b .. b + NS - 1
    if (sg.l_begin > 0) n_acc = sg.nacc_state[k];
    else if (tid == 0) hist_len[p0] = 0;
    // one trace step: (t0, g0) = point l - 1, (t1, g1) = point l; afterwards the set of point l - 1 is refilled
#if HIST_PROF
    long long hp[5] = {0, 0, 0, 0, 0}, hp_last = clock64();
#define HP_STAMP(k) do { const long long t_ = clock64(); hp[k] += t_ - hp_last; hp_last = t_; } while (0)
#else
#define HP_STAMP(k) do { } while (0)
#endif
    auto step = [&](const int l, double (&t0)[EPT], double (&g0)[EPT], const double (&t1)[EPT], const double (&g1)[EPT]) {
        HP_STAMP(4);
        double v[4] = {0.0, 0.0, 0.0, 0.0};   // y.s, y.y, y'diag(a)y, s'diag(1/a)s
        double sv[EPT], yv[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const bool in = tid + HIST_NT * e < d;
            const double s = in ? t1[e] - t0[e] : 0.0, y = in ? g0[e] - g1[e] : 0.0;   // :45-46
            sv[e] = s; yv[e] = y;
            v[0] += y * s;
            v[1] += y * y;
            v[2] += y * al[e] * y;
            v[3] += s * ial[e] * s;
        }
        load_point(l - 1 + NS, t0, g0);                                         // refill: first needed NS - 2 steps from now
        HP_STAMP(0);
        pf_block_sum_mv<4, 4, (HIST_NT <= 256 ? HIST_NT / 64 : 0)>(v, red, flip);   // static wave count only for <= 4 waves (16 waves: register pressure)
        HP_STAMP(1);
        const bool accept = v[0] > eps * v[1];                                  // :47
        if (accept && sg.hinit == 1) {                                          // Hinit = (alpha, s, y) -> fill(y's / y'y)  (test/inverse_hessian.jl:49)
            const double an = v[0] / v[1], ian = 1.0 / an;
#pragma unroll
            for (int e = 0; e < EPT; ++e) { al[e] = an; ial[e] = ian; }
            n_acc += 1;
        } else if (accept) {                                                    // gilbert_init :5-10
            const double a = v[2], b = v[0], c = v[3], aoc = a / c, rb = 1.0 / b;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const double sa = sv[e] * ial[e];
                const double x = a * ial[e] + yv[e] * yv[e] - aoc * sa * sa;
                // the reference's formula (alpha' = b / (a/alpha + y^2 - (a/c)(s/alpha)^2)) with 1/alpha CARRIED as x * (1/b) instead of
                // recomputed by a division: a / alpha -> a * ial and s / alpha -> s * ial differ from the reference's quotients in the last
                // bit or two per accepted step.  The oracle divides like the reference; tests bound the drift over full-length
                // traces (alpha and everything downstream <= 1e-10, status / j_eff array-equal: test_config3_whole_trace_...)
                al[e] = b / x;
                ial[e] = x * rb;                                                // 1 / alpha for the next step without a second division
            }
            n_acc += 1;
        }
        HP_STAMP(2);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + HIST_NT * e;
            if (i < d) alpha_all[(size_t)(p0 + l) * d + i] = al[e];
        }
        if (tid == 0) {
            if (accept) acc[n_acc - 1] = l - 1;
            hist_len[p0 + l] = n_acc < J ? n_acc : J;                           // r_eff = max r_ind so far (:49-50)
            hist_src[(size_t)(p0 + l) * J] = n_acc;                             // parked in the row's first slot until the expansion below
        }
        HP_STAMP(3);
    };
    int l = b + 1;
    for (; l + NS - 1 <= L; l += NS) {                                          // :43
#pragma unroll
        for (int q = 0; q < NS; ++q) step(l + q, tq[q], gq[q], tq[(q + 1) % NS], gq[(q + 1) % NS]);
    }
    const int rem = L - l + 1;                                                  // 0 .. NS - 1 steps left
#pragma unroll
    for (int q = 0; q < NS - 1; ++q)
        if (q < rem) step(l + q, tq[q], gq[q], tq[(q + 1) % NS], gq[(q + 1) % NS]);
#if HIST_PROF
    if (k == 0 && (tid & 63) == 0)
        printf("HIST_PROF wave %d: %d steps, cycles per step: dots+loads %lld block sum %lld update %lld stores %lld loop %lld\n", tid >> 6, L,
               hp[0] / L, hp[1] / L, hp[2] / L, hp[3] / L, hp[4] / L);
#endif
    if (tid == 0) n_rej[k] = L > 0 ? L - n_acc : 0;                             // :57  (a later segment of the same path overwrites it)
    if (sg.ial_state) {                                                         // hand the recurrence on to the next segment
#pragma unroll
        for (int e = 0; e < EPT; ++e) { const int i = tid + HIST_NT * e; if (i < d) sg.ial_state[(size_t)k * d + i] = ial[e]; }
        if (tid == 0) sg.nacc_state[k] = n_acc;
    }
    // ---- hist_inds (:105) of every point from the accepted list: row l = the last min(n_acc(l), J) accepted steps, oldest first
    __threadfence_block();
    __syncthreads();
    for (int q = b + 1 + tid; q <= L; q += HIST_NT) {
        int *row = hist_src + (size_t)(p0 + q) * J;
        const int na = row[0], re = na < J ? na : J;
        int first = 0;
        for (int c = 0; c < re; ++c) {
            const int v = acc[na - re + c];
            if (c == 0) first = v; else row[c] = v;
        }
        row[0] = re > 0 ? first : 0;
    }
}

// Lean variant of the walk for 2048 < d <= 10 240 (1024 threads, EPT = 3 .. 10, 128 registers per thread).  The prefetching kernel
// above keeps alpha, 1 / alpha and three register sets of rows: at EPT = 10 that is 273 spilled registers and 25 us per trace step
// (200 steps of config 5's shape = 4.9 ms, 4.5 % of the step).  Here a thread carries alpha, the current point's rows and, in the
// registers of the previous point, s and y in place (4 + 1 vectors); 1 / alpha lives in LDS (<= 80 KB); the next point is loaded
// when the step needs it.  Same arithmetic, same reduction, same thread -> element map: bit-identical to the kernel above
// (tests/probes/history_ab.py: 3.4 against 4.9 ms at d = 10^4, 1.1 against 2.3 ms at d = 6000, 0.92 against 0.97 ms at d = 3000).
template <int EPT>
__global__ __launch_bounds__(1024) void pf_history_lean_kernel(
    int d, int J, double eps, const int64_t *__restrict__ off, const double *__restrict__ theta,
    const double *__restrict__ grad, double *__restrict__ alpha_all, int *__restrict__ hist_len,
    int *__restrict__ hist_src, int *__restrict__ n_rej, int *__restrict__ acc_list, HistSeg sg) {
    constexpr int HIST_NT = 1024;
    extern __shared__ double hl_ial[];                        // [EPT][1024]: 1 / alpha, element e of thread tid at e * 1024 + tid
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = off[k];
    const int Ltot = sg.npts ? sg.npts[k] - 1 : (int)(off[k + 1] - p0 - 1);
    const int L = Ltot < sg.l_end - 1 ? Ltot : sg.l_end - 1;
    const int b = sg.l_begin > 0 ? sg.l_begin - 1 : 0;      // (HistSeg: the steps of this launch are b + 1 .. L)
    if (sg.l_begin > 0 && b >= L) return;
    __shared__ double red[2 * 4 * (HIST_NT / 64)];
    int flip = 0;
    int n_acc = 0;
    int *const acc = acc_list + p0;
    // EPT >= 9 (d > 8192): the previous point's rows are RE-READ (they were fetched one step earlier: L2 hits) instead of carried --
    // 5 EPT doubles of state did not fit the 128 registers of a 1024-thread workgroup (24 spilled, 17 us per step at d = 10^4
    // against 6 us at d = 7500; round 4)
    constexpr bool REREAD = EPT >= 9;
    double al[EPT], tc[REREAD ? 1 : EPT], gc[REREAD ? 1 : EPT], sv[EPT], yv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + HIST_NT * e, ic = i < d ? i : d - 1;
        if (sg.l_begin > 0) {
            al[e] = alpha_all[(size_t)(p0 + b) * d + ic]; hl_ial[e * HIST_NT + tid] = sg.ial_state[(size_t)k * d + ic];
        } else {
            al[e] = 1.0; hl_ial[e * HIST_NT + tid] = 1.0;                      // H0 = I  (:38-39)
            if (i < d) alpha_all[(size_t)p0 * d + i] = 1.0;
        }
        if (!REREAD) { tc[REREAD ? 0 : e] = theta[(size_t)(p0 + b) * d + ic]; gc[REREAD ? 0 : e] = grad[(size_t)(p0 + b) * d + ic]; }
    }
    if (sg.l_begin > 0) n_acc = sg.nacc_state[k];
    else if (tid == 0) hist_len[p0] = 0;
    for (int l = b + 1; l <= L; ++l) {                                          // :43
        const size_t row = (size_t)(p0 + l) * d;
        double v[4] = {0.0, 0.0, 0.0, 0.0};   // y.s, y.y, y'diag(a)y, s'diag(1/a)s
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + HIST_NT * e, ic = i < d ? i : d - 1;
            sv[e] = theta[row + ic]; yv[e] = grad[row + ic];
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const bool in = tid + HIST_NT * e < d;
            const double t1 = sv[e], g1 = yv[e];
            double t0, g0;
            if (REREAD) {
                const int i = tid + HIST_NT * e, ic = i < d ? i : d - 1;
                t0 = theta[row - d + ic]; g0 = grad[row - d + ic];
            } else { t0 = tc[REREAD ? 0 : e]; g0 = gc[REREAD ? 0 : e]; }
            const double s = in ? t1 - t0 : 0.0, y = in ? g0 - g1 : 0.0;         // :45-46
            if (!REREAD) { tc[REREAD ? 0 : e] = t1; gc[REREAD ? 0 : e] = g1; }
            sv[e] = s; yv[e] = y;
            const double ia = hl_ial[e * HIST_NT + tid];
            v[0] += y * s;
            v[1] += y * y;
            v[2] += y * al[e] * y;
            v[3] += s * ia * s;
        }
        pf_block_sum_mv<4, 4, 0>(v, red, flip);
        const bool accept = v[0] > eps * v[1];                                  // :47
        if (accept && sg.hinit == 1) {                                          // Hinit = fill(y's / y'y)  (test/inverse_hessian.jl:49)
            const double an = v[0] / v[1], ian = 1.0 / an;
#pragma unroll
            for (int e = 0; e < EPT; ++e) { al[e] = an; hl_ial[e * HIST_NT + tid] = ian; }
            n_acc += 1;
        } else if (accept) {                                                    // gilbert_init :5-10
            const double a = v[2], b = v[0], c = v[3], aoc = a / c, rb = 1.0 / b;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const double ia = hl_ial[e * HIST_NT + tid];
                const double sa = sv[e] * ia;
                const double x = a * ia + yv[e] * yv[e] - aoc * sa * sa;
                al[e] = b / x;
                hl_ial[e * HIST_NT + tid] = x * rb;
            }
            n_acc += 1;
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + HIST_NT * e;
            if (i < d) alpha_all[(size_t)(p0 + l) * d + i] = al[e];
        }
        if (tid == 0) {
            if (accept) acc[n_acc - 1] = l - 1;
            hist_len[p0 + l] = n_acc < J ? n_acc : J;
            hist_src[(size_t)(p0 + l) * J] = n_acc;
        }
    }
    if (tid == 0) n_rej[k] = L > 0 ? L - n_acc : 0;                             // :57
    if (sg.ial_state) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) { const int i = tid + HIST_NT * e; if (i < d) sg.ial_state[(size_t)k * d + i] = hl_ial[e * HIST_NT + tid]; }
        if (tid == 0) sg.nacc_state[k] = n_acc;
    }
    __threadfence_block();
    __syncthreads();
    for (int q = b + 1 + tid; q <= L; q += HIST_NT) {                           // hist_inds (:105), as above
        int *row = hist_src + (size_t)(p0 + q) * J;
        const int na = row[0], re = na < J ? na : J;
        int first = 0;
        for (int c = 0; c < re; ++c) {
            const int v = acc[na - re + c];
            if (c == 0) first = v; else row[c] = v;
        }
        row[0] = re > 0 ? first : 0;
    }
}

// Memory-resident walk for ANY d (the default beyond 16 384 coordinates, where neither register kernel applies; PFMI_HISTORY_KERNEL=mem
// selects it at every d for the tests).  Nothing is carried in registers: alpha of the previous point is read from its row of alpha_all,
// the rows of the two points are swept twice per step (dot products, then the gilbert_init update; the second sweep hits L2).  The
// divisions are the reference's own (a / alpha, s / alpha: src/inverse_hessian.jl:5-10), not the carried reciprocal of the register kernels.
__global__ __launch_bounds__(1024) void pf_history_mem_kernel(
    int d, int J, double eps, const int64_t *__restrict__ off, const double *__restrict__ theta,
    const double *__restrict__ grad, double *__restrict__ alpha_all, int *__restrict__ hist_len,
    int *__restrict__ hist_src, int *__restrict__ n_rej, int *__restrict__ acc_list, HistSeg sg) {
    constexpr int HIST_NT = 1024;
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = off[k];
    const int Ltot = sg.npts ? sg.npts[k] - 1 : (int)(off[k + 1] - p0 - 1);
    const int L = Ltot < sg.l_end - 1 ? Ltot : sg.l_end - 1;
    const int b = sg.l_begin > 0 ? sg.l_begin - 1 : 0;      // (HistSeg: the steps of this launch are b + 1 .. L)
    if (sg.l_begin > 0 && b >= L) return;
    __shared__ double red[2 * 4 * (HIST_NT / 64)];
    int flip = 0;
    int n_acc = 0;
    int *const acc = acc_list + p0;
    if (sg.l_begin > 0) n_acc = sg.nacc_state[k];
    else {
        for (int i = tid; i < d; i += HIST_NT) alpha_all[(size_t)p0 * d + i] = 1.0;  // H0 = I  (:38-39)
        if (tid == 0) hist_len[p0] = 0;
    }
    for (int l = b + 1; l <= L; ++l) {                                              // :43
        const size_t row = (size_t)(p0 + l) * d;
        double v[4] = {0.0, 0.0, 0.0, 0.0};   // y.s, y.y, y'diag(a)y, s'diag(1/a)s
        for (int i = tid; i < d; i += HIST_NT) {                                    // (a thread reads back only what it wrote itself: no barrier)
            const double s = theta[row + i] - theta[row - d + i], y = grad[row - d + i] - grad[row + i];   // :45-46
            const double al = alpha_all[row - d + i];
            v[0] += y * s;
            v[1] += y * y;
            v[2] += y * al * y;
            v[3] += s * s / al;
        }
        pf_block_sum_mv<4, 4, 0>(v, red, flip);
        const bool accept = v[0] > eps * v[1];                                      // :47
        const double a = v[2], b = v[0], c = v[3], aoc = a / c;
        for (int i = tid; i < d; i += HIST_NT) {
            const double al = alpha_all[row - d + i];
            double an = al;
            if (accept) {                                                           // gilbert_init :5-10
                const double s = theta[row + i] - theta[row - d + i], y = grad[row - d + i] - grad[row + i];
                const double sa = s / al;
                an = sg.hinit == 1 ? v[0] / v[1] : b / (a / al + y * y - aoc * sa * sa);       // (hinit 1: fill(y's / y'y), test/inverse_hessian.jl:49)
            }
            alpha_all[row + i] = an;
        }
        if (accept) n_acc += 1;
        if (tid == 0) {
            if (accept) acc[n_acc - 1] = l - 1;
            hist_len[p0 + l] = n_acc < J ? n_acc : J;
            hist_src[(size_t)(p0 + l) * J] = n_acc;
        }
    }
    if (tid == 0) n_rej[k] = L > 0 ? L - n_acc : 0;                                 // :57
    if (sg.nacc_state && tid == 0) sg.nacc_state[k] = n_acc;
    __threadfence_block();
    __syncthreads();
    for (int q = b + 1 + tid; q <= L; q += HIST_NT) {                               // hist_inds (:105), as above
        int *rowp = hist_src + (size_t)(p0 + q) * J;
        const int na = rowp[0], re = na < J ? na : J;
        int first = 0;
        for (int cc = 0; cc < re; ++cc) {
            const int vv = acc[na - re + cc];
            if (cc == 0) first = vv; else rowp[cc] = vv;
        }
        rowp[0] = re > 0 ? first : 0;
    }
}

// ---------------------------------------------------------------------------------------------------
#include "fit_args.h"

// acc[cc] = sum_{i >= i0} M[i][c] * M[i][cc]  for all cc (block-wide, every thread gets the result)
template <int KPAD>
__device__ __forceinline__ void pf_multi_dot(const double *M, int d, int i0, int c, double (&acc)[KPAD],
                                             double *red) {
#pragma unroll
    for (int cc = 0; cc < KPAD; ++cc) acc[cc] = 0.0;
    for (int i = i0 + (int)threadIdx.x; i < d; i += (int)blockDim.x) {
        const double *row = M + (size_t)i * KPAD;
        const double xc = row[c];
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) acc[cc] += xc * row[cc];
    }
    pf_block_sum<KPAD>(acc, red);
}
// acc[cc] = sum_i x_i * M[i][cc], x_i produced by f(i)
template <int KPAD, typename F>
__device__ __forceinline__ void pf_vec_dot(const double *M, int d, F f, double (&acc)[KPAD], double *red) {
#pragma unroll
    for (int cc = 0; cc < KPAD; ++cc) acc[cc] = 0.0;
    for (int i = (int)threadIdx.x; i < d; i += (int)blockDim.x) {
        const double *row = M + (size_t)i * KPAD;
        const double x = f(i, row);
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) acc[cc] += x * row[cc];
    }
    pf_block_sum<KPAD>(acc, red);
}

template <int KPAD>
__global__ __launch_bounds__(FIT_THREADS) void pf_fit_kernel(FitArgs A) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int64_t p;
    if (!pf_fit_point(A, blockIdx.x, p)) { if (tid == 0) A.status[p] = PFMI_FIT_ABSENT; return; }
    const int d = A.d, J = A.J;
    const int path = A.path_of[p];
    const int64_t p0 = A.off[path];
    const int j = A.hist_len[p], m = 2 * j, k = d < m ? d : m;
    const double *alpha = A.alpha_all + (size_t)p * d;
    double *Vh = A.vh + (size_t)p * d * KPAD;
    double *sqa = A.sqrt_alpha + (size_t)p * d;
    double *mu = A.mu + (size_t)p * d;
    const double *theta_p = A.theta + (size_t)p * d, *grad_p = A.grad + (size_t)p * d;

    __shared__ double red[(FIT_THREADS / 64) * KPAD];
    __shared__ double sP[KPAD], sW[KPAD], sHead[KPAD], sTau[KPAD];
    // the five small matrices: LDS up to KPAD = 32; KPAD = 64 (history_length 17 .. 32, the slow-but-correct route): 5 x 32 KB do not fit, the
    // kernel works directly in the fit's OUTPUT blocks (T, V, R, D of this point in global memory) and a per-workgroup scratch for G
    constexpr bool BIG = KPAD > 32;
    constexpr int MSZ = BIG ? 1 : KPAD * KPAD;
    __shared__ double sD_[MSZ], sR_[MSZ], sT_[MSZ], sV_[MSZ], sG_[MSZ];
    double *sD = sD_, *sR = sR_, *sT = sT_, *sV = sV_, *sG = sG_;
    if constexpr (BIG) {
        const size_t sm0 = (size_t)p * KPAD * KPAD;
        sD = A.dmat + sm0; sR = A.rq + sm0; sT = A.tmat + sm0; sV = A.vchol + sm0;
        sG = A.big + (size_t)blockIdx.x * KPAD * KPAD;
    }
    __shared__ double sScal, sBeta, sLogdetV;
    __shared__ int sStatus;

    // ---- U = sqrt(alpha), A positive definite?  (src/woodbury.jl:202-203)
    double bad = 0.0, ldu = 0.0;
    for (int i = tid; i < d; i += nt) {
        const double al = alpha[i];
        if (!(al > 0.0) || !isfinite(al)) bad = 1.0;
        const double s = sqrt(al);
        sqa[i] = s;
        ldu += log(s);
    }
    {
        double v[2] = {bad, ldu};
        pf_block_sum<2>(v, red);
        bad = v[0]; ldu = v[1];
    }
    if (tid == 0) sStatus = (bad > 0.0) ? PFMI_FIT_A_NOT_PD : PFMI_FIT_OK;
    for (int t = tid; t < KPAD * KPAD; t += nt) { sD[t] = 0.0; sR[t] = 0.0; sT[t] = 0.0; sV[t] = 0.0; sG[t] = 0.0; }
    __syncthreads();
    if (sStatus != PFMI_FIT_OK) {
        for (int i = tid; i < d; i += nt) {
            mu[i] = NAN;
            for (int c = 0; c < KPAD; ++c) Vh[(size_t)i * KPAD + c] = 0.0;
        }
        for (int t = tid; t < KPAD * KPAD; t += nt) {
            A.tmat[(size_t)p * KPAD * KPAD + t] = 0.0; A.vchol[(size_t)p * KPAD * KPAD + t] = 0.0;
            A.rq[(size_t)p * KPAD * KPAD + t] = 0.0; A.dmat[(size_t)p * KPAD * KPAD + t] = 0.0;
        }
        if (tid == 0) { A.status[p] = sStatus; A.logdet[p] = NAN; }
        return;
    }

    // ---- rows of B~ = U' \ [alpha.Y  S]   (src/inverse_hessian.jl:117-118, src/woodbury.jl:204)
    for (int i = tid; i < d; i += nt) {
        double *row = Vh + (size_t)i * KPAD;
        const double al = alpha[i], sa = sqa[i];
        for (int c = 0; c < j; ++c) {
            const int src = A.hist_src[(size_t)p * J + c];
            const size_t q0 = (size_t)(p0 + src) * d + i, q1 = (size_t)(p0 + src + 1) * d + i;
            const double y = A.grad[q0] - A.grad[q1];     // y = grad_l - grad_{l+1}   :46
            const double s = A.theta[q1] - A.theta[q0];   // s = theta_{l+1} - theta_l :45
            row[c] = (al * y) / sa;
            row[j + c] = s / sa;
        }
        for (int c = m; c < KPAD; ++c) row[c] = 0.0;
    }
    __threadfence_block();
    __syncthreads();

    double acc[KPAD];
    // ---- Gram matrix G = B~'B~ : G[c][b] (c < j, b < j) = Y'alpha Y ; G[j+a][b] = S'Y
    // four Gram rows per sweep over the block (the kernel is bound by these sweeps at large d)
    constexpr int GR = 4;
    for (int c = 0; c < m; c += GR) {
        double accg[GR][KPAD];
#pragma unroll
        for (int g = 0; g < GR; ++g)
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) accg[g][cc] = 0.0;
        for (int i = tid; i < d; i += nt) {
            const double *row = Vh + (size_t)i * KPAD;
            double r[KPAD], x[GR];
#pragma unroll
            for (int g = 0; g < GR; ++g) x[g] = 0.0;
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) {
                r[cc] = row[cc];
#pragma unroll
                for (int g = 0; g < GR; ++g) if (cc == c + g) x[g] = r[cc];
            }
#pragma unroll
            for (int g = 0; g < GR; ++g)
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) accg[g][cc] += x[g] * r[cc];
        }
#pragma unroll
        for (int g = 0; g < GR; ++g) {
            pf_block_sum<KPAD>(accg[g], red);
            if (tid == 0 && c + g < m) {
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) sG[(c + g) * KPAD + cc] = accg[g][cc];
            }
        }
    }
    __syncthreads();
    // ---- D (m x m)   (src/inverse_hessian.jl:119-130), thread 0; sT/sV used as scratch
    if (tid == 0 && j > 0) {
        double *R = sT, *nRinv = sV, *Mm = sR;   // j x j each, row-major with stride KPAD
        for (int a = 0; a < j; ++a)
            for (int b = 0; b < j; ++b) {
                R[a * KPAD + b] = (b >= a) ? sG[(j + a) * KPAD + b] : 0.0;     // triu(S'Y)   :119-121
                nRinv[a * KPAD + b] = 0.0;
            }
        for (int c = 0; c < j; ++c)                                              // -R^{-1}     :122-124
            for (int r = c; r >= 0; --r) {
                double rhs = (r == c) ? -1.0 : 0.0;
                for (int t = r + 1; t <= c; ++t) rhs -= R[r * KPAD + t] * nRinv[t * KPAD + c];
                nRinv[r * KPAD + c] = rhs / R[r * KPAD + r];
            }
        for (int a = 0; a < j; ++a)
            for (int b = 0; b < j; ++b) {
                sD[a * KPAD + (j + b)] = nRinv[a * KPAD + b];                    // D12         :122
                sD[(j + a) * KPAD + b] = nRinv[b * KPAD + a];                    // D21         :125
                double v = (a <= b) ? sG[a * KPAD + b] : sG[b * KPAD + a];       // Y'alpha Y   :127-128
                if (a == b) v += R[a * KPAD + a];                                // + diag(R)   :126
                Mm[a * KPAD + b] = v;
            }
        // D22 = nRinv' (M nRinv)                                                   :129-130
        for (int a = 0; a < j; ++a)
            for (int b = 0; b < j; ++b) {
                double v = 0.0;
                for (int t = 0; t <= b; ++t) v += Mm[a * KPAD + t] * nRinv[t * KPAD + b];
                sG[a * KPAD + b] = v;   // G no longer needed
            }
        for (int a = 0; a < j; ++a)
            for (int b = 0; b < j; ++b) {
                double v = 0.0;
                for (int t = 0; t <= a; ++t) v += nRinv[t * KPAD + a] * sG[t * KPAD + b];
                sD[(j + a) * KPAD + (j + b)] = v;
            }
        for (int t = 0; t < KPAD * KPAD; ++t) { sT[t] = 0.0; sV[t] = 0.0; sR[t] = 0.0; }
    }
    __syncthreads();

    // ---- Householder QR of B~ (d x m), dgeqr2/dlarfg convention, + compact-WY T (dlarft)
    // The sweep that applies reflector c to the trailing block also accumulates the dot products reflector c + 1 needs
    // (column c + 1 against every column, rows > c + 1): one read + write of the block per column instead of 2 reads + 1 write.
    for (int c = 0; c < k; ++c) {
        if (c == 0) pf_multi_dot<KPAD>(Vh, d, c + 1, c, acc, red);
        if (tid == 0) {
            const double *rowc = Vh + (size_t)c * KPAD;
            double xn2 = 0.0;
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) if (cc == c) xn2 = acc[cc];
            const double alpha_c = rowc[c];
            const double xnorm = sqrt(xn2);
            double tau, scal, beta;
            if (xnorm == 0.0) { tau = 0.0; scal = 0.0; beta = alpha_c; }
            else {
                beta = -copysign(hypot(alpha_c, xnorm), alpha_c);
                tau = (beta - alpha_c) / beta;
                scal = 1.0 / (alpha_c - beta);
            }
            sTau[c] = tau; sScal = scal; sBeta = beta;
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) {
                const double vdot = rowc[cc] + scal * acc[cc];   // v_c . column cc
                sP[cc] = (cc > c) ? tau * vdot : vdot;
            }
            sT[c * KPAD + c] = tau;                              // T[0:c, c] = -tau T[0:c,0:c] (Vh' v_c)
            for (int a = 0; a < c; ++a) {
                double v = 0.0;
                for (int b = a; b < c; ++b) v += sT[a * KPAD + b] * sP[b];
                sT[a * KPAD + c] = -tau * v;
            }
        }
        __syncthreads();
        {
            const double scal = sScal;
            double acc2[KPAD];
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) acc2[cc] = 0.0;
            for (int i = c + 1 + tid; i < d; i += nt) {
                double *row = Vh + (size_t)i * KPAD;
                double r[KPAD];
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) r[cc] = row[cc];
                double v = 0.0;
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc == c) v = r[cc] * scal;
                double xn = 0.0;
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) {
                    if (cc == c) { r[cc] = v; row[cc] = v; }
                    else if (cc > c) { r[cc] -= sP[cc] * v; row[cc] = r[cc]; }
                    if (cc == c + 1) xn = r[cc];
                }
                if (i > c + 1) {
#pragma unroll
                    for (int cc = 0; cc < KPAD; ++cc) acc2[cc] += xn * r[cc];
                }
            }
            if (tid == 0) {
                double *rowc = Vh + (size_t)c * KPAD;
                rowc[c] = sBeta;
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc > c) rowc[cc] -= sP[cc];
            }
            if (c + 1 < k) {
                pf_block_sum<KPAD>(acc2, red);
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) acc[cc] = acc2[cc];
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    // ---- split the in-place factor: R (k x m) -> sR, Householder vectors keep an explicit unit diagonal
    for (int t = tid; t < k * KPAD; t += nt) {
        const int a = t / KPAD, b = t % KPAD;
        if (b >= a) {
            sR[a * KPAD + b] = (b < m) ? Vh[(size_t)a * KPAD + b] : 0.0;
            Vh[(size_t)a * KPAD + b] = (b == a) ? 1.0 : 0.0;
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- C = I + R D R' (k x k), V = chol(C).U     (src/woodbury.jl:205)
    if (tid == 0) {
        int st = PFMI_FIT_OK;
        double ldv = 0.0;
        for (int a = 0; a < k; ++a)                       // sG <- R D   (k x m)
            for (int b = 0; b < m; ++b) {
                double v = 0.0;
                for (int t = a; t < m; ++t) v += sR[a * KPAD + t] * sD[t * KPAD + b];
                sG[a * KPAD + b] = v;
            }
        for (int a = 0; a < k; ++a)
            for (int b = a; b < k; ++b) {
                double v = (a == b) ? 1.0 : 0.0;
                for (int t = b; t < m; ++t) v += sG[a * KPAD + t] * sR[b * KPAD + t];
                sV[a * KPAD + b] = v;
            }
        for (int c = 0; c < k && st == PFMI_FIT_OK; ++c) {
            double diag = sV[c * KPAD + c];
            for (int t = 0; t < c; ++t) diag -= sV[t * KPAD + c] * sV[t * KPAD + c];
            if (!(diag > 0.0) || !isfinite(diag)) { st = PFMI_FIT_C_NOT_PD; break; }
            diag = sqrt(diag);
            sV[c * KPAD + c] = diag;
            ldv += log(diag);
            for (int b = c + 1; b < k; ++b) {
                double v = sV[c * KPAD + b];
                for (int t = 0; t < c; ++t) v -= sV[t * KPAD + c] * sV[t * KPAD + b];
                sV[c * KPAD + b] = v / diag;
            }
        }
        for (int a = k; a < KPAD; ++a) sV[a * KPAD + a] = 1.0;   // identity padding
        sStatus = st;
        sLogdetV = ldv;
    }
    __syncthreads();
    const size_t sm = (size_t)p * KPAD * KPAD;
    if constexpr (!BIG) {
        for (int t = tid; t < KPAD * KPAD; t += nt) {
            A.tmat[sm + t] = sT[t]; A.vchol[sm + t] = sV[t]; A.rq[sm + t] = sR[t]; A.dmat[sm + t] = sD[t];
        }
    } else { (void)sm; __threadfence_block(); }
    if (sStatus != PFMI_FIT_OK) {
        for (int i = tid; i < d; i += nt) mu[i] = NAN;
        if (tid == 0) { A.status[p] = sStatus; A.logdet[p] = NAN; }
        return;
    }
    // ---- mu = theta + Sigma grad,  Sigma g = U' Q [V'V 0;0 I] Q' U g   (src/mvnormal.jl:16-19,
    //      src/woodbury.jl:64-68,129-143);  Q = I - Vh T Vh',  Q' = I - Vh T' Vh'
    pf_vec_dot<KPAD>(Vh, d, [&](int i, const double *) { return sqa[i] * grad_p[i]; }, acc, red);
    if (tid == 0) {
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) sP[cc] = acc[cc];
        for (int a = 0; a < KPAD; ++a) {   // t1 = T' w1
            double v = 0.0;
            for (int b = 0; b <= a; ++b) v += sT[b * KPAD + a] * sP[b];
            sW[a] = v;
        }
    }
    __syncthreads();
    // head rows of b = Q' U g
    for (int i = tid; i < k; i += nt) {
        const double *row = Vh + (size_t)i * KPAD;
        double v = sqa[i] * grad_p[i];
        for (int cc = 0; cc < KPAD; ++cc) v -= row[cc] * sW[cc];
        sHead[i] = v;
    }
    __syncthreads();
    if (tid == 0) {   // head <- V'(V head)
        for (int a = 0; a < k; ++a) {
            double v = 0.0;
            for (int b = a; b < k; ++b) v += sV[a * KPAD + b] * sHead[b];
            sHead[a] = v;
        }
        for (int a = k - 1; a >= 0; --a) {
            double v = 0.0;
            for (int b = 0; b <= a; ++b) v += sV[b * KPAD + a] * sHead[b];
            sHead[a] = v;
        }
    }
    __syncthreads();
    auto bvec = [&](int i, const double *row) {
        if (i < k) return sHead[i];
        double v = sqa[i] * grad_p[i];
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) v -= row[cc] * sW[cc];
        return v;
    };
    pf_vec_dot<KPAD>(Vh, d, bvec, acc, red);
    if (tid == 0) {
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) sP[cc] = acc[cc];
        for (int a = 0; a < KPAD; ++a) {   // t2 = T w2
            double v = 0.0;
            for (int b = a; b < KPAD; ++b) v += sT[a * KPAD + b] * sP[b];
            sTau[a] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < d; i += nt) {
        const double *row = Vh + (size_t)i * KPAD;
        double v = bvec(i, row);
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) v -= row[cc] * sTau[cc];
        mu[i] = theta_p[i] + sqa[i] * v;
    }
    if (tid == 0) {
        A.status[p] = PFMI_FIT_OK;
        A.logdet[p] = 2.0 * (ldu + sLogdetV);    // src/woodbury.jl:76-80
    }
}

// ---------------------------------------------------------------------------------------------------
// Register-resident variant of pf_fit_kernel for d <= 256 * RPT: thread t owns rows {t + 256 i}, the whole
// d x KC block lives in VGPRs (RPT * KC doubles per thread), so the ~25 sweeps of the Gram / Householder /
// mean computation touch no memory at all -- only ~27 block reductions remain.  Same operations and the same
// LAPACK reflector convention as pf_fit_kernel; only the summation order inside the block reductions differs
// (fp64 roundoff), which the parity tests cover.
template <int KPAD, int RPT, int NT>
__global__ __launch_bounds__(NT, (KPAD <= 8 ? 4 : KPAD <= 12 ? 3 : 2)) void pf_fit_reg_kernel(FitArgs A) {   // measured best occupancy per KPAD (J = 6: 3 waves per SIMD, 168 VGPRs + 19 spilled dwords beat 2 waves with 221)
    // XCD-aware point order (round 4): workgroup b runs on XCD b % 8 and every XCD has its own L2.  A fit reads 4 J trace rows, all but two
    // of which its neighbours p - 1, p + 1, ... read as well: with p = b those neighbours sit on eight different L2s and the rows came from
    // HBM / MALL again and again (FETCH_SIZE 1.5 GB per launch for 0.18 GB of distinct rows; the row loads were a third of the kernel).  XCD x
    // now walks its own contiguous eighth of the points, so the ring's rows stay in that XCD's L2.
    int64_t p;
    {
        const int P = (int)gridDim.x, x = blockIdx.x & 7, slot = blockIdx.x >> 3, q8 = P >> 3, r8 = P & 7;
        int idx = x * q8 + (x < r8 ? x : r8) + slot;
        if (FITREG_XCD == 0) idx = blockIdx.x;
        if (!pf_fit_point(A, idx, p)) { if (threadIdx.x == 0) A.status[p] = PFMI_FIT_ABSENT; return; }
    }
    const int tid = threadIdx.x;
    const int d = A.d, J = A.J;
    const int path = A.path_of[p];
    const int64_t p0 = A.off[path];
    const int j = A.hist_len[p], m = 2 * j, k = d < m ? d : m;
    const double *alpha = A.alpha_all + (size_t)p * d;
    double *Vh = A.vh + (size_t)p * d * KPAD;
    double *sqa = A.sqrt_alpha + (size_t)p * d;
    double *mu = A.mu + (size_t)p * d;
    const double *theta_p = A.theta + (size_t)p * d, *grad_p = A.grad + (size_t)p * d;

    constexpr int NVMAX = KPAD;
    __shared__ __attribute__((aligned(16))) double red[2 * (NT / 64) * NVMAX];       // two halves: one barrier per block reduction (pf_block_sum_pp)
    int flip = 0;
    __shared__ double sRow[2][KPAD];
    __shared__ double sD[KPAD * KPAD], sR[KPAD * KPAD], sT[KPAD * KPAD], sV[KPAD * KPAD], sG[KPAD * KPAD];
    __shared__ double sX1[(KPAD / 2) * KPAD], sX2[(KPAD / 2) * KPAD], sX3[(KPAD / 2) * KPAD];     // j x j scratch of the D computation
    __shared__ double sGv[KPAD * KPAD], sVt[KPAD * KPAD];   // V'V (by-product of the column reductions) and the top k x k block of V: the mean needs no second sweep
    __shared__ double sT12[KPAD], sDl[KPAD];
    __shared__ double sLogdetV;
    __shared__ int sStatus;

#if FITREG_PROF
    long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#endif
    double a[RPT][KPAD];       // this thread's rows of B~ (later: Householder vectors)
    double bad = 0.0, ldu = 0.0;
    double isa_r[RPT], al_r[RPT];                                 // alpha and 1 / sqrt(alpha) of this thread's rows (kept for the rows of B~)
    int rl_r[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) { const int row = tid + NT * i; rl_r[i] = row < d ? row : d - 1; }
#pragma unroll
    for (int i = 0; i < RPT; ++i) al_r[i] = alpha[rl_r[i]];      // (all RPT loads in flight: guarded loads were one round trip each)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = tid + NT * i;
        const double al = al_r[i];
        const double s = sqrt(al);
        isa_r[i] = 1.0 / s;                                        // one reciprocal per row instead of 2 j divisions
        if (row < d) {
            if (!(al > 0.0) || !isfinite(al)) bad = 1.0;
            sqa[row] = s;
            ldu += log(s);
        }
    }
    {
        double v[2] = {bad, ldu};
        pf_block_sum_pp<2, NVMAX>(v, red, flip);
        bad = v[0]; ldu = v[1];
    }
    for (int t = tid; t < KPAD * KPAD; t += NT) { sD[t] = 0.0; sR[t] = 0.0; sT[t] = 0.0; sV[t] = 0.0; sG[t] = 0.0; sGv[t] = 0.0; sVt[t] = 0.0; }
    const size_t sm = (size_t)p * KPAD * KPAD;
    if (bad > 0.0) {                                           // A not positive definite (src/woodbury.jl:202)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = tid + NT * i;
            if (row < d) {
                mu[row] = NAN;
                for (int c = 0; c < KPAD; ++c) Vh[(size_t)row * KPAD + c] = 0.0;
            }
        }
        for (int t = tid; t < KPAD * KPAD; t += NT) { A.tmat[sm + t] = 0.0; A.vchol[sm + t] = 0.0; A.rq[sm + t] = 0.0; A.dmat[sm + t] = 0.0; }
        if (tid == 0) { A.status[p] = PFMI_FIT_A_NOT_PD; A.logdet[p] = NAN; }
        return;
    }
    FR_STAMP(0);                                               // prologue: sqrt / log of alpha, first reduction, LDS zero fill
    // ---- rows of B~ = U' \ [alpha.Y  S]   (src/inverse_hessian.jl:117-118, src/woodbury.jl:204)
    // (round 4: the four rows of a history pair are loaded for all RPT rows of the thread at once, unconditionally from clamped
    //  addresses -- 16 loads in flight, 6 round trips per fit.  With a guarded block per (row, pair) the compiler waited with vmcnt(0) after
    //  every 4 loads: 24 dependent round trips, a third of the kernel.  Same arithmetic.)
#pragma unroll
    for (int i = 0; i < RPT; ++i)
#pragma unroll
        for (int c = 0; c < KPAD; ++c) a[i][c] = 0.0;
#pragma unroll
    for (int c = 0; c < KPAD / 2; ++c) {
        if (c >= j) continue;                                               // wave-uniform
        const int src = A.hist_src[(size_t)p * J + c];
        const double *g0 = A.grad + (size_t)(p0 + src) * d, *g1 = g0 + d, *t0 = A.theta + (size_t)(p0 + src) * d, *t1 = t0 + d;
        double vg0[RPT], vg1[RPT], vt0[RPT], vt1[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) { vg0[i] = g0[rl_r[i]]; vg1[i] = g1[rl_r[i]]; vt0[i] = t0[rl_r[i]]; vt1[i] = t1[rl_r[i]]; }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const bool in = tid + NT * i < d;
            const double y = vg0[i] - vg1[i];                               // y = grad_l - grad_{l+1}   :46
            const double sx = vt1[i] - vt0[i];                              // s = theta_{l+1} - theta_l :45
            const double by = in ? (al_r[i] * y) * isa_r[i] : 0.0, bs = in ? sx * isa_r[i] : 0.0;
            // columns c and j + c.  j is a run-time value: the usual full history (j = KPAD / 2: all but the first points of a
            // path) takes a static slot, shorter ones select among the statically unrolled targets
            a[i][c] = by;
            if (j == KPAD / 2) a[i][KPAD / 2 + c] = bs;
            else {
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc == j + c) a[i][cc] = bs;
            }
        }
    }
    FR_STAMP(1);                                               // rows of B~
    double acc[KPAD];
    // ---- Householder QR, one block reduction per column.  Thread aa < KPAD keeps row aa of the compact-WY T in
    //      registers (dlarft: T[0:c, c] = -tau T[0:c,0:c] (Vh' v_c)), so the column loop has no serial section.
    // (row aa of T lives in LDS and is only ever touched by thread aa: no barrier needed, and 2 KPAD fewer VGPRs)
    // (fully unrolled: with a run-time column index every "column c of my rows" is a chain of KPAD v_cndmask pairs per row --
    //  a quarter of the kernel's instructions at KPAD = 12, and the kernel is issue-bound with 3 waves per SIMD)
#pragma unroll
    for (int c = 0; c < KPAD; ++c) {
        if (c >= k) continue;                 // (not `break`: an early exit keeps the optimizer from unrolling around the barrier)
        double *srow = sRow[c & 1];
        if (tid == c) {                       // row c belongs to thread c (slot 0): publish it
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) srow[cc] = a[0][cc];
        }
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) acc[cc] = 0.0;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = tid + NT * i;
            if (row > c) {
                double xc = 0.0;
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc == c) xc = a[i][cc];
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) acc[cc] += xc * a[i][cc];
            }
        }
        pf_block_sum_mv<KPAD, NVMAX>(acc, red, flip);       // its barriers also publish srow (double buffered: no trailing barrier)
        double xn2 = 0.0;
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) if (cc == c) xn2 = acc[cc];
        const double alpha_c = srow[c];
        double tau, scal, beta;
        if (xn2 == 0.0) { tau = 0.0; scal = 0.0; beta = alpha_c; }     // dlarfg's xnorm == 0 test (sqrt(xn2) == 0 iff xn2 == 0: xn2 is a sum of squares)
        else {
            beta = -copysign(sqrt(fma(alpha_c, alpha_c, xn2)), alpha_c);
            tau = (beta - alpha_c) / beta;
            scal = 1.0 / (alpha_c - beta);
        }
        double (&wv)[KPAD] = acc;            // in place -- cc > c: tau * (v_c . column cc); cc < c: v_c . v_cc
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) {
            const double vdot = srow[cc] + scal * acc[cc];
            wv[cc] = (cc > c) ? tau * vdot : vdot;
        }
        if (tid <= c && tid < KPAD) {        // T column c, one row per thread
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < KPAD; ++b) if (b >= tid && b < c) v += sT[tid * KPAD + b] * wv[b];
            const double tnew = (tid == c) ? tau : -tau * v;
            sT[tid * KPAD + c] = tnew;
            if (tid == c) {                  // row / column c of V'V: wv[cc < c] = v_c . v_cc, |v_c|^2 = 1 + scal^2 |x|^2
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc < c) { sGv[c * KPAD + cc] = wv[cc]; sGv[cc * KPAD + c] = wv[cc]; }
                sGv[c * KPAD + c] = fma(scal * scal, xn2, 1.0);
            }
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = tid + NT * i;
            if (row > c) {
                double v = 0.0;
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) if (cc == c) v = a[i][cc] * scal;
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) {
                    if (cc == c) a[i][cc] = v;
                    else if (cc > c) a[i][cc] -= wv[cc] * v;
                }
            } else if (row == c) {
#pragma unroll
                for (int cc = 0; cc < KPAD; ++cc) {
                    if (cc == c) a[i][cc] = beta;
                    else if (cc > c) a[i][cc] -= wv[cc];
                }
            }
        }
    }
    FR_STAMP(2);                                               // QR
    // ---- split: R (k x m) -> sR; Householder vectors get an explicit unit diagonal
    if (tid < k) {
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) {
            if (cc >= tid) {
                sR[tid * KPAD + cc] = (cc < m) ? a[0][cc] : 0.0;
                a[0][cc] = (cc == tid) ? 1.0 : 0.0;
            }
            if (cc < k) sVt[tid * KPAD + cc] = a[0][cc];            // top k x k block of V (unit lower triangular)
        }
    }
    __syncthreads();
    // ---- Gram matrix from the triangular factor: G = B~'B~ = R'R (G[c][b], c, b < j: Y'alpha Y; G[j + a][b]: S'Y) -- round 1 swept
    //      the register block for it (KPAD (KPAD + 1) / 2 products per row and three block reductions, 15 % of the kernel)
    for (int t = tid; t < m * m; t += NT) {
        const int aa = t / m, b = t % m, u1 = (aa < b ? aa : b) < k - 1 ? (aa < b ? aa : b) : k - 1;
        double v = 0.0;
        for (int u = 0; u <= u1; ++u) v += sR[u * KPAD + aa] * sR[u * KPAD + b];
        sG[aa * KPAD + b] = v;
    }
    __syncthreads();
    // ---- D (m x m)   (src/inverse_hessian.jl:119-130)
    if (j > 0) {
        double *R = sX1, *nRinv = sX2;
        for (int t = tid; t < j * j; t += NT) {
            const int aa = t / j, b = t % j;
            R[aa * KPAD + b] = (b >= aa) ? sG[(j + aa) * KPAD + b] : 0.0;      // triu(S'Y)   :119-121
            nRinv[aa * KPAD + b] = 0.0;
        }
    }
    __syncthreads();
    if (tid < j) {                                   // -R^{-1}: lane c solves column c by back substitution :122-124
        const double *R = sX1;
        double *nRinv = sX2;
        const int c = tid;
        for (int r = c; r >= 0; --r) {
            double rhs = (r == c) ? -1.0 : 0.0;
            for (int t = r + 1; t <= c; ++t) rhs -= R[r * KPAD + t] * nRinv[t * KPAD + c];
            nRinv[r * KPAD + c] = rhs / R[r * KPAD + r];
        }
    }
    __syncthreads();
    if (j > 0) {   // M = Y'alpha Y + diag(R); D12, D21 -- one entry per thread
        for (int t = tid; t < j * j; t += NT) {
            const int aa = t / j, b = t % j;
            sD[aa * KPAD + (j + b)] = sX2[aa * KPAD + b];
            sD[(j + aa) * KPAD + b] = sX2[b * KPAD + aa];
            double v = (aa <= b) ? sG[aa * KPAD + b] : sG[b * KPAD + aa];
            if (aa == b) v += sX1[aa * KPAD + aa];
            sX3[aa * KPAD + b] = v;                                  // M
        }
    }
    __syncthreads();
    if (j > 0) {   // T1 = M nRinv  -> sG (G is no longer needed)
        for (int t = tid; t < j * j; t += NT) {
            const int aa = t / j, b = t % j;
            double v = 0.0;
            for (int u = 0; u <= b; ++u) v += sX3[aa * KPAD + u] * sX2[u * KPAD + b];
            sG[aa * KPAD + b] = v;
        }
    }
    __syncthreads();
    if (j > 0) {   // D22 = nRinv' T1
        for (int t = tid; t < j * j; t += NT) {
            const int aa = t / j, b = t % j;
            double v = 0.0;
            for (int u = 0; u <= aa; ++u) v += sX2[u * KPAD + aa] * sG[u * KPAD + b];
            sD[(j + aa) * KPAD + (j + b)] = v;
        }
    }
    __syncthreads();
    FR_STAMP(3);                                               // Gram, D
    // ---- C = I + R D R' (k x k): RD -> sG, then C -> sV (upper), all threads; Cholesky on one lane
    for (int t = tid; t < k * m; t += NT) {
        const int aa = t / m, b = t % m;
        double v = 0.0;
        for (int u = aa; u < m; ++u) v += sR[aa * KPAD + u] * sD[u * KPAD + b];
        sG[aa * KPAD + b] = v;
    }
    __syncthreads();
    for (int t = tid; t < k * k; t += NT) {
        const int aa = t / k, b = t % k;
        if (b >= aa) {
            double v = (aa == b) ? 1.0 : 0.0;
            for (int u = b; u < m; ++u) v += sG[aa * KPAD + u] * sR[b * KPAD + u];
            sV[aa * KPAD + b] = v;
        }
    }
    __syncthreads();
    FR_STAMP(4);                                               // C
    if (tid < 64) {                          // wave 0: Cholesky C = V'V in REGISTERS, lane b owns column b of the upper factor.
        // Row c of V needs column c (lane c's registers) in every lane: v_readlane broadcasts instead of the LDS round trips of the
        // left-looking loop this replaces (round 4: 37 000 of a fit's 257 000 cycles, and the fit kernel's time follows its critical
        // path -- without this section it ran 12 % faster).  Same operations in the same order as before: the same bits.
        const int b = tid;
        double col[KPAD];
#pragma unroll
        for (int t = 0; t < KPAD; ++t) col[t] = (b < k && t <= b) ? sV[t * KPAD + b] : 0.0;
        int status = PFMI_FIT_OK;
        double ldv = 0.0;
#pragma unroll
        for (int c = 0; c < ((FITREG_ABLATE & 1) ? 0 : KPAD); ++c) {
            if (c >= k || status != PFMI_FIT_OK) continue;          // wave-uniform
            double v = col[c];                                     // C[c][b]
#pragma unroll
            for (int t = 0; t < c; ++t) {
                const double vtc = pf_readlane_f64(col[t], c);     // V[t][c]
                v -= vtc * col[t];                                 // (lanes b < c hold zeros there: harmless)
            }
            const double diag = pf_readlane_f64(v, c);
            if (!(diag > 0.0) || !isfinite(diag)) { status = PFMI_FIT_C_NOT_PD; continue; }    // src/woodbury.jl:205
            const double dc = sqrt(diag);
            ldv = ldv + log(dc);
            col[c] = (b == c) ? dc : (b > c ? v / dc : col[c]);
        }
#pragma unroll
        for (int t = 0; t < KPAD; ++t) if (b < k && t <= b) sV[t * KPAD + b] = col[t];
        if (b >= k && b < KPAD) sV[b * KPAD + b] = 1.0;                             // identity padding
        if (b == 0) { sStatus = status; sLogdetV = ldv; }
    }
    __syncthreads();
    FR_STAMP(5);                                               // Cholesky
    for (int t = tid; t < KPAD * KPAD; t += NT) {
        A.tmat[sm + t] = sT[t]; A.vchol[sm + t] = sV[t]; A.rq[sm + t] = sR[t]; A.dmat[sm + t] = sD[t];
    }
    // Householder block out (row-major [d][KPAD], 96 B rows)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = tid + NT * i;
        if (row < d) {
            double2 *o = reinterpret_cast<double2 *>(Vh + (size_t)row * KPAD);       // 16-byte stores (rows are KPAD * 8 bytes, KPAD even)
#pragma unroll
            for (int cc = 0; cc < KPAD; cc += 2) o[cc >> 1] = make_double2(a[i][cc], a[i][cc + 1]);
        }
    }
    if (sStatus != PFMI_FIT_OK) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) { const int row = tid + NT * i; if (row < d) mu[row] = NAN; }
        if (tid == 0) { A.status[p] = sStatus; A.logdet[p] = NAN; }
        return;
    }
    FR_STAMP(6);                                               // outputs
    if (FITREG_ABLATE & 2) return;
    // ---- mu = theta + U' Q [V'V 0;0 I] Q' U g     (sqrt(alpha) and sqrt(alpha) * grad are re-read rather than kept in VGPRs)
    // b = Q'(U g) = U g - V t1 (t1 = T'w1, w1 = V'U g: the ONE sweep + block reduction); head' = Vc'Vc head; x = Q b' = b' - V t2,
    // t2 = T V'b'.  V'b' needs no second sweep (round 4): V'b = w1 - (V'V) t1 with V'V from the column reductions, and b' differs from
    // b in the head rows only: V'b' = V'b + Vtop'(head' - head).  So x = U g - V (t1 + t2) + e_head (head' - head): every thread needs
    // t1 + t2 and the head rows their increment, both made by wave 0 with one entry per lane (broadcasts by v_readlane).
    double agv[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = tid + NT * i;
        agv[i] = (row < d) ? sqa[row] * grad_p[row] : 0.0;
    }
#pragma unroll
    for (int cc = 0; cc < KPAD; ++cc) acc[cc] = 0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i)
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) acc[cc] += agv[i] * a[i][cc];
    pf_block_sum_mv<KPAD, NVMAX>(acc, red, flip);
    if (tid < 64) {
        const int aa = tid < KPAD ? tid : KPAD - 1;                // lanes >= KPAD compute a harmless copy of lane KPAD - 1
        const bool live = tid < k;
        double w1l = 0.0, t1l = 0.0;
#pragma unroll
        for (int b = 0; b < KPAD; ++b) {
            w1l = (b == aa) ? acc[b] : w1l;
            t1l += sT[b * KPAD + aa] * acc[b];                     // t1 = T'w1 (T is upper triangular, zero filled)
        }
        double hbl = live ? agv[0] : 0.0;                          // head of b = U g - V t1: row aa of thread aa
#pragma unroll
        for (int cc = 0; cc < KPAD; ++cc) hbl -= sVt[aa * KPAD + cc] * pf_readlane_f64(t1l, cc);
        hbl = live ? hbl : 0.0;
        double tmpl = 0.0;
#pragma unroll
        for (int b = 0; b < KPAD; ++b) tmpl += sV[aa * KPAD + b] * pf_readlane_f64(hbl, b);       // Vc head
        double hdl = 0.0;
#pragma unroll
        for (int b = 0; b < KPAD; ++b) hdl += sV[b * KPAD + aa] * pf_readlane_f64(tmpl, b);       // Vc'(Vc head)
        const double dll = live ? hdl - hbl : 0.0;
        double w2l = w1l;                                          // V'b' = w1 - (V'V) t1 + Vtop'(head' - head)
#pragma unroll
        for (int b = 0; b < KPAD; ++b) w2l -= sGv[aa * KPAD + b] * pf_readlane_f64(t1l, b);
#pragma unroll
        for (int r = 0; r < KPAD; ++r) w2l += sVt[r * KPAD + aa] * pf_readlane_f64(dll, r);
        double t2l = 0.0;
#pragma unroll
        for (int b = 0; b < KPAD; ++b) t2l += sT[aa * KPAD + b] * pf_readlane_f64(w2l, b);        // t2 = T V'b'
        if (tid < KPAD) { sT12[tid] = t1l + t2l; sDl[tid] = dll; }
    }
    __syncthreads();
    double t12[KPAD];
#pragma unroll
    for (int cc = 0; cc < KPAD; ++cc) t12[cc] = sT12[cc];
    const double dl0 = (tid < k) ? sDl[tid < KPAD ? tid : 0] : 0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = tid + NT * i;
        if (row < d) {
            double v = agv[i];
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) v -= a[i][cc] * t12[cc];
            if (i == 0) v += dl0;
            mu[row] = theta_p[row] + sqa[row] * v;
        }
    }
    FR_STAMP(7);                                               // mean
#if FITREG_PROF
    if ((blockIdx.x & 511) == 7 && tid == 0)
        printf("FITREG_PROF wg %d (10 ns ticks): prologue %lld rows %lld QR %lld gram+D %lld C %lld chol %lld out %lld mean %lld\n", (int)blockIdx.x, prof[0],
               prof[1], prof[2], prof[3], prof[4], prof[5], prof[6], prof[7]);
#endif
    if (tid == 0) {
        A.status[p] = PFMI_FIT_OK;
        A.logdet[p] = 2.0 * (ldu + sLogdetV);
    }
}

// ---------------------------------------------------------------------------------------------------
int32_t pf_launch_history(pfmi_ctx *c, double eps, const HistSeg *seg) {
    PF_CHECK(c->J <= 64, PFMI_ERR_UNSUPPORTED, "history_length %d > 64 unsupported", c->J);
    HistSeg sg = seg ? *seg : HistSeg{nullptr, 0, INT_MAX, nullptr, nullptr, 0};
    sg.hinit = c->hinit;
    pf_kernel_begin(c);
    {
        const char *hk = pf_debug_get("PFMI_HISTORY_KERNEL");            // "mem": the memory-resident walk at every d (tests)
        if (c->d > 16 * 1024 || (hk && hk[0] == 'm')) {
            hipLaunchKernelGGL(pf_history_mem_kernel, dim3(c->K), dim3(1024), 0, c->stream, c->d, c->J, eps, c->d_off.as<int64_t>(),
                               c->th(), c->gr(), c->alpha_all.as<double>(), c->hist_len.as<int>(),
                               c->hist_src.as<int>(), c->n_rej.as<int>(), c->hist_acc.as<int>(), sg);
            pf_kernel_end(c, "history");
            PF_HIP(hipGetLastError());
            return PFMI_OK;
        }
    }
#define PF_HIST(E, NT)                                                                                           \
    hipLaunchKernelGGL((pf_history_kernel<E, NT>), dim3(c->K), dim3(NT), 0, c->stream, c->d, c->J, eps,           \
                       c->d_off.as<int64_t>(), c->th(), c->gr(),                      \
                       c->alpha_all.as<double>(), c->hist_len.as<int>(), c->hist_src.as<int>(), c->n_rej.as<int>(),          \
                       c->hist_acc.as<int>(), sg)
    // the walk is sequential in l: one block reduction per iteration, cheaper across 4 waves than across 16
    // (a single wave for d <= 256: no cross-wave exchange at all)
    if (c->d <= 64) PF_HIST(1, 64); else if (c->d <= 128) PF_HIST(2, 64); else if (c->d <= 256) PF_HIST(4, 64);
    else if (c->d <= 512) PF_HIST(2, 256); else if (c->d <= 1024) PF_HIST(4, 256);
    else { const int ept = (c->d + 1023) / 1024;
           const char *hk = pf_debug_get("PFMI_HISTORY_KERNEL");        // "prefetch": the register-set kernel at every d (tests, A/B)
           const bool lean = !(hk && hk[0] == 'p');
#define PF_HIST_LEAN(E)                                                                                                  \
    do {                                                                                                                \
        auto kern = pf_history_lean_kernel<E>;                                                                          \
        PF_TRY(pf_raise_lds_limit(c, reinterpret_cast<const void *>(kern), (int)(sizeof(double) * E * 1024)));           \
        hipLaunchKernelGGL(kern, dim3(c->K), dim3(1024), sizeof(double) * E * 1024, c->stream, c->d, c->J, eps,         \
                           c->d_off.as<int64_t>(), c->th(), c->gr(), c->alpha_all.as<double>(), \
                           c->hist_len.as<int>(), c->hist_src.as<int>(), c->n_rej.as<int>(), c->hist_acc.as<int>(), sg);  \
    } while (0)
           if (ept <= 2) PF_HIST(2, 1024);
           else if (ept <= 4) { if (lean) PF_HIST_LEAN(4); else PF_HIST(4, 1024); }
           else if (ept <= 6) { if (lean) PF_HIST_LEAN(6); else PF_HIST(6, 1024); }
           else if (ept <= 8) { if (lean) PF_HIST_LEAN(8); else PF_HIST(8, 1024); }
           else if (ept <= 10) { if (lean) PF_HIST_LEAN(10); else PF_HIST(10, 1024); }
           else if (ept <= 12) PF_HIST(12, 1024); else PF_HIST(16, 1024); }
#undef PF_HIST_LEAN
#undef PF_HIST
    pf_kernel_end(c, "history");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

template <int KPAD>
static void launch_fit_t(pfmi_ctx *c, const FitArgs &a) {
    const char *force = pf_debug_get("PFMI_FIT_KERNEL");            // "mem" forces the general (memory-resident) kernel
    const bool allow_reg = !(force && force[0] == 'm');
    dim3 grid((unsigned)a.P), block(FIT_THREADS);
    if constexpr (KPAD <= 16) {
        // register-resident kernel: 256 threads x RPT rows (measured faster than 512 x 2: the kernel is bound by
        // block-reduction / serial O(m^3) latency, and 8-wave barriers cost more than the extra occupancy buys)
        if (allow_reg && a.d <= 256) { hipLaunchKernelGGL((pf_fit_reg_kernel<KPAD, 1, 256>), grid, dim3(256), 0, c->stream, a); return; }
        if (allow_reg && a.d <= 512) { hipLaunchKernelGGL((pf_fit_reg_kernel<KPAD, 2, 256>), grid, dim3(256), 0, c->stream, a); return; }
        if (allow_reg && a.d <= 1024) { hipLaunchKernelGGL((pf_fit_reg_kernel<KPAD, 4, 256>), grid, dim3(256), 0, c->stream, a); return; }
    }
    hipLaunchKernelGGL(pf_fit_kernel<KPAD>, grid, block, 0, c->stream, a);
}

int32_t pf_launch_fit_panel(pfmi_ctx *c, const FitArgs &a, bool *handled);   // fit_panel_kernel.hip
int32_t pf_launch_fit_tsqr(pfmi_ctx *c, const FitArgs &a, bool *handled);    // fit_tsqr_kernel.hip

int32_t pf_launch_fit(pfmi_ctx *c, int seg_l0, int seg_len) {
    FitArgs a;
    a.d = c->d; a.J = c->J;
    a.seg_l0 = seg_l0; a.seg_len = seg_len; a.vcap = c->vcap; a.npts = c->st_npts.as<int32_t>();
    PF_CHECK(seg_len == 0 || c->virt, PFMI_ERR_STATE, "fit: segment launches need the streaming layout");
    a.off = c->d_off.as<int64_t>(); a.path_of = c->d_path_of.as<int32_t>();
    a.theta = c->th(); a.grad = c->gr(); a.alpha_all = c->alpha_all.as<double>();
    a.hist_len = c->hist_len.as<int32_t>(); a.hist_src = c->hist_src.as<int32_t>();
    a.vh = c->vh.as<double>(); a.tmat = c->tmat.as<double>(); a.vchol = c->vchol.as<double>();
    a.rq = c->rq.as<double>(); a.dmat = c->dmat.as<double>(); a.sqrt_alpha = c->sqrt_alpha.as<double>();
    a.mu = c->mu.as<double>(); a.logdet = c->logdet.as<double>(); a.status = c->status.as<int32_t>();
    a.P = seg_len > 0 ? (int64_t)c->K * seg_len : c->P;        // work items of this launch
    a.big = nullptr;
    pf_kernel_begin(c);
    {
        // d > 1024: the TSQR + Householder-reconstruction kernel (round 6: the block crosses HBM once each way); PFMI_FIT_KERNEL (debug hook)
        // = "panel": the left-looking panel kernel of rounds 2 - 5, "mem": the column-by-column memory-resident kernel
        const char *force = pf_debug_get("PFMI_FIT_KERNEL");
        bool handled = false;
        if (!(force && (force[0] == 'm' || force[0] == 'p'))) PF_TRY(pf_launch_fit_tsqr(c, a, &handled));
        if (!handled && !(force && force[0] == 'm')) PF_TRY(pf_launch_fit_panel(c, a, &handled));
        if (handled) {
            pf_kernel_end(c, "fit");
            PF_HIP(hipGetLastError());
            return PFMI_OK;
        }
    }
    switch (c->kpad) {
        case 4: launch_fit_t<4>(c, a); break;
        case 8: launch_fit_t<8>(c, a); break;
        case 12: launch_fit_t<12>(c, a); break;
        case 16: launch_fit_t<16>(c, a); break;
        case 20: launch_fit_t<20>(c, a); break;
        case 32: launch_fit_t<32>(c, a); break;
        case 64:
            PF_TRY(c->fit_scratch.ensure(sizeof(double) * (size_t)a.P * 64 * 64));
            a.big = c->fit_scratch.as<double>();
            launch_fit_t<64>(c, a);
            break;
        default: PF_CHECK(false, PFMI_ERR_UNSUPPORTED, "unsupported kpad %d", c->kpad);
    }
    pf_kernel_end(c, "fit");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
