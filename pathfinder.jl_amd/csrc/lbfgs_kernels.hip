// lbfgs_kernels.hip -- batched on-device L-BFGS trajectory generation.
//
// Plays the role of reference src/optimize.jl:35-121 (optimize_with_trace + OptimizationCallback): it PRODUCES the hot
// path's input, the trace (theta_l, logp_l, grad logp_l), directly in HBM so that fit_batch can consume it without a
// host round trip.  The reference delegates the optimisation itself to Optim.LBFGS + HagerZhang (third party); this is
// this repo's own driver -- L-BFGS direction with gamma = s'y / y'y, strong-Wolfe bracketing + bisection zoom,
// maxiters as src/optimize.jl:40, stop at |g|_inf <= g_tol -- the same iteration as pfmi/optimize.py (host, for
// callback targets); tests check it against the scalar C restatement under oracle/.  Built-in targets only
// (analytic gradients).
//
// One persistent workgroup per path (paths are independent, src/multipath.jl:190-208); every vector lives in
// registers (thread t owns elements t, t+NT, ...), the (s, y) ring lives in LDS when it fits and in an L2-resident
// scratch otherwise (element i of every history vector is only ever touched by its owner thread, so the ring needs no
// barriers).  All scalars that steer control flow come out of fixed-order block reductions and are bit-identical in
// every thread, so the whole workgroup walks the same line-search branches.
//
// One path = one workgroup = one chain of dependent steps with a single wave per SIMD: the kernel is bound by LATENCY, so the
// iteration is arranged around few, wide steps (round 3; round 2 ran the textbook two-loop recursion: 2 h + 5 block reductions
// and 2 h one-slot-at-a-time sweeps of the ring through flat loads per iteration):
//   * H g is applied in the compact form of Byrd, Nocedal & Schnabel (1994): H = gamma I + [S gamma Y] M [S'; gamma Y'] with M built from
//     R = triu(S'Y), D = diag(R) and Y'Y.  Mathematically the two-loop recursion; all it needs are inner products.
//   * ONE fused reduction when a step is accepted gathers everything the next direction needs: for every pair c of the ring
//     (s_c'y_new, y_c'y_new, s_c'g_new, y_c'g_new) -- the new column of R and Y'Y and the vectors S'g, Y'g -- plus the curvature test, the
//     convergence test (count of |g_i| > g_tol), g'g and the "moved / finite" flags: 4 CH + 8 values per CH pairs of the ring.
//   * the (h x h) triangular solves run in every thread redundantly on a wave-private LDS copy of the Gram data (uniform
//     addresses = broadcast reads; no barrier, no cross-lane traffic, bit-identical everywhere).
//   * a function evaluation is ONE reduction: the directional derivative g(x + a p)'p is accumulated in the same sweep
//     (Gaussian: sum a_i e_i p_i - h'(W~'p); funnel: g_0 p_0 + e^-tau sum x_i p_i) instead of a second dot product.
//   * the ring is addressed as LDS (ds_read) or as global memory by a template flag, never through a pointer that could be
//     either, and its rows are padded to EPT * NT so that every sweep is a batch of unconditional loads.
//   * block reductions: wave butterfly, one barrier, then lane l of every wave sums the waves' partials of value l (instead of every
//     thread reading all NV x nwaves partials); the few totals a thread needs as scalars are broadcast with v_readlane.
// 2 + (line-search evaluations) reductions per iteration instead of 2 h + 5.
#include "pfmi_common.h"

struct LbfgsArgs {
    int d, J, maxiters, kind, r, hist_in_lds;
    int reject_every;                       // test hook (PFMI_LBFGS_REJECT_EVERY): every n-th pair is treated as failing the curvature test
    double g_tol, offset;
    const double *x0;                       // [K][d]
    const double *mean, *a, *wd, *gm;       // target (TargetDev layout)
    double *hs, *hy;                        // [K][J][EPT * NT] scratch ring (used when !hist_in_lds)
    double *tr_theta, *tr_grad, *tr_lp;     // staging trace [K][maxiters+1][d], [K][maxiters+1]
    int32_t *npts;                          // [K]
    // streaming (pfmi_stream_enqueue): the number of recorded points is PUBLISHED whenever it reaches a multiple of pub_mask + 1 -- to npts[k]
    // (device) and to h_prog[k] (page-locked HOST memory the calling thread polls: it launches the fits and scans of the points every path
    // has produced so far on the other CUs while the paths are still being optimised) -- and h_prog[K + k] is raised behind the final count.
    // Nobody on the device ever waits for these: the host is the scheduler.  pub_mask < 0: no publication (the packed route).
    int32_t *h_prog;                        // [2 K] host-visible (hipHostMalloc, coherent), or null
    int pub_mask;
};

// XGM (the 1024-thread variants, 128 registers per thread): the gradient g of the current point and the target's mean / diagonal are
// not held in registers but re-read where they are used -- g from the trace row the kernel has just written (record() stores -g of
// every accepted point; an L2 hit, needed once per iteration), so that x, p, xn, gn (4 EPT doubles) are the vectors a thread carries.
template <int EPT, int NT, int RPAD>
struct LbfgsState {
    static constexpr bool XGM = NT > 256;
    static constexpr int NS = XGM ? 1 : EPT;
    double x[EPT], g[NS], p[EPT], gn[EPT], mean[NS], av[NS];
    double a_last;                               // the trial point of the last evaluation is XN(e) = fma(a_last, p, x): recomputed, never stored
    __device__ __forceinline__ double XN(int e) const { return fma(a_last, p[e], x[e]); }
    const double *grow, *tmean, *ta;             // XGM: trace row of the current point's gradient, the target's vectors
    __device__ __forceinline__ double G(int e, int i, int d) const { if constexpr (XGM) return i < d ? -grow[i] : 0.0; else return g[e]; }
    __device__ __forceinline__ double M(int e, int i, int d) const { if constexpr (XGM) return (tmean && i < d) ? tmean[i] : 0.0; else return mean[e]; }
    __device__ __forceinline__ double AV(int e, int i, int d) const { if constexpr (XGM) return (ta && i < d) ? ta[i] : 0.0; else return av[e]; }
    // this thread's rows of the low-rank target factor, when they fit in registers (256-thread workgroups run one wave per SIMD: 512
    // registers each): both sweeps of every function evaluation used to re-load them from L2 inside the line search's critical path
    static constexpr bool WD_REG = (RPAD > 0) && (EPT * RPAD <= 64);
    double wdr[WD_REG ? EPT : 1][WD_REG ? RPAD : 1];
    const double *gm;                        // r x r factor (LDS copy)
    double fn;
#ifdef LB_PROF
    long long ec[4] = {0, 0, 0, 0}, et;
    int nev = 0;
#endif
};
#ifdef LB_PROF
#define LB_E0(S) (S).et = clock64()
#define LB_E(S, i) do { const long long now_ = clock64(); (S).ec[i] += now_ - (S).et; (S).et = now_; } while (0)
#else
#define LB_E0(S)
#define LB_E(S, i)
#endif

// element e of thread tid.  d <= 1024: tid + e NT (LDS ring: consecutive lanes, consecutive 8-byte slots).  XGM variants (ring and trace
// rows in global memory): thread tid owns the PAIRS 2 (tid + k NT) + {0, 1}, so that its two loads of a pair merge into one 16-byte
// load -- the ring sweeps of these variants are bound by bytes in flight per wave (43 GB/s per CU with 8-byte loads).
#define LB_IX(e) (LbfgsState<EPT, NT, RPAD>::XGM ? 2 * (tid + ((e) >> 1) * NT) + ((e) & 1) : tid + (e) * NT)

template <int RPAD> struct LbNv {
    static constexpr int EVAL = (2 * RPAD + 2 + 3) / 4 * 4 < 4 ? 4 : (2 * RPAD + 2 + 3) / 4 * 4;      // values of one function evaluation
};
template <int NT> struct LbCh { static constexpr int CH = NT <= 256 ? 6 : 2, LB = NT <= 256 ? 3 : 1; };       // ring pairs per fused reduction / per batch of loads                  // ring pairs per fused reduction
template <int NT, int RPAD> struct LbRed {
    static constexpr int G = 4 * LbCh<NT>::CH + 8;
    static constexpr int NVMAX = LbNv<RPAD>::EVAL > G ? LbNv<RPAD>::EVAL : G;
};

__device__ __forceinline__ double lb_rl(double v, int lane) {          // value of a (uniform) lane, in every lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Block-wide sums of NV values per thread (NV % 4 == 0, NV <= 64), one barrier (ping-pong scratch of 2 * (NT / 64) * NVMAX doubles, as
// pf_block_sum_mv).  Returns, in lane l of EVERY wave, the total of value l % NVP (NVP = NV rounded up to a power of two; garbage-free
// zero for l % NVP >= NV): wave stage = pf_block_sum_mv's butterfly; then lane l adds the partials of its value over the waves
// {g, g + G, ...} (g = l / NVP, G = 64 / NVP) in increasing order and the G groups are combined by xor butterflies (fp addition is
// commutative: all lanes of a value, in all waves, end with the same bits).
template <int NV, int NVMAX, int NT>
__device__ __forceinline__ double lb_block_sum(double (&v)[NV], double *red, int &flip) {
    static_assert(NV % 4 == 0 && NV <= 64, "lb_block_sum: NV must be a multiple of 4, at most 64");
    constexpr int NW = NT / 64, NVP = NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64, G = 64 / NVP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *buf = red + flip * (NW * NVMAX);
    flip ^= 1;
    double r[NV / 2], q[NV / 4];
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) r[j] = pf_swap32_add(v[j], v[j + NV / 2]);
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) q[j] = pf_swap16_add(r[j], r[j + NV / 4]);
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
        q[j] = pf_dpp_add<0x111, 0xf>(q[j]);   // row_shr:1
        q[j] = pf_dpp_add<0x112, 0xf>(q[j]);   // row_shr:2
        q[j] = pf_dpp_add<0x114, 0xf>(q[j]);   // row_shr:4
        q[j] = pf_dpp_add<0x118, 0xf>(q[j]);   // row_shr:8  -> lane 15 of each row
    }
    if ((lane & 15) == 15) {
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) buf[wave * NV + (lane >> 4) * (NV / 4) + j] = q[j];
    }
    __syncthreads();
    const int i = lane & (NVP - 1), g = lane / NVP;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < (NW + G - 1) / G; ++w) {
        const int ww = g + w * G;
        const double part = buf[(ww < NW ? ww : 0) * NV + (i < NV ? i : 0)];
        s += (ww < NW && i < NV) ? part : 0.0;
    }
    if (G >= 2) s = pf_add_xor32(s);
    if (G >= 4) s = pf_add_xor16(s);
    if (G >= 8) s += __shfl_xor(s, 8, 64);
    if (G >= 16) s += __shfl_xor(s, 4, 64);
    return s;
}

// f = -logp at xn = x + a p, gn = grad f(xn), dphi = gn . p  -- one block reduction
template <int EPT, int NT, int RPAD>
__device__ __forceinline__ void lb_eval(const LbfgsArgs &A, LbfgsState<EPT, NT, RPAD> &S, double a, double *red, int &flip, double &f,
                                        double &dphi) {
    const int tid = threadIdx.x, d = A.d;
    constexpr int NVP = LbNv<RPAD>::EVAL, NVMAX = LbRed<NT, RPAD>::NVMAX;
    double v[NVP];
    S.a_last = a;
#ifdef LB_PROF
    ++S.nev;
#endif
#pragma unroll
    for (int j = 0; j < NVP; ++j) v[j] = 0.0;
    if (A.kind == PFMI_TARGET_FUNNEL) {
        LB_E0(S);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = LB_IX(e);
            const double xn = S.XN(e);
            if (i == 0) { v[1] = xn; v[3] = S.p[e]; }
            else if (i < d) { v[0] += xn * xn; v[2] += xn * S.p[e]; }
        }
        LB_E(S, 0);
        const double tot = lb_block_sum<NVP, NVMAX, NT>(v, red, flip);
        LB_E(S, 1);
        const double ss = lb_rl(tot, 0), tau = lb_rl(tot, 1), xp = lb_rl(tot, 2), p0 = lb_rl(tot, 3);
        const double ee = exp(-tau), dm1 = (double)(d - 1);
        f = 0.5 * ((tau / 3.0) * (tau / 3.0) + dm1 * tau + ee * ss);
        const double g0v = 0.5 * (2.0 * tau / 9.0 + dm1 - ee * ss);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = LB_IX(e);
            S.gn[e] = (i == 0) ? g0v : (i < d ? ee * S.XN(e) : 0.0);
        }
        dphi = g0v * p0 + ee * xp;
        LB_E(S, 3);
    } else {
        double ev[EPT];
        LB_E0(S);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = LB_IX(e);
            ev[e] = S.XN(e) - S.M(e, i, d);
            const double ae = S.AV(e, i, d) * ev[e];
            v[0] += ae * ev[e];
            v[1] += ae * S.p[e];
            if (RPAD > 0 && i < d) {
                if constexpr (LbfgsState<EPT, NT, RPAD>::WD_REG) {
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) { v[2 + j] += S.wdr[e][j] * ev[e]; v[2 + RPAD + j] += S.wdr[e][j] * S.p[e]; }
                } else {
                    const double *row = A.wd + (size_t)i * RPAD;
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) { const double wj = row[j]; v[2 + j] += wj * ev[e]; v[2 + RPAD + j] += wj * S.p[e]; }
                }
            }
        }
        LB_E(S, 0);
        const double tot = lb_block_sum<NVP, NVMAX, NT>(v, red, flip);
        const double qa = lb_rl(tot, 0), qp = lb_rl(tot, 1);
        LB_E(S, 1);
        double corr = 0.0, hp = 0.0;
        double hh[RPAD > 0 ? RPAD : 1];
        if (RPAD > 0) {
            double we[RPAD > 0 ? RPAD : 1], gg[RPAD > 0 ? RPAD : 1];
#pragma unroll
            for (int j = 0; j < RPAD; ++j) we[j] = lb_rl(tot, 2 + j);
#pragma unroll
            for (int j = 0; j < RPAD; ++j) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l <= j; ++l) s += S.gm[j * RPAD + l] * we[l];
                gg[j] = s;
                corr += s * s;
            }
#pragma unroll
            for (int l = 0; l < RPAD; ++l) {
                double s = 0.0;
#pragma unroll
                for (int j = l; j < RPAD; ++j) s += S.gm[j * RPAD + l] * gg[j];
                hh[l] = s;
                hp += s * lb_rl(tot, 2 + RPAD + l);
            }
        }
        f = 0.5 * (qa - corr) - A.offset;
        dphi = qp - hp;                                     // (a e - W~ h)'p
        LB_E(S, 2);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = LB_IX(e);
            double gv = S.AV(e, i, d) * ev[e];
            if (RPAD > 0 && i < d) {
                if constexpr (LbfgsState<EPT, NT, RPAD>::WD_REG) {
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) gv -= S.wdr[e][j] * hh[j];
                } else {
                    const double *row = A.wd + (size_t)i * RPAD;
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) gv -= row[j] * hh[j];
                }
            }
            S.gn[e] = gv;
        }
        LB_E(S, 3);
    }
    S.fn = f;
}

// Strong-Wolfe line search: bracketing (step doubling) then bisection zoom, as pfmi/optimize.py.  ONE call site of lb_eval (a state
// machine instead of nested loops): the function evaluation is the bulk of the iteration's code; with the nested form (three inlined
// copies, 37 instead of 26 KB) the kernel is 9 % slower (1.64 against 1.50 ms at config 3: register allocation across the copies --
// NOT the instruction cache: SQC_ICACHE_MISSES stay ~1000 per launch either way, profiles/r03_experiments.md section 6).
template <int EPT, int NT, int RPAD>
__device__ __forceinline__ void lb_search(const LbfgsArgs &A, LbfgsState<EPT, NT, RPAD> &S, double *red, int &flip, double f0, double g0,
                                          double a_init) {
    const double c1 = 1e-4, c2 = 0.9, amax = 1e10;
    double a_prev = 0.0, f_prev = f0, a = a_init, lo = 0.0, hi = 0.0, f_lo = 0.0;
    int it = 0, zit = 0;
    bool zoom = false;
    for (;;) {
        double f, g;
        lb_eval<EPT, NT, RPAD>(A, S, a, red, flip, f, g);
        if (!zoom) {
            const int it0 = it++;
            if (!isfinite(f)) { a = 0.5 * (a_prev + a); if (it >= 25) return; continue; }
            if ((f > f0 + c1 * a * g0) || (it0 > 0 && f >= f_prev)) { zoom = true; lo = a_prev; hi = a; f_lo = f_prev; }
            else if (fabs(g) <= -c2 * g0) return;
            else if (g >= 0) { zoom = true; lo = a; hi = a_prev; f_lo = f; }
            else { a_prev = a; f_prev = f; a = fmin(2 * a, amax); if (it >= 25) return; continue; }
        } else {
            if ((f > f0 + c1 * a * g0) || (f >= f_lo)) {
                hi = a;
            } else {
                if (fabs(g) <= -c2 * g0) return;
                if (g * (hi - lo) >= 0) hi = lo;
                lo = a; f_lo = f;
            }
            if (++zit >= 30) return;
        }
        a = 0.5 * (lo + hi);
    }
}

// HL: the (s, y) ring lives in LDS
template <int EPT, int NT, int RPAD, bool HL>
__global__ __launch_bounds__(NT) void pf_lbfgs_kernel(LbfgsArgs A) {
    constexpr int CH = LbCh<NT>::CH, LB = LbCh<NT>::LB, NVG = LbRed<NT, RPAD>::G, NVMAX = LbRed<NT, RPAD>::NVMAX, DP = EPT * NT;
    extern __shared__ double lb_dyn[];
    __shared__ double red[2 * (NT / 64) * NVMAX];        // two halves: one barrier per block reduction
    int flip = 0;
    __shared__ double s_gm[RPAD > 0 ? RPAD * RPAD : 1];      // the target's r x r Cholesky factor: read in every function evaluation
    if (RPAD > 0) { for (int t = threadIdx.x; t < RPAD * RPAD; t += NT) s_gm[t] = A.gm[t]; __syncthreads(); }
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, d = A.d, J = A.J;
    // wave-private copy of the Gram data: SY[J][J] (s_c'y_c', slot indexed), YY[J][J], U[J] = S'g, W[J] = Y'g, RI[J] = 1 / (s_c'y_c).
    // Written by lanes of this wave, read by all of them: LDS operations of one wave execute in program order.
    const int gstride = 2 * J * J + 3 * J;
    double *GSY = lb_dyn + (size_t)(tid >> 6) * gstride, *GYY = GSY + J * J, *GU = GYY + J * J, *GW = GU + J, *GRI = GW + J;
    double *ring = lb_dyn + (size_t)(NT / 64) * gstride;            // LDS ring: s at [slot][DP], y behind the J slots of s
    double *const ghs = A.hs + (size_t)k * J * DP, *const ghy = A.hy + (size_t)k * J * DP;
    auto ld_s = [&](int slot, int i) -> double { if constexpr (HL) return ring[slot * DP + i]; else return ghs[(size_t)slot * DP + i]; };
    auto ld_y = [&](int slot, int i) -> double { if constexpr (HL) return ring[(J + slot) * DP + i]; else return ghy[(size_t)slot * DP + i]; };
    auto st_sy = [&](int slot, int i, double sv_, double yv_) {
        if constexpr (HL) { ring[slot * DP + i] = sv_; ring[(J + slot) * DP + i] = yv_; }
        else { ghs[(size_t)slot * DP + i] = sv_; ghy[(size_t)slot * DP + i] = yv_; }
    };
    const size_t tcap = (size_t)A.maxiters + 1;
    double *tr_theta = A.tr_theta + (size_t)k * tcap * d, *tr_grad = A.tr_grad + (size_t)k * tcap * d;
    double *tr_lp = A.tr_lp + (size_t)k * tcap;

    LbfgsState<EPT, NT, RPAD> S;
    constexpr bool XGM = LbfgsState<EPT, NT, RPAD>::XGM;
    S.gm = s_gm;
    S.grow = nullptr;
    S.tmean = A.kind == PFMI_TARGET_GAUSS ? A.mean : nullptr; S.ta = A.kind == PFMI_TARGET_GAUSS ? A.a : nullptr;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = LB_IX(e);
        const bool act = i < d;
        S.p[e] = 0.0;
        S.x[e] = act ? A.x0[(size_t)k * d + i] : 0.0;
        if constexpr (!XGM) {
            S.mean[e] = (act && A.kind == PFMI_TARGET_GAUSS) ? A.mean[i] : 0.0;
            S.av[e] = (act && A.kind == PFMI_TARGET_GAUSS) ? A.a[i] : 0.0;
        }
        if constexpr (LbfgsState<EPT, NT, RPAD>::WD_REG) {
#pragma unroll
            for (int j = 0; j < RPAD; ++j) S.wdr[e][j] = act ? A.wd[(size_t)i * RPAD + j] : 0.0;
        }
    }
    double f, dphi;
    lb_eval<EPT, NT, RPAD>(A, S, 0.0, red, flip, f, dphi);
    if constexpr (!XGM) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) S.g[e] = S.gn[e];
    }
    int n = 0, h = 0, head = 0;
    double gam = 1.0;
    auto record = [&]() {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = LB_IX(e);
            if (i < d) { tr_theta[(size_t)n * d + i] = S.x[e]; tr_grad[(size_t)n * d + i] = -S.gn[e]; }      // (gn: the gradient at x, see the callers)
        }
        if (tid == 0) tr_lp[n] = -f;
        S.grow = tr_grad + (size_t)n * d;                                      // (XGM: every thread re-reads only what it wrote)
        ++n;
        if (A.pub_mask >= 0 && (n & A.pub_mask) == 0) {
            // rows 0 .. n - 1 reach memory: every wave writes its own stores back (agent-scope fence), the barrier orders all of them before
            // thread 0 publishes the count.  The consumers are kernels the host launches AFTER it has read the count (a launch acquires).
            __threadfence();
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(A.npts + k, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                // RELEASE at system scope: the host must not see count n before npts[k] and the rows behind it are visible to the
                // kernels it then launches -- a relaxed store is not ordered after a release store to another address
                __hip_atomic_store(A.h_prog + k, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    record();
    // state of the current point that the fused reduction of the previous acceptance left behind: g'g, #{|g_i| > g_tol}, non-finite flag
    double gg, nbig, nbad;
    {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            v[0] += S.gn[e] * S.gn[e];
            v[1] += (fabs(S.gn[e]) > A.g_tol) ? 1.0 : 0.0;
            v[2] += isfinite(S.gn[e]) ? 0.0 : 1.0;
        }
        const double tot = lb_block_sum<4, NVMAX, NT>(v, red, flip);
        gg = lb_rl(tot, 0); nbig = lb_rl(tot, 1); nbad = lb_rl(tot, 2);
    }
#ifdef LB_PROF
    long long pc[7] = {0, 0, 0, 0, 0, 0, 0}, pt0 = clock64();
#define LB_T(i) do { const long long now_ = clock64(); pc[i] += now_ - pt0; pt0 = now_; } while (0)
#else
#define LB_T(i)
#endif
    for (int it = 0; it < A.maxiters; ++it) {
        LB_T(5);
        if (!isfinite(f) || nbad > 0.0) break;                       // src/optimize.jl:103-105
        if (nbig == 0.0) break;                                     // |g|_inf <= g_tol
        // ---- direction p = -H g
        double g0 = 0.0;
        if (h > 0) {
            // the two h x h triangular solves on the lanes of every wave (lane i < h = the pair of age i, 0 = oldest; run-time loops: the
            // unrolled per-thread form on registers was four times slower):  t = R^-1 (S'g);  a = R^-T ((D + gamma Y'Y) t - gamma Y'g);
            // H g = gamma g + S a - gamma Y t
            const int li = lane < h ? lane : 0;
            const int sl = head + li - (head + li >= J ? J : 0);
            double u = GU[sl];
            const double w = GW[sl], ri = GRI[sl], u0 = u, dd = GSY[sl * J + sl];
            double t = 0.0, acc = 0.0;
            auto slot_of = [&](int j) { const int jj = j < 0 ? 0 : (j >= h ? h - 1 : j); return head + jj - (head + jj >= J ? J : 0); };
            // (the Gram entries of step j -/+ 1 are fetched while step j computes: the recurrence is a chain of cross-lane reads and
            //  must not wait for an LDS round trip per step as well)
            double rij = GSY[sl * J + slot_of(h - 1)], yij = GYY[sl * J + slot_of(h - 1)];
            for (int j = h - 1; j >= 0; --j) {
                const int sn = slot_of(j - 1);
                const double rnext = GSY[sl * J + sn], ynext = GYY[sl * J + sn];
                const double tj = lb_rl(u, j) * lb_rl(ri, j);
                if (lane == j) t = tj;
                if (lane < j) u -= rij * tj;
                acc += yij * tj;
                rij = rnext; yij = ynext;
            }
            double z = dd * t + gam * (acc - w), ca = 0.0;
            double rji = GSY[slot_of(0) * J + sl];
            for (int j = 0; j < h; ++j) {
                const double rnext = GSY[slot_of(j + 1) * J + sl];
                const double aj = lb_rl(z, j) * lb_rl(ri, j);
                if (lane == j) ca = aj;
                if (lane > j) z -= rji * aj;
                rji = rnext;
            }
            double s1 = (lane < h) ? u0 * ca - gam * (w * t) : 0.0;          // sum over the h lanes, in lane order
            {
                double tot1 = 0.0;
                for (int i = 0; i < h; ++i) tot1 += lb_rl(s1, i);
                s1 = tot1;
            }
            g0 = -(gam * gg + s1);                                   // g'p
            LB_T(0);
            double q[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) q[e] = gam * S.G(e, LB_IX(e), d);
            if constexpr (XGM) {                                     // register-starved variants: no staging, one pair at a time
                for (int c = 0; c < h; ++c) {
                    const int slot = head + c - (head + c >= J ? J : 0);
                    const double cs = lb_rl(ca, c), cy = -gam * lb_rl(t, c);
                    if constexpr (EPT <= 20) {                        // all loads of the pair in flight before the first use
                        double sr[EPT], yr[EPT];                      // (32 rows per thread: the staging itself spills -- 167 against 134 us)
#pragma unroll
                        for (int e = 0; e < EPT; ++e) { sr[e] = ld_s(slot, LB_IX(e)); yr[e] = ld_y(slot, LB_IX(e)); }
#pragma unroll
                        for (int e = 0; e < EPT; ++e) q[e] += cs * sr[e] + cy * yr[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < EPT; ++e) q[e] += cs * ld_s(slot, LB_IX(e)) + cy * ld_y(slot, LB_IX(e));
                    }
                }
            } else
            for (int c0 = 0; c0 < h; c0 += LB) {                     // LB pairs per batch of loads
                double sr[LB][EPT], yr[LB][EPT], cs[LB], cy[LB];
#pragma unroll
                for (int cc = 0; cc < LB; ++cc) {
                    const bool on = c0 + cc < h;
                    const int c = on ? c0 + cc : 0;
                    const int slot = head + c - (head + c >= J ? J : 0);
                    cs[cc] = on ? lb_rl(ca, c) : 0.0;
                    cy[cc] = on ? -gam * lb_rl(t, c) : 0.0;
#pragma unroll
                    for (int e = 0; e < EPT; ++e) { sr[cc][e] = ld_s(slot, LB_IX(e)); yr[cc][e] = ld_y(slot, LB_IX(e)); }
                }
#pragma unroll
                for (int cc = 0; cc < LB; ++cc)
#pragma unroll
                    for (int e = 0; e < EPT; ++e) q[e] += cs[cc] * sr[cc][e] + cy[cc] * yr[cc][e];
            }
#pragma unroll
            for (int e = 0; e < EPT; ++e) S.p[e] = -q[e];
        }
        if (h == 0 || !(g0 < 0)) {                                   // first step, or not a descent direction: restart
            h = 0; head = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) S.p[e] = -S.G(e, LB_IX(e), d);
            g0 = -gg;
        }
        double a0 = 1.0;
        if (!h) a0 = fmin(1.0, 1.0 / fmax(sqrt(gg), 1e-300));
        LB_T(1);
        lb_search<EPT, NT, RPAD>(A, S, red, flip, f, g0, a0);
        LB_T(2);
        // ---- accept the last evaluated point: one fused reduction (per CH pairs of the ring) for the curvature test, the stopping
        //      tests and every inner product of the next direction.  Values 4 cc + {0, 1, 2, 3} = (s_c'y, y_c'y, s_c'g, y_c'g) of ring
        //      slot c = c0 + cc against the new pair (s, y) and the new gradient g; values 4 CH + {0..7} (first batch only) =
        //      s'y, y'y, #moved, #non-finite, g'g, #{|g_i| > g_tol}, s'g, y'g.
        double yv[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) yv[e] = S.gn[e] - S.G(e, LB_IX(e), d);
        const int slot_new = (h == J) ? head : head + h - (head + h >= J ? J : 0);       // h == J: the oldest pair is replaced
        double sy = 0.0, yy = 0.0, moved = 0.0;
        bool take = false, stop = false;
        for (int pass = 0; pass < 2 && !stop; ++pass) {              // second pass only when the pair fails the curvature test (rare):
            const int snew = pass == 0 ? slot_new : -1;              // S'g, Y'g of the unchanged ring
            for (int c0 = 0; c0 < J; c0 += CH) {
                double v[NVG];
#pragma unroll
                for (int j = 0; j < NVG; ++j) v[j] = 0.0;
                if constexpr (XGM) {
#pragma unroll
                    for (int cc = 0; cc < CH; ++cc) {
                        const int c = c0 + cc;
                        const int age = c - head + (c < head ? J : 0);
                        if (c < J && c != snew && age < h) {
                            constexpr int EB = EPT <= 20 ? EPT : 1;               // rows whose loads are issued together
#pragma unroll
                            for (int e0 = 0; e0 < EPT; e0 += EB) {
                                double sr[EB], yr[EB];
#pragma unroll
                                for (int e = 0; e < EB; ++e) { sr[e] = ld_s(c, LB_IX(e0 + e)); yr[e] = ld_y(c, LB_IX(e0 + e)); }
#pragma unroll
                                for (int e = 0; e < EB; ++e) {
                                    v[4 * cc + 0] += sr[e] * yv[e0 + e];
                                    v[4 * cc + 1] += yr[e] * yv[e0 + e];
                                    v[4 * cc + 2] += sr[e] * S.gn[e0 + e];
                                    v[4 * cc + 3] += yr[e] * S.gn[e0 + e];
                                }
                            }
                        }
                    }
                } else
#pragma unroll
                for (int cb = 0; cb < CH; cb += LB) {                          // LB pairs per batch of loads
                    double sr[LB][EPT], yr[LB][EPT];
                    bool on[LB];
#pragma unroll
                    for (int cc = 0; cc < LB; ++cc) {
                        const int c = c0 + cb + cc;
                        const int age = c - head + (c < head ? J : 0);
                        on[cc] = c < J && c != snew && age < h;               // a live pair of the ring (uniform)
                        if (on[cc]) {
#pragma unroll
                            for (int e = 0; e < EPT; ++e) { sr[cc][e] = ld_s(c, LB_IX(e)); yr[cc][e] = ld_y(c, LB_IX(e)); }
                        }
                    }
#pragma unroll
                    for (int cc = 0; cc < LB; ++cc) {
                        if (on[cc]) {
#pragma unroll
                            for (int e = 0; e < EPT; ++e) {
                                v[4 * (cb + cc) + 0] += sr[cc][e] * yv[e];
                                v[4 * (cb + cc) + 1] += yr[cc][e] * yv[e];
                                v[4 * (cb + cc) + 2] += sr[cc][e] * S.gn[e];
                                v[4 * (cb + cc) + 3] += yr[cc][e] * S.gn[e];
                            }
                        }
                    }
                }
                const bool first = c0 == 0 && pass == 0;
                if (first) {
#pragma unroll
                    for (int e = 0; e < EPT; ++e) {
                        const double xe = S.x[e], xne = S.XN(e), sve = xne - xe;
                        v[4 * CH + 0] += yv[e] * sve;
                        v[4 * CH + 1] += yv[e] * yv[e];
                        v[4 * CH + 2] += (xne != xe) ? 1.0 : 0.0;
                        v[4 * CH + 3] += isfinite(S.gn[e]) ? 0.0 : 1.0;
                        v[4 * CH + 4] += S.gn[e] * S.gn[e];
                        v[4 * CH + 5] += (fabs(S.gn[e]) > A.g_tol) ? 1.0 : 0.0;
                        v[4 * CH + 6] += sve * S.gn[e];
                        v[4 * CH + 7] += yv[e] * S.gn[e];
                    }
                }
                LB_T(3);
                const double tot = lb_block_sum<NVG, NVMAX, NT>(v, red, flip);
                LB_T(4);
                if (first) {
                    sy = lb_rl(tot, 4 * CH + 0); yy = lb_rl(tot, 4 * CH + 1); moved = lb_rl(tot, 4 * CH + 2); nbad = lb_rl(tot, 4 * CH + 3);
                    gg = lb_rl(tot, 4 * CH + 4); nbig = lb_rl(tot, 4 * CH + 5);
                    if (!isfinite(S.fn) || nbad > 0.0) { stop = true; break; }
                    take = sy > 1e-10 * yy && !(A.reject_every > 0 && (it + 1) % A.reject_every == 0);
                    if (!take) break;
                }
                // the lanes that own a total file it in this wave's Gram data
                const double ri_new = 1.0 / sy;                               // (uniform: every lane has s'y)
                if (lane < 4 * CH) {
                    const int cc = lane >> 2, t = lane & 3, c = c0 + cc;
                    const int age = c - head + (c < head ? J : 0);
                    if (c < J && c != snew && age < h) {
                        if (t == 2) GU[c] = tot;
                        else if (t == 3) GW[c] = tot;
                        else if (pass == 0) {
                            if (t == 0) GSY[c * J + slot_new] = tot;
                            else { GYY[c * J + slot_new] = tot; GYY[slot_new * J + c] = tot; }
                        }
                    }
                } else if (first && lane < NVG) {
                    const int e = lane - 4 * CH;
                    if (e == 0) { GSY[slot_new * J + slot_new] = tot; GRI[slot_new] = ri_new; }
                    else if (e == 1) GYY[slot_new * J + slot_new] = tot;
                    else if (e == 6) GU[slot_new] = tot;
                    else if (e == 7) GW[slot_new] = tot;
                }
            }
            if (take) break;
        }
        LB_T(6);
        if (stop) {                                                  // src/optimize.jl:96-105: the offending iterate is recorded, then the run stops
#pragma unroll
            for (int e = 0; e < EPT; ++e) S.x[e] = S.XN(e);
            S.a_last = 0.0;
            f = S.fn;
            record();
            break;
        }
        if (take) {
            if (h == J) head = head + 1 == J ? 0 : head + 1; else ++h;
#pragma unroll
            for (int e = 0; e < EPT; ++e) st_sy(slot_new, LB_IX(e), S.XN(e) - S.x[e], yv[e]);
            gam = sy / yy;
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) { S.x[e] = S.XN(e); if constexpr (!XGM) S.g[e] = S.gn[e]; }
        S.a_last = 0.0;                                              // XN(e) = x again (the accepted point)
        f = S.fn;
        record();
        if (!(moved > 0.0)) break;
    }
#ifdef LB_PROF
    if (tid == 0 && k == 0) printf("lbprof n=%d evals=%d cycles/iter: algebra %lld combine %lld search %lld (eval: sweep1 %lld reduce %lld small %lld sweep2 %lld) accept: sweep %lld reduce %lld writes %lld rest %lld\n", n, S.nev, pc[0] / n, pc[1] / n, pc[2] / n, S.ec[0] / n, S.ec[1] / n, S.ec[2] / n, S.ec[3] / n, pc[3] / n, pc[4] / n, pc[6] / n, pc[5] / n);
#endif
    if (A.pub_mask >= 0) {
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(A.npts + k, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.h_prog + k, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(A.h_prog + gridDim.x + k, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // final: behind the count
        }
    } else if (tid == 0) A.npts[k] = n;
}

// compaction: staging [K][cap][d] -> packed [P][d] (the layout pfmi_set_traces uploads)
__global__ void pf_trace_pack_kernel(int d, int64_t cap, const int64_t *__restrict__ off, const int32_t *__restrict__ path_of,
                                     const double *__restrict__ st_theta, const double *__restrict__ st_grad,
                                     const double *__restrict__ st_lp, double *__restrict__ theta, double *__restrict__ grad,
                                     double *__restrict__ lp) {
    const int64_t p = blockIdx.x;
    const int k = path_of[p];
    const int64_t l = p - off[k];
    const size_t src = ((size_t)k * cap + l) * d, dst = (size_t)p * d;
    for (int i = threadIdx.x; i < d; i += blockDim.x) { theta[dst + i] = st_theta[src + i]; grad[dst + i] = st_grad[src + i]; }
    if (threadIdx.x == 0) lp[p] = st_lp[(size_t)k * cap + l];
}

// ---------------------------------------------------------------------------------------------------
template <int EPT, int NT, int RPAD>
static int32_t launch_lb(pfmi_ctx *c, const LbfgsArgs &A, int K, size_t dyn) {
    auto kern = A.hist_in_lds ? pf_lbfgs_kernel<EPT, NT, RPAD, true> : pf_lbfgs_kernel<EPT, NT, RPAD, false>;
    if (dyn > 0) PF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kern, dim3(K), dim3(NT), dyn, c->stream, A);
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_lbfgs(pfmi_ctx *c, int K, int J, int maxiters, double g_tol, const double *d_x0, int pub_mask, int32_t *h_prog) {
    const TargetDev &T = c->target;
    const int d = T.d;
    LbfgsArgs A;
    A.d = d; A.J = J; A.maxiters = maxiters; A.kind = T.kind; A.r = T.r; A.g_tol = g_tol; A.offset = T.offset;
    A.x0 = d_x0;
    { const char *rj = pf_debug_get("PFMI_LBFGS_REJECT_EVERY"); A.reject_every = rj ? atoi(rj) : 0; }
    A.mean = T.mean.as<double>(); A.a = T.a.as<double>(); A.wd = T.wd.as<double>(); A.gm = T.g.as<double>();
    const int nt = d <= 256 ? 64 : d <= 1024 ? 256 : 512, ept = d <= 1024 ? 4 : d <= 10240 ? 20 : 32;
    const size_t hist_bytes = sizeof(double) * 2 * (size_t)J * nt * ept;          // rows padded to EPT * NT
    const size_t gram_bytes = sizeof(double) * (size_t)(nt / 64) * (2 * J * J + 3 * J);      // wave-private Gram data
    A.hist_in_lds = hist_bytes + gram_bytes <= 140 * 1024;
    if (!A.hist_in_lds) {
        PF_TRY(c->lb_hs.ensure(hist_bytes / 2 * K));
        PF_TRY(c->lb_hy.ensure(hist_bytes / 2 * K));
    }
    A.hs = c->lb_hs.as<double>(); A.hy = c->lb_hy.as<double>();
    A.tr_theta = c->st_theta.as<double>(); A.tr_grad = c->st_grad.as<double>(); A.tr_lp = c->st_lp.as<double>();
    A.npts = c->st_npts.as<int32_t>();
    A.h_prog = h_prog; A.pub_mask = h_prog ? pub_mask : -1;
    const size_t dyn = gram_bytes + (A.hist_in_lds ? hist_bytes : 0);
    const int rp = T.kind == PFMI_TARGET_GAUSS ? T.rpad : 0;
#define PF_LB(EPT, NT)                                                                    \
    (rp == 0 ? launch_lb<EPT, NT, 0>(c, A, K, dyn)                                        \
             : rp == 8 ? launch_lb<EPT, NT, 8>(c, A, K, dyn) : launch_lb<EPT, NT, 16>(c, A, K, dyn))
    if (d <= 256) return PF_LB(4, 64);                       // a single wave: block reductions need no cross-wave exchange
    if (d <= 1024) return PF_LB(4, 256);                     // (2 x 512 threads: 8.9 instead of 6.8 us per iteration -- the reductions span twice the waves)
    if (d <= 10240) return PF_LB(20, 512);                   // (10 x 1024 threads: 128 registers per thread, the state spills)
    PF_CHECK(d <= 16384, PFMI_ERR_UNSUPPORTED, "optimize_batch: d = %d > 16384 unsupported", d);
    return PF_LB(32, 512);
#undef PF_LB
}

int32_t pf_launch_trace_pack(pfmi_ctx *c, int64_t cap) {
    hipLaunchKernelGGL(pf_trace_pack_kernel, dim3((unsigned)c->P), dim3(256), 0, c->stream, c->d, cap, c->d_off.as<int64_t>(),
                       c->d_path_of.as<int32_t>(), c->st_theta.as<double>(), c->st_grad.as<double>(), c->st_lp.as<double>(),
                       c->theta.as<double>(), c->grad.as<double>(), c->trace_lp.as<double>());
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
