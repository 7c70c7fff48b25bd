// lbfgs_kernels.hip -- batched on-device L-BFGS trajectory generation.
//
// Plays the role of reference src/optimize.jl:35-121 (optimize_with_trace + OptimizationCallback): it PRODUCES the hot
// path's input, the trace (theta_l, logp_l, grad logp_l), directly in HBM so that fit_batch can consume it without a
// host round trip.  The reference delegates the optimisation itself to Optim.LBFGS + HagerZhang (third party); this is
// this repo's own driver -- two-loop recursion with gamma = s'y / y'y, strong-Wolfe bracketing + bisection zoom,
// maxiters as src/optimize.jl:40, stop at |g|_inf <= g_tol -- the same algorithm as pfmi/optimize.py (host, for
// callback targets); tests check it against the scalar C restatement under oracle/.  Built-in targets only
// (analytic gradients).
//
// One persistent workgroup per path (paths are independent, src/multipath.jl:190-208); every vector lives in
// registers (thread t owns elements t, t+NT, ...), the (s, y) ring lives in LDS when J*d*16 B fits and in an
// L2-resident scratch otherwise (element i of every history vector is only ever touched by its owner thread, so the
// ring needs no barriers).  All scalars that steer control flow come out of fixed-order block reductions and are
// bit-identical in every thread, so the whole workgroup walks the same line-search branches.
#include "pfmi_common.h"

struct LbfgsArgs {
    int d, J, maxiters, kind, r, hist_in_lds;
    double g_tol, offset;
    const double *x0;                       // [K][d]
    const double *mean, *a, *wd, *gm;       // target (TargetDev layout)
    double *hs, *hy;                        // [K][J][d] scratch ring (used when !hist_in_lds)
    double *tr_theta, *tr_grad, *tr_lp;     // staging trace [K][maxiters+1][d], [K][maxiters+1]
    int32_t *npts;                          // [K]
};

template <int EPT, int NT, int RPAD>
struct LbfgsState {
    double x[EPT], g[EPT], p[EPT], xn[EPT], gn[EPT], mean[EPT], av[EPT];
    // this thread's rows of the low-rank target factor, when they fit in registers (256-thread workgroups run one wave per SIMD: 512
    // registers each): both sweeps of every function evaluation used to re-load them from L2 inside the line search's critical path
    static constexpr bool WD_REG = (RPAD > 0) && (EPT * RPAD <= 64);
    double wdr[WD_REG ? EPT : 1][WD_REG ? RPAD : 1];
    const double *gm;                        // r x r factor (LDS copy)
    double fn;
};

// f = -logp at xn = x + a p, gn = grad f(xn), dphi = gn . p
template <int EPT, int NT, int RPAD>
__device__ __forceinline__ void lb_eval(const LbfgsArgs &A, LbfgsState<EPT, NT, RPAD> &S, double a, double *red, int &flip, double &f,
                                        double &dphi) {
    const int tid = threadIdx.x, d = A.d;
    constexpr int NVP = (RPAD + 2 + 3) / 4 * 4;            // padded to a multiple of 4: multi-value butterfly reduction
    double v[NVP];
#pragma unroll
    for (int j = 0; j < NVP; ++j) v[j] = 0.0;
    if (A.kind == PFMI_TARGET_FUNNEL) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * NT;
            S.xn[e] = S.x[e] + a * S.p[e];
            if (i == 0) v[1] = S.xn[e];
            else if (i < d) v[0] += S.xn[e] * S.xn[e];
        }
        pf_block_sum_pp<NVP, RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(v, red, flip);
        const double tau = v[1], ss = v[0], ee = exp(-tau), dm1 = (double)(d - 1);
        f = 0.5 * ((tau / 3.0) * (tau / 3.0) + dm1 * tau + ee * ss);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * NT;
            S.gn[e] = (i == 0) ? 0.5 * (2.0 * tau / 9.0 + dm1 - ee * ss) : (i < d ? ee * S.xn[e] : 0.0);
        }
    } else {
        double ev[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * NT;
            S.xn[e] = S.x[e] + a * S.p[e];
            ev[e] = S.xn[e] - S.mean[e];
            v[0] += S.av[e] * ev[e] * ev[e];
            if (RPAD > 0 && i < d) {
                if constexpr (LbfgsState<EPT, NT, RPAD>::WD_REG) {
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) v[2 + j] += S.wdr[e][j] * ev[e];
                } else {
                    const double *row = A.wd + (size_t)i * RPAD;
#pragma unroll
                    for (int j = 0; j < RPAD; ++j) v[2 + j] += row[j] * ev[e];
                }
            }
        }
        pf_block_sum_pp<NVP, RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(v, red, flip);
        double corr = 0.0;
        double hh[RPAD > 0 ? RPAD : 1];
        if (RPAD > 0) {
            double gg[RPAD > 0 ? RPAD : 1];
#pragma unroll
            for (int j = 0; j < RPAD; ++j) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l <= j; ++l) s += S.gm[j * RPAD + l] * v[2 + l];
                gg[j] = s;
                corr += s * s;
            }
#pragma unroll
            for (int l = 0; l < RPAD; ++l) {
                double s = 0.0;
#pragma unroll
                for (int j = l; j < RPAD; ++j) s += S.gm[j * RPAD + l] * gg[j];
                hh[l] = s;
            }
        }
        f = 0.5 * (v[0] - corr) - A.offset;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * NT;
            double gv = S.av[e] * ev[e];
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
    }
    double dp = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) dp += S.gn[e] * S.p[e];
    dphi = pf_block_sum1_pp<RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(dp, red, flip);
    S.fn = f;
}

template <int EPT, int NT, int RPAD>
__device__ __forceinline__ void lb_zoom(const LbfgsArgs &A, LbfgsState<EPT, NT, RPAD> &S, double *red, int &flip, double lo, double hi,
                                        double f_lo, double f0, double g0) {
    const double c1 = 1e-4, c2 = 0.9;
    for (int it = 0; it < 30; ++it) {
        const double a = 0.5 * (lo + hi);
        double f, g;
        lb_eval<EPT, NT, RPAD>(A, S, a, red, flip, f, g);
        if ((f > f0 + c1 * a * g0) || (f >= f_lo)) {
            hi = a;
        } else {
            if (fabs(g) <= -c2 * g0) return;
            if (g * (hi - lo) >= 0) hi = lo;
            lo = a; f_lo = f;
        }
    }
}

template <int EPT, int NT, int RPAD>
__device__ __forceinline__ void lb_search(const LbfgsArgs &A, LbfgsState<EPT, NT, RPAD> &S, double *red, int &flip, double f0, double g0,
                                          double a_init) {
    const double c1 = 1e-4, c2 = 0.9, amax = 1e10;
    double a_prev = 0.0, f_prev = f0, a = a_init;
    for (int it = 0; it < 25; ++it) {
        double f, g;
        lb_eval<EPT, NT, RPAD>(A, S, a, red, flip, f, g);
        if (!isfinite(f)) { a = 0.5 * (a_prev + a); continue; }
        if ((f > f0 + c1 * a * g0) || (it > 0 && f >= f_prev)) { lb_zoom<EPT, NT, RPAD>(A, S, red, flip, a_prev, a, f_prev, f0, g0); return; }
        if (fabs(g) <= -c2 * g0) return;
        if (g >= 0) { lb_zoom<EPT, NT, RPAD>(A, S, red, flip, a, a_prev, f, f0, g0); return; }
        a_prev = a; f_prev = f;
        a = fmin(2 * a, amax);
    }
}

template <int EPT, int NT, int RPAD>
__global__ __launch_bounds__(NT) void pf_lbfgs_kernel(LbfgsArgs A) {
    extern __shared__ double lb_dyn[];
    __shared__ double red[2 * (NT / 64) * (RPAD + 4)];   // two halves: one barrier per block reduction (pf_block_sum_pp)
    int flip = 0;
    __shared__ double s_rho[16], s_al[16];
    __shared__ double s_gm[RPAD > 0 ? RPAD * RPAD : 1];      // the target's r x r Cholesky factor: read in every function evaluation
    if (RPAD > 0) { for (int t = threadIdx.x; t < RPAD * RPAD; t += NT) s_gm[t] = A.gm[t]; __syncthreads(); }
    const int k = blockIdx.x, tid = threadIdx.x, d = A.d, J = A.J;
    double *hs = A.hist_in_lds ? lb_dyn : A.hs + (size_t)k * J * d;
    double *hy = A.hist_in_lds ? lb_dyn + (size_t)J * d : A.hy + (size_t)k * J * d;
    const size_t tcap = (size_t)A.maxiters + 1;
    double *tr_theta = A.tr_theta + (size_t)k * tcap * d, *tr_grad = A.tr_grad + (size_t)k * tcap * d;
    double *tr_lp = A.tr_lp + (size_t)k * tcap;

    LbfgsState<EPT, NT, RPAD> S;
    S.gm = s_gm;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * NT;
        const bool act = i < d;
        S.x[e] = act ? A.x0[(size_t)k * d + i] : 0.0;
        S.p[e] = 0.0;
        S.mean[e] = (act && A.kind == PFMI_TARGET_GAUSS) ? A.mean[i] : 0.0;
        S.av[e] = (act && A.kind == PFMI_TARGET_GAUSS) ? A.a[i] : 0.0;
        if constexpr (LbfgsState<EPT, NT, RPAD>::WD_REG) {
#pragma unroll
            for (int j = 0; j < RPAD; ++j) S.wdr[e][j] = act ? A.wd[(size_t)i * RPAD + j] : 0.0;
        }
    }
    double f, dphi;
    lb_eval<EPT, NT, RPAD>(A, S, 0.0, red, flip, f, dphi);
#pragma unroll
    for (int e = 0; e < EPT; ++e) S.g[e] = S.gn[e];
    int n = 0, h = 0, head = 0;
    double gam = 1.0;
    auto record = [&]() {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * NT;
            if (i < d) { tr_theta[(size_t)n * d + i] = S.x[e]; tr_grad[(size_t)n * d + i] = -S.g[e]; }
        }
        if (tid == 0) tr_lp[n] = -f;
        ++n;
    };
    record();
    for (int it = 0; it < A.maxiters; ++it) {
        double gm = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) gm = fmax(gm, isfinite(S.g[e]) ? fabs(S.g[e]) : INFINITY);
        gm = pf_block_max1_pp<RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(gm, red, flip);
        if (!isfinite(f) || !(gm < INFINITY)) break;                   // src/optimize.jl:103-105
        if (gm <= A.g_tol) break;
        // ---- two-loop recursion: q = H g
        double q[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) q[e] = S.g[e];
        for (int c = h - 1; c >= 0; --c) {
            const int slot = (head + c) % J;
            const double *s = hs + (size_t)slot * d, *y = hy + (size_t)slot * d;
            double sq = 0.0, yv[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int i = tid + e * NT;
                yv[e] = i < d ? y[i] : 0.0;
                sq += (i < d ? s[i] : 0.0) * q[e];
            }
            sq = pf_block_sum1_pp<RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(sq, red, flip);
            const double al = s_rho[slot] * sq;
            if (tid == 0) s_al[slot] = al;
#pragma unroll
            for (int e = 0; e < EPT; ++e) q[e] -= al * yv[e];
        }
        if (h) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) q[e] *= gam;
        }
        __syncthreads();                                               // s_al visible
        for (int c = 0; c < h; ++c) {
            const int slot = (head + c) % J;
            const double *s = hs + (size_t)slot * d, *y = hy + (size_t)slot * d;
            double yq = 0.0, sv[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int i = tid + e * NT;
                sv[e] = i < d ? s[i] : 0.0;
                yq += (i < d ? y[i] : 0.0) * q[e];
            }
            yq = pf_block_sum1_pp<RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(yq, red, flip);
            const double co = s_al[slot] - s_rho[slot] * yq;
#pragma unroll
            for (int e = 0; e < EPT; ++e) q[e] += co * sv[e];
        }
        double v2[2] = {0.0, 0.0};
#pragma unroll
        for (int e = 0; e < EPT; ++e) { S.p[e] = -q[e]; v2[0] += S.g[e] * S.p[e]; v2[1] += S.g[e] * S.g[e]; }
        pf_block_sum_pp<2, RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(v2, red, flip);
        double g0 = v2[0];
        if (g0 >= 0) {                                                  // not a descent direction: restart
            h = 0; head = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) S.p[e] = -S.g[e];
            g0 = -v2[1];
        }
        double a0 = 1.0;
        if (!h) a0 = fmin(1.0, 1.0 / fmax(sqrt(v2[1]), 1e-300));
        lb_search<EPT, NT, RPAD>(A, S, red, flip, f, g0, a0);
        // ---- accept the last evaluated point
        double v4[4] = {0.0, 0.0, 0.0, 0.0};
        double sv[EPT], yv[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            sv[e] = S.xn[e] - S.x[e];
            yv[e] = S.gn[e] - S.g[e];
            v4[0] += yv[e] * sv[e];
            v4[1] += yv[e] * yv[e];
            v4[2] += (S.xn[e] != S.x[e]) ? 1.0 : 0.0;
            v4[3] += isfinite(S.gn[e]) ? 0.0 : 1.0;
        }
        pf_block_sum_pp<4, RPAD + 4, (NT <= 256 ? NT / 64 : 0)>(v4, red, flip);
        if (!isfinite(S.fn) || v4[3] > 0.0) break;
        if (v4[0] > 1e-10 * v4[1]) {
            if (h == J) { head = (head + 1) % J; h = J - 1; }
            const int slot = (head + h) % J;
            double *s = hs + (size_t)slot * d, *y = hy + (size_t)slot * d;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int i = tid + e * NT;
                if (i < d) { s[i] = sv[e]; y[i] = yv[e]; }
            }
            if (tid == 0) s_rho[slot] = 1.0 / v4[0];
            gam = v4[0] / v4[1];
            ++h;
            __syncthreads();                                           // s_rho visible
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) { S.x[e] = S.xn[e]; S.g[e] = S.gn[e]; }
        f = S.fn;
        record();
        if (!(v4[2] > 0.0)) break;
    }
    if (tid == 0) A.npts[k] = n;
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
    auto kern = pf_lbfgs_kernel<EPT, NT, RPAD>;
    if (dyn > 0) PF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kern, dim3(K), dim3(NT), dyn, c->stream, A);
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_lbfgs(pfmi_ctx *c, int K, int J, int maxiters, double g_tol, const double *d_x0) {
    const TargetDev &T = c->target;
    const int d = T.d;
    LbfgsArgs A;
    A.d = d; A.J = J; A.maxiters = maxiters; A.kind = T.kind; A.r = T.r; A.g_tol = g_tol; A.offset = T.offset;
    A.x0 = d_x0;
    A.mean = T.mean.as<double>(); A.a = T.a.as<double>(); A.wd = T.wd.as<double>(); A.gm = T.g.as<double>();
    const size_t hist_bytes = sizeof(double) * 2 * (size_t)J * d;
    A.hist_in_lds = hist_bytes <= 144 * 1024;
    if (!A.hist_in_lds) {
        PF_TRY(c->lb_hs.ensure(hist_bytes / 2 * K));
        PF_TRY(c->lb_hy.ensure(hist_bytes / 2 * K));
    }
    A.hs = c->lb_hs.as<double>(); A.hy = c->lb_hy.as<double>();
    A.tr_theta = c->st_theta.as<double>(); A.tr_grad = c->st_grad.as<double>(); A.tr_lp = c->st_lp.as<double>();
    A.npts = c->st_npts.as<int32_t>();
    const size_t dyn = A.hist_in_lds ? hist_bytes : 0;
    const int rp = T.kind == PFMI_TARGET_GAUSS ? T.rpad : 0;
#define PF_LB(EPT, NT)                                                                    \
    (rp == 0 ? launch_lb<EPT, NT, 0>(c, A, K, dyn)                                        \
             : rp == 8 ? launch_lb<EPT, NT, 8>(c, A, K, dyn) : launch_lb<EPT, NT, 16>(c, A, K, dyn))
    if (d <= 256) return PF_LB(4, 64);                       // a single wave: block reductions need no cross-wave exchange
    if (d <= 1024) return PF_LB(4, 256);
    if (d <= 10240) return PF_LB(10, 1024);
    PF_CHECK(d <= 16384, PFMI_ERR_UNSUPPORTED, "optimize_batch: d = %d > 16384 unsupported", d);
    return PF_LB(16, 1024);
#undef PF_LB
}

int32_t pf_launch_trace_pack(pfmi_ctx *c, int64_t cap) {
    hipLaunchKernelGGL(pf_trace_pack_kernel, dim3((unsigned)c->P), dim3(256), 0, c->stream, c->d, cap, c->d_off.as<int64_t>(),
                       c->d_path_of.as<int32_t>(), c->st_theta.as<double>(), c->st_grad.as<double>(), c->st_lp.as<double>(),
                       c->theta.as<double>(), c->grad.as<double>(), c->trace_lp.as<double>());
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
