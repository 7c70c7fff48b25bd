// elbo_kernels.hip -- Monte Carlo ELBO over batches of fitted MvNormals (gfx950).
//
//   pf_elbo_draws_kernel : the dominant kernel.  rand_and_logpdf (reference src/mvnormal.jl:24-39)
//       fused with the target evaluation of elbo_and_samples (src/elbo.jl:12-16):
//         u ~ N(0, I) (counter-based Philox, or read from HBM in parity mode), |u|^2,
//         x = mu + U' Q [V'u_1; u_2]   (unwhiten!, src/woodbury.jl:401-406,136-143), logq, logp(x).
//       Mapping: one LANE per draw, rows of the d-vector walked sequentially.  Every coefficient of a
//       row (Householder row, sqrt(alpha_i), mu_i, target row) is wave-uniform -> scalar loads, the
//       per-draw state (Q'z accumulators, target accumulators) lives in VGPRs, and there is no
//       cross-lane reduction at all.  Q is applied in compact-WY form Q z = z - Vh (T (Vh' z)): pass 1
//       accumulates w = Vh' z, pass 2 regenerates z from the counter-based generator (nothing is
//       spilled to HBM) and emits x row by row.  Draws are written only when asked for (the winning
//       fit of a path); the ELBO scan over all iterations keeps them in registers.
//   pf_elbo_reduce_kernel : mean and corrected standard error of logp - logq per fit
//       (src/elbo.jl:16-18), fixed reduction tree.
//   pf_elbo_argmax_kernel : _findmax_skipnan per path (src/utils.jl:55-72).
//   pf_logpdf_kernel      : Distributions.logpdf of arbitrary points through the factor
//       (src/resample.jl:85-89 -> invquad, src/woodbury.jl:378-382,158-165).
#include "pfmi_common.h"
#include "elbo_args.h"
#include <stdlib.h>

#define ELBO_THREADS 256


// TGT: 0 none (host callback evaluates later), 1 Gaussian family with RPAD low-rank columns, 2 funnel
template <int TGT, int RPAD>
struct TargetAcc {
    double q = 0.0, tau = 0.0;
    double tr[RPAD > 0 ? RPAD : 1];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < (RPAD > 0 ? RPAD : 1); ++j) tr[j] = 0.0;
    }
    __device__ __forceinline__ void row(const ElboArgs &A, int i, double x) {
        if (TGT == 1) {
            const double e = x - A.t_mean[i];
            q += A.t_a[i] * e * e;
            if (RPAD > 0) {
                const double *wr = A.t_wd + (size_t)i * RPAD;
#pragma unroll
                for (int j = 0; j < RPAD; ++j) tr[j] += wr[j] * e;
            }
        } else if (TGT == 2) {
            if (i == 0) tau = x; else q += x * x;
        }
    }
    __device__ __forceinline__ double finish(const ElboArgs &A) {
        if (TGT == 1) {
            double corr = 0.0;
            if (RPAD > 0) {
#pragma unroll
                for (int j = 0; j < RPAD; ++j) {
                    double g = 0.0;
#pragma unroll
                    for (int l = 0; l <= j; ++l) g += A.t_g[j * RPAD + l] * tr[l];
                    corr += g * g;
                }
            }
            return A.t_offset - 0.5 * (q - corr);
        } else if (TGT == 2) {
            const double t3 = tau / 3.0;
            return (t3 * t3 + (double)(A.d - 1) * tau + q * exp(-tau)) / -2.0;
        }
        return NAN;
    }
};

template <int KPAD, int TGT, int RPAD, bool MEM>
__global__ __launch_bounds__(ELBO_THREADS) void pf_elbo_draws_kernel(ElboArgs A) {
    __shared__ double2 icdf_s[MEM ? 2 : 2 * PF_ICDF_LDS_ENTRIES];       // inverse-CDF table of the in-kernel generator
    if (!MEM) {
        pf_icdf_load(icdf_s);
        __syncthreads();
    }
    const int slot = blockIdx.y;
    const int p = A.points[slot];
    const int64_t nl = (int64_t)blockIdx.x * ELBO_THREADS + threadIdx.x;
    if (nl >= A.N) return;
    const int d = A.d;
    const size_t blk = A.by_point ? (size_t)p : (size_t)slot;
    double *out_lp = A.logp + blk * A.log_stride + nl;
    double *out_lq = A.logq + blk * A.log_stride + nl;
    if (A.status[p] != PFMI_FIT_OK) { *out_lp = NAN; *out_lq = NAN; return; }

    const double *__restrict__ Vh = A.vh + (size_t)p * d * KPAD;
    const double *__restrict__ T = A.tmat + (size_t)p * KPAD * KPAD;
    const double *__restrict__ Vc = A.vchol + (size_t)p * KPAD * KPAD;
    const double *__restrict__ sqa = A.sqrt_alpha + (size_t)p * d;
    const double *__restrict__ mu = A.mu + (size_t)p * d;
    const uint64_t seed = A.seeds[slot];
    const uint32_t n = (uint32_t)(A.n0 + nl);
    const double *U = MEM ? (A.u + blk * A.u_stride + (size_t)nl * d) : nullptr;
    double *X = A.x ? (A.x + (size_t)slot * A.x_stride + (size_t)nl * d) : nullptr;
    const int ngroups = (d + 3) >> 2;

    auto gen4 = [&](int g, double (&z)[4]) {
        if (MEM) {
#pragma unroll
            for (int t = 0; t < 4; ++t) z[t] = (4 * g + t < d) ? U[4 * g + t] : 0.0;
        } else {
            pf_randn4(seed, (uint32_t)g, n, 0u, icdf_s, z);
#pragma unroll
            for (int t = 0; t < 4; ++t) if (4 * g + t >= d) z[t] = 0.0;
        }
    };

    // ---- head rows 0..KPAD-1: z_head = V' u_head (lmul!(V', x[1:k]), src/woodbury.jl:139; V is
    //      identity-padded beyond k), kept in registers for pass 2
    double zh[KPAD], w[KPAD];
    double usq = 0.0;
#pragma unroll
    for (int g = 0; g < KPAD / 4; ++g) {
        double z4[4];
        if (g < ngroups) gen4(g, z4); else { z4[0] = z4[1] = z4[2] = z4[3] = 0.0; }
#pragma unroll
        for (int t = 0; t < 4; ++t) { zh[4 * g + t] = z4[t]; usq += z4[t] * z4[t]; }   // src/mvnormal.jl:31
    }
#pragma unroll
    for (int a = KPAD - 1; a >= 0; --a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b <= a; ++b) s += Vc[b * KPAD + a] * zh[b];
        zh[a] = s;
    }
#pragma unroll
    for (int j = 0; j < KPAD; ++j) w[j] = 0.0;
#pragma unroll
    for (int i = 0; i < KPAD; ++i) {
        if (i < d) {
            const double *row = Vh + (size_t)i * KPAD;
#pragma unroll
            for (int j = 0; j < KPAD; ++j) w[j] += row[j] * zh[i];
        }
    }
    // ---- pass 1 over the remaining rows: |u|^2 and w = Vh' z
    for (int g = KPAD / 4; g < ngroups; ++g) {
        double z4[4];
        gen4(g, z4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * g + t;
            if (i < d) {
                usq += z4[t] * z4[t];
                const double *row = Vh + (size_t)i * KPAD;
#pragma unroll
                for (int j = 0; j < KPAD; ++j) w[j] += row[j] * z4[t];
            }
        }
    }
    // ---- tv = T w  (Q = I - Vh T Vh')
    double tv[KPAD];
#pragma unroll
    for (int a = 0; a < KPAD; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = a; b < KPAD; ++b) s += T[a * KPAD + b] * w[b];
        tv[a] = s;
    }
    // ---- pass 2: x_i = mu_i + sqrt(alpha_i) (z_i - Vh[i,:] tv)   (lmul!(Q), lmul!(U'), .+= mu)
    TargetAcc<TGT, RPAD> tg;
    tg.init();
#pragma unroll
    for (int i = 0; i < KPAD; ++i) {
        if (i < d) {
            const double *row = Vh + (size_t)i * KPAD;
            double v = zh[i];
#pragma unroll
            for (int j = 0; j < KPAD; ++j) v -= row[j] * tv[j];
            const double xi = mu[i] + sqa[i] * v;
            tg.row(A, i, xi);
            if (X) X[i] = xi;
        }
    }
    for (int g = KPAD / 4; g < ngroups; ++g) {
        double z4[4];
        gen4(g, z4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * g + t;
            if (i < d) {
                const double *row = Vh + (size_t)i * KPAD;
                double v = z4[t];
#pragma unroll
                for (int j = 0; j < KPAD; ++j) v -= row[j] * tv[j];
                const double xi = mu[i] + sqa[i] * v;
                tg.row(A, i, xi);
                if (X) X[i] = xi;
            }
        }
    }
    *out_lq = ((double)d * PF_LOG2PI + A.logdet[p] + usq) / -2.0;       // src/mvnormal.jl:36
    *out_lp = tg.finish(A);
}

// ---------------------------------------------------------------------------------------------------
// mean(logr), sqrt(var(logr; corrected) / N)  per fit.  Grid = P points.   src/elbo.jl:16-18
__global__ __launch_bounds__(256) void pf_elbo_reduce_kernel(int64_t N, const int64_t *__restrict__ off,
                                                             const int32_t *__restrict__ path_of,
                                                             const int32_t *__restrict__ status,
                                                             const double *__restrict__ logp,
                                                             const double *__restrict__ logq,
                                                             double *__restrict__ elbo, double *__restrict__ se,
                                                             int64_t vcap, int seg_len, const int32_t *__restrict__ npts) {
    // (seg_len > 0: the streaming layout -- block b is position b % seg_len of path b / seg_len, slot k * vcap + l; positions a path never
    //  reached keep the NaN they were initialised with)
    int p = blockIdx.x;
    const int tid = threadIdx.x;
    if (seg_len > 0) {
        const int k = blockIdx.x / seg_len, l = blockIdx.x - k * seg_len;
        if (l >= npts[k]) return;
        p = (int)((int64_t)k * vcap + l);
    }
    __shared__ double red[8];
    const bool first = ((int64_t)p == off[path_of[p]]);
    if (first || status[p] != PFMI_FIT_OK || N <= 0) {
        if (tid == 0) { elbo[p] = NAN; se[p] = NAN; }
        return;
    }
    const double *lp = logp + (size_t)p * N, *lq = logq + (size_t)p * N;
    double s = 0.0;
    for (int64_t n = tid; n < N; n += 256) s += lp[n] - lq[n];
    s = pf_block_sum1(s, red);
    const double mean = s / (double)N;
    double v = 0.0;
    for (int64_t n = tid; n < N; n += 256) { const double r = (lp[n] - lq[n]) - mean; v += r * r; }
    v = pf_block_sum1(v, red);
    if (tid == 0) {
        elbo[p] = mean;
        se[p] = sqrt((v / (double)(N - 1)) / (double)N);
    }
}

// _findmax_skipnan over the iterations 1..L of each path (first element seeds the state even if NaN,
// later NaNs are skipped, strict > so the first maximum wins).  1-based result, 0 when L == 0.
// One wave per path (round 1 walked each path with one thread: ~175 dependent loads, 65 us): lane-local candidates, then a
// butterfly that keeps the larger value and, among equal values, the smaller index.
// (npts != nullptr: the streaming layout -- path k owns npts[k] points from off[k] on, not everything up to off[k + 1])
__global__ __launch_bounds__(64) void pf_elbo_argmax_kernel(int K, const int64_t *__restrict__ off, const int32_t *__restrict__ npts,
                                                           const double *__restrict__ elbo, int64_t *__restrict__ best_iter) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= K) return;
    const int64_t p0 = off[k];
    const int L = npts ? npts[k] - 1 : (int)(off[k + 1] - p0 - 1);
    if (L <= 0) { if (lane == 0) best_iter[k] = 0; return; }
    double xmax = 0.0;
    int imax = 0x7FFFFFFF, have = 0;
    for (int l = 1 + lane; l <= L; l += 64) {
        const double xi = elbo[p0 + l];
        if (isnan(xi)) continue;
        if (!have || xi > xmax) { xmax = xi; imax = l; have = 1; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double xo = __shfl_xor(xmax, o, 64);
        const int io = __shfl_xor(imax, o, 64), ho = __shfl_xor(have, o, 64);
        const bool take = ho && (!have || xo > xmax || (xo == xmax && io < imax));
        if (take) { xmax = xo; imax = io; have = 1; }
    }
    if (lane == 0) best_iter[k] = have ? imax : 1;       // all NaN: the first element seeded the state
}

// log_ratios = logp - logq
__global__ void pf_logratio_kernel(int64_t n, const double *__restrict__ lp, const double *__restrict__ lq,
                                   double *__restrict__ lr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lr[i] = lp[i] - lq[i];
}

// dst[points[s] * N + n] = src[s * N + n]: the callback path's per-fit log densities into the point-indexed table (one launch
// instead of one device-to-device copy per fit)
// (a failed fit has no draws: its closure values are computed on stale memory and replaced by NaN, like an exception in the reference)
__global__ void pf_scatter_rows_kernel(int64_t ns, int64_t N, const int32_t *__restrict__ points, const int32_t *__restrict__ status,
                                       const double *__restrict__ src, double *__restrict__ dst) {
    const int64_t s = blockIdx.y;
    if (s >= ns) return;
    const int p = points[s];
    const bool ok = status[p] == PFMI_FIT_OK;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
        dst[(size_t)p * N + n] = ok ? src[(size_t)s * N + n] : NAN;
}
__global__ void pf_nan_failed_kernel(int64_t ns, int64_t N, const int32_t *__restrict__ points, const int32_t *__restrict__ status,
                                     double *__restrict__ lp) {
    const int64_t s = blockIdx.y;
    if (s >= ns || status[points[s]] == PFMI_FIT_OK) return;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
        lp[(size_t)s * N + n] = NAN;
}

// Winners of the ELBO scan picked on the device (pfmi_pool_build_best): fit_distributions[fit_iteration + 1] (src/singlepath.jl:224),
// success = L > 0 && ELBO finite and != -Inf (src/singlepath.jl:299, 309-314); a successful path reuses the seed of its winning
// fit (src/singlepath.jl:226-230), a failed one takes fail_seeds[k] (rand(rng, fit_distribution, ndraws), :231-233) -- or, streaming
// layout without fail_seeds, the value its run's predrawn stream holds behind the L it consumed: stream_tab[k * vcap + L].
__global__ void pf_pool_pick_kernel(int K, const int64_t *__restrict__ off, const int32_t *__restrict__ npts, const int64_t *__restrict__ best_iter,
                                    const double *__restrict__ elbo, const uint64_t *__restrict__ seeds,
                                    const uint64_t *__restrict__ fail_seeds, const uint64_t *__restrict__ stream_tab, int64_t vcap,
                                    int32_t *__restrict__ points, uint64_t *__restrict__ pseeds, int32_t *__restrict__ ok) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int64_t p0 = off[k], L = npts ? (int64_t)npts[k] - 1 : off[k + 1] - p0 - 1, b = best_iter[k];
    const int64_t p = p0 + b;
    const double v = (b > 0) ? elbo[p] : NAN;
    const int good = (L > 0 && b > 0 && !isnan(v) && v != -INFINITY) ? 1 : 0;
    points[k] = (int32_t)p;
    pseeds[k] = good ? seeds[p] : fail_seeds ? fail_seeds[k] : (stream_tab && L >= 0) ? stream_tab[(int64_t)k * vcap + L] : seeds[p];
    ok[k] = good;
}

// ---------------------------------------------------------------------------------------------------
// logpdf(MvNormal(mu, W), x) = -(d log2pi + logdet)/2 - |L \ (x - mu)|^2 / 2, one lane per column.
// ldiv!(L): z = U'^{-1}(x - mu); z <- Q'z = z - Vh T'(Vh' z); z[1:k] <- V'^{-1} z[1:k]  (src/woodbury.jl:158-165)
template <int KPAD>
__global__ __launch_bounds__(ELBO_THREADS) void pf_logpdf_kernel(int d, int p, int64_t N, const double *__restrict__ Xall,
                                                                const double *__restrict__ vh, const double *__restrict__ tmat,
                                                                const double *__restrict__ vchol, const double *__restrict__ sqrt_alpha,
                                                                const double *__restrict__ mu_all, const double *__restrict__ logdet,
                                                                const int32_t *__restrict__ status, double *__restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * ELBO_THREADS + threadIdx.x;
    if (n >= N) return;
    if (status[p] != PFMI_FIT_OK) { out[n] = NAN; return; }
    const double *Vh = vh + (size_t)p * d * KPAD, *T = tmat + (size_t)p * KPAD * KPAD, *Vc = vchol + (size_t)p * KPAD * KPAD;
    const double *sqa = sqrt_alpha + (size_t)p * d, *mu = mu_all + (size_t)p * d;
    const double *X = Xall + (size_t)n * d;
    double w[KPAD], tv[KPAD], zh[KPAD];
#pragma unroll
    for (int j = 0; j < KPAD; ++j) w[j] = 0.0;
    for (int i = 0; i < d; ++i) {
        const double e = (X[i] - mu[i]) / sqa[i];
        const double *row = Vh + (size_t)i * KPAD;
#pragma unroll
        for (int j = 0; j < KPAD; ++j) w[j] += row[j] * e;
    }
#pragma unroll
    for (int a = 0; a < KPAD; ++a) {   // tv = T' w
        double s = 0.0;
#pragma unroll
        for (int b = 0; b <= a; ++b) s += T[b * KPAD + a] * w[b];
        tv[a] = s;
    }
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < KPAD; ++i) {
        zh[i] = 0.0;
        if (i < d) {
            const double *row = Vh + (size_t)i * KPAD;
            double v = (X[i] - mu[i]) / sqa[i];
#pragma unroll
            for (int j = 0; j < KPAD; ++j) v -= row[j] * tv[j];
            zh[i] = v;
        }
    }
    for (int i = KPAD; i < d; ++i) {
        const double *row = Vh + (size_t)i * KPAD;
        double v = (X[i] - mu[i]) / sqa[i];
#pragma unroll
        for (int j = 0; j < KPAD; ++j) v -= row[j] * tv[j];
        ss += v * v;
    }
#pragma unroll
    for (int a = 0; a < KPAD; ++a) {   // forward substitution V' y = zh (identity padded)
        double v = zh[a];
#pragma unroll
        for (int b = 0; b < a; ++b) v -= Vc[b * KPAD + a] * zh[b];
        zh[a] = v / Vc[a * KPAD + a];
        ss += zh[a] * zh[a];
    }
    out[n] = -((double)d * PF_LOG2PI + logdet[p]) / 2.0 - ss / 2.0;
}

// ---------------------------------------------------------------------------------------------------
template <int KPAD, int TGT, int RPAD>
static void launch_draws_tm(const ElboArgs &a, dim3 grid, hipStream_t s, bool mem) {
    if (mem) hipLaunchKernelGGL((pf_elbo_draws_kernel<KPAD, TGT, RPAD, true>), grid, dim3(ELBO_THREADS), 0, s, a);
    else hipLaunchKernelGGL((pf_elbo_draws_kernel<KPAD, TGT, RPAD, false>), grid, dim3(ELBO_THREADS), 0, s, a);
}
template <int KPAD>
static int32_t launch_draws_k(const ElboArgs &a, dim3 grid, hipStream_t s, bool mem, int tgt, int rpad) {
    if (tgt == 0) launch_draws_tm<KPAD, 0, 0>(a, grid, s, mem);
    else if (tgt == 2) launch_draws_tm<KPAD, 2, 0>(a, grid, s, mem);
    else if (rpad == 0) launch_draws_tm<KPAD, 1, 0>(a, grid, s, mem);
    else if (rpad == 8) launch_draws_tm<KPAD, 1, 8>(a, grid, s, mem);
    else if (rpad == 16) launch_draws_tm<KPAD, 1, 16>(a, grid, s, mem);
    else { pf_set_error("unsupported target rank padding %d", rpad); return PFMI_ERR_UNSUPPORTED; }
    return PFMI_OK;
}

int32_t pf_launch_elbo_draws(pfmi_ctx *c, const int32_t *d_points, const uint64_t *d_seeds, int64_t nfits,
                             int64_t n0, int64_t N, const double *d_u, int64_t u_stride, double *d_x,
                             int64_t x_stride, double *d_logp, double *d_logq, int64_t log_stride,
                             bool with_target, bool by_point) {
    if (nfits <= 0 || N <= 0) return PFMI_OK;
    ElboArgs a;
    a.d = c->d; a.points = d_points; a.seeds = d_seeds; a.n0 = n0; a.N = N;
    a.vh = c->vh.as<double>(); a.tmat = c->tmat.as<double>(); a.vchol = c->vchol.as<double>();
    a.sqrt_alpha = c->sqrt_alpha.as<double>(); a.mu = c->mu.as<double>(); a.logdet = c->logdet.as<double>();
    a.status = c->status.as<int32_t>();
    a.u = d_u; a.u_stride = u_stride; a.x = d_x; a.x_stride = x_stride;
    a.logp = d_logp; a.logq = d_logq; a.log_stride = log_stride; a.by_point = by_point ? 1 : 0;
    const TargetDev &t = c->target;
    a.t_mean = t.mean.as<double>(); a.t_a = t.a.as<double>(); a.t_wd = t.wd.as<double>(); a.t_g = t.g.as<double>(); a.t_wd16 = t.wd16.as<double>();
    a.t_offset = t.offset;
    int tgt = 0, rpad = 0;
    if (with_target) {
        if (t.kind == PFMI_TARGET_GAUSS) { tgt = 1; rpad = t.rpad; }
        else if (t.kind == PFMI_TARGET_FUNNEL) tgt = 2;
    }
    const int64_t gx = (N + ELBO_THREADS - 1) / ELBO_THREADS;
    const bool mem = d_u != nullptr;
    // production (in-kernel RNG) path: MFMA kernel with register-resident normals when the shape allows;
    // PFMI_ELBO_KERNEL=lane forces the general lane-per-draw kernel (used by the tests to cross-check)
    const char *force = pf_debug_get("PFMI_ELBO_KERNEL");
    if (!mem && !(force && force[0] == 'l')) {
        bool handled = false;
        pf_kernel_begin(c);
        int32_t rc = PFMI_OK;
        // ELBO scans (no draws written) take the single-pass quadratic-form kernel: one wave per 16-draw group, so it wants
        // N >= 64; tiny scans (ndraws_elbo = 5 default) and draw-writing launches take the two-pass MFMA kernel, which
        // splits the rows of ONE 16-draw group over the waves (d <= 1024, J <= 8); PFMI_ELBO_KERNEL = qf | mfma | lane forces one
        const bool mf_shape = a.d <= 1024 && c->kpad <= 16;
        const bool want_qf = (force && force[0] == 'q') || (!(force && force[0] == 'm') && (N >= 64 || (!mf_shape && N >= 16)));
        if (want_qf && !d_x) rc = pf_launch_elbo_qf(c, a, nfits, tgt, rpad, &handled);
        // launches that WRITE draws (pool, pfmi_draws, device-closure scans): the streaming writer (any d), then -- for a built-in
        // target -- the scan on the same (fit, seed, n0, N) for logp: a draw's logp through the expanded form, its logq the same bits
        // from both kernels.  PFMI_ELBO_KERNEL = xw forces it, = mfma | lane keep round 2's writers (tests cross-check them).
        // (a built-in target whose factor fits the two-pass kernel -- d <= 1024, J <= 8 -- keeps that kernel: it evaluates logp in the same
        // launch, 0.26 ms for the pool of config 3 against 0.49 ms for writer + scan)
        const bool want_xw = d_x && ((force && force[0] == 'x') || (!(force && force[0] == 'm') && !(tgt != 0 && mf_shape)));   // by shape, never by N:
        // a draw is the same bits whether it is made alone (top-up, pfmi_draws) or as part of a pool
        if (want_xw) {
            rc = pf_launch_elbo_xw(c, a, nfits, &handled);
            if (handled && rc == PFMI_OK && tgt != 0) {
                ElboArgs b = a;
                b.x = nullptr; b.x_stride = 0;
                bool h2 = false;
                rc = pf_launch_elbo_qf(c, b, nfits, tgt, rpad, &h2);
                if (rc == PFMI_OK && !h2) { pf_set_error("elbo_draws: scan kernel unavailable for kpad %d", c->kpad); rc = PFMI_ERR_UNSUPPORTED; }
            }
        }
        if (!handled) rc = pf_launch_elbo_mfma(c, a, nfits, tgt, rpad, &handled);
        if (handled) {
            pf_kernel_end(c, d_x ? "elbo_draws_x" : "elbo_draws");
            PF_TRY(rc);
            PF_HIP(hipGetLastError());
            return PFMI_OK;
        }
    }
    // gridDim.y is limited to 65535: chunk the fit list
    for (int64_t s0 = 0; s0 < nfits; s0 += 32768) {
        const int64_t ns = (nfits - s0 < 32768) ? (nfits - s0) : 32768;
        ElboArgs b = a;
        b.points = d_points + s0; b.seeds = d_seeds + s0;
        if (b.u && !by_point) b.u += s0 * u_stride;
        if (b.x) b.x += s0 * x_stride;
        if (!by_point) { b.logp += s0 * log_stride; b.logq += s0 * log_stride; }
        dim3 grid((unsigned)gx, (unsigned)ns);
        pf_kernel_begin(c);
        int32_t rc = PFMI_OK;
        switch (c->kpad) {
            case 4: rc = launch_draws_k<4>(b, grid, c->stream, mem, tgt, rpad); break;
            case 8: rc = launch_draws_k<8>(b, grid, c->stream, mem, tgt, rpad); break;
            case 12: rc = launch_draws_k<12>(b, grid, c->stream, mem, tgt, rpad); break;
            case 16: rc = launch_draws_k<16>(b, grid, c->stream, mem, tgt, rpad); break;
            case 20: rc = launch_draws_k<20>(b, grid, c->stream, mem, tgt, rpad); break;
            case 32: rc = launch_draws_k<32>(b, grid, c->stream, mem, tgt, rpad); break;
            case 64: rc = launch_draws_k<64>(b, grid, c->stream, mem, tgt, rpad); break;     // history_length 17 .. 32: this kernel only
            default: pf_set_error("unsupported kpad %d", c->kpad); rc = PFMI_ERR_UNSUPPORTED;
        }
        pf_kernel_end(c, d_x ? "elbo_draws_x" : "elbo_draws");
        PF_TRY(rc);
        PF_HIP(hipGetLastError());
    }
    return PFMI_OK;
}

int32_t pf_launch_elbo_reduce(pfmi_ctx *c) {
    pf_kernel_begin(c);
    int seg_len = 0;
    if (c->virt) for (int k = 0; k < c->K; ++k) if (c->npts_h[(size_t)k] > seg_len) seg_len = c->npts_h[(size_t)k];
    hipLaunchKernelGGL(pf_elbo_reduce_kernel, dim3((unsigned)(c->virt ? (int64_t)c->K * seg_len : c->P)), dim3(256), 0, c->stream, c->N_e,
                       c->d_off.as<int64_t>(), c->d_path_of.as<int32_t>(), c->status.as<int32_t>(),
                       c->logp.as<double>(), c->logq.as<double>(), c->elbo.as<double>(), c->se.as<double>(),
                       c->vcap, seg_len, c->st_npts.as<int32_t>());
    hipLaunchKernelGGL(pf_elbo_argmax_kernel, dim3((unsigned)c->K), dim3(64), 0, c->stream, c->K,
                       c->d_off.as<int64_t>(), c->virt ? c->st_npts.as<int32_t>() : (const int32_t *)nullptr, c->elbo.as<double>(),
                       c->best_iter.as<int64_t>());
    pf_kernel_end(c, "elbo_reduce");
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_scatter_rows(pfmi_ctx *c, int64_t ns, int64_t N, const int32_t *d_points, const double *d_src, double *d_dst) {
    if (ns <= 0 || N <= 0) return PFMI_OK;
    const unsigned gx = (unsigned)((N + 255) / 256 < 64 ? (N + 255) / 256 : 64);
    hipLaunchKernelGGL(pf_scatter_rows_kernel, dim3(gx, (unsigned)ns), dim3(256), 0, c->stream, ns, N, d_points, c->status.as<int32_t>(),
                       d_src, d_dst);
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_nan_failed(pfmi_ctx *c, int64_t ns, int64_t N, const int32_t *d_points, double *d_lp) {
    if (ns <= 0 || N <= 0) return PFMI_OK;
    const unsigned gx = (unsigned)((N + 255) / 256 < 64 ? (N + 255) / 256 : 64);
    for (int64_t s0 = 0; s0 < ns; s0 += 32768) {
        const int64_t n1 = ns - s0 < 32768 ? ns - s0 : 32768;
        hipLaunchKernelGGL(pf_nan_failed_kernel, dim3(gx, (unsigned)n1), dim3(256), 0, c->stream, n1, N, d_points + s0,
                           c->status.as<int32_t>(), d_lp + (size_t)s0 * N);
    }
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_pool_pick(pfmi_ctx *c, int have_fail_seeds) {
    hipLaunchKernelGGL(pf_pool_pick_kernel, dim3((unsigned)((c->K + 63) / 64)), dim3(64), 0, c->stream, c->K, c->d_off.as<int64_t>(),
                       c->virt ? c->st_npts.as<int32_t>() : (const int32_t *)nullptr, c->best_iter.as<int64_t>(), c->elbo.as<double>(), c->seeds.as<uint64_t>(),
                       have_fail_seeds ? c->fail_seeds.as<uint64_t>() : (const uint64_t *)nullptr,
                       c->virt ? c->d_stream_tab : (const uint64_t *)nullptr, c->vcap, c->pool_points.as<int32_t>(),
                       c->pool_seeds.as<uint64_t>(), c->pool_ok.as<int32_t>());
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_logratio(pfmi_ctx *c, int64_t n) {
    if (n <= 0) return PFMI_OK;
    hipLaunchKernelGGL(pf_logratio_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n,
                       c->pool_lp.as<double>(), c->pool_lq.as<double>(), c->pool_lr.as<double>());
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_logpdf(pfmi_ctx *c, int64_t point, int64_t N, const double *d_x, double *d_out) {
    if (N <= 0) return PFMI_OK;
    dim3 grid((unsigned)((N + ELBO_THREADS - 1) / ELBO_THREADS));
#define PF_LPDF(KP)                                                                                         \
    hipLaunchKernelGGL(pf_logpdf_kernel<KP>, grid, dim3(ELBO_THREADS), 0, c->stream, c->d, (int)point, N, d_x, \
                       c->vh.as<double>(), c->tmat.as<double>(), c->vchol.as<double>(),                     \
                       c->sqrt_alpha.as<double>(), c->mu.as<double>(), c->logdet.as<double>(),              \
                       c->status.as<int32_t>(), d_out)
    switch (c->kpad) {
        case 4: PF_LPDF(4); break;
        case 8: PF_LPDF(8); break;
        case 12: PF_LPDF(12); break;
        case 16: PF_LPDF(16); break;
        case 20: PF_LPDF(20); break;
        case 32: PF_LPDF(32); break;
        case 64: PF_LPDF(64); break;
        default: PF_CHECK(false, PFMI_ERR_UNSUPPORTED, "unsupported kpad %d", c->kpad);
    }
#undef PF_LPDF
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
