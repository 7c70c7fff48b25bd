// woodbury_kernels.hip -- the remaining WoodburyPDMat / PDMats operator surface on the device:
// unwhiten, whiten, invunwhiten, R*x, W*x, W\x, quad, invquad, diag   (reference src/woodbury.jl:129-165, 326-423).
// These are what the AdvancedHMC / DynamicHMC extensions call on a fitted metric (ext/PathfinderAdvancedHMCExt.jl:17-23,
// ext/PathfinderDynamicHMCExt.jl:7-15).  They reuse the factor produced by pf_fit_kernel: Q = I - Vh T Vh' (compact WY).
//
// One lane per column of X (d x N column-major), rows walked sequentially in two sweeps (w = Vh' z, then the
// rank-KPAD correction), every row coefficient wave-uniform.  Family A applies Q  (head transform first, T):
//   lmul!(L): x <- U' Q [V'x1; x2]      (:136-143)      ldiv!(R): x <- U^-1 Q [V^-1 x1; x2]   (:151-157)
// family B applies Q' (T', head transform last):
//   ldiv!(L): x <- [V'^-1 0;0 I] Q' U'^-1 x  (:158-165)  lmul!(R): x <- [V 0;0 I] Q' U x      (:129-135)
#include "pfmi_common.h"

#define WB_THREADS 256

// MODE 0: lmul_L (unwhiten)  1: ldiv_L (whiten)  2: lmul_R  3: ldiv_R (invunwhiten)
template <int KPAD, int MODE>
__global__ __launch_bounds__(WB_THREADS) void pf_woodbury_kernel(int d, int p, int64_t N, const double *__restrict__ Xin,
                                                                  double *__restrict__ Xout, const double *__restrict__ vh,
                                                                  const double *__restrict__ tmat, const double *__restrict__ vchol,
                                                                  const double *__restrict__ sqrt_alpha,
                                                                  const int32_t *__restrict__ status) {
    const int64_t n = (int64_t)blockIdx.x * WB_THREADS + threadIdx.x;
    if (n >= N) return;
    const double *X = Xin + (size_t)n * d;
    double *O = Xout + (size_t)n * d;
    if (status[p] != PFMI_FIT_OK) { for (int i = 0; i < d; ++i) O[i] = NAN; return; }
    const double *Vh = vh + (size_t)p * d * KPAD, *T = tmat + (size_t)p * KPAD * KPAD, *Vc = vchol + (size_t)p * KPAD * KPAD;
    const double *sqa = sqrt_alpha + (size_t)p * d;
    constexpr bool FAM_A = (MODE == 0 || MODE == 3);
    double zh[KPAD], w[KPAD], tv[KPAD];
    // ---- head block: rows 0 .. KPAD-1 (rows >= d are zero; V is identity padded beyond k)
#pragma unroll
    for (int i = 0; i < KPAD; ++i) {
        double v = (i < d) ? X[i] : 0.0;
        if (MODE == 1 && i < d) v /= sqa[i];
        if (MODE == 2 && i < d) v *= sqa[i];
        zh[i] = v;
    }
    if (MODE == 0) {                               // z_head = V' x_head
#pragma unroll
        for (int a = KPAD - 1; a >= 0; --a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b <= a; ++b) s += Vc[b * KPAD + a] * zh[b];
            zh[a] = s;
        }
    } else if (MODE == 3) {                        // z_head = V^-1 x_head (back substitution)
#pragma unroll
        for (int a = KPAD - 1; a >= 0; --a) {
            double s = zh[a];
#pragma unroll
            for (int b = a + 1; b < KPAD; ++b) s -= Vc[a * KPAD + b] * zh[b];
            zh[a] = s / Vc[a * KPAD + a];
        }
    }
    // ---- sweep 1: w = Vh' z
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
    for (int i = KPAD; i < d; ++i) {
        double v = X[i];
        if (MODE == 1) v /= sqa[i];
        if (MODE == 2) v *= sqa[i];
        const double *row = Vh + (size_t)i * KPAD;
#pragma unroll
        for (int j = 0; j < KPAD; ++j) w[j] += row[j] * v;
    }
    // ---- tv = T w (Q) or T' w (Q')
#pragma unroll
    for (int a = 0; a < KPAD; ++a) {
        double s = 0.0;
        if (FAM_A) {
#pragma unroll
            for (int b = a; b < KPAD; ++b) s += T[a * KPAD + b] * w[b];
        } else {
#pragma unroll
            for (int b = 0; b <= a; ++b) s += T[b * KPAD + a] * w[b];
        }
        tv[a] = s;
    }
    // ---- sweep 2: z - Vh tv, post scaling / head transform
#pragma unroll
    for (int i = 0; i < KPAD; ++i) {
        if (i < d) {
            const double *row = Vh + (size_t)i * KPAD;
            double v = zh[i];
#pragma unroll
            for (int j = 0; j < KPAD; ++j) v -= row[j] * tv[j];
            zh[i] = v;
        } else {
            zh[i] = 0.0;
        }
    }
    if (MODE == 1) {                               // head <- V'^-1 head (forward substitution)
#pragma unroll
        for (int a = 0; a < KPAD; ++a) {
            double s = zh[a];
#pragma unroll
            for (int b = 0; b < a; ++b) s -= Vc[b * KPAD + a] * zh[b];
            zh[a] = s / Vc[a * KPAD + a];
        }
    } else if (MODE == 2) {                        // head <- V head
#pragma unroll
        for (int a = 0; a < KPAD; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = a; b < KPAD; ++b) s += Vc[a * KPAD + b] * zh[b];
            zh[a] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < KPAD; ++i) {
        if (i < d) {
            double v = zh[i];
            if (MODE == 0) v *= sqa[i];
            if (MODE == 3) v /= sqa[i];
            O[i] = v;
        }
    }
    for (int i = KPAD; i < d; ++i) {
        double v = X[i];
        if (MODE == 1) v /= sqa[i];
        if (MODE == 2) v *= sqa[i];
        const double *row = Vh + (size_t)i * KPAD;
#pragma unroll
        for (int j = 0; j < KPAD; ++j) v -= row[j] * tv[j];
        if (MODE == 0) v *= sqa[i];
        if (MODE == 3) v /= sqa[i];
        O[i] = v;
    }
}

// out[n] = sum_i X[i, n]^2
__global__ void pf_colsumsq_kernel(int d, int64_t N, const double *__restrict__ X, double *__restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double *x = X + (size_t)n * d;
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += x[i] * x[i];
    out[n] = s;
}

// diag(W)_i = alpha_i + b_i' D b_i,  b_i = row i of B = [alpha.Y  S]   (src/woodbury.jl:326-329)
template <int KPAD>
__global__ void pf_woodbury_diag_kernel(int d, int J, int p, int64_t p0, const double *__restrict__ theta,
                                        const double *__restrict__ grad, const double *__restrict__ alpha_all,
                                        const int32_t *__restrict__ hist_len, const int32_t *__restrict__ hist_src,
                                        const double *__restrict__ dmat, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    const int j = hist_len[p], m = 2 * j;
    const double al = alpha_all[(size_t)p * d + i];
    const double *D = dmat + (size_t)p * KPAD * KPAD;
    double b[KPAD];
#pragma unroll
    for (int c = 0; c < KPAD; ++c) b[c] = 0.0;
#pragma unroll
    for (int c = 0; c < KPAD / 2; ++c) {
        if (c < j) {
            const int src = hist_src[(size_t)p * J + c];
            const size_t q0 = (size_t)(p0 + src) * d + i, q1 = q0 + d;
            const double by = al * (grad[q0] - grad[q1]), bs = theta[q1] - theta[q0];
#pragma unroll
            for (int cc = 0; cc < KPAD; ++cc) {
                if (cc == c) b[cc] = by;
                if (cc == j + c) b[cc] = bs;
            }
        }
    }
    double q = 0.0;
#pragma unroll
    for (int a = 0; a < KPAD; ++a) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < KPAD; ++c) s += D[a * KPAD + c] * b[c];
        q += b[a] * s;
    }
    (void)m;
    out[i] = al + q;
}

// ---------------------------------------------------------------------------------------------------
template <int KPAD>
static void launch_wb(pfmi_ctx *c, int mode, int64_t p, int64_t N, const double *in, double *out) {
    dim3 grid((unsigned)((N + WB_THREADS - 1) / WB_THREADS)), block(WB_THREADS);
#define PF_WB(M)                                                                                                   \
    hipLaunchKernelGGL((pf_woodbury_kernel<KPAD, M>), grid, block, 0, c->stream, c->d, (int)p, N, in, out,          \
                       c->vh.as<double>(), c->tmat.as<double>(), c->vchol.as<double>(), c->sqrt_alpha.as<double>(), \
                       c->status.as<int32_t>())
    if (mode == 0) PF_WB(0); else if (mode == 1) PF_WB(1); else if (mode == 2) PF_WB(2); else PF_WB(3);
#undef PF_WB
}

int32_t pf_launch_woodbury_prim(pfmi_ctx *c, int mode, int64_t p, int64_t N, const double *d_in, double *d_out) {
    if (N <= 0) return PFMI_OK;
    switch (c->kpad) {
        case 4: launch_wb<4>(c, mode, p, N, d_in, d_out); break;
        case 8: launch_wb<8>(c, mode, p, N, d_in, d_out); break;
        case 12: launch_wb<12>(c, mode, p, N, d_in, d_out); break;
        case 16: launch_wb<16>(c, mode, p, N, d_in, d_out); break;
        case 20: launch_wb<20>(c, mode, p, N, d_in, d_out); break;
        case 32: launch_wb<32>(c, mode, p, N, d_in, d_out); break;
        case 64: launch_wb<64>(c, mode, p, N, d_in, d_out); break;
        default: PF_CHECK(false, PFMI_ERR_UNSUPPORTED, "unsupported kpad %d", c->kpad);
    }
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_colsumsq(pfmi_ctx *c, int64_t N, const double *d_x, double *d_out) {
    if (N <= 0) return PFMI_OK;
    hipLaunchKernelGGL(pf_colsumsq_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, c->d, N, d_x, d_out);
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}

int32_t pf_launch_woodbury_diag(pfmi_ctx *c, int64_t p, double *d_out) {
    const int64_t p0 = c->off[(size_t)c->path_of[(size_t)p]];
    dim3 grid((unsigned)((c->d + 255) / 256)), block(256);
#define PF_WD(KP)                                                                                                   \
    hipLaunchKernelGGL(pf_woodbury_diag_kernel<KP>, grid, block, 0, c->stream, c->d, c->J, (int)p, p0,               \
                       c->th(), c->gr(), c->alpha_all.as<double>(),                       \
                       c->hist_len.as<int32_t>(), c->hist_src.as<int32_t>(), c->dmat.as<double>(), d_out)
    switch (c->kpad) {
        case 4: PF_WD(4); break;
        case 8: PF_WD(8); break;
        case 12: PF_WD(12); break;
        case 16: PF_WD(16); break;
        case 20: PF_WD(20); break;
        case 32: PF_WD(32); break;
        case 64: PF_WD(64); break;
        default: PF_CHECK(false, PFMI_ERR_UNSUPPORTED, "unsupported kpad %d", c->kpad);
    }
#undef PF_WD
    PF_HIP(hipGetLastError());
    return PFMI_OK;
}
