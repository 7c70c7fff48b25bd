// logp_kernels.hip -- USER-SIDE example of a PFMI_TARGET_DEVICE_CALLBACK closure (include/pfmi.h: pfmi_logp_dev_fn).
//
// The reference hands `logp` to the hot path as an arbitrary closure (src/elbo.jl:15, src/resample.jl:90-92).  A host closure costs
// a PCIe round trip per draw; a DEVICE closure is a function that enqueues the user's own kernel on the stream it is given:
//     void fn(const double *X_dev, int32_t d, int64_t n, double *out_dev, void *stream, void *user)
// X_dev is d x n column-major in HBM (a column = a draw), out_dev[n] receives logp.  This file is such a closure for the
// Gaussian family of SURVEY.md 8(d) (diagonal + rank-r, r <= 16) and for the funnel, written the way a user who cares about the
// HBM roofline would write it: every byte of X is read exactly once, fully coalesced (a wave reads 512 contiguous bytes of a
// column per load), the rank-r projection Wd'e runs on v_mfma_f64_16x16x4 (a true contraction over d), and nothing but out[n] is
// written.  It is NOT part of libpfmi: tests and bench.py load it as "the user's kernel" (the built-in targets never form x at all).
//
//   pfx_gauss_create / pfx_gauss_destroy   parameters -> device-resident handle (the closure's `user`)
//   pfx_gauss_logp                         the closure (pfmi_logp_dev_fn)
//   pfx_funnel_logp                        the closure for the funnel (user = NULL)
//   pfx_host_gauss_create / _destroy / pfx_host_gauss_logp    the same Gaussian family as a compiled HOST closure (pfmi_logp_fn): what a
//                                          C / Julia caller's `logp` looks like to the library -- re-entrant, so pfmi_set_callback_threads
//                                          (the reference's ntasks) may call it from several threads at once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

struct Gauss {
    int d, r, rows64;              // rows64 = d rounded up to a multiple of 64
    double offset;
    double *mean, *a;              // [rows64], zero padded
    double *wd16;                  // [rows64][16] row-major, zero padded: Wd = diag(a) W
    double *g;                     // [16][16] row-major lower triangular, zero padded
};

#define PFX_THREADS 256
#define PFX_WAVES 4
#define PFX_LD 66                  // leading dimension of a wave's transposition tile (64 rows + 2: conflict-free reads)

// One workgroup = 16 columns (draws).  The d rows are walked in 64-row tiles, wave w takes tiles w, w + 4, ...: lane l loads row
// r0 + l of the 16 columns (16 coalesced 512-byte reads), e = x - m, q_c += a e^2 stays in the lane; e goes through a wave-private
// LDS tile into the B-operand layout of v_mfma_f64_16x16x4 (lane (k, col) <- e[r0 + 4 s + k][col]) and t[j][col] += Wd[row][j] e
// accumulates on the matrix core, 16 MFMAs per tile.  At the end the four waves' partial t and q are summed through LDS and 16
// lanes finish logp = offset - (q - |G t|^2) / 2.
template <bool LOWRANK>
__global__ __launch_bounds__(PFX_THREADS) void pfx_gauss_kernel(Gauss P, const double *__restrict__ X, long long n, double *__restrict__ out) {
    __shared__ double tile[PFX_WAVES][16 * PFX_LD];
    __shared__ double red_t[PFX_WAVES][256];
    __shared__ double red_q[PFX_WAVES][16];
    __shared__ double tt[16][17];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, k = lane >> 4, col = lane & 15;
    const int d = P.d, ntiles = P.rows64 >> 6;
    for (long long grp = blockIdx.x; grp * 16 < n; grp += gridDim.x) {
        const long long c0 = grp * 16;
        double q[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) q[c] = 0.0;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        // software pipeline: the 16 loads of tile t + 4 are in flight while tile t is consumed (a wave then has 2 x 8 KB outstanding)
        double xn[16];
        auto fetch = [&](const int t, double (&x)[16]) {
            const int row = t * 64 + lane;
            const bool rv = t < ntiles && row < d;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const long long cc = c0 + c;
                x[c] = (rv && cc < n) ? __builtin_nontemporal_load(&X[(size_t)cc * d + row]) : 0.0;
            }
        };
        fetch(wv, xn);
        for (int t = wv; t < ntiles; t += PFX_WAVES) {
            const int row = t * 64 + lane;
            const double m = P.mean[row], av = P.a[row];
            double e[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) e[c] = (row < d && c0 + c < n) ? xn[c] - m : 0.0;       // (a NaN draw stays NaN)
            fetch(t + PFX_WAVES, xn);
#pragma unroll
            for (int c = 0; c < 16; ++c) q[c] = fma(av * e[c], e[c], q[c]);
            if (LOWRANK) {
                double *tl = tile[wv];
#pragma unroll
                for (int c = 0; c < 16; ++c) tl[c * PFX_LD + lane] = e[c];
                __builtin_amdgcn_wave_barrier();
                const double *wp = P.wd16 + ((size_t)t * 64 + k) * 16 + col;       // A[j = col][k] = Wd[row 4 s + k][j]
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const double b = tl[col * PFX_LD + 4 * s + k];                  // B[k][col] = e[4 s + k][col]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wp[(size_t)s * 64], b, acc, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- sums over the lanes of a wave (q) and over the four waves (q, t)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            double v = q[c];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            q[c] = v;
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) red_q[wv][c] = q[c];
        }
        if (LOWRANK) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red_t[wv][r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (LOWRANK && wv == 0) {                           // D layout: lane (k, col), register r holds t[j = k + 4 r][col]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = red_t[0][r * 64 + lane];
#pragma unroll
                for (int w = 1; w < PFX_WAVES; ++w) v += red_t[w][r * 64 + lane];
                tt[col][k + 4 * r] = v;
            }
        }
        __syncthreads();
        if (tid < 16 && c0 + tid < n) {
            double qq = red_q[0][tid];
#pragma unroll
            for (int w = 1; w < PFX_WAVES; ++w) qq += red_q[w][tid];
            double corr = 0.0;
            if (LOWRANK) {
                for (int j = 0; j < P.r; ++j) {
                    double gsum = 0.0;
                    for (int l = 0; l <= j; ++l) gsum = fma(P.g[j * 16 + l], tt[tid][l], gsum);
                    corr = fma(gsum, gsum, corr);
                }
            }
            out[c0 + tid] = P.offset - 0.5 * (qq - corr);
        }
        __syncthreads();
    }
}

// funnel (docs/src/examples/quickstart.md:229-234): logp = -[(tau/3)^2 + (d-1) tau + exp(-tau) sum_{i>=2} x_i^2] / 2, tau = x_1.
// One wave per column, coalesced; 4 columns per workgroup.
__global__ __launch_bounds__(PFX_THREADS) void pfx_funnel_kernel(int d, const double *__restrict__ X, long long n, double *__restrict__ out) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long long c = (long long)blockIdx.x * PFX_WAVES + wv; c < n; c += (long long)gridDim.x * PFX_WAVES) {
        const double *x = X + (size_t)c * d;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int i = 1 + lane;
        for (; i + 192 < d; i += 256) {
            const double v0 = x[i], v1 = x[i + 64], v2 = x[i + 128], v3 = x[i + 192];
            s0 = fma(v0, v0, s0); s1 = fma(v1, v1, s1); s2 = fma(v2, v2, s2); s3 = fma(v3, v3, s3);
        }
        for (; i < d; i += 64) { const double v = x[i]; s0 = fma(v, v, s0); }
        double s = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            const double tau = x[0], t3 = tau / 3.0;
            out[c] = (t3 * t3 + (double)(d - 1) * tau + s * exp(-tau)) / -2.0;
        }
    }
}

bool ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fprintf(stderr, "device_logp: %s failed: %s\n", what, hipGetErrorString(e));
    return false;
}

}  // namespace

extern "C" {

// mean[d], a[d] (diagonal precision part), Wd[d x r] column-major (= diag(a) W), G[r x r] column-major lower triangular:
// logp(x) = offset - ( e'diag(a) e - |G Wd'e|^2 ) / 2, e = x - mean  -- the parameterisation of PFMI_TARGET_GAUSS.
void *pfx_gauss_create(int32_t d, int32_t r, const double *mean, const double *a, const double *Wd, const double *G, double offset) {
    if (d < 1 || r < 0 || r > 16 || !mean || !a || (r > 0 && (!Wd || !G))) return nullptr;
    Gauss *P = new Gauss();
    P->d = d; P->r = r; P->rows64 = (d + 63) / 64 * 64; P->offset = offset;
    const size_t R = (size_t)P->rows64;
    std::vector<double> hm(R, 0.0), ha(R, 0.0), hw(R * 16, 0.0), hg(256, 0.0);
    for (int i = 0; i < d; ++i) { hm[(size_t)i] = mean[i]; ha[(size_t)i] = a[i]; }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < r; ++j) hw[(size_t)i * 16 + j] = Wd[i + (size_t)d * j];
    for (int j = 0; j < r; ++j)
        for (int l = 0; l <= j; ++l) hg[(size_t)j * 16 + l] = G[j + (size_t)r * l];
    bool good = ok(hipMalloc(&P->mean, R * 8), "hipMalloc") && ok(hipMalloc(&P->a, R * 8), "hipMalloc") &&
                ok(hipMalloc(&P->wd16, R * 16 * 8), "hipMalloc") && ok(hipMalloc(&P->g, 256 * 8), "hipMalloc");
    good = good && ok(hipMemcpy(P->mean, hm.data(), R * 8, hipMemcpyHostToDevice), "hipMemcpy") &&
           ok(hipMemcpy(P->a, ha.data(), R * 8, hipMemcpyHostToDevice), "hipMemcpy") &&
           ok(hipMemcpy(P->wd16, hw.data(), R * 16 * 8, hipMemcpyHostToDevice), "hipMemcpy") &&
           ok(hipMemcpy(P->g, hg.data(), 256 * 8, hipMemcpyHostToDevice), "hipMemcpy");
    if (!good) { delete P; return nullptr; }
    return P;
}

void pfx_gauss_destroy(void *h) {
    Gauss *P = reinterpret_cast<Gauss *>(h);
    if (!P) return;
    (void)hipFree(P->mean); (void)hipFree(P->a); (void)hipFree(P->wd16); (void)hipFree(P->g);
    delete P;
}

static unsigned grid_for(long long units) {
    const long long cap = 256LL * 12;                      // a few workgroups per CU, grid-stride beyond
    return (unsigned)(units < cap ? (units > 0 ? units : 1) : cap);
}

// pfmi_logp_dev_fn: enqueue on `stream`, never synchronise
void pfx_gauss_logp(const double *X_dev, int32_t d, int64_t n, double *out_dev, void *stream, void *user) {
    const Gauss *P = reinterpret_cast<const Gauss *>(user);
    if (!P || d != P->d || n <= 0) return;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = grid_for((n + 15) / 16);
    if (P->r > 0) hipLaunchKernelGGL(pfx_gauss_kernel<true>, dim3(grid), dim3(PFX_THREADS), 0, s, *P, X_dev, (long long)n, out_dev);
    else hipLaunchKernelGGL(pfx_gauss_kernel<false>, dim3(grid), dim3(PFX_THREADS), 0, s, *P, X_dev, (long long)n, out_dev);
}

void pfx_funnel_logp(const double *X_dev, int32_t d, int64_t n, double *out_dev, void *stream, void *user) {
    (void)user;
    if (n <= 0) return;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pfx_funnel_kernel, dim3(grid_for((n + PFX_WAVES - 1) / PFX_WAVES)), dim3(PFX_THREADS), 0, s, (int)d, X_dev,
                       (long long)n, out_dev);
}

}  // extern "C"


// ---- the same target as a compiled HOST closure (include/pfmi.h: pfmi_logp_fn) ---------------------------------------------------
// logp(x) = offset - (e' diag(a) e - |G Wd' e|^2) / 2, one column at a time like the reference calls its closure (src/elbo.jl:15);
// plain scalar loops, no shared mutable state: thread-safe as src/multipath.jl:104-108 requires when ntasks > 1.
namespace {
struct HostGauss { int d, r; double offset; std::vector<double> mean, a, wd, g; };   // wd [d][r] column-major, g [r][r] column-major lower
}
extern "C" void *pfx_host_gauss_create(int32_t d, int32_t r, const double *mean, const double *a, const double *Wd, const double *G, double offset) {
    if (d <= 0 || r < 0 || r > 64 || !mean || !a || (r > 0 && (!Wd || !G))) return nullptr;
    HostGauss *h = new HostGauss();
    h->d = d; h->r = r; h->offset = offset;
    h->mean.assign(mean, mean + d); h->a.assign(a, a + d);
    if (r > 0) { h->wd.assign(Wd, Wd + (size_t)d * r); h->g.assign(G, G + (size_t)r * r); }
    return h;
}
extern "C" void pfx_host_gauss_destroy(void *user) { delete static_cast<HostGauss *>(user); }
extern "C" void pfx_host_gauss_logp(const double *X, int32_t d, int64_t n, double *out, void *user) {
    const HostGauss *h = static_cast<const HostGauss *>(user);
    const int r = h->r;
    for (int64_t j = 0; j < n; ++j) {
        const double *x = X + (size_t)j * (size_t)d;
        double q = 0.0, t[64];
        for (int b = 0; b < r; ++b) t[b] = 0.0;
        for (int i = 0; i < d; ++i) {
            const double e = x[i] - h->mean[(size_t)i];
            q += h->a[(size_t)i] * e * e;
            for (int b = 0; b < r; ++b) t[b] += h->wd[(size_t)i + (size_t)d * b] * e;
        }
        double s = 0.0;
        for (int a_ = 0; a_ < r; ++a_) {
            double v = 0.0;
            for (int b = 0; b <= a_; ++b) v += h->g[(size_t)a_ + (size_t)r * b] * t[b];
            s += v * v;
        }
        out[j] = h->offset - 0.5 * (q - s);
    }
}
