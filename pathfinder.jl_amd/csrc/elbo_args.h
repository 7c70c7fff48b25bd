// elbo_args.h -- argument block shared by the two ELBO draw kernels (lane-per-draw and MFMA).
#pragma once
#include <stdint.h>

struct ElboArgs {
    int d;
    const int32_t *points;     // [nfits] trace point (= fit) per slot
    const uint64_t *seeds;     // [nfits]
    int64_t n0, N;             // draws n0 .. n0+N-1
    const double *vh, *tmat, *vchol, *sqrt_alpha, *mu, *logdet;
    const int32_t *status;
    const double *u;           // parity mode: slot s reads u + s*u_stride, d x N column-major
    int64_t u_stride;
    double *x;                 // optional: slot s writes x + s*x_stride, d x N column-major
    int64_t x_stride;
    double *logp, *logq;       // slot s writes + s*log_stride
    int64_t log_stride;
    int by_point;              // != 0: u / logp / logq blocks are indexed by the trace point, not the slot
    // target
    const double *t_mean, *t_a, *t_wd, *t_g;
    const double *t_wd16;      // [ceil(d/16)*16][16] zero-padded copy of t_wd (MFMA kernel)
    double t_offset;
};

// implemented in elbo_mfma_kernel.hip; returns PFMI_ERR_UNSUPPORTED when the shape is outside its range
int32_t pf_launch_elbo_mfma(struct pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad, bool *handled);
// implemented in elbo_qf_kernel.hip: single-pass quadratic-form scan (in-kernel RNG, no draws written), any d
int32_t pf_launch_elbo_qf(struct pfmi_ctx *c, const ElboArgs &a, int64_t nfits, int tgt, int rpad, bool *handled);
// implemented in elbo_xw_kernel.hip: the draw writer (x materialised in HBM, in-kernel RNG, logq; no target), any d
int32_t pf_launch_elbo_xw(struct pfmi_ctx *c, const ElboArgs &a, int64_t nfits, bool *handled);
