// fit_args.h -- argument block of the fit kernels (fit_kernels.hip)
#pragma once
#include <stdint.h>

struct FitArgs {
    int d, J;
    const int64_t *off;
    const int32_t *path_of;
    const double *theta, *grad, *alpha_all;
    const int32_t *hist_len, *hist_src;
    double *vh, *tmat, *vchol, *rq, *dmat, *sqrt_alpha, *mu, *logdet;
    int32_t *status;
    int64_t P;
};
