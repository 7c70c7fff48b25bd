// fit_args.h -- argument block shared by the fit kernels (fit_kernels.hip, fit_cluster_kernel.hip)
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
    // cluster kernel only
    int64_t P;
    unsigned long long *cl_counter;   // [nclusters] arrival counters (zeroed before the launch)
    double *cl_buf;                   // [nclusters][2][nwg][CL_NVMAX] partial sums
    int cl_nwg, cl_nclusters;
};

// implemented in fit_cluster_kernel.hip: large-d fits, one CLUSTER of workgroups per fit; *handled = false when the shape is
// outside its range (the caller then uses the memory-resident kernel)
int32_t pf_launch_fit_cluster(struct pfmi_ctx *c, FitArgs a, bool *handled);
