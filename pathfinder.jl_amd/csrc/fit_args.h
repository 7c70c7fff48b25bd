// fit_args.h -- argument block of the fit kernels (fit_kernels.hip)
#pragma once
#include <stdint.h>

struct FitArgs {
    int d, J;
    // streaming pipeline (pfmi_stream_enqueue): seg_len > 0 -- the launch covers the points l in [seg_l0, seg_l0 + seg_len) of every path,
    // workgroup index idx -> path k = idx / seg_len, l = seg_l0 + idx % seg_len, p = k * vcap + l; a point the path never reached
    // (l >= npts[k]) gets status PFMI_FIT_ABSENT and nothing else.  seg_len == 0: p = idx (the packed route).
    int seg_l0, seg_len;
    int64_t vcap;
    const int32_t *npts;
    const int64_t *off;
    const int32_t *path_of;
    const double *theta, *grad, *alpha_all;
    const int32_t *hist_len, *hist_src;
    double *vh, *tmat, *vchol, *rq, *dmat, *sqrt_alpha, *mu, *logdet;
    int32_t *status;
    int64_t P;
    double *big;                   // KPAD = 64 only: per-workgroup scratch [P][64][64] (the Gram block; the other small matrices live in the outputs)
};
#ifdef __HIPCC__
// work item idx of a fit launch -> trace point p; false: the path never reached that point (the caller marks it PFMI_FIT_ABSENT)
__device__ __forceinline__ bool pf_fit_point(const FitArgs &A, int64_t idx, int64_t &p) {
    if (A.seg_len == 0) { p = idx; return true; }
    const int k = (int)(idx / A.seg_len), l = A.seg_l0 + (int)(idx - (int64_t)k * A.seg_len);
    p = (int64_t)k * A.vcap + l;
    return l < __hip_atomic_load(A.npts + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// Segment of a walk (streaming pipeline, pfmi_stream_enqueue): the launch processes the steps l in [max(1, l_begin), l_end) of every path and
// hands the state of the recurrence to the next segment's launch -- alpha in its row of alpha_all, the CARRIED reciprocal in ial_state
// [K][d] (it is x * (1 / b), not 1 / alpha: it must travel bit for bit), the number of accepted updates in nacc_state[K], the accepted
// list in acc_list.  npts[k] (device) = points of path k recorded so far (>= l_end, or final).  The packed route passes
// {nullptr, 0, INT_MAX, nullptr, nullptr}: one launch walks the whole path, exactly as before.
// hinit: the `Hinit` keyword of lbfgs_inverse_hessians (src/inverse_hessian.jl:25): PFMI_HINIT_GILBERT (0, the default, :5-10) or
// PFMI_HINIT_SCALAR_YS_OVER_YY (1: alpha = fill(y's / y'y), the Nocedal-Wright scaling of test/inverse_hessian.jl:49).
struct HistSeg { const int32_t *npts; int l_begin, l_end; double *ial_state; int *nacc_state; int hinit; };
