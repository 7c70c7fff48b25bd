/* c_abi_demo.c -- the whole multipathfinder hot path through the C ABI of libpfmi.so, no Python, no torch:
 * what a Julia `ccall` (or cgo / JNI) binding does, in plain C.
 *
 *   gcc -O2 -I include examples/c_abi_demo.c -o c_abi_demo -L pathfinder.jl_amd/lib -lpfmi -Wl,-rpath,$PWD/pathfinder.jl_amd/lib -lm
 *
 * Target: N(m, diag(sigma^2)), d = 20, log sigma ~ U(-0.4, 0.4) (a mild "diagonal Gaussian", cf. BASELINE config 2).  The K traces are
 * made on the device (pfmi_optimize_batch); a host-made trace would go through pfmi_set_traces instead.
 * Prints one line per path and a final line "OK ..." after checking the invariants the reference's tests check. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pfmi.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int32_t rc_ = (call);                                                                \
        if (rc_ != PFMI_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pfmi_last_error()); return 1; } \
    } while (0)

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double unif(uint64_t *s) { return (double)(splitmix(s) >> 11) / 9007199254740992.0; }

int main(void) {
    enum { K = 8, D = 20, J = 6, NE = 1000, NDRAWS = 1000 };
    uint64_t st = 20260928ull;
    double *mean = malloc(sizeof(double) * D), *prec = malloc(sizeof(double) * D), *x0 = malloc(sizeof(double) * K * D);
    for (int i = 0; i < D; ++i) { mean[i] = 2.0 * unif(&st) - 1.0; prec[i] = exp(-2.0 * (0.8 * unif(&st) - 0.4)); }
    for (int i = 0; i < K * D; ++i) x0[i] = 4.0 * unif(&st) - 2.0;                 /* U[-2, 2]  (src/singlepath.jl:158-159) */

    pfmi_ctx *ctx = NULL;
    CHECK(pfmi_create(0, &ctx));
    pfmi_target tg = {0};
    tg.kind = PFMI_TARGET_GAUSS; tg.d = D; tg.r = 0; tg.mean = mean; tg.a = prec;
    CHECK(pfmi_set_target(ctx, &tg));

    int64_t npts[K], P = 0, off[K + 1];
    CHECK(pfmi_optimize_batch(ctx, K, x0, J, 1000, 1e-8, npts));                       /* optimize_with_trace */
    off[0] = 0;
    for (int k = 0; k < K; ++k) { off[k + 1] = off[k] + npts[k]; }
    P = off[K];
    CHECK(pfmi_fit_batch(ctx, J, 1e-12));                                               /* fit_mvnormals */
    int32_t *status = malloc(sizeof(int32_t) * P);
    int64_t nrej[K];
    CHECK(pfmi_get_fit_status(ctx, status, NULL, NULL, nrej));

    uint64_t *seeds = malloc(sizeof(uint64_t) * P);
    for (int64_t p = 0; p < P; ++p) seeds[p] = splitmix(&st);                           /* rand!(rng, UInt64[L]) src/elbo.jl:2 */
    double *elbo = malloc(sizeof(double) * P), *se = malloc(sizeof(double) * P);
    int64_t best[K], pts[K];
    uint64_t pseeds[K];
    CHECK(pfmi_elbo_batch(ctx, NE, seeds, NULL, elbo, se, best));                       /* maximize_elbo */
    for (int k = 0; k < K; ++k) {
        pts[k] = off[k] + best[k]; pseeds[k] = seeds[pts[k]];
        printf("path %d: %lld iterations, %lld rejected updates, fit_iteration %lld, ELBO %.3f +- %.3f\n", k,
               (long long)(npts[k] - 1), (long long)nrej[k], (long long)best[k], elbo[pts[k]], se[pts[k]]);
        if (best[k] < 1 || status[pts[k]] != PFMI_FIT_OK || !isfinite(elbo[pts[k]])) { fprintf(stderr, "path %d failed\n", k); return 1; }
    }
    CHECK(pfmi_pool_build(ctx, NE, pts, pseeds));                                       /* draws_per_component */
    void *lr_dev = NULL;
    int64_t S = 0, tail = 0;
    CHECK(pfmi_pool_log_ratios_dev(ctx, &lr_dev, &S));
    double *w = malloc(sizeof(double) * S), khat = 0.0, wsum = 0.0;
    CHECK(pfmi_psis_dev(ctx, lr_dev, S, w, NULL, &khat, &tail));                        /* _compute_psis_result */
    for (int64_t i = 0; i < S; ++i) wsum += w[i];
    int64_t *idx = malloc(sizeof(int64_t) * NDRAWS);
    CHECK(pfmi_resample_indices(ctx, S, NDRAWS, 1, 1, splitmix(&st), NULL, idx));       /* _resample */
    double *draws = malloc(sizeof(double) * D * NDRAWS);
    CHECK(pfmi_pool_gather(ctx, NDRAWS, idx, 0, draws));
    /* the resampled draws should look like the target: compare the marginal means / variances */
    double worst_m = 0.0, worst_v = 0.0;
    for (int i = 0; i < D; ++i) {
        double m = 0.0, v = 0.0;
        for (int n = 0; n < NDRAWS; ++n) m += draws[i + (size_t)D * n];
        m /= NDRAWS;
        for (int n = 0; n < NDRAWS; ++n) { double e = draws[i + (size_t)D * n] - m; v += e * e; }
        v /= NDRAWS - 1;
        double sd = sqrt(1.0 / prec[i]);
        if (fabs(m - mean[i]) / sd > worst_m) worst_m = fabs(m - mean[i]) / sd;
        if (fabs(v * prec[i] - 1.0) > worst_v) worst_v = fabs(v * prec[i] - 1.0);
    }
    int ok = fabs(wsum - 1.0) < 1e-10 && worst_m < 0.5 && worst_v < 0.6;
    for (int n = 0; n < NDRAWS; ++n) if (idx[n] < 0 || idx[n] >= S) ok = 0;
    printf("%s pool %lld draws, pareto k = %.3f (tail %lld), sum w = %.12f, worst |mean err|/sd = %.3f, worst |var ratio - 1| = %.3f\n",
           ok ? "OK" : "FAILED", (long long)S, khat, (long long)tail, wsum, worst_m, worst_v);
    CHECK(pfmi_destroy(ctx));
    return ok ? 0 : 1;
}
