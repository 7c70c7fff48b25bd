/*
 * julia_sequence.c -- the exact sequence of C-ABI calls that pathfinder.jl_amd/julia/PathfinderMI355X.jl makes for
 * `multipathfinder(eng, fun, ndraws; ...)`, `resample(...)` and the multi-GPU `Comm`, replayed from plain C with the results the
 * Julia side relies on checked after every step.  No Julia toolchain exists in this repository's images, so this program is
 * the executed stand-in for the wrapper (tests/test_gpu_abi.py::test_julia_call_sequence_in_c builds and runs it on the
 * GPU box).  Each block names the Julia function whose ccalls it replays.
 *
 *   gcc -O2 -Iinclude examples/julia_sequence.c -o julia_sequence -Lpathfinder.jl_amd/lib -lpfmi -Wl,-rpath,... -lm -ldl
 *
 * Round 3 (second half of main): `multipathfinder(engines::Vector{Engine}, target::DeviceTarget, ...)` -- built-in targets through
 * CTarget kinds 0 / 1, the enqueue / wait entry points, winners picked on the device, the fused pooled stage; `julia_sequence G`
 * shards the runs over G engines (G > 1 on a 1-GPU box: PFMI_RCCL_LIB = the in-process stand-in, PFMI_COMM_ALLOW_SHARED_GPU = 1) and
 * demands the single-engine result bit for bit; with PFMI_DEMO_LIB = examples/device_logp/liblogp_demo.so also a
 * `DeviceClosureTarget` (kind 3: the closure is a kernel launcher, draws never leave HBM).
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pfmi.h"

#define D 24
#define K 5
#define NITER 14
#define NPTS (NITER + 1)
#define N_ELBO 200
#define N_R 300
#define NDRAWS 128

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int32_t rc__ = (call);                                                                        \
        if (rc__ != 0) { printf("FAIL %s -> %d: %s\n", #call, rc__, pfmi_last_error()); return 1; } \
    } while (0)
#define REQUIRE(cond, ...)                                   \
    do {                                                     \
        if (!(cond)) { printf("FAIL " __VA_ARGS__); printf("\n"); return 1; } \
    } while (0)

/* the "Julia closure": logp(x) = -1/2 sum a_i (x_i - m_i)^2, reached through the pfmi_logp_fn trampoline one column at a time
 * (PathfinderMI355X._logp_trampoline; reference src/elbo.jl:15) */
static double g_a[D], g_m[D];
static long g_calls = 0, g_cols = 0;
static double logp1(const double *x) {
    double s = 0.0;
    for (int i = 0; i < D; ++i) s += g_a[i] * (x[i] - g_m[i]) * (x[i] - g_m[i]);
    return -0.5 * s;
}
static void trampoline(const double *X, int32_t d, int64_t n, double *out, void *user) {
    (void)user;
    g_calls += 1; g_cols += n;
    for (int64_t j = 0; j < n; ++j) out[j] = logp1(X + (size_t)j * d);
}

static uint64_t splitmix(uint64_t *s) {     /* plays the host rng (Julia: rand!(rng, UInt64[...])) */
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double unif(uint64_t *s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }

/* plays Pathfinder.optimize_with_trace on the host: steepest ascent with exact line search; trace = (theta_l, grad logp(theta_l)) */
static void host_trace(uint64_t *rng, double *theta, double *grad) {
    double x[D], g[D];
    for (int i = 0; i < D; ++i) x[i] = 4.0 * unif(rng) - 2.0;                 /* init_sampler: U[-2, 2] */
    for (int l = 0; l < NPTS; ++l) {
        double gg = 0.0, gag = 0.0;
        for (int i = 0; i < D; ++i) { g[i] = -g_a[i] * (x[i] - g_m[i]); gg += g[i] * g[i]; gag += g_a[i] * g[i] * g[i]; }
        memcpy(theta + (size_t)l * D, x, sizeof(x));
        memcpy(grad + (size_t)l * D, g, sizeof(g));
        const double t = gg / gag;
        for (int i = 0; i < D; ++i) x[i] += t * g[i];
    }
}

int main(int argc, char **argv) {
    uint64_t rng = 20260928ull;
    const int G = argc > 1 ? atoi(argv[1]) : 1;
    for (int i = 0; i < D; ++i) { g_a[i] = exp(-1.0 + 2.0 * unif(&rng)); g_m[i] = 2.0 * unif(&rng) - 1.0; }

    /* ---- Engine(device) ------------------------------------------------------------------------------------------------- */
    pfmi_ctx *ctx = NULL;
    CHECK(pfmi_create(0, &ctx));

    /* ---- multipathfinder: set_target!(eng, logp, d) ------------------------------------------------------------------------ */
    pfmi_target tg;
    memset(&tg, 0, sizeof(tg));
    tg.kind = PFMI_TARGET_HOST_CALLBACK; tg.d = D; tg.fn = trampoline; tg.user = NULL;
    CHECK(pfmi_set_target(ctx, &tg));

    /* ---- run_seeds = rand!(rng, UInt64[nruns]); per-run rng copies; host optimisations ------------------------------------------- */
    uint64_t run_rng[K];
    for (int k = 0; k < K; ++k) run_rng[k] = splitmix(&rng);
    static double theta[K * NPTS * D], grad[K * NPTS * D];
    int64_t npts[K], off[K + 1];
    off[0] = 0;
    for (int k = 0; k < K; ++k) {
        host_trace(&run_rng[k], theta + (size_t)k * NPTS * D, grad + (size_t)k * NPTS * D);
        npts[k] = NPTS; off[k + 1] = off[k] + NPTS;
    }
    const int64_t P = off[K];

    /* ---- fit_mvnormals(eng, traces): pfmi_set_traces, pfmi_fit_batch, pfmi_get_fit_status ------------------------------------------ */
    CHECK(pfmi_set_traces(ctx, K, npts, D, theta, grad));
    CHECK(pfmi_fit_batch(ctx, 6, 1e-12));
    static int32_t status[K * NPTS], jeff[K * NPTS];
    int64_t nrej[K];
    CHECK(pfmi_get_fit_status(ctx, status, jeff, NULL, nrej));
    for (int64_t p = 0; p < P; ++p) REQUIRE(status[p] == 0, "fit %lld failed (status %d)", (long long)p, status[p]);
    for (int k = 0; k < K; ++k) {
        REQUIRE(jeff[off[k]] == 0 && jeff[off[k] + 1] == 1 && jeff[off[k + 1] - 1] == 6, "effective history of run %d", k);
        REQUIRE(nrej[k] == 0, "rejected updates in run %d", k);
    }

    /* ---- maximize_elbo(b, rngs, N): seeds = rand!(rng_k, UInt64[L_k]) per run, pfmi_elbo_batch --------------------------------------- */
    static uint64_t seeds[K * NPTS];
    for (int k = 0; k < K; ++k) {
        seeds[off[k]] = 0;
        for (int l = 1; l < NPTS; ++l) seeds[off[k] + l] = splitmix(&run_rng[k]);
    }
    static double elbo[K * NPTS], se[K * NPTS];
    int64_t best[K];
    const long calls0 = g_calls;
    CHECK(pfmi_elbo_batch(ctx, N_ELBO, seeds, NULL, elbo, se, best));
    REQUIRE(g_calls > calls0 && g_cols == (long)(P - K) * N_ELBO, "callback saw %ld columns, expected %ld", g_cols, (long)(P - K) * N_ELBO);
    for (int k = 0; k < K; ++k) {
        REQUIRE(isnan(elbo[off[k]]), "first point has no ELBO (fit_distributions[2:end])");
        REQUIRE(best[k] >= 1 && best[k] <= NITER, "iteration_opt of run %d = %lld", k, (long long)best[k]);
        for (int l = 1; l < NPTS; ++l)            /* _findmax_skipnan: first maximum wins (src/utils.jl:55-72) */
            REQUIRE(!(elbo[off[k] + l] > elbo[off[k] + best[k]]), "argmax of run %d", k);
    }

    /* ---- LazyELBOEstimates[i]: pfmi_draws(p, seed, 0, N) regenerates the ELBO draws; value = mean(logp - logq) ------------------------ */
    {
        const int64_t p = off[2] + best[2];
        static double X[D * N_ELBO], lp[N_ELBO], lq[N_ELBO];
        CHECK(pfmi_draws(ctx, p, seeds[p], 0, N_ELBO, NULL, X, lp, lq));
        double s = 0.0;
        for (int n = 0; n < N_ELBO; ++n) {
            s += lp[n] - lq[n];
            REQUIRE(fabs(lp[n] - logp1(X + (size_t)n * D)) <= 1e-12 * (1 + fabs(lp[n])), "logp of a regenerated draw");
        }
        REQUIRE(fabs(s / N_ELBO - elbo[p]) <= 1e-10 * (1 + fabs(elbo[p])), "ELBOEstimate.value %.15g vs %.15g", s / N_ELBO, elbo[p]);
    }

    /* ---- fit_distribution(b, p): pfmi_get_fit -> MvNormal(mu, WoodburyPDMat(Diagonal(alpha), B, D, WoodburyPDFactorization(U, Q, V))) --- */
    int64_t fit_points[K];
    uint64_t draw_seeds[K];
    for (int k = 0; k < K; ++k) {
        const int64_t p = off[k] + best[k];
        fit_points[k] = p; draw_seeds[k] = seeds[p];       /* success: the winner's ELBO draws are reused (src/singlepath.jl:226-233) */
        const int j = jeff[p], m = 2 * j, kk = m < D ? m : D;
        double alpha[D], mu[D], logdet;
        double *B = malloc(sizeof(double) * D * m), *Dm = malloc(sizeof(double) * m * m), *qrf = malloc(sizeof(double) * D * m);
        double *T = malloc(sizeof(double) * kk * kk), *V = malloc(sizeof(double) * kk * kk);
        CHECK(pfmi_get_fit(ctx, p, alpha, B, Dm, qrf, T, V, mu, &logdet));
        /* mu = theta + Sigma grad with Sigma = diag(alpha) + B D B'  (src/mvnormal.jl:14-21) */
        const double *th = theta + (size_t)p * D, *gr = grad + (size_t)p * D;
        double Btg[32], DBtg[32];
        for (int a = 0; a < m; ++a) { Btg[a] = 0.0; for (int i = 0; i < D; ++i) Btg[a] += B[i + (size_t)D * a] * gr[i]; }
        for (int a = 0; a < m; ++a) { DBtg[a] = 0.0; for (int c = 0; c < m; ++c) DBtg[a] += Dm[a + (size_t)m * c] * Btg[c]; }
        for (int i = 0; i < D; ++i) {
            double v = th[i] + alpha[i] * gr[i];
            for (int a = 0; a < m; ++a) v += B[i + (size_t)D * a] * DBtg[a];
            REQUIRE(fabs(v - mu[i]) <= 1e-9 * (1 + fabs(mu[i])), "mu[%d] of run %d: %.15g vs %.15g", i, k, v, mu[i]);
            REQUIRE(alpha[i] > 0.0, "alpha > 0");
        }
        for (int a = 0; a < kk; ++a) {                      /* V upper triangular with positive diagonal, T upper triangular */
            REQUIRE(V[a + (size_t)kk * a] > 0.0, "V diagonal");
            for (int c = 0; c < a; ++c) REQUIRE(V[a + (size_t)kk * c] == 0.0 && T[a + (size_t)kk * c] == 0.0, "triangularity");
        }
        double ld = 0.0;                                    /* logdet = sum log alpha + 2 sum log V_ii (src/woodbury.jl:76-80) */
        for (int i = 0; i < D; ++i) ld += log(alpha[i]);
        for (int a = 0; a < kk; ++a) ld += 2.0 * log(V[a + (size_t)kk * a]);
        REQUIRE(fabs(ld - logdet) <= 1e-10 * (1 + fabs(ld)), "logdet");
        free(B); free(Dm); free(qrf); free(T); free(V);
    }

    /* ---- _compute_psis_result: pfmi_pool_build, pfmi_pool_log_ratios_dev, pfmi_psis_dev ------------------------------------------------- */
    const int64_t S = (int64_t)K * N_R;
    CHECK(pfmi_pool_build(ctx, N_R, fit_points, draw_seeds));
    void *lr_dev = NULL;
    int64_t cnt = 0;
    CHECK(pfmi_pool_log_ratios_dev(ctx, &lr_dev, &cnt));
    REQUIRE(cnt == S && lr_dev != NULL, "log-ratio shard");
    static double w[K * N_R], lw[K * N_R];
    double khat = NAN;
    int64_t M = 0;
    CHECK(pfmi_psis_dev(ctx, lr_dev, S, w, lw, &khat, &M));
    double wsum = 0.0;
    for (int64_t i = 0; i < S; ++i) wsum += w[i];
    REQUIRE(fabs(wsum - 1.0) < 1e-12 && M == 117 && isfinite(khat), "PSIS: sum w = %.15g, M = %lld, k = %g", wsum, (long long)M, khat);

    /* ---- _resample: rand(rng, UInt64) -> pfmi_resample_indices, pfmi_pool_gather; ids = cld(idx + 1, N_r) -------------------------------- */
    int64_t idx[NDRAWS];
    static double draws[D * NDRAWS];
    const uint64_t rs_seed = splitmix(&rng);
    CHECK(pfmi_resample_indices(ctx, S, NDRAWS, 1, 1, rs_seed, NULL, idx));
    CHECK(pfmi_pool_gather(ctx, NDRAWS, idx, 0, draws));
    /* DevicePathfinderResult.draws of run k == the pool block of run k == pfmi_draws(fit_point, draw_seed, 0, N_r) */
    {
        static double X[D * N_R], lp[N_R], lq[N_R];
        for (int t = 0; t < NDRAWS; t += 17) {
            REQUIRE(idx[t] >= 0 && idx[t] < S, "index range");
            const int k = (int)(idx[t] / N_R), n = (int)(idx[t] % N_R);          /* component id = cld(idx + 1, N_r) - 1 */
            CHECK(pfmi_draws(ctx, fit_points[k], draw_seeds[k], 0, N_R, NULL, X, lp, lq));
            REQUIRE(memcmp(X + (size_t)n * D, draws + (size_t)t * D, sizeof(double) * D) == 0, "gathered column %d", t);
        }
    }

    /* ---- _resample(...; statsbase = true): u = rand(rng, ndraws) on the host, StatsBase.direct_sample! scan on the device ------------------ */
    {
        double u[NDRAWS];
        int64_t idx_sb[NDRAWS];
        for (int t = 0; t < NDRAWS; ++t) u[t] = unif(&rng);
        CHECK(pfmi_resample_indices_direct(ctx, S, NDRAWS, u, idx_sb));
        for (int t = 0; t < NDRAWS; ++t) {                   /* the literal loop of StatsBase.sample(rng, wv) */
            int64_t i = 0;
            double cw = w[0];
            while (cw < u[t] && i < S - 1) { i += 1; cw += w[i]; }
            REQUIRE(i == idx_sb[t], "direct_sample! index %d: %lld vs %lld", t, (long long)i, (long long)idx_sb[t]);
        }
    }

    /* ---- resample(result, n; ndraws_per_run = M): fresh candidates, seeds = rand(rng, UInt64, K) ------------------------------------------- */
    {
        uint64_t fresh[K];
        for (int k = 0; k < K; ++k) fresh[k] = splitmix(&rng);
        CHECK(pfmi_pool_build(ctx, 150, fit_points, fresh));
        CHECK(pfmi_pool_log_ratios_dev(ctx, &lr_dev, &cnt));
        REQUIRE(cnt == (int64_t)K * 150, "fresh pool size");
        CHECK(pfmi_psis_dev(ctx, lr_dev, cnt, NULL, NULL, &khat, &M));
        int64_t idx2[64];
        static double d2[D * 64];
        CHECK(pfmi_resample_indices(ctx, cnt, 64, 1, 0, splitmix(&rng), NULL, idx2));     /* replace = false */
        CHECK(pfmi_pool_gather(ctx, 64, idx2, 0, d2));
        for (int a = 0; a < 64; ++a)
            for (int c = 0; c < a; ++c) REQUIRE(idx2[a] != idx2[c], "replace = false returned a duplicate");
        /* stored draws again (ndraws_per_run === nothing): re-pooled from the run's own seeds, PSIS reproduces the weights bit for bit */
        CHECK(pfmi_pool_build(ctx, N_R, fit_points, draw_seeds));
        CHECK(pfmi_pool_log_ratios_dev(ctx, &lr_dev, &cnt));
        static double w2[K * N_R];
        CHECK(pfmi_psis_dev(ctx, lr_dev, cnt, w2, NULL, &khat, &M));
        REQUIRE(memcmp(w, w2, sizeof(w)) == 0, "stored-draws PSIS weights changed");
    }

    /* ---- Comm([eng]): pfmi_comm_init_all, pfmi_comm_pool_psis, pfmi_comm_resample (one GPU here; G engines on a node) ---------------------- */
    {
        pfmi_comm *comm = NULL;
        pfmi_ctx *ctxs[1] = {ctx};
        int32_t rc = pfmi_comm_init_all(1, ctxs, &comm);
        if (rc == PFMI_ERR_UNSUPPORTED) {
            printf("note: RCCL not available on this box (%s); collective sequence skipped\n", pfmi_last_error());
        } else {
            REQUIRE(rc == 0, "pfmi_comm_init_all: %s", pfmi_last_error());
            int32_t world = 0, nlocal = 0, ver = 0;
            CHECK(pfmi_comm_info(comm, &world, &nlocal, &ver));
            REQUIRE(world == 1 && nlocal == 1, "comm info");
            double k2;
            int64_t M2;
            CHECK(pfmi_comm_pool_psis(comm, &k2, &M2));
            int64_t idx3[NDRAWS];
            static double d3[D * NDRAWS];
            CHECK(pfmi_comm_resample(comm, NDRAWS, 1, 1, rs_seed, NULL, idx3, d3));
            REQUIRE(memcmp(idx3, idx, sizeof(idx)) == 0 && memcmp(d3, draws, sizeof(draws)) == 0, "collective path differs from the local one");
            CHECK(pfmi_comm_destroy(comm));
        }
    }

    /* ---- retry loop (src/singlepath.jl:259-283): run 1 is re-optimised from a fresh init, every run is refitted in ONE batch; the
     *      finished runs keep their seeds, so their ELBOs come back bit-identical ---------------------------------------------------- */
    {
        static double elbo2[K * NPTS], se2[K * NPTS];
        int64_t best2[K];
        host_trace(&run_rng[1], theta + (size_t)1 * NPTS * D, grad + (size_t)1 * NPTS * D);
        for (int l = 1; l < NPTS; ++l) seeds[off[1] + l] = splitmix(&run_rng[1]);
        CHECK(pfmi_set_traces(ctx, K, npts, D, theta, grad));
        CHECK(pfmi_fit_batch(ctx, 6, 1e-12));
        CHECK(pfmi_get_fit_status(ctx, status, jeff, NULL, nrej));
        CHECK(pfmi_elbo_batch(ctx, N_ELBO, seeds, NULL, elbo2, se2, best2));
        for (int k = 0; k < K; ++k) {
            if (k == 1) continue;
            REQUIRE(best2[k] == best[k], "finished run %d changed its winner", k);
            for (int l = 1; l < NPTS; ++l) REQUIRE(elbo2[off[k] + l] == elbo[off[k] + l], "finished run %d was not reproduced", k);
        }
    }

    /* ================= round 3: multipathfinder(engines, target::DeviceTarget, ndraws) =================================================== */
    {
#define K3 8
#define MAXIT 60
        REQUIRE(G >= 1 && K3 % G == 0, "G = %d must divide %d runs", G, K3);
        /* set_target!(eng, GaussTarget(mean, a)): CTarget(kind = 0, d, r = 0, mean, a, ...) -- the library copies the parameters */
        pfmi_target gt;
        memset(&gt, 0, sizeof(gt));
        gt.kind = PFMI_TARGET_GAUSS; gt.d = D; gt.r = 0; gt.mean = g_m; gt.a = g_a; gt.offset = 0.0;
        uint64_t rng3 = 777ull, run3[K3], fail3[K3];
        static double x0[K3 * D];
        for (int k = 0; k < K3; ++k) run3[k] = splitmix(&rng3);
        for (int k = 0; k < K3; ++k)
            for (int i = 0; i < D; ++i) x0[k * D + i] = 4.0 * unif(&run3[k]) - 2.0;              /* init_sampler(rngs[k], ...) */
        const uint64_t rs3 = splitmix(&rng3);                                                   /* rand(rng, UInt64) of _resample */
        /* ---- reference: ONE engine, the blocking entry points, winners chosen by the host */
        pfmi_ctx *e1 = NULL;
        CHECK(pfmi_create(0, &e1));
        CHECK(pfmi_set_target(e1, &gt));
        int64_t np1[K3], off1[K3 + 1];
        CHECK(pfmi_optimize_batch(e1, K3, x0, 6, MAXIT, 1e-8, np1));
        off1[0] = 0;
        for (int k = 0; k < K3; ++k) off1[k + 1] = off1[k] + np1[k];
        const int64_t P1 = off1[K3];
        uint64_t *sd1 = calloc((size_t)P1, sizeof(uint64_t));
        uint64_t r3[K3];
        memcpy(r3, run3, sizeof(r3));
        for (int k = 0; k < K3; ++k) {
            for (int64_t l = 1; l < np1[k]; ++l) sd1[off1[k] + l] = splitmix(&r3[k]);            /* rand!(rng_k, UInt64[L_k]) */
            uint64_t peek = r3[k];
            fail3[k] = splitmix(&peek);                                                         /* rand(copy(rng_k), UInt64) */
        }
        CHECK(pfmi_fit_batch(e1, 6, 1e-12));
        double *el1 = malloc(sizeof(double) * P1), *se1 = malloc(sizeof(double) * P1);
        int64_t best1[K3], fp1[K3];
        uint64_t ds1[K3];
        CHECK(pfmi_elbo_batch(e1, N_ELBO, sd1, NULL, el1, se1, best1));
        for (int k = 0; k < K3; ++k) {
            REQUIRE(best1[k] >= 1 && isfinite(el1[off1[k] + best1[k]]), "run %d of the built-in target failed", k);
            fp1[k] = off1[k] + best1[k]; ds1[k] = sd1[fp1[k]];
        }
        CHECK(pfmi_pool_build(e1, N_R, fp1, ds1));
        void *lrd = NULL;
        int64_t c1 = 0;
        CHECK(pfmi_pool_log_ratios_dev(e1, &lrd, &c1));
        static double w1[K3 * N_R];
        double k1 = NAN;
        int64_t M1 = 0, idx1[NDRAWS];
        static double dr1[D * NDRAWS];
        CHECK(pfmi_psis_dev(e1, lrd, c1, w1, NULL, &k1, &M1));
        CHECK(pfmi_resample_indices(e1, c1, NDRAWS, 1, 1, rs3, NULL, idx1));
        CHECK(pfmi_pool_gather(e1, NDRAWS, idx1, 0, dr1));

        /* ---- the Julia sequence: G engines, every stage enqueued on all of them before the first wait */
        pfmi_ctx *eg[K3];
        int ndev = 0;
        CHECK(pfmi_device_count(&ndev));
        const int Kl = K3 / G;
        for (int g = 0; g < G; ++g) {
            CHECK(pfmi_create(g % ndev, &eg[g]));
            CHECK(pfmi_set_target(eg[g], &gt));
        }
        for (int g = 0; g < G; ++g) CHECK(pfmi_optimize_batch_enqueue(eg[g], Kl, x0 + (size_t)g * Kl * D, 6, MAXIT, 1e-8));
        int64_t npg[K3];
        for (int g = 0; g < G; ++g) CHECK(pfmi_optimize_batch_wait(eg[g], npg + g * Kl));
        for (int k = 0; k < K3; ++k) REQUIRE(npg[k] == np1[k], "trace length of run %d depends on the sharding", k);
        for (int g = 0; g < G; ++g) CHECK(pfmi_fit_batch(eg[g], 6, 1e-12));
        for (int g = 0; g < G; ++g) {                       /* seeds of the block = the corresponding slice of the single-engine table */
            CHECK(pfmi_elbo_batch_enqueue(eg[g], N_ELBO, sd1 + off1[g * Kl], NULL));
            CHECK(pfmi_pool_build_best(eg[g], N_R, fail3 + g * Kl));
        }
        pfmi_comm *cm = NULL;
        CHECK(pfmi_comm_init_all(G, eg, &cm));
        int32_t world = 0, nlocal = 0, ver = -1;
        CHECK(pfmi_comm_info(cm, &world, &nlocal, &ver));
        REQUIRE(world == G && nlocal == G, "comm of %d engines reports world %d", G, world);
        double kg = NAN;
        int64_t Mg = 0, idxg[NDRAWS];
        static double drg[D * NDRAWS];
        CHECK(pfmi_comm_psis_resample(cm, NDRAWS, 1, 1, rs3, NULL, &kg, &Mg, idxg, drg));           /* ONE synchronisation */
        REQUIRE(kg == k1 && Mg == M1, "pooled PSIS depends on the sharding: k %.17g vs %.17g", kg, k1);
        REQUIRE(memcmp(idxg, idx1, sizeof(idx1)) == 0 && memcmp(drg, dr1, sizeof(dr1)) == 0, "resampled draws depend on the sharding (G = %d)", G);
        for (int g = 0; g < G; ++g) {                       /* the downloads come last */
            const int64_t Pg = off1[(g + 1) * Kl] - off1[g * Kl];
            double *elg = malloc(sizeof(double) * Pg);
            int64_t bg[K3], pg[K3];
            uint64_t sg[K3];
            int32_t okg[K3], *stg = malloc(sizeof(int32_t) * Pg);
            CHECK(pfmi_get_fit_status(eg[g], stg, NULL, NULL, NULL));
            CHECK(pfmi_elbo_batch_wait(eg[g], elg, NULL, bg));
            CHECK(pfmi_pool_winners(eg[g], pg, sg, okg));
            for (int j = 0; j < Kl; ++j) {
                const int k = g * Kl + j;
                REQUIRE(bg[j] == best1[k] && okg[j] == 1 && sg[j] == ds1[k], "winner of run %d", k);
                REQUIRE(pg[j] + off1[g * Kl] == fp1[k], "fit point of run %d", k);
            }
            REQUIRE(memcmp(elg + 1, el1 + off1[g * Kl] + 1, sizeof(double) * (np1[g * Kl] - 1)) == 0, "ELBO table of engine %d", g);
            free(elg); free(stg);
        }
        static double wg[K3 * N_R];
        CHECK(pfmi_psis_weights(eg[0], (int64_t)K3 * N_R, wg, NULL));                             /* DevicePSISResult.weights */
        REQUIRE(memcmp(wg, w1, sizeof(w1)) == 0, "PSIS weights depend on the sharding");
        CHECK(pfmi_comm_destroy(cm));

        /* ---- round 5: the streaming pipeline (PathfinderMI355X.jl: stream_enqueue!, stream_seeds!, stream_pump!, stream_wait!,
         *      psis_resample_enqueue!, defer_downloads!, psis_resample_wait!) -- the optimisations, the fits and the scans as ONE dataflow per
         *      engine, the calling thread scheduling all of them; fixed-stride slots; ONE wait for every result.  Same bits as above. */
        {
            const int64_t cap = MAXIT + 1;
            for (int g = 0; g < G; ++g)                      /* the optimisers start at once; the seed streams are drawn while they run */
                CHECK(pfmi_stream_enqueue(eg[g], Kl, x0 + (size_t)g * Kl * D, 6, MAXIT, 1e-8, 1e-12, N_ELBO, NULL));
            uint64_t *tab = calloc((size_t)K3 * cap, sizeof(uint64_t));
            memcpy(r3, run3, sizeof(r3));
            for (int k = 0; k < K3; ++k)                     /* rand!(copy(rng_k), UInt64[maxiters + 1]): value l - 1 seeds fit l, value L a failed run */
                for (int64_t i = 0; i < cap; ++i) tab[k * cap + i] = splitmix(&r3[k]);
            for (int g = 0; g < G; ++g) CHECK(pfmi_stream_seeds(eg[g], tab + (size_t)g * Kl * cap));
            int left = G, done5[K3];
            memset(done5, 0, sizeof(done5));
            while (left > 0)                                 /* one scheduling pass per engine in turn, never blocking */
                for (int g = 0; g < G; ++g) {
                    if (done5[g]) continue;
                    int32_t fin = 0;
                    CHECK(pfmi_stream_pump(eg[g], &fin));
                    if (fin) { done5[g] = 1; --left; }
                }
            int64_t np5[K3];
            for (int g = 0; g < G; ++g) CHECK(pfmi_stream_wait(eg[g], np5 + g * Kl));
            for (int k = 0; k < K3; ++k) REQUIRE(np5[k] == np1[k], "streamed trace length of run %d", k);
            for (int g = 0; g < G; ++g) CHECK(pfmi_pool_build_best(eg[g], N_R, NULL));      /* failed runs: value L of their stream, looked up on the device */
            pfmi_comm *cm5 = NULL;
            CHECK(pfmi_comm_init_all(G, eg, &cm5));
            CHECK(pfmi_comm_psis_resample_enqueue(cm5, NDRAWS, 1, 1, rs3, NULL));
            double *el5[K3], *se5[K3];
            int32_t *st5[K3];
            int64_t b5[K3];
            for (int g = 0; g < G; ++g) {                    /* queued behind the pooled stage, delivered by its ONE wait */
                const int64_t Pg = Kl * cap;
                el5[g] = malloc(sizeof(double) * Pg); se5[g] = malloc(sizeof(double) * Pg); st5[g] = malloc(sizeof(int32_t) * Pg);
                CHECK(pfmi_defer_downloads(eg[g], 1));
                CHECK(pfmi_get_fit_status(eg[g], st5[g], NULL, NULL, NULL));
                CHECK(pfmi_elbo_batch_wait(eg[g], el5[g], se5[g], b5 + g * Kl));
                CHECK(pfmi_defer_downloads(eg[g], 0));
            }
            double k5 = NAN;
            int64_t M5 = 0, idx5[NDRAWS];
            static double dr5[D * NDRAWS];
            CHECK(pfmi_comm_psis_resample_wait(cm5, &k5, &M5, idx5, dr5));
            REQUIRE(k5 == k1 && M5 == M1, "streamed pooled PSIS: k %.17g vs %.17g", k5, k1);
            REQUIRE(memcmp(idx5, idx1, sizeof(idx1)) == 0 && memcmp(dr5, dr1, sizeof(dr1)) == 0, "streamed resampled draws (G = %d)", G);
            for (int k = 0; k < K3; ++k) {
                const int g = k / Kl, j = k % Kl;
                REQUIRE(b5[k] == best1[k], "streamed winner of run %d", k);
                for (int64_t l = 0; l < cap; ++l) {
                    const int64_t slot = (int64_t)j * cap + l;
                    if (l < np1[k]) {
                        REQUIRE(st5[g][slot] == PFMI_FIT_OK, "status of run %d point %lld", k, (long long)l);
                        if (l >= 1) REQUIRE(memcmp(&el5[g][slot], &el1[off1[k] + l], sizeof(double)) == 0 &&
                                            memcmp(&se5[g][slot], &se1[off1[k] + l], sizeof(double)) == 0, "streamed ELBO of run %d fit %lld", k, (long long)l);
                    } else REQUIRE(st5[g][slot] == PFMI_FIT_ABSENT && isnan(el5[g][slot]), "slot %lld of run %d is not marked absent", (long long)l, k);
                }
            }
            for (int g = 0; g < G; ++g) { free(el5[g]); free(se5[g]); free(st5[g]); }
            CHECK(pfmi_comm_destroy(cm5));
            free(tab);
            printf("round-5 streaming sequence ok: G=%d engines\n", G);
        }

        /* ---- DeviceClosureTarget (kind 3): the closure is a kernel launcher from a user library; same draws, logp evaluated in HBM */
        const char *demo = getenv("PFMI_DEMO_LIB");
        if (demo && demo[0]) {
            void *h = dlopen(demo, RTLD_NOW | RTLD_LOCAL);
            REQUIRE(h != NULL, "dlopen %s: %s", demo, dlerror());
            void *(*create)(int32_t, int32_t, const double *, const double *, const double *, const double *, double) =
                (void *(*)(int32_t, int32_t, const double *, const double *, const double *, const double *, double))dlsym(h, "pfx_gauss_create");
            pfmi_logp_dev_fn fn = (pfmi_logp_dev_fn)dlsym(h, "pfx_gauss_logp");
            REQUIRE(create && fn, "demo library symbols");
            void *user = create(D, 0, g_m, g_a, NULL, NULL, 0.0);
            REQUIRE(user != NULL, "pfx_gauss_create");
            pfmi_target dt;
            memset(&dt, 0, sizeof(dt));
            dt.kind = PFMI_TARGET_DEVICE_CALLBACK; dt.d = D; dt.dev_fn = fn; dt.user = user;
            CHECK(pfmi_set_target(e1, &dt));               /* same engine, same fits: only logp's route changes */
            double *el3 = malloc(sizeof(double) * P1);
            int64_t b3[K3];
            CHECK(pfmi_elbo_batch(e1, N_ELBO, sd1, NULL, el3, NULL, b3));
            double hb = 0.0;
            CHECK(pfmi_callback_stats_dev(e1, &hb));
            REQUIRE(hb == 8.0 * D * N_ELBO * (double)(P1 - K3), "bytes materialised for the closure: %g", hb);
            for (int64_t p = 0; p < P1; ++p)
                REQUIRE((isnan(el3[p]) && isnan(el1[p])) || fabs(el3[p] - el1[p]) <= 1e-9 * (1 + fabs(el1[p])), "closure ELBO of fit %lld", (long long)p);
            for (int k = 0; k < K3; ++k) REQUIRE(b3[k] == best1[k], "closure argmax of run %d", k);
            free(el3);
            printf("device closure ok (%s)\n", demo);
        }
        for (int g = 0; g < G; ++g) CHECK(pfmi_destroy(eg[g]));
        CHECK(pfmi_destroy(e1));
        free(sd1); free(el1); free(se1);
        printf("round-3 sequence ok: G=%d engines, rccl_version=%d\n", G, ver);
    }

    CHECK(pfmi_destroy(ctx));
    printf("OK julia_sequence: K=%d d=%d pareto_k=%.4f callback calls=%ld columns=%ld G=%d\n", K, D, khat, g_calls, g_cols, G);
    return 0;
}
