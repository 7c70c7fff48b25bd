/*
 * pfmi.h -- C ABI of libpfmi.so: the MI355X-native (gfx950, HIP) engine for the ELBO / sampling /
 * inverse-Hessian / PSIS-resampling hot path of Pathfinder.
 *
 * The reference (mlcolab/Pathfinder.jl v0.10.7) has no FFI boundary; the seam this library fills is
 * the four Julia call sites of the hot path:
 *     fit_mvnormals(points, gradients; history_length)            src/singlepath.jl:301-303
 *     maximize_elbo(rng, logp, fit_distributions[2:end], N, ntasks) src/singlepath.jl:306-308
 *     _compute_psis_result(logp, fit_distributions, draws)        src/multipath.jl:221, src/resample.jl:35
 *     _resample(rng, draws_per_component, psis_result, ndraws)    src/multipath.jl:225, src/resample.jl:42-44
 * Each entry point below cites the reference function it replaces.  INTEGRATION.md shows the
 * Julia `ccall` stubs a maintainer would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; every pointer is a HOST pointer unless the
 *     function name ends in `_dev` (then it is a device pointer on the ctx's GPU);
 *   - all matrices are column-major Float64 (Julia's layout); trace points are stored point-major:
 *     theta[p*d + i] is coordinate i of point p (== hcat(points...) in Julia memory);
 *   - indices are 0-based here (the Julia wrapper adds 1);
 *   - the caller owns every host buffer; the library owns device memory inside the ctx and keeps
 *     no host pointer after a call returns; callbacks run on the calling thread during the call;
 *   - return value 0 = ok, < 0 = error (message: pfmi_last_error()); per-fit numerical failures
 *     (src/woodbury.jl:202,205 PosDefException) are reported in status[] and as NaN ELBOs, never
 *     as a process abort;
 *   - one ctx per host thread; a ctx owns one HIP stream on one GPU.
 */
#ifndef PFMI_H
#define PFMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfmi_ctx pfmi_ctx;

/* return codes */
#define PFMI_OK 0
#define PFMI_ERR_ARG (-1)
#define PFMI_ERR_HIP (-2)
#define PFMI_ERR_STATE (-3)
#define PFMI_ERR_UNSUPPORTED (-4)
#define PFMI_ERR_NUMERIC (-5)
#define PFMI_ERR_COMM (-6)
#define PFMI_ERR_RETRY (-7)   /* transient: the enqueued step was discarded (e.g. the GPU is shared and an in-kernel hand-over timed out); enqueue it again */

/* per-fit status (src/woodbury.jl:189-190, 202, 205) */
#define PFMI_FIT_OK 0
#define PFMI_FIT_A_NOT_PD 1
#define PFMI_FIT_C_NOT_PD 2
#define PFMI_FIT_NONFINITE 3
#define PFMI_FIT_ABSENT 4       /* streaming layout (pfmi_stream_enqueue): the path ended before this slot -- there is no such trace point */

/* target kinds: the hot path only ever sees logp(x) = -f(x) (src/singlepath.jl:186, src/multipath.jl:159) */
#define PFMI_TARGET_GAUSS 0          /* offset - 1/2 [ sum a_i e_i^2 - || G Wd' e ||^2 ], e = x - mean  */
#define PFMI_TARGET_FUNNEL 1         /* docs/src/examples/quickstart.md:229-234                         */
#define PFMI_TARGET_HOST_CALLBACK 2  /* arbitrary host closure, evaluated on a host copy of the draws    */
#define PFMI_TARGET_DEVICE_CALLBACK 3 /* arbitrary DEVICE closure: the draws never leave HBM               */

/* host callback: X is d x n column-major, out[n] receives logp of every column (src/elbo.jl:15) */
typedef void (*pfmi_logp_fn)(const double *X, int32_t d, int64_t n, double *out, void *user);
/* device callback (the reference's general `logp` closure, src/elbo.jl:15, src/resample.jl:90-92, kept on the GPU): X_dev is a
 * DEVICE pointer to d x n column-major draws in HBM, out_dev a DEVICE pointer to n doubles.  The function is called on the host and
 * must ENQUEUE its kernel(s) on `stream` (a hipStream_t of the ctx) and return without synchronising -- e.g. an AMDGPU.jl kernel
 * launch or a HIP launcher; the library orders its own work behind it on the same stream.  No PCIe traffic.  `stream` is NOT the same
 * on every call: an ELBO scan hands the blocks of fits to two streams in turn, so that the closure of one block runs beside the library's
 * draw writer of the next (round 6: 2.87 -> 3.27 TB/s of moved bytes with the example closure) -- the closure must use the stream it is
 * given and keep no per-call state in `user` that a concurrent call on the other stream could clobber.  (The gain comes from each kernel
 * filling the other's launch tails and idle memory pipe; a closure squeezed into the registers / LDS the writer leaves on a CU -- <= 96
 * vector registers, <= 24 KB -- was measured and is NOT worth its slower loads: write the closure for its own bandwidth.) */
typedef void (*pfmi_logp_dev_fn)(const double *X_dev, int32_t d, int64_t n, double *out_dev, void *stream, void *user);

typedef struct {
    int32_t kind;        /* PFMI_TARGET_*                                      */
    int32_t d;           /* dimension                                          */
    int32_t r;           /* GAUSS: rank of the low-rank part (0 = diagonal)    */
    int32_t reserved;
    const double *mean;  /* GAUSS: d                                           */
    const double *a;     /* GAUSS: d, diagonal precision part                  */
    const double *Wd;    /* GAUSS: d x r column-major (NULL if r == 0)         */
    const double *G;     /* GAUSS: r x r column-major, lower triangular        */
    double offset;       /* GAUSS: additive constant                           */
    pfmi_logp_fn fn;     /* HOST_CALLBACK                                      */
    void *user;          /* HOST_CALLBACK / DEVICE_CALLBACK                    */
    pfmi_logp_dev_fn dev_fn; /* DEVICE_CALLBACK                                */
} pfmi_target;

/* ---- library / context ----------------------------------------------------------------------- */
const char *pfmi_last_error(void);
int32_t pfmi_version(void);
int32_t pfmi_device_count(int32_t *count);
int32_t pfmi_create(int32_t device, pfmi_ctx **out);
int32_t pfmi_destroy(pfmi_ctx *ctx);   /* communicators that hold ctx are closed first: any finaliser order is safe */
int32_t pfmi_sync(pfmi_ctx *ctx);
/* Test / tuning hooks (kernel selection PFMI_ELBO_KERNEL / PFMI_FIT_KERNEL / PFMI_HISTORY_KERNEL / PFMI_PSIS_KERNEL / PFMI_QF_NO_TAIL,
 * PFMI_DEVCB_CHUNK_MB, PFMI_LBFGS_REJECT_EVERY, the collective path's PFMI_RCCL_LIB / PFMI_COMM_ALLOW_SHARED_GPU / PFMI_COMM_FORCE_RCCL;
 * INTEGRATION.md section 3).  A hook is read from this explicit table, or from the environment variable of the same name ONLY when
 * the process was started with PFMI_DEBUG_HOOKS=1: a production process is never re-configured by a stray environment variable.
 * value == NULL unsets.  Process-global; set hooks before other threads use the library.  No reference counterpart. */
int32_t pfmi_debug_set(const char *key, const char *value);

/* device-time instrumentation (hipEvents on the ctx stream).  pfmi_timer_* bracket any sequence of
 * calls; pfmi_kernel_time returns accumulated time and launch count of one named kernel family
 * ("history", "fit", "elbo_draws" = ELBO scan, "elbo_draws_x" = draw launches that also write x, "elbo_reduce",
 * "psis", "resample") since pfmi_profile(ctx, mode).  mode 1: the host waits after every stage (each stage starts on an idle
 * GPU: its figure includes the host's launch latency); mode 2: the event pairs stay in the stream and are read by
 * pfmi_kernel_time (which waits for them) -- the pipeline runs as it does unprofiled and a stage's figure is its kernels'
 * time; mode 0: off. */
int32_t pfmi_timer_start(pfmi_ctx *ctx);
int32_t pfmi_timer_stop(pfmi_ctx *ctx, double *milliseconds);
int32_t pfmi_profile(pfmi_ctx *ctx, int32_t enable);
int32_t pfmi_kernel_time(pfmi_ctx *ctx, const char *name, double *milliseconds, int64_t *launches);

/* ---- inputs ------------------------------------------------------------------------------------ */
/* The target log density (what `logp` is in src/elbo.jl:12-20 and src/resample.jl:81-95). */
int32_t pfmi_set_target(pfmi_ctx *ctx, const pfmi_target *target);

/* K optimisation traces (OptimizationTrace.points / .gradients, src/optimize.jl:110-114), path k
 * has npoints[k] = L_k + 1 points; theta/grad hold all P = sum npoints points, point-major. */
int32_t pfmi_set_traces(pfmi_ctx *ctx, int32_t K, const int64_t *npoints, int32_t d,
                        const double *theta, const double *grad);

/* ---- trajectory generation on the device ---------------------------------- */
/* Plays optimize_with_trace (src/optimize.jl:35-121) for the BUILT-IN targets: runs K independent L-BFGS
 * optimisations of -logp from x0[k] (K x d, point-major), one persistent workgroup per path, and leaves the
 * traces resident in HBM exactly as pfmi_set_traces would (so pfmi_fit_batch follows without a host round trip).
 * This repo's own driver (two-loop recursion, strong-Wolfe line search, maxiters as src/optimize.jl:40, stop at
 * |grad|_inf <= g_tol); Optim.LBFGS + HagerZhang of the reference is third party: trajectory parity unpinned.
 * npoints[k] = L_k + 1 out.  Callback targets -> PFMI_ERR_UNSUPPORTED (optimise on the host, pfmi_set_traces). */
int32_t pfmi_optimize_batch(pfmi_ctx *ctx, int32_t K, const double *x0, int32_t history_length, int32_t maxiters,
                            double g_tol, int64_t *npoints);
/* The same in two halves, for ONE host thread that drives several contexts (one per GPU, src/multipath.jl:190-208 fans the runs
 * out over tasks): _enqueue uploads x0 and launches the optimisations without waiting, _wait blocks until this ctx's paths are done,
 * packs the traces and returns npoints[K].  pfmi_optimize_batch == _enqueue + _wait. */
int32_t pfmi_optimize_batch_enqueue(pfmi_ctx *ctx, int32_t K, const double *x0, int32_t history_length, int32_t maxiters,
                                    double g_tol);
int32_t pfmi_optimize_batch_wait(pfmi_ctx *ctx, int64_t *npoints);
/* OptimizationTrace of path k (src/optimize.jl:94-100): theta/grad (L_k+1) x d point-major, logp L_k+1; any may be
 * NULL.  logp is only available for traces made by pfmi_optimize_batch. */
int32_t pfmi_get_trace(pfmi_ctx *ctx, int32_t k, double *theta, double *logp, double *grad);

/* ---- streaming pipeline (round 5): optimise + fit + ELBO scan as ONE enqueued dataflow ---------------------------------------
 * What src/singlepath.jl:285-325 does for one run -- optimise (:285-297), fit_mvnormals (:301-303), maximize_elbo (:306-308) -- for K
 * runs at once, with the fits and scans of the trace points a path has ALREADY produced running while the paths are still being
 * optimised (the reference overlaps whole runs over tasks, src/multipath.jl:190-208).  Equivalent to, and bit-identical with,
 *     pfmi_optimize_batch(K, x0, J, maxiters, g_tol) ; pfmi_fit_batch(J, eps) ; pfmi_elbo_batch_enqueue(N, seeds')
 * except for the LAYOUT: trace point l of path k is slot  p = k * (maxiters + 1) + l  in every per-point array of the context (status,
 * j_eff, logdet, elbo, se, ... have K * (maxiters + 1) entries; the slots a path did not reach carry status PFMI_FIT_ABSENT / NaN), and
 * `seeds` holds the runs' predrawn seed streams: seeds[k * (maxiters + 1) + i], i = 0 .. maxiters, = the (i + 1)-th UInt64 a COPY of run k's
 * rng yields (src/elbo.jl:2 draws L of them AFTER the optimisation, when L is known; here maxiters + 1 are drawn up front and the host
 * advances the run's rng by the L the path turned out to have).  The ELBO estimate of fit l (l = 1 .. L, slot k * (maxiters + 1) + l)
 * uses value l - 1; a later pfmi_pool_build_best(ctx, N_r, NULL) gives a FAILED run the value behind the ones it consumed, index L (what
 * rand(rng, fit_distribution, ndraws) would start from, src/singlepath.jl:231-233).
 * The calling thread SCHEDULES the dataflow: the optimiser (one workgroup per path) publishes its progress into page-locked host memory,
 * pfmi_stream_pump / pfmi_stream_wait launch the walk, fits and scan of each segment of trace positions once every path has produced it -- no
 * kernel ever waits for another.  After pfmi_stream_wait: pfmi_pool_build_best / pfmi_comm_psis_resample as usual, pfmi_get_fit_status and
 * pfmi_elbo_batch_wait return the slot arrays.  Built-in targets, history_length <= 16, d <= 16384, room for the fixed-stride layout;
 * PFMI_ERR_UNSUPPORTED otherwise (use the three calls above). */
int32_t pfmi_stream_enqueue(pfmi_ctx *ctx, int32_t K, const double *x0, int32_t history_length, int32_t maxiters, double g_tol,
                            double eps, int64_t N, const uint64_t *seeds);
/* seeds may be NULL in pfmi_stream_enqueue: the optimiser starts at once and the host draws the streams while it runs; pfmi_stream_seeds
 * hands them over before the first pfmi_stream_pump / pfmi_stream_wait (nothing but the optimiser is launched until then). */
int32_t pfmi_stream_seeds(pfmi_ctx *ctx, const uint64_t *seeds);
/* pfmi_stream_pump: ONE scheduling pass of the calling thread (never blocks): launches the walk, fits and scan of the trace positions every
 * path has produced since the last pass; *finished = 1 once the last segment and the reduction are enqueued.  pfmi_stream_wait pumps until
 * then and returns the points per path -- it does NOT wait for the GPU.  A host that drives several contexts pumps them in turn. */
int32_t pfmi_stream_pump(pfmi_ctx *ctx, int32_t *finished);
int32_t pfmi_stream_wait(pfmi_ctx *ctx, int64_t *npoints);
/* Give up an outstanding streaming call -- for a host that failed between pfmi_stream_enqueue and pfmi_stream_wait (an exception while it
 * drew the seed streams or pumped another context; the reference's task-based fan-out simply propagates the exception, src/multipath.jl:190-208).
 * Drains what is in flight, forgets the half-made results; the context is usable again.  No call outstanding: no-op.  (pfmi_stream_enqueue,
 * pfmi_set_traces and pfmi_optimize_batch_enqueue do the same implicitly when they find a stale call.) */
int32_t pfmi_stream_cancel(pfmi_ctx *ctx);

/* ---- fit_mvnormals / lbfgs_inverse_hessians / pdfactorize --------------------------------------- */
/* replaces fit_mvnormals (src/mvnormal.jl:14-21) = lbfgs_inverse_hessians (src/inverse_hessian.jl:25-66)
 * + lbfgs_inverse_hessian (:98-133) + WoodburyPDMat/pdfactorize (src/woodbury.jl:259-263, 201-207)
 * + mu = theta + Sigma*grad, for every point of every trace, batched on the GPU. */
int32_t pfmi_fit_batch(pfmi_ctx *ctx, int32_t history_length, double eps);
/* The `Hinit` keyword of lbfgs_inverse_hessians (src/inverse_hessian.jl:25, forwarded by fit_mvnormals, src/mvnormal.jl:14-16): how the
 * diagonal H0 is updated at every ACCEPTED step (:55).  A Julia closure cannot cross the boundary; the two the reference itself uses can:
 *   PFMI_HINIT_GILBERT            gilbert_init (src/inverse_hessian.jl:5-10), the default;
 *   PFMI_HINIT_SCALAR_YS_OVER_YY  (alpha, s, y) -> fill(y's / y'y): the Nocedal-Wright scaling Optim.LBFGS starts from, which the
 *                                 reference's own test passes as Hinit (test/inverse_hessian.jl:49, :62-66: H * grad is then the step
 *                                 the optimiser took).
 * pfmi_fit_batch_ex = pfmi_fit_batch with Hinit for THIS call; pfmi_set_hinit sets the context's default (pfmi_fit_batch and the
 * streaming pipeline use it). */
#define PFMI_HINIT_GILBERT 0
#define PFMI_HINIT_SCALAR_YS_OVER_YY 1
int32_t pfmi_fit_batch_ex(pfmi_ctx *ctx, int32_t history_length, double eps, int32_t hinit);
int32_t pfmi_set_hinit(pfmi_ctx *ctx, int32_t hinit);

/* status[P], j_eff[P] (effective history length), logdet[P], n_rejected[K]; any may be NULL */
int32_t pfmi_get_fit_status(pfmi_ctx *ctx, int32_t *status, int32_t *j_eff, double *logdet,
                            int64_t *n_rejected);

/* Materialise one fitted MvNormal{WoodburyPDMat} (so the host can build Sigma.A/.B/.D/.F.{U,Q,V}):
 * alpha[d]; B[d*2j]; D[2j*2j]; qr_factors[d*2j] + T[k*k] = QRCompactWY of U' \ B (Householder
 * vectors below the diagonal, R above; T upper triangular); V[k*k] upper Cholesky; mu[d]; k = min(d,2j).
 * Any output may be NULL. */
int32_t pfmi_get_fit(pfmi_ctx *ctx, int64_t point, double *alpha, double *B, double *D,
                     double *qr_factors, double *T, double *V, double *mu, double *logdet);

/* ---- maximize_elbo / elbo_and_samples / rand_and_logpdf ------------------------------------------ */
/* replaces maximize_elbo (src/elbo.jl:1-10) over fit_distributions[2:end] of every path:
 * for each point p that is not the first of its path: N draws x = mu + L u (src/mvnormal.jl:24-39),
 * logq, logp, ELBO mean and standard error (src/elbo.jl:12-20); then the NaN-skipping first-max
 * argmax per path (src/utils.jl:55-72).
 * seeds[P]: per-fit UInt64 seed (src/elbo.jl:2), keys the counter-based generator;
 * u_host: NULL (production: normals generated in-kernel) or P*d*N doubles, block p = the d x N
 *         standard normals of fit p (parity mode; blocks of first points are ignored);
 * elbo[P], se[P] (NaN for first points / failed fits), best_iter[K] (1-based iteration index
 * like the reference's fit_iteration; 0 when the path has no iterations). */
int32_t pfmi_elbo_batch(pfmi_ctx *ctx, int64_t N, const uint64_t *seeds, const double *u_host,
                        double *elbo, double *se, int64_t *best_iter);

/* The same in two halves (one host thread, several GPUs; or a host that wants to overlap its own work with the scan):
 * _enqueue uploads the seeds and launches the scan, the reduction and the per-path argmax without waiting (HOST_CALLBACK targets
 * evaluate on the calling thread, so for them _enqueue does the whole job); _wait blocks and downloads.  elbo / se / best_iter may
 * be NULL.  pfmi_elbo_batch == _enqueue + _wait. */
int32_t pfmi_elbo_batch_enqueue(pfmi_ctx *ctx, int64_t N, const uint64_t *seeds, const double *u_host);
int32_t pfmi_elbo_batch_wait(pfmi_ctx *ctx, double *elbo, double *se, int64_t *best_iter);

/* The reference's `ntasks` for HOST_CALLBACK targets (src/elbo.jl:3-6 and src/resample.jl:85-92 evaluate `logp` over tasks through
 * src/utils.jl:33-49; "the log-density function must be thread-safe", src/multipath.jl:104-108): with nthreads > 1 every staged block of
 * draws is cut into nthreads contiguous column ranges and `fn` is called on each range from its own host thread (one of them the calling
 * thread) -- `fn` must then be re-entrant and callable from threads the library creates; `user` is shared.  The results do not depend on
 * nthreads (same columns, one value per column; test/singlepath.jl:173-203).  Default 1.  A closure that parallelises internally (the
 * Julia trampoline with Threads.@threads) leaves this at 1. */
int32_t pfmi_set_callback_threads(pfmi_ctx *ctx, int32_t nthreads);

/* HOST_CALLBACK targets only: wall time spent inside the user's callback and bytes of draws handed to it (device -> pinned host)
 * during the last pfmi_elbo_batch.  The draws of a block of fits are generated and downloaded while the host evaluates the
 * previous block, so elbo_batch time ~ max(callback time, generation + PCIe time). */
int32_t pfmi_callback_stats(pfmi_ctx *ctx, double *callback_seconds, double *bytes_to_host);
/* DEVICE_CALLBACK targets: bytes of draws materialised in HBM for the callback during the last pfmi_elbo_batch (written once by the
 * draw kernel, read once by the user's kernel: 16 d bytes per draw physically move) */
int32_t pfmi_callback_stats_dev(pfmi_ctx *ctx, double *bytes_in_hbm);

/* per-draw log densities of one fit from the last pfmi_elbo_batch: logp[N], logq[N] */
int32_t pfmi_get_elbo_logs(pfmi_ctx *ctx, int64_t point, double *logp, double *logq);

/* draws n0 .. n0+N-1 of fit `point` (ELBOEstimate.draws, src/elbo.jl:19; also rand(rng, dist, n),
 * src/singlepath.jl:226-233): X[d*N], logp[N], logq[N]; u_host NULL or d*N normals. */
int32_t pfmi_draws(pfmi_ctx *ctx, int64_t point, uint64_t seed, int64_t n0, int64_t N,
                   const double *u_host, double *X, double *logp, double *logq);

/* Distributions.logpdf(MvNormal(mu_p, Sigma_p), X) for arbitrary X[d*N] through the factor
 * (src/resample.jl:85-89 -> src/woodbury.jl:378-382,158-165) */
int32_t pfmi_logpdf(pfmi_ctx *ctx, int64_t point, int64_t N, const double *X, double *out);

/* ---- remaining WoodburyPDMat / PDMats operator surface of a fitted covariance -------------- */
/* What the HMC integrations call on a fitted metric (ext/PathfinderAdvancedHMCExt.jl:17-23,
 * ext/PathfinderDynamicHMCExt.jl:7-15).  X is d x N column-major; out is d x N, or N values for the quadratic forms. */
#define PFMI_OP_UNWHITEN 0     /* unwhiten!(r, W, x) = L x            src/woodbury.jl:401-406, 136-143 */
#define PFMI_OP_WHITEN 1       /* whiten!(r, W, x)   = L \ x          src/woodbury.jl:410-415, 158-165 */
#define PFMI_OP_RMUL 2         /* R x  (lmul!(R, x))                  src/woodbury.jl:129-135          */
#define PFMI_OP_INVUNWHITEN 3  /* invunwhiten!(r, W, x) = R \ x       src/woodbury.jl:417-422, 151-157 */
#define PFMI_OP_MUL 4          /* mul!(y, W, x) = L (R x)             src/woodbury.jl:340-349, 64-68   */
#define PFMI_OP_SOLVE 5        /* W \ x = R \ (L \ x)                src/woodbury.jl:344, 70-74       */
#define PFMI_OP_QUAD 6         /* quad(W, x)    = |R x|^2 per column  src/woodbury.jl:384-397          */
#define PFMI_OP_INVQUAD 7      /* invquad(W, x) = |L \ x|^2           src/woodbury.jl:369-382          */
int32_t pfmi_woodbury_apply(pfmi_ctx *ctx, int64_t point, int32_t op, int64_t N, const double *X, double *out);
/* diag(W) = diag(A) + rowwise b' D b   (src/woodbury.jl:326-329); diag[d] */
int32_t pfmi_woodbury_diag(pfmi_ctx *ctx, int64_t point, double *diag);

/* ---- pooling, _compute_psis_result, _resample ---------------------------------------------------- */
/* draws_per_component = stack(draws) (src/multipath.jl:217): for path k take N_r draws of fit
 * `points[k]` with seed seeds[k] into the device-resident pool (d, N_r, K) and
 * log_ratios[k*N_r + n] = logp - logq (src/resample.jl:81-95; n fastest, k slowest). */
int32_t pfmi_pool_build(pfmi_ctx *ctx, int64_t N_r, const int64_t *points, const uint64_t *seeds);
/* The same with the winners picked ON THE DEVICE from the last pfmi_elbo_batch[_enqueue] -- no host round trip between the scan and
 * the pool: for path k the fit is fit_distributions[fit_iteration + 1] (src/singlepath.jl:224) = point off[k] + best_iter[k]; a
 * successful path (L > 0, ELBO finite and != -Inf, src/singlepath.jl:309-314) reuses its ELBO draws, i.e. the seed of the winning
 * fit (src/singlepath.jl:226-230, N_r > N_e continues the same counter); a failed path draws afresh with fail_seeds[k]
 * (rand(rng, fit_distribution, ndraws), src/singlepath.jl:231-233; fail_seeds NULL: the seed of the fit is used there too).
 * Only enqueues.  pfmi_pool_winners reads back what was used (blocks): points[K], seeds[K], success[K]; any may be NULL. */
int32_t pfmi_pool_build_best(pfmi_ctx *ctx, int64_t N_r, const uint64_t *fail_seeds);
int32_t pfmi_pool_winners(pfmi_ctx *ctx, int64_t *points, uint64_t *seeds, int32_t *success);
int32_t pfmi_pool_get(pfmi_ctx *ctx, double *draws, double *log_ratios);
/* device pointer to the local log-ratio shard (K_local * N_r doubles) for the RCCL all-gather */
int32_t pfmi_pool_log_ratios_dev(pfmi_ctx *ctx, void **dev_ptr, int64_t *count);

/* PSIS.psis(log_ratios) (src/resample.jl:78): log_ratios_dev is a DEVICE buffer of S doubles (the
 * all-gathered pool), weights are kept device-resident; host outputs may be NULL. */
int32_t pfmi_psis_dev(pfmi_ctx *ctx, const void *log_ratios_dev, int64_t S, double *weights,
                      double *log_weights, double *pareto_k, int64_t *tail_len);
int32_t pfmi_psis(pfmi_ctx *ctx, const double *log_ratios, int64_t S, double *weights,
                  double *log_weights, double *pareto_k, int64_t *tail_len);
/* the ctx's current PSIS result (PSISResult.weights / .log_weights, S doubles each; either may be NULL) -- for hosts that leave
 * the vectors on the device until somebody looks at them */
int32_t pfmi_psis_weights(pfmi_ctx *ctx, int64_t S, double *weights, double *log_weights);

/* _resample index selection (src/resample.jl:58-66): ndraws indices into 0..S-1.
 * importance != 0: weighted by the ctx's current PSIS weights; == 0: uniform (psis_result === nothing).
 * replace: with / without replacement.  uniforms: NULL (Philox(seed)) or ndraws doubles in [0,1). */
int32_t pfmi_resample_indices(pfmi_ctx *ctx, int64_t S, int64_t ndraws, int32_t importance,
                              int32_t replace, uint64_t seed, const double *uniforms, int64_t *idx);

/* StatsBase-compatible index selection for hosts that must match a live Julia run bit for bit: exactly
 * StatsBase.direct_sample!(rng, 1:S, ProbabilityWeights(weights, 1), x) (the weighted branch of src/resample.jl:61-66 ->
 * StatsBase.sample(rng, wv): t = rand(rng) * 1; i = 1; cw = w[1]; while cw < t && i < S: i += 1; cw += w[i]) on the ctx's
 * current PSIS weights, with the running sum accumulated sequentially in fp64 like the Julia loop and uniforms[t] = the
 * host's rand(rng) values in [0, 1).  idx is 0-based.  (pfmi_resample_indices is this library's own fixed-point
 * inverse-CDF sampler: deterministic for any launch geometry / GPU count, but not StatsBase's algorithm.) */
int32_t pfmi_resample_indices_direct(pfmi_ctx *ctx, int64_t S, int64_t ndraws, const double *uniforms, int64_t *idx);

/* draws = draws_all[:, inds] (src/resample.jl:68).  Global pool column g = k_global*N_r + n is owned by this ctx iff
 * col_offset <= g < col_offset + K_local*N_r.  Host variant: every index must be owned (PFMI_ERR_ARG otherwise -- a
 * stale or out-of-range index is an error, never a silent zero column).  draws[d*ndraws]. */
int32_t pfmi_pool_gather(pfmi_ctx *ctx, int64_t ndraws, const int64_t *idx, int64_t col_offset,
                         double *draws);
/* variant into a DEVICE buffer for hosts that assemble a result themselves: columns not owned are written as zeros (a sum over the
 * ranks' buffers is then the result; pfmi_comm_* itself moves only the owned columns) */
int32_t pfmi_pool_gather_dev(pfmi_ctx *ctx, int64_t ndraws, const int64_t *idx, int64_t col_offset,
                             void *draws_dev);

/* ---- multi-GPU: the pooled stage over paths sharded across GPUs (RCCL over xGMI) ----------------------------------- */
/* Runs are independent until pooling (src/multipath.jl:190-208); draws_per_component = stack(draws), _compute_psis_result and
 * _resample (src/multipath.jl:215-225) see every run.  Each GPU owns a contiguous block of paths (its ctx was fed only those
 * traces; pool order stays k-major, src/resample.jl:93) and has called pfmi_pool_build.  The blocks may have DIFFERENT lengths (any nruns
 * over any number of GPUs, src/multipath.jl:131-146: e.g. 20 runs over 8 GPUs = 3 3 3 3 2 2 2 2); draws per run and dimension must agree.
 * A pfmi_comm joins the contexts:
 *   pfmi_comm_init_all   ONE host process drives G contexts, one per GPU (ncclCommInitAll) -- a single Julia / C caller;
 *   pfmi_comm_init_rank  one process per GPU: every process passes the same 128-byte id (pfmi_comm_unique_id on rank 0,
 *                        shipped by the host's launcher) -- e.g. under torch.distributed.run or MPI.
 * The result is identical for every G (extension of the ntasks invariance of test/multipath.jl:107-140). */
typedef struct pfmi_comm pfmi_comm;
int32_t pfmi_comm_unique_id(uint8_t *id128);
int32_t pfmi_comm_init_all(int32_t ngpus, pfmi_ctx *const *ctxs, pfmi_comm **out);
int32_t pfmi_comm_init_rank(pfmi_ctx *ctx, int32_t world, int32_t rank, const uint8_t *id128, pfmi_comm **out);
int32_t pfmi_comm_destroy(pfmi_comm *comm);
/* world = ranks RCCL itself reports (ncclCommCount), nlocal = contexts driven by this process, rccl_version = ncclGetVersion */
int32_t pfmi_comm_info(pfmi_comm *comm, int32_t *world, int32_t *nlocal, int32_t *rccl_version);
/* _compute_psis_result over all runs (src/multipath.jl:221): ONE all-gather of the fp64 log-ratio shards (K_r * N_r doubles from GPU r;
 * unequal shards travel padded to the largest one and are compacted back to the pool order), then PSIS.psis replicated on every GPU
 * (weights stay device resident). */
int32_t pfmi_comm_pool_psis(pfmi_comm *comm, double *pareto_k, int64_t *tail_len);
/* _resample over all runs (src/multipath.jl:225, src/resample.jl:58-72): index selection replicated on every GPU (same
 * arguments as pfmi_resample_indices; identical indices by construction, checked); the selected columns reach the caller by
 * OWNER-ONLY transfers -- no GPU other than the one that owns a column touches it, no rank allocates d x ndraws unless it returns it:
 *   pfmi_comm_init_all   every GPU stores the columns it owns straight into draws[d * ndraws] (zero-copy when `draws` is page-locked
 *                        memory from pfmi_host_alloc; through a page-locked staging block of the communicator + one host copy otherwise);
 *   pfmi_comm_init_rank  the owners send their columns to RANK 0 (ncclSend / ncclRecv of exactly the owned columns): `draws` is filled
 *                        on rank 0 only -- pass NULL on the other ranks (a non-NULL array there comes back all-NaN, never stale); idx,
 *                        pareto_k, tail_len are replicated and valid on every rank.  Costs one extra host round trip (the transfers are
 *                        sized from the indices).
 * idx: global 0-based pool columns (component id = idx / N_r). */
int32_t pfmi_comm_resample(pfmi_comm *comm, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed,
                           const double *uniforms, int64_t *idx, double *draws);
/* Both stages in one call: [all-gather + PSIS when importance != 0] -> index selection -> owner-only transfer of the selected columns;
 * under pfmi_comm_init_all everything is enqueued on every local context before the ONE wait, so a single host thread keeps all its
 * GPUs busy (src/multipath.jl:221-225).  A world of one context needs no RCCL at all (the single-GPU hot path takes this route too).
 * pareto_k / tail_len are NaN / 0 when importance == 0.
 * One process per GPU (pfmi_comm_init_rank): every rank must make the same sequence of pfmi_comm_* calls; a PFMI_ERR_* from any of
 * them is fatal for the whole group -- the ranks first agree on their shard size and local status, so a local precondition failure
 * or a size mismatch is reported on EVERY rank instead of leaving the others blocked in a collective. */
int32_t pfmi_comm_psis_resample(pfmi_comm *comm, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed,
                                const double *uniforms, double *pareto_k, int64_t *tail_len, int64_t *idx, double *draws);
/* The same in two halves: _enqueue launches all-gather, PSIS and index selection; _wait adds the owner-only transfers into `draws` (known
 * only now) and is the host round trip.  Between the two the caller may queue downloads on the member contexts with pfmi_defer_downloads:
 * _wait delivers them too.  (The library's own staged downloads of this stage are never affected by the caller's defer mode.) */
int32_t pfmi_comm_psis_resample_enqueue(pfmi_comm *comm, int64_t ndraws, int32_t importance, int32_t replace, uint64_t seed, const double *uniforms);
int32_t pfmi_comm_psis_resample_wait(pfmi_comm *comm, double *pareto_k, int64_t *tail_len, int64_t *idx, double *draws);
/* pfmi_defer_downloads(ctx, 1): from now on the download-only entry points -- pfmi_get_fit_status, pfmi_elbo_batch_wait, pfmi_psis_weights --
 * queue their copies behind whatever is enqueued and return at once; the destination arrays are filled (and a PFMI_ERR_RETRY of the scan
 * is reported) by the NEXT entry point that waits on this context (pfmi_sync, pfmi_comm_psis_resample_wait, ...): one host round trip for
 * all of them.  The arrays must stay valid until then.  (ctx, 0): back to normal, what is queued stays queued.  (ctx, -1): drop everything
 * queued (after a failure between queueing and waiting).  A queued download delivers the values of the call it was queued for: entry
 * points that enqueue work rewriting those results issue the pending copies first, in stream order. */
int32_t pfmi_defer_downloads(pfmi_ctx *ctx, int32_t mode);

/* ---- host utility --------------------------------------------------------------------------------------------------- */
/* The counter-based generator the Python / C host mirrors use for the reference's seed hierarchy (run_seeds =
 * rand!(rng, UInt64[nruns]) src/multipath.jl:162; seeds = rand!(rng, UInt64[L]) src/elbo.jl:2): out[i] = low 64 bits of
 * Philox4x32-10(counter (t0 + i, stream), key seed).  Host code only (a Julia host uses its own rng instead). */
int32_t pfmi_host_rand_u64(uint64_t seed, uint64_t t0, int64_t n, uint32_t stream, uint64_t *out);
/* m generators in one call: out[j * n + i] = value i of generator (seeds[j], first counter t0[j]) */
int32_t pfmi_host_rand_u64_multi(int32_t m, const uint64_t *seeds, const uint64_t *t0, int64_t n, uint32_t stream, uint64_t *out);

/* ---- device utilities for hosts that keep buffers on the GPU (bench, multi-GPU) ---------------- */
int32_t pfmi_malloc_dev(pfmi_ctx *ctx, int64_t bytes, void **dev_ptr);
int32_t pfmi_free_dev(pfmi_ctx *ctx, void *dev_ptr);
int32_t pfmi_memcpy_h2d(pfmi_ctx *ctx, void *dev_dst, const void *host_src, int64_t bytes);
int32_t pfmi_memcpy_d2h(pfmi_ctx *ctx, void *host_dst, const void *dev_src, int64_t bytes);

/* Page-locked host memory for LARGE result buffers (the draws of pfmi_pool_gather / pfmi_comm_resample / pfmi_comm_psis_resample, the
 * pool of pfmi_pool_get).  Every entry point accepts any host pointer; into ordinary (pageable) memory the runtime copies a large result
 * in 32 MB pieces through its own staging buffer, DMA and host copy one after the other (measured 16 GB/s: 9.7 ms for the 160 MB of draws
 * of a d = 10^4, ndraws = 2000 call), into a buffer from pfmi_host_alloc it is one DMA transfer at the link rate (54 GB/s).  Below ~16 MB
 * there is nothing to gain (measured at 8 MB).  Allocation costs milliseconds: recycle the buffers (the Python host keeps a pool). */
int32_t pfmi_host_alloc(int64_t bytes, void **host_ptr);
int32_t pfmi_host_free(void *host_ptr);

#ifdef __cplusplus
}
#endif
#endif /* PFMI_H */
