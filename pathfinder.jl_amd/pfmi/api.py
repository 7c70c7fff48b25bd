"""Host-side mirror of the reference's public API for the hot path:

    pathfinder()       reference src/singlepath.jl:142-257
    multipathfinder()  reference src/multipath.jl:118-245
    resample()         reference src/resample.jl:20-46
    fit_mvnormals()    reference src/mvnormal.jl:14-21
    maximize_elbo()    reference src/elbo.jl:1-10

Same names, argument meaning and failure behaviour; the numerics run in libpfmi.so on the GPU.  Runs
are batched: the host performs the K optimisations (each with its own seeded rng copy, as
src/multipath.jl:190-193), then ONE fit_batch / elbo_batch / pool_build covers every path.

(The north-star host language is Julia; no Julia toolchain exists in this image, so the host mirror
is Python and the Julia `ccall` wrapper lives, untested, in pathfinder.jl_amd/julia/ -- INTEGRATION.md.)
"""
import gc
import os
import warnings
from collections.abc import Sequence
from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np

from ._lib import PfmiError, PfmiRetry
from .core import Engine, StaleHandleError  # noqa: F401
from .hostrng import HostRNG, rand_u64_multi
from .optimize import OptimizationTrace, optimize_with_trace

DEFAULT_HISTORY_LENGTH = 6     # src/Pathfinder.jl:24
DEFAULT_NDRAWS_ELBO = 5        # src/Pathfinder.jl:27
_RETRIES = 2                   # how often a call is re-enqueued after PFMI_ERR_RETRY (pfmi._lib.PfmiRetry)


class _LazySeq(Sequence):
    """list-like whose items are built on first access (the reference materialises L fits / ELBO estimates per path;
    at 10^4 fits that is pure host overhead when only a few are ever looked at)"""

    def __init__(self, n, make):
        self._n, self._make, self._items = n, make, {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if i not in self._items:
            self._items[i] = self._make(i)
        return self._items[i]


class PosDefException(ArithmeticError):
    """mirrors LinearAlgebra.PosDefException thrown by pdfactorize (src/woodbury.jl:202,205)"""


# ---- distribution objects -------------------------------------------------------------------------------
@dataclass
class WoodburyPDFactorization:      # src/woodbury.jl:12-21
    U: np.ndarray                   # diag of the (diagonal) Cholesky factor of A = sqrt(alpha)
    Q_factors: np.ndarray           # QRCompactWY factors (d x 2j)
    Q_T: np.ndarray                 # compact-WY T (k x k)
    V: np.ndarray                   # upper Cholesky factor (k x k)


@dataclass
class WoodburyPDMat:                # src/woodbury.jl:246-257
    A: np.ndarray                   # diagonal (d,)
    B: np.ndarray                   # (d, 2j)
    D: np.ndarray                   # (2j, 2j)
    F: WoodburyPDFactorization
    logdet: float

    engine: Any = field(repr=False, default=None)
    point: int = -1
    token: Any = field(repr=False, default=None)      # engine.fit_token() this handle was made under

    def dense(self):
        return np.diag(self.A) + self.B @ self.D @ self.B.T

    def _op(self, op, x):
        if self.engine is None:
            raise RuntimeError("this WoodburyPDMat was built on the host (inv / scaling): it has no device factor")
        self.engine.check_token(self.token, "WoodburyPDMat")
        return self.engine.woodbury_apply(self.point, op, x)

    # PDMats surface on the device (reference src/woodbury.jl:326-423)
    def unwhiten(self, x): return self._op("unwhiten", x)
    def whiten(self, x): return self._op("whiten", x)
    def invunwhiten(self, x): return self._op("invunwhiten", x)
    def mul(self, x): return self._op("mul", x)
    def solve(self, x): return self._op("solve", x)
    def quad(self, x): return self._op("quad", x)
    def invquad(self, x): return self._op("invquad", x)

    def diag(self):
        if self.engine is None:
            return self.A + np.einsum("ij,jk,ik->i", self.B, self.D, self.B)
        self.engine.check_token(self.token, "WoodburyPDMat")
        return self.engine.woodbury_diag(self.point)

    # inv(W) and W * c build NEW WoodburyPDMat objects on the host from the downloaded pieces, exactly as the reference does
    # (src/woodbury.jl:50-52, 218-223, 317-321, 357-360); they carry no engine handle (dense() / diag() work, the device operators do not).
    def thin_Q(self):
        """first k columns of Q = I - Vh T Vh' (Vh = unit-lower Householder vectors of F.Q_factors)"""
        d, k = self.B.shape[0], self.F.V.shape[0]
        Vh = np.tril(self.F.Q_factors[:, :k], -1) + np.eye(d, k)
        Q1 = -Vh @ (self.F.Q_T @ Vh[:k, :].T)
        Q1[:k, :] += np.eye(k)
        return Q1

    def inv(self):
        """inv(W) = WoodburyPDMat(pdunfactorize(inv(F))...): F^-1 = (U'^-1, Q, V'^-1), A = U'U, B = U'Q, D = V'V - I"""
        Ui = 1.0 / self.F.U                                       # inv(U') of the diagonal U
        k = self.F.V.shape[0]
        Vi = np.linalg.solve(self.F.V.T, np.eye(k)) if k else np.zeros((0, 0))     # inv(V'): lower triangular
        Fi = WoodburyPDFactorization(Ui, self.F.Q_factors, self.F.Q_T, Vi)
        B = Ui[:, None] * self.thin_Q()
        D = Vi.T @ Vi - np.eye(k)
        return WoodburyPDMat(Ui * Ui, B, D, Fi, -self.logdet)

    def __mul__(self, c):
        """W * c (src/woodbury.jl:357-360): c > 0 -> WoodburyPDMat(A c, B, D c), otherwise the dense matrix times c"""
        c = float(c)
        if not c > 0:
            return self.dense() * c
        sc = np.sqrt(c)
        F = WoodburyPDFactorization(self.F.U * sc, self.F.Q_factors, self.F.Q_T, None)     # V of the scaled matrix is not formed
        return WoodburyPDMat(self.A * c, self.B.copy(), self.D * c, F, self.logdet + len(self.A) * np.log(c))

    __rmul__ = __mul__

    # the rest of the surface the reference tests (test/woodbury.jl:228-309)
    @property
    def dim(self): return len(self.A)                               # PDMats.dim, src/woodbury.jl:367

    @property
    def T(self): return self                                        # adjoint / transpose: W is symmetric (src/woodbury.jl:313-315)

    def __add__(self, c):
        """W + c I (src/woodbury.jl:333-338: through PDMats' ScalMat sum, i.e. a dense matrix)"""
        return self.dense() + float(c) * np.eye(len(self.A))

    __radd__ = __add__

    def rdiv(self, X):
        """X / W for row vectors / row blocks (n,) or (r, n): (W \\ X')' with the device solve (test/woodbury.jl:299-305)"""
        X = np.asarray(X, dtype=np.float64)
        return self.solve(X.T.copy()).T if X.ndim == 2 else self.solve(X)


@dataclass
class MvNormal:
    """MvNormal{WoodburyPDMat} (src/mvnormal.jl:18).  mu / Sigma may be left to be downloaded from the engine on first access
    (multipathfinder keeps ~10^4 of these as handles)."""
    mu_: Any
    Sigma_: Any
    engine: Any = field(repr=False, default=None)
    point: int = -1
    j: int = -1                      # effective history length of this fit (needed to size B, D when materialising)
    token: Any = field(repr=False, default=None)      # engine.fit_token() this handle was made under

    def _live(self):
        self.engine.check_token(self.token, "MvNormal")

    def _materialise(self):
        if self.Sigma_ is None and self.engine is not None and self.j >= 0:
            self._live()
            f = self.engine.get_fit(self.point, self.j)
            F = WoodburyPDFactorization(np.sqrt(f["alpha"]), f["qr_factors"], f["T"], f["V"])
            self.Sigma_ = WoodburyPDMat(f["alpha"], f["B"], f["D"], F, f["logdet"], self.engine, self.point, self.token)
            self.mu_ = f["mu"]
        return self

    @property
    def mu(self): return self._materialise().mu_
    @property
    def Sigma(self): return self._materialise().Sigma_

    def logpdf(self, X):
        self._live()
        return self.engine.logpdf(self.point, X)

    def rand(self, seed, n, n0=0):
        self._live()
        return self.engine.draws(self.point, seed, n, n0)[0]


@dataclass
class ELBOEstimate:                 # src/elbo.jl:22-29
    value: float
    std_err: float
    engine: Any = field(repr=False, default=None)
    point: int = -1
    seed: int = 0
    ndraws: int = 0
    _cache: Any = field(repr=False, default=None)
    token: Any = field(repr=False, default=None)

    def _materialise(self):         # draws are regenerated on demand from the seed
        if self._cache is None:
            self.engine.check_token(self.token, "ELBOEstimate")
            self._cache = self.engine.draws(self.point, self.seed, self.ndraws)
        return self._cache

    @property
    def draws(self): return self._materialise()[0]
    @property
    def log_densities_target(self): return self._materialise()[1]
    @property
    def log_densities_fit(self): return self._materialise()[2]
    @property
    def log_density_ratios(self):
        x = self._materialise()
        return x[1] - x[2]

    def __str__(self):
        return f"ELBO estimate: {self.value:.2f} ± {self.std_err:.2f}"


@dataclass
class PSISResult:
    weights: np.ndarray
    log_weights: np.ndarray
    pareto_shape: float
    tail_length: int


@dataclass
class PathfinderResult:             # src/singlepath.jl:53-70
    input: Any
    rng: Any
    logp: Any
    fit_distribution: MvNormal
    draws_: Any = field(repr=False)     # (d, ndraws) array, or a thunk that downloads / regenerates it on first access
    fit_iteration: int = 0
    num_tries: int = 0
    optim_trace: Any = None
    fit_distributions: Any = None       # sequence of MvNormal
    elbo_estimates: Any = None          # sequence of ELBOEstimate
    num_bfgs_updates_rejected: int = 0
    success: bool = True
    draw_seed: int = 0
    ndraws_per_run: int = 0             # number of columns of `draws` (the stored candidates resample() reuses)

    @property
    def draws(self):
        if callable(self.draws_):
            self.draws_ = self.draws_()
        return self.draws_

    def __str__(self):                  # Base.show, src/singlepath.jl:72-83
        lines = ["Single-path Pathfinder result", f"  tries: {self.num_tries}", f"  draws: {self.draws.shape[1]}",
                 f"  fit iteration: {self.fit_iteration} (total: {len(self.optim_trace) - 1})"]
        if self.fit_iteration >= 1:
            lines.append(f"  fit ELBO: {str(self.elbo_estimates[self.fit_iteration - 1]).replace('ELBO estimate: ', '')}")
        lines.append(f"  fit distribution: MvNormal(dim = {self.draws.shape[0]}, Sigma = WoodburyPDMat)")
        return "\n".join(lines)

    @property
    def fit_distribution_transformed(self): return self.fit_distribution
    @property
    def draws_transformed(self): return self.draws


@dataclass
class MultiPathfinderResult:        # src/multipath.jl:31-44
    input: Any
    rng: Any
    logp: Any
    fit_distribution: List[MvNormal]        # components of the uniform mixture (src/multipath.jl:215-216)
    draws: np.ndarray
    draw_component_ids: np.ndarray          # 1-based like the reference
    pathfinder_results: List[PathfinderResult]
    psis_result: Optional[PSISResult]
    engine: Any = field(repr=False, default=None)
    ndraws_per_run: int = 0
    engines: Any = field(repr=False, default=None)      # the engines the runs are sharded over (contiguous blocks)

    @property
    def fit_distribution_transformed(self): return self.fit_distribution
    @property
    def draws_transformed(self): return self.draws

    def __str__(self):                  # Base.show, src/multipath.jl:46-65
        lines = ["Multi-path Pathfinder result", f"  runs: {len(self.pathfinder_results)}", f"  draws: {self.draws.shape[1]}"]
        if self.psis_result is not None:
            k = self.psis_result.pareto_shape
            assessment = "very bad" if k > 1 else "bad" if k > 0.7 else "ok" if k > 0.5 else "good"
            lines.append(f"  Pareto shape diagnostic: {round(k, 2)} ({assessment})")
        return "\n".join(lines)


# ---- helpers --------------------------------------------------------------------------------------------
_STATUS_MSG = {1: "A = diag(alpha) is not positive definite", 2: "C = I + R D R' is not positive definite",
               3: "non-finite factor"}


def _make_dist(eng, p, jeff, materialise=True):
    dist = MvNormal(None, None, eng, p, int(jeff[p]), eng.fit_token())
    return dist._materialise() if materialise else dist


def _make_dists(eng, p0, npts, status, jeff, materialise=True):
    if not materialise:                       # handles only: built on first access
        return _LazySeq(npts, lambda l: _make_dist(eng, p0 + l, jeff, False))
    return [_make_dist(eng, p0 + l, jeff, True) for l in range(npts)]


def fit_mvnormals(points, gradients, history_length=5, engine=None, eps=1e-12, Hinit="gilbert"):
    """fit_mvnormals(θs, ∇logpθs; history_length, Hinit, ϵ) -> (dists, num_bfgs_updates_rejected)
    reference src/mvnormal.jl:14-21 (keywords forwarded to lbfgs_inverse_hessians, src/inverse_hessian.jl:25).  Hinit: "gilbert" (the
    reference's default gilbert_init) or "nocedal_wright" ((α, s, y) -> fill(y's / y'y), test/inverse_hessian.jl:49).  Raises
    PosDefException like WoodburyPDMat's constructor."""
    eng = engine or Engine()
    eng.set_traces([np.asarray(points)], [np.asarray(gradients)])
    eng.fit_batch(history_length, eps, hinit=Hinit)
    status, jeff, _, nrej = eng.fit_status()
    bad = np.nonzero(status)[0]
    if len(bad):
        raise PosDefException(f"fit {int(bad[0])}: {_STATUS_MSG.get(int(status[bad[0]]), 'failed')}")
    return _make_dists(eng, 0, len(points), status, jeff), int(nrej[0])


def maximize_elbo(rng, target, dists, ndraws, ntasks=1):
    """maximize_elbo(rng, logp, dists, ndraws, ntasks) -> (iteration_opt, estimates)
    reference src/elbo.jl:1-10.  `dists` must be a contiguous slice of one fit_mvnormals result."""
    if len(dists) == 0:
        return 0, []                                        # src/elbo.jl:7
    eng = dists[0].engine
    seeds_l = rng.rand_u64(len(dists))                      # src/elbo.jl:2
    seeds = np.zeros(eng.P, dtype=np.uint64)
    for dist, s in zip(dists, seeds_l):
        seeds[dist.point] = s
    eng.set_target(target)
    eng.set_callback_threads(ntasks)                        # src/elbo.jl:3-6: logp over `ntasks` tasks (host closures; no-op for device targets)
    elbo, se, _ = eng.elbo_batch(ndraws, seeds)
    ests = [ELBOEstimate(float(elbo[d.point]), float(se[d.point]), eng, d.point, int(seeds[d.point]), ndraws,
                         token=eng.fit_token()) for d in dists]
    return _findmax_skipnan([e.value for e in ests])[1], ests


def _findmax_skipnan(xs):
    """reference src/utils.jl:55-72 (1-based index)"""
    state = None
    for i, xi in enumerate(xs, 1):
        if state is None:
            state = (xi, i)
            continue
        if np.isnan(xi):
            continue
        if np.isnan(state[0]) or xi > state[0]:
            state = (xi, i)
    return state


class UniformSampler:               # src/singlepath.jl:332-344
    def __init__(self, scale):
        if not scale > 0:
            raise ValueError("scale of uniform sampler must be positive.")   # DomainError
        self.scale = scale

    def __call__(self, rng, point):
        point[:] = rng.rand(len(point)) * 2 * self.scale - self.scale
        return point


class DeviceOptimizationTrace:
    """OptimizationTrace (src/optimize.jl:94-100) of a path optimised on the device; the arrays are downloaded on
    first access (valid while the engine still holds this batch of traces)."""

    def __init__(self, engine, k, n):
        self._eng, self._k, self._n, self._cache = engine, k, n, None
        self._token = engine.trace_token()

    def __len__(self):
        return self._n

    def materialise(self):
        if self._cache is None:
            self._eng.check_token(self._token, "DeviceOptimizationTrace")
            self._cache = self._eng.get_trace(self._k)
        return self

    @property
    def points(self): return self.materialise()._cache[0]
    @property
    def log_densities(self): return self.materialise()._cache[1]
    @property
    def gradients(self): return self.materialise()._cache[2]


def _use_device_optimizer(target, optimizer, history_length=DEFAULT_HISTORY_LENGTH):
    builtin = getattr(target, "kind", 2) in (0, 1)
    if optimizer == "device" and not builtin:
        raise ValueError("optimizer='device' needs a built-in target (analytic gradient on the GPU)")
    if optimizer == "auto":                                  # the device L-BFGS keeps a path's vectors in registers (d <= 16 384) and its
        return builtin and getattr(target, "d", 0) <= 16384 and history_length <= 16      # ring in LDS / a scratch of 16 pairs
    return optimizer == "device"


# ---- batched driver shared by pathfinder / multipathfinder ------------------------------------------------
def _comm_for(engs):
    """the pfmi_comm joining `engs` (cached on the first engine; a single engine forms a world of one without RCCL)"""
    from .core import Comm
    key = tuple(id(e) for e in engs)
    cache = engs[0].__dict__.setdefault("_comm_cache", {})
    if key not in cache or cache[key].h is None:
        cache[key] = Comm.init_all(engs)
    return cache[key]


def _blocks(K, G):
    """contiguous blocks of paths, one per engine, the first K % G one path longer (pool order stays k-major, src/resample.jl:93; any
    nruns over any number of engines, like the reference's nruns over ntasks, src/multipath.jl:131-146, 190-208)"""
    if K < G:
        raise ValueError(f"nruns={K} is smaller than the number of engines {G}: every engine needs at least one run")
    base, rem = divmod(K, G)
    out, k0 = [], 0
    for g in range(G):
        k1 = k0 + base + (1 if g < rem else 0)
        out.append((k0, k1))
        k0 = k1
    return out


def _run_paths(engs, target, inits, run_rngs, *, dim, history_length, ndraws_elbo, ntries, init_sampler,
               optimizer_kwargs, materialise, optimizer="auto", strict=True, pool=None):
    """Runs every path to success (or ntries), batching the GPU work over the engines (contiguous blocks of paths, one engine per
    GPU, driven by THIS host thread: every stage is enqueued on all engines before the first wait).  Returns per-path dicts.
    The K optimisations run on the device for built-in targets (pfmi_optimize_batch, all paths of an engine in one launch) and
    through the host driver for callback targets (the reference's general case, src/optimize.jl:35-59).
    pool = dict(N_r, ndraws, importance, replace, seed): the pooled stage (winners picked on the device, pool, PSIS, index selection,
    gather) is enqueued right behind the ELBO scan -- no host round trip in between; its result is valid when no path needed a retry
    and is recomputed with the retry otherwise."""
    K = len(inits)
    G = len(engs)
    blocks = _blocks(K, G)
    state = [dict(itry=0, done=False) for _ in range(K)]
    for g, (k0, k1) in enumerate(blocks):
        for k in range(k0, k1):
            state[k].update(eng=engs[g], g=g, kl=k - k0)
    pending = list(range(K))
    on_device = _use_device_optimizer(target, optimizer, history_length)
    stream_ok = on_device and history_length <= 16 and os.environ.get("PFMI_NO_STREAM") != "1"
    okw = {k: v for k, v in optimizer_kwargs.items() if k in ("maxiters", "g_tol")}
    pooled = None
    while pending:
        pooled_error = None
        tables = None                                               # per engine: (fit_status, elbo_batch_wait, psis_weights) delivered by the pooled wait
        need = []
        for k in pending:
            st = state[k]
            st["itry"] += 1
            if st["itry"] == 1 and inits[k] is not None:
                st["x0"] = np.array(inits[k], dtype=np.float64)
            else:
                need.append(k)
        if need and type(init_sampler) is UniformSampler:          # src/singlepath.jl:167-168, 277 -- all runs in one Philox batch
            u = np.stack(rand_u64_multi([run_rngs[k] for k in need], [dim] * len(need)))
            x0s = ((u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)) * 2 * init_sampler.scale - init_sampler.scale
            for i, k in enumerate(need):
                state[k]["x0"] = x0s[i]
        else:
            for k in need:
                state[k]["x0"] = init_sampler(run_rngs[k], np.empty(dim))
        predrawn = None
        streamed = False
        if on_device and stream_ok:
            # ONE dataflow per engine (pfmi_stream_enqueue): the fits and ELBO scans of the trace points a path has already produced run
            # while the paths are still being optimised.  The per-fit seeds of every run for the LONGEST possible trace are drawn up front
            # from copies of the runs' rngs (a run that was finished in an earlier try keeps the stream it had).
            cap = int(okw.get("maxiters", 1000)) + 1
            try:
                for eng, (k0, k1) in zip(engs, blocks):             # the optimiser starts at once; the seed streams follow while it runs
                    eng.stream_enqueue(np.stack([s["x0"] for s in state[k0:k1]]), ndraws_elbo, None, history_length, **okw)
                streamed = True
            except PfmiError as ex:
                for e2 in engs:                                     # engines enqueued before the refusal / failure are released
                    e2.stream_cancel()
                if getattr(ex, "code", 0) != -4:                    # PFMI_ERR_UNSUPPORTED: the packed route below (on every engine)
                    raise
                stream_ok = False
            except BaseException:
                for e2 in engs:                                     #  an engine enqueued before the failure must not stay "active":
                    e2.stream_cancel()                              # a caller's persistent Engine has to survive a failed call
                raise
        if streamed:
            try:
                for k, sd in zip(pending, rand_u64_multi([run_rngs[k].copy() for k in pending], [cap] * len(pending))):
                    state[k]["stream_tab"] = sd
                predrawn = {k: state[k]["stream_tab"] for k in pending}
                for eng, (k0, k1) in zip(engs, blocks):
                    eng.stream_seeds(np.concatenate([s["stream_tab"] for s in state[k0:k1]]))
                active = list(zip(engs, blocks))
                gc_was_on = gc.isenabled()
                gc.disable()                # this thread IS the pipeline's scheduler for the next millisecond or two: a full collection (tens of
                try:                        # milliseconds) in here would starve every GPU it drives (profiles/r05_experiments.md section 5)
                    while active:                                   # the calling thread schedules every engine's pipeline
                        active = [(eng, b) for eng, b in active if not eng.stream_pump()]
                finally:
                    if gc_was_on:
                        gc.enable()
                for eng, (k0, k1) in zip(engs, blocks):
                    npts = eng.stream_wait()
                    for k in range(k0, k1):
                        state[k]["trace"] = DeviceOptimizationTrace(eng, k - k0, int(npts[k - k0]))
                        if materialise:
                            state[k]["trace"].materialise()
            except BaseException:           # KeyboardInterrupt while pumping, a failure in the seed draw, an engine's pump error: no engine
                for e2 in engs:             # may be left with a streaming call outstanding
                    try:
                        e2.stream_cancel()
                    except Exception:
                        pass
                raise
        elif on_device:     # every path in one launch per engine; finished paths are recomputed identically from their x0
            for eng, (k0, k1) in zip(engs, blocks):
                eng.optimize_batch_enqueue(np.stack([s["x0"] for s in state[k0:k1]]), history_length, **okw)
            # while the optimisations run: the per-fit seeds of every pending run for the LONGEST possible trace, drawn from copies of
            # the runs' rngs (counter-based: the real rng is advanced by L_k once L_k is known, so the stream consumption is the
            # reference's: rand!(rng_k, UInt64[L_k]), src/elbo.jl:2)
            cap = int(okw.get("maxiters", 1000)) + 1
            predrawn = dict(zip(pending, rand_u64_multi([run_rngs[k].copy() for k in pending], [cap] * len(pending))))
            for eng, (k0, k1) in zip(engs, blocks):
                npts = eng.optimize_batch_wait()
                for k in range(k0, k1):
                    state[k]["trace"] = DeviceOptimizationTrace(eng, k - k0, int(npts[k - k0]))
                    if materialise:
                        state[k]["trace"].materialise()
        else:
            for k in pending:
                state[k]["trace"] = optimize_with_trace(target, state[k]["x0"], history_length=history_length, **optimizer_kwargs)
            for eng, (k0, k1) in zip(engs, blocks):
                eng.set_traces([s["trace"].points for s in state[k0:k1]], [s["trace"].gradients for s in state[k0:k1]])
        # one batched fit + ELBO over every path (finished paths are recomputed identically from their seeds).  fit_batch only
        # ENQUEUES the history walk and the fits; the per-fit seeds are drawn on the host while they run
        if not streamed:
            for eng in engs:
                eng.fit_batch(history_length)
        if predrawn is not None:
            for k in pending:
                L = len(state[k]["trace"]) - 1
                state[k]["seeds"] = np.concatenate([[np.uint64(0)], predrawn[k][:L]]).astype(np.uint64)
                state[k]["fail_seed"] = np.uint64(predrawn[k][L])     # the value the rng would produce next (see below)
                run_rngs[k].counter += L
        else:
            fresh = rand_u64_multi([run_rngs[k] for k in pending], [len(state[k]["trace"]) - 1 for k in pending])
            for k, sd in zip(pending, fresh):                       # seeds = rand!(rng_k, UInt64[L_k])  (src/elbo.jl:2)
                state[k]["seeds"] = np.concatenate([[np.uint64(0)], sd]).astype(np.uint64)
                # what rand(rng_k, fit_distribution, ndraws) would use if this try ends in failure (src/singlepath.jl:231-233): peeked
                # from a copy, the run's rng only advances when the path really fails (_assemble_path)
                state[k]["fail_seed"] = np.uint64(run_rngs[k].copy().rand_u64(1)[0])
        if not streamed:
            for eng, (k0, k1) in zip(engs, blocks):
                eng.elbo_batch_enqueue(ndraws_elbo, np.concatenate([s["seeds"] for s in state[k0:k1]]))
        if pool is not None:                                        # optimistic: right behind the scan, no host round trip
            for eng, (k0, k1) in zip(engs, blocks):
                eng.pool_build_best(pool["N_r"], np.array([s["fail_seed"] for s in state[k0:k1]], dtype=np.uint64))
            comm = _comm_for(engs)
            # the pooled stage is ENQUEUED; the tables of this try (fit statuses, ELBO table, PSIS weights) are queued behind it and the one
            # wait below delivers everything: a single host round trip per try
            try:
                comm.psis_resample_enqueue(pool["ndraws"], importance=pool["importance"], replace=pool.get("replace", True), seed=pool["seed"])
                queued = []
                try:
                    for eng in engs:
                        eng.defer(1)
                    for g, eng in enumerate(engs):
                        queued.append((eng.fit_status(), eng.elbo_batch_wait(),
                                       eng.psis_weights(len(inits) * pool["N_r"]) if (g == 0 and pool["importance"]) else None))
                except BaseException:
                    for eng in engs:
                        eng.defer(-1)                               # nothing queued may be delivered into arrays that are going away
                    raise
                for eng in engs:
                    eng.defer(0)
            except PfmiRetry:
                raise
            except Exception as ex:                                 # surfaced AFTER the fit statuses: the reference would have thrown
                pooled_error = ex                                   # PosDefException from fit_mvnormals before it ever pooled
            else:
                try:
                    res, idx, draws = comm.psis_resample_wait()
                    pooled = dict(psis=res, idx=idx, draws=draws, comm=comm, weights=queued[0][2])
                    tables = queued
                except PfmiRetry:
                    for eng in engs:
                        eng.defer(-1)
                    raise
                except Exception as ex:
                    pooled_error = ex
                    try:                                            # whatever the failed wait did not deliver
                        for eng in engs:
                            eng.sync()
                        tables = queued
                    except PfmiRetry:
                        raise
                    except Exception:
                        for eng in engs:
                            eng.defer(-1)
        # ---- first wait of this try: everything above is in flight on every engine
        new_pending = []
        all_status, all_jeff = [], []
        for g, (eng, (k0, k1)) in enumerate(zip(engs, blocks)):
            status, jeff, logdet, nrej = tables[g][0] if tables is not None else eng.fit_status()
            all_status.append(status); all_jeff.append(jeff)
            bad = np.flatnonzero((status != 0) & (status != 4))      # (4 = PFMI_FIT_ABSENT: a slot of the streaming layout no path reached)
            if len(bad) and strict:
                # WoodburyPDMat's constructor throws inside fit_mvnormals (src/woodbury.jl:202,205) and nothing in
                # _pathfinder / the retry loop catches it (src/singlepath.jl:259-314): the reference's call fails as a whole
                p = int(bad[0])
                kl = int(np.searchsorted(eng.offsets, p, side="right") - 1)
                raise PosDefException(f"run {k0 + kl + 1}, fit {p - int(eng.offsets[kl]) + 1}: {_STATUS_MSG.get(int(status[p]), 'failed')} "
                                      f"({len(bad)} of {len(status)} fits failed; strict=False keeps them as NaN ELBOs instead)")
            elbo, se, best = tables[g][1] if tables is not None else eng.elbo_batch_wait()
            for k in range(k0, k1):
                st = state[k]
                kl = k - k0
                p0 = int(eng.offsets[kl])
                p1 = p0 + eng.path_len(kl)
                L = p1 - p0 - 1
                fit_it = int(best[kl])
                ok = L > 0                                           # src/singlepath.jl:299
                if fit_it > 0:
                    v = elbo[p0 + fit_it]
                    ok = ok and (not np.isnan(v)) and v != -np.inf   # :309-314
                else:
                    ok = False
                st.update(p0=p0, L=L, fit_iteration=fit_it, success=ok, status=status[p0:p1], jeff=jeff[p0:p1],
                          elbo=elbo[p0:p1], se=se[p0:p1], nrej=int(nrej[kl]), status_all=status, jeff_all=jeff)
                if not ok and st["itry"] < ntries and not st["done"]:
                    new_pending.append(k)
                else:
                    st["done"] = True
        if new_pending:
            # some run failed this try and is retried (src/singlepath.jl:259-283: the reference retries every run up to `ntries`
            # BEFORE it ever pools): the optimistic pooled stage above ran on that failed try's draws -- whatever it returned or
            # raised (e.g. "weights are all zero" when every run's ELBO was NaN) is stale; the retry iteration recomputes it
            pooled, pooled_error = None, None
        elif pooled_error is not None:                              # no retry pending: the pooled stage's failure is the call's failure
            raise pooled_error
        pending = new_pending
    return state, pooled


def _assemble_path(st, rng, ndraws_elbo, materialise, warn=True):
    eng = st["eng"]
    p0, L = st["p0"], st["L"]
    if not st["success"] and warn:
        warnings.warn(f"Pathfinder failed after {st['itry']} tries. Increase `ntries`, inspect the model for "
                      "numerical instability, or provide a more suitable `init_sampler`.")
    if st["nrej"] > 0 and warn:
        perc = round(st["nrej"] * 100 / (L + 1), 1)
        warnings.warn(f"{st['nrej']} ({perc}%) updates to the inverse Hessian estimate were rejected to keep it "
                      "positive definite.")
    dists = _make_dists(eng, p0, L + 1, st["status_all"], st["jeff_all"], materialise)
    tok = eng.fit_token()

    def _est(i, st=st):
        l = i + 1
        return ELBOEstimate(float(st["elbo"][l]), float(st["se"][l]), eng, p0 + l, int(st["seeds"][l]), ndraws_elbo,
                            token=tok)

    ests = _LazySeq(L, _est) if not materialise else [_est(i) for i in range(L)]
    fit_it = st["fit_iteration"]
    fit_point = p0 + fit_it                                       # fit_distributions[fit_iteration + 1]
    if st["success"]:
        draw_seed = int(st["seeds"][fit_it])                      # reuse the ELBO draws, top up if needed
    else:
        draw_seed = int(rng.rand_u64(1)[0])                       # rand(rng, fit_distribution, ndraws)
        assert draw_seed == int(st["fail_seed"])                  # what pfmi_pool_build_best was given for this path
    return dict(dists=dists, ests=ests, fit_point=fit_point, draw_seed=draw_seed)


def pathfinder(target, *, rng=None, init=None, dim=-1, init_scale=2, init_sampler=None, input=None,
               history_length=DEFAULT_HISTORY_LENGTH, ndraws_elbo=DEFAULT_NDRAWS_ELBO, ndraws=None, ntries=1000,
               ntasks=1, engine=None, materialise=True, optimizer="auto", strict=True, **optimizer_kwargs):
    """Single-path Pathfinder (reference src/singlepath.jl:142-257).  optimizer: "auto" (device L-BFGS for built-in
    targets, host driver for callbacks), "device" or "host".  strict: a non-positive-definite fit raises PosDefException like
    the reference (src/woodbury.jl:202,205); strict=False keeps the library's per-fit status / NaN-ELBO behaviour."""
    rng = rng if rng is not None else HostRNG(0)
    ndraws = ndraws_elbo if ndraws is None else ndraws
    init_sampler = init_sampler or UniformSampler(init_scale)
    if init is None:
        if dim <= 0:
            dim = getattr(target, "d", -1)
        if dim <= 0:
            raise ValueError("An initial point `init` or dimension `dim` must be provided.")   # :171
    else:
        dim = len(init)
    eng = engine or Engine()
    eng.set_target(target)
    eng.set_callback_threads(ntasks)                        # `ntasks` (src/singlepath.jl:114-117 -> src/elbo.jl:3-6): a host closure is called from ntasks threads
    rng_state = (rng.seed, rng.counter) if isinstance(rng, HostRNG) else None
    for attempt in range(_RETRIES + 1):
        try:
            state, _ = _run_paths([eng], target, [init], [rng], dim=dim, history_length=history_length,
                                  ndraws_elbo=ndraws_elbo, ntries=ntries, init_sampler=init_sampler,
                                  optimizer_kwargs=optimizer_kwargs, materialise=materialise, optimizer=optimizer, strict=strict)
            break
        except PfmiRetry:       # transient (GPU shared / dispatch serialised): the library switched to its wait-free route; same seeds again
            if attempt == _RETRIES or rng_state is None:
                raise
            rng.seed, rng.counter = rng_state
    st = state[0]
    a = _assemble_path(st, rng, ndraws_elbo, materialise)
    X = eng.draws(a["fit_point"], a["draw_seed"], ndraws)[0]     # src/singlepath.jl:226-233
    return PathfinderResult(input if input is not None else target, rng, target.logp,
                            a["dists"][st["fit_iteration"]], X, st["fit_iteration"], st["itry"], st["trace"],
                            a["dists"], a["ests"], st["nrej"], st["success"], a["draw_seed"], ndraws)


def multipathfinder(target, ndraws, *, init=None, nruns=-1, ndraws_elbo=DEFAULT_NDRAWS_ELBO, ndraws_per_run=None,
                    rng=None, history_length=DEFAULT_HISTORY_LENGTH, importance=True, dim=-1, init_scale=2,
                    init_sampler=None, ntries=1000, ntasks=1, ntasks_per_run=1, input=None, engine=None, engines=None,
                    materialise=False, optimizer="auto", strict=True, **optimizer_kwargs):
    """Multi-path Pathfinder (reference src/multipath.jl:118-245).  optimizer, strict: see pathfinder().
    engines=[Engine(0), Engine(1), ...]: the runs are sharded in contiguous blocks over several GPUs driven by this one host thread
    (the reference fans them out over tasks, src/multipath.jl:190-208); the pooled stage (:215-225) goes through the RCCL group of
    the engines.  The result does not depend on the number of engines (test/multipath.jl:107-140 extended to GPUs)."""
    if init is None:
        if nruns <= 0:
            raise ValueError("A positive `nruns` must be set or `init` must be provided.")     # :148-150
        inits = [None] * nruns
    else:
        inits = list(init)
    nruns = len(inits)
    if ndraws_per_run is None:
        ndraws_per_run = max(ndraws_elbo, -(-ndraws // max(nruns, 1)))                          # :138
    if ndraws > ndraws_per_run * nruns:
        warnings.warn("More draws requested than total number of draws across replicas. Draws will not be unique.")
    rng = rng if rng is not None else HostRNG(0)
    init_sampler = init_sampler or UniformSampler(init_scale)
    if dim <= 0:
        dim = getattr(target, "d", -1) if inits[0] is None else len(inits[0])
    run_seeds = rng.rand_u64(nruns)                                                             # :162
    run_rngs = [rng.copy().seed_(int(s)) for s in run_seeds]                                    # :189-193
    engs = list(engines) if engines else [engine or Engine()]
    for e in engs:
        e.set_target(target)
        # the reference spreads the runs over `ntasks` tasks and each run's logp evaluations over `ntasks_per_run` (src/multipath.jl:104-108,
        # 190-208); here the runs of an engine are ONE batch, so a host closure sees ntasks * ntasks_per_run threads per staged block
        e.set_callback_threads(max(1, int(ntasks)) * max(1, int(ntasks_per_run)))
    resample_seed = int(rng.rand_u64(1)[0])                                                     # _resample's draw from the top-level rng (:225)
    # draws_per_component = stack(draws) (:217), _compute_psis_result (:221), _resample (:225): enqueued behind the ELBO scan
    for attempt in range(_RETRIES + 1):
        try:
            state, pooled = _run_paths(engs, target, inits, run_rngs, dim=dim, history_length=history_length,
                                       ndraws_elbo=ndraws_elbo, ntries=ntries, init_sampler=init_sampler,
                                       optimizer_kwargs=optimizer_kwargs, materialise=materialise, optimizer=optimizer, strict=strict,
                                       pool=dict(N_r=ndraws_per_run, ndraws=ndraws, importance=importance, seed=resample_seed))
            break
        except PfmiRetry:       # transient (GPU shared / dispatch serialised): every run again from its own seed, the same answer
            if attempt == _RETRIES:
                raise
            run_rngs = [rng.copy().seed_(int(s)) for s in run_seeds]
    parts = [_assemble_path(st, r, ndraws_elbo, materialise) for st, r in zip(state, run_rngs)]
    # the per-run draws (d, N_r, K) stay on the device.  A run's block is a pure function of (fit, draw_seed, N_r): when it is
    # looked at it is REGENERATED from the seed (bit-identical to the pool block, tests: pool == eng.draws), so the handle
    # does not depend on what a later resample() put into the pool; it does depend on the fits, hence the token.
    results = []

    def _run_draws(eng, tok, fit_point, seed):
        eng.check_token(tok, "PathfinderResult.draws")
        return eng.draws(fit_point, seed, ndraws_per_run)[0]

    for k, (st, a) in enumerate(zip(state, parts)):
        eng = st["eng"]
        thunk = (lambda eng=eng, tok=eng.fit_token(), a=a: _run_draws(eng, tok, a["fit_point"], a["draw_seed"]))
        results.append(PathfinderResult(input if input is not None else target, run_rngs[k], target.logp,
                                        a["dists"][st["fit_iteration"]], thunk, st["fit_iteration"],
                                        st["itry"], st["trace"], a["dists"], a["ests"], st["nrej"], st["success"],
                                        a["draw_seed"], ndraws_per_run))
        if materialise:
            results[-1].draws
    S = nruns * ndraws_per_run
    psis_result = None
    if importance:                                                                               # :220-224
        w, lw = pooled["weights"] if pooled.get("weights") is not None else engs[0].psis_weights(S)
        psis_result = PSISResult(w, lw, pooled["psis"]["pareto_shape"], pooled["psis"]["tail_length"])
    ids = pooled["idx"] // ndraws_per_run + 1                                                   # cld.(inds, N) with 1-based inds
    return MultiPathfinderResult(input if input is not None else target, rng, target.logp,
                                 [r.fit_distribution for r in results], pooled["draws"], ids, results, psis_result, engs[0],
                                 ndraws_per_run, engs)


def _resample(rng, comm, psis, ndraws_per_component, ndraws, replace=True):
    """_compute_psis_result + _resample (reference src/resample.jl:58-79) over the engines of `comm`: pooled PSIS, indices on the
    device, owner gather, component ids -- one synchronisation."""
    seed = int(rng.rand_u64(1)[0])
    res, idx, draws = comm.psis_resample(ndraws, importance=psis, replace=replace, seed=seed)
    ids = idx // ndraws_per_component + 1                          # cld.(inds, N) with 1-based inds
    return res, draws, ids


def resample(result, ndraws, *, rng=None, replace=True, importance=True, ndraws_per_run=None, ntasks=1):
    """resample(result::MultiPathfinderResult, ndraws; ...)  (reference src/resample.jl:20-46)"""
    rng = rng if rng is not None else result.rng
    engs = result.engines or [result.engine]
    runs = result.pathfinder_results
    K = len(runs)
    blocks = _blocks(K, len(engs))
    for r in runs:
        r.fit_distribution._live()                                  # the fits must still be the engine's current ones
    if ndraws_per_run is not None:                                  # fresh candidates (:102-109)
        seeds = rng.rand_u64(K)
        npr = ndraws_per_run
    else:                                                           # reuse stored draws (:97-101): the candidates are
        npr = runs[0].ndraws_per_run                                # pathfinder_results[k].draws, whatever an earlier
        seeds = [r.draw_seed for r in runs]                         # resample() drew fresh
    for eng, (k0, k1) in zip(engs, blocks):
        eng.set_callback_threads(ntasks)                            # src/resample.jl:85-92: logp over `ntasks` tasks
        eng.pool_build(npr, [r.fit_distribution.point for r in runs[k0:k1]], seeds[k0:k1])
    S = K * npr
    # PSIS of the (re)built pool: for stored draws this reproduces result.psis_result bit for bit
    res, draws, ids = _resample(rng, _comm_for(engs), importance, npr, ndraws, replace=replace)
    psis_result = None
    if importance:
        if ndraws_per_run is None and result.psis_result is not None and len(result.psis_result.weights) == S:
            # stored draws + stored PSIS: the reference hands the SAME object on (:31-41).  (A result of a fresh-candidate resample
            # stores weights of ITS candidates, not of pathfinder_results[k].draws: the reference would pair them with the wrong pool
            # and fail on the length; here the stored pool's PSIS is recomputed instead.)
            psis_result = result.psis_result
        else:
            w, lw = engs[0].psis_weights(S)
            psis_result = PSISResult(w, lw, res["pareto_shape"], res["tail_length"])
    return MultiPathfinderResult(result.input, result.rng, result.logp, result.fit_distribution, draws, ids,
                                 runs, psis_result, engs[0], npr, engs)
