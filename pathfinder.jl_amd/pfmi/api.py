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
import warnings
from collections.abc import Sequence
from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np

from .core import Engine, StaleHandleError  # noqa: F401
from .hostrng import HostRNG, rand_u64_multi
from .optimize import OptimizationTrace, optimize_with_trace

DEFAULT_HISTORY_LENGTH = 6     # src/Pathfinder.jl:24
DEFAULT_NDRAWS_ELBO = 5        # src/Pathfinder.jl:27


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


def fit_mvnormals(points, gradients, history_length=5, engine=None, eps=1e-12):
    """fit_mvnormals(θs, ∇logpθs; history_length) -> (dists, num_bfgs_updates_rejected)
    reference src/mvnormal.jl:14-21.  Raises PosDefException like WoodburyPDMat's constructor."""
    eng = engine or Engine()
    eng.set_traces([np.asarray(points)], [np.asarray(gradients)])
    eng.fit_batch(history_length, eps)
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


def _use_device_optimizer(target, optimizer):
    builtin = getattr(target, "kind", 2) in (0, 1)
    if optimizer == "device" and not builtin:
        raise ValueError("optimizer='device' needs a built-in target (analytic gradient on the GPU)")
    return builtin if optimizer == "auto" else optimizer == "device"


# ---- batched driver shared by pathfinder / multipathfinder ------------------------------------------------
def _run_paths(eng, target, inits, run_rngs, *, dim, history_length, ndraws_elbo, ntries, init_sampler,
               optimizer_kwargs, materialise, optimizer="auto", strict=True):
    """Runs every path to success (or ntries), batching the GPU work.  Returns per-path dicts.
    The K optimisations run on the device for built-in targets (pfmi_optimize_batch, all paths in one launch) and
    through the host driver for callback targets (the reference's general case, src/optimize.jl:35-59)."""
    K = len(inits)
    state = [dict(itry=0, done=False) for _ in range(K)]
    pending = list(range(K))
    on_device = _use_device_optimizer(target, optimizer)
    okw = {k: v for k, v in optimizer_kwargs.items() if k in ("maxiters", "g_tol")}
    while pending:
        need = []
        for k in pending:
            st = state[k]
            st["itry"] += 1
            if st["itry"] == 1 and inits[k] is not None:
                st["x0"] = np.array(inits[k], dtype=np.float64)
            else:
                need.append(k)
        if need and type(init_sampler) is UniformSampler:          # src/singlepath.jl:167-168, 277 -- all runs in one Philox batch
            for k, u in zip(need, rand_u64_multi([run_rngs[k] for k in need], [dim] * len(need))):
                state[k]["x0"] = ((u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)) * 2 * init_sampler.scale \
                    - init_sampler.scale
        else:
            for k in need:
                state[k]["x0"] = init_sampler(run_rngs[k], np.empty(dim))
        if not on_device:
            for k in pending:
                state[k]["trace"] = optimize_with_trace(target, state[k]["x0"], history_length=history_length, **optimizer_kwargs)
        if on_device:     # every path in one launch; finished paths are recomputed identically from their x0
            npts = eng.optimize_batch(np.stack([s["x0"] for s in state]), history_length, **okw)
            for k in range(K):
                state[k]["trace"] = DeviceOptimizationTrace(eng, k, int(npts[k]))
                if materialise:
                    state[k]["trace"].materialise()
        else:
            eng.set_traces([s["trace"].points for s in state], [s["trace"].gradients for s in state])
        # one batched fit + ELBO over every path (finished paths are recomputed identically from their seeds).  fit_batch only
        # ENQUEUES the history walk and the fits; the per-fit seeds are drawn on the host while they run
        eng.fit_batch(history_length)
        fresh = rand_u64_multi([run_rngs[k] for k in pending], [len(state[k]["trace"]) - 1 for k in pending])
        for k, sd in zip(pending, fresh):                           # seeds = rand!(rng_k, UInt64[L_k])  (src/elbo.jl:2)
            state[k]["seeds"] = np.concatenate([[np.uint64(0)], sd]).astype(np.uint64)
        status, jeff, logdet, nrej = eng.fit_status()
        bad = np.flatnonzero(status)
        if len(bad) and strict:
            # WoodburyPDMat's constructor throws inside fit_mvnormals (src/woodbury.jl:202,205) and nothing in
            # _pathfinder / the retry loop catches it (src/singlepath.jl:259-314): the reference's call fails as a whole
            p = int(bad[0])
            k = int(np.searchsorted(eng.offsets, p, side="right") - 1)
            raise PosDefException(f"run {k + 1}, fit {p - int(eng.offsets[k]) + 1}: {_STATUS_MSG.get(int(status[p]), 'failed')} "
                                  f"({len(bad)} of {len(status)} fits failed; strict=False keeps them as NaN ELBOs instead)")
        seeds = np.concatenate([s["seeds"] for s in state])
        elbo, se, best = eng.elbo_batch(ndraws_elbo, seeds)
        new_pending = []
        for k in range(K):
            st = state[k]
            p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
            L = p1 - p0 - 1
            fit_it = int(best[k])
            ok = L > 0                                           # src/singlepath.jl:299
            if fit_it > 0:
                v = elbo[p0 + fit_it]
                ok = ok and (not np.isnan(v)) and v != -np.inf   # :309-314
            else:
                ok = False
            st.update(p0=p0, L=L, fit_iteration=fit_it, success=ok, status=status[p0:p1], jeff=jeff[p0:p1],
                      elbo=elbo[p0:p1], se=se[p0:p1], nrej=int(nrej[k]))
            if not ok and st["itry"] < ntries and not st["done"]:
                new_pending.append(k)
            else:
                st["done"] = True
        pending = new_pending
    return state, status, jeff


def _assemble_path(eng, target, st, rng, ndraws, ndraws_elbo, input_, status, jeff, materialise, warn=True):
    p0, L = st["p0"], st["L"]
    if not st["success"] and warn:
        warnings.warn(f"Pathfinder failed after {st['itry']} tries. Increase `ntries`, inspect the model for "
                      "numerical instability, or provide a more suitable `init_sampler`.")
    if st["nrej"] > 0 and warn:
        perc = round(st["nrej"] * 100 / (L + 1), 1)
        warnings.warn(f"{st['nrej']} ({perc}%) updates to the inverse Hessian estimate were rejected to keep it "
                      "positive definite.")
    dists = _make_dists(eng, p0, L + 1, status, jeff, materialise)
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
    state, status, jeff = _run_paths(eng, target, [init], [rng], dim=dim, history_length=history_length,
                                     ndraws_elbo=ndraws_elbo, ntries=ntries, init_sampler=init_sampler,
                                     optimizer_kwargs=optimizer_kwargs, materialise=materialise, optimizer=optimizer, strict=strict)
    st = state[0]
    a = _assemble_path(eng, target, st, rng, ndraws, ndraws_elbo, input, status, jeff, materialise)
    X = eng.draws(a["fit_point"], a["draw_seed"], ndraws)[0]     # src/singlepath.jl:226-233
    return PathfinderResult(input if input is not None else target, rng, target.logp,
                            a["dists"][st["fit_iteration"]], X, st["fit_iteration"], st["itry"], st["trace"],
                            a["dists"], a["ests"], st["nrej"], st["success"], a["draw_seed"], ndraws)


def multipathfinder(target, ndraws, *, init=None, nruns=-1, ndraws_elbo=DEFAULT_NDRAWS_ELBO, ndraws_per_run=None,
                    rng=None, history_length=DEFAULT_HISTORY_LENGTH, importance=True, dim=-1, init_scale=2,
                    init_sampler=None, ntries=1000, ntasks=1, ntasks_per_run=1, input=None, engine=None,
                    materialise=False, optimizer="auto", strict=True, **optimizer_kwargs):
    """Multi-path Pathfinder (reference src/multipath.jl:118-245).  optimizer, strict: see pathfinder()."""
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
    eng = engine or Engine()
    eng.set_target(target)
    state, status, jeff = _run_paths(eng, target, inits, run_rngs, dim=dim, history_length=history_length,
                                     ndraws_elbo=ndraws_elbo, ntries=ntries, init_sampler=init_sampler,
                                     optimizer_kwargs=optimizer_kwargs, materialise=materialise, optimizer=optimizer, strict=strict)
    parts = [_assemble_path(eng, target, st, r, ndraws_per_run, ndraws_elbo, input, status, jeff, materialise)
             for st, r in zip(state, run_rngs)]
    # draws_per_component = stack(draws)   (:217) -- device resident
    eng.pool_build(ndraws_per_run, [a["fit_point"] for a in parts], [a["draw_seed"] for a in parts])
    # the per-run draws (d, N_r, K) stay on the device.  A run's block is a pure function of (fit, draw_seed, N_r): when it is
    # looked at it is REGENERATED from the seed (bit-identical to the pool block, tests: pool == eng.draws), so the handle
    # does not depend on what a later resample() put into the pool; it does depend on the fits, hence the token.
    results = []
    tok = eng.fit_token()

    def _run_draws(fit_point, seed):
        eng.check_token(tok, "PathfinderResult.draws")
        return eng.draws(fit_point, seed, ndraws_per_run)[0]

    for k, (st, a) in enumerate(zip(state, parts)):
        thunk = (lambda a=a: _run_draws(a["fit_point"], a["draw_seed"]))
        results.append(PathfinderResult(input if input is not None else target, run_rngs[k], target.logp,
                                        a["dists"][st["fit_iteration"]], thunk, st["fit_iteration"],
                                        st["itry"], st["trace"], a["dists"], a["ests"], st["nrej"], st["success"],
                                        a["draw_seed"], ndraws_per_run))
        if materialise:
            results[-1].draws
    S = nruns * ndraws_per_run
    psis_result = None
    if importance:                                                                               # :220-224
        ptr, cnt = eng.pool_log_ratios_dev()
        psis_result = PSISResult(**eng.psis_dev(ptr, cnt))
    draws, ids = _resample(rng, eng, S, ndraws_per_run, psis_result, ndraws)                     # :225
    return MultiPathfinderResult(input if input is not None else target, rng, target.logp,
                                 [r.fit_distribution for r in results], draws, ids, results, psis_result, eng,
                                 ndraws_per_run)


def _resample(rng, eng, S, ndraws_per_component, psis_result, ndraws, replace=True):
    """_resample (reference src/resample.jl:58-72): indices on the device, gather, component ids."""
    seed = int(rng.rand_u64(1)[0])
    idx = eng.resample_indices(S, ndraws, importance=psis_result is not None, replace=replace, seed=seed)
    draws = eng.pool_gather(idx)
    ids = idx // ndraws_per_component + 1                          # cld.(inds, N) with 1-based inds
    return draws, ids


def resample(result, ndraws, *, rng=None, replace=True, importance=True, ndraws_per_run=None, ntasks=1):
    """resample(result::MultiPathfinderResult, ndraws; ...)  (reference src/resample.jl:20-46)"""
    rng = rng if rng is not None else result.rng
    eng = result.engine
    K = len(result.pathfinder_results)
    for r in result.pathfinder_results:
        r.fit_distribution._live()                                  # the fits must still be the engine's current ones
    if ndraws_per_run is not None:                                  # fresh candidates (:102-109)
        seeds = rng.rand_u64(K)
        eng.pool_build(ndraws_per_run, [r.fit_distribution.point for r in result.pathfinder_results], seeds)
        npr = ndraws_per_run
    else:                                                           # reuse stored draws (:97-101): the candidates are
        npr = result.pathfinder_results[0].ndraws_per_run           # pathfinder_results[k].draws, whatever an earlier
        eng.pool_build(npr, [r.fit_distribution.point for r in result.pathfinder_results],   # resample() drew fresh
                       [r.draw_seed for r in result.pathfinder_results])
    S = K * npr
    if importance:                                                  # PSIS of the (re)built pool: for stored draws this
        psis_result = PSISResult(**eng.psis_dev(*eng.pool_log_ratios_dev()))   # reproduces result.psis_result bit for bit
    else:
        psis_result = None
    draws, ids = _resample(rng, eng, S, npr, psis_result, ndraws, replace=replace)
    return MultiPathfinderResult(result.input, result.rng, result.logp, result.fit_distribution, draws, ids,
                                 result.pathfinder_results, psis_result, eng, npr)
