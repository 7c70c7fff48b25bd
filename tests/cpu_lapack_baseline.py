"""CPU baseline #2 for bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE, never on the product path): the per-fit pipeline of the
reference as the reference itself runs it on a CPU -- through LAPACK / BLAS-3 -- instead of the scalar C port of `oracle/pf_oracle.c`.

The reference's hot loop is `rand_and_logpdf` (src/mvnormal.jl:24-39): `randn!(d, N)`, `unwhiten!` = `lmul!(Q, .)` on the d x N block
(src/woodbury.jl:136-143: LAPACK `gemqrt`, BLAS-3), `logp` on every column (src/elbo.jl:15), mean / var (src/elbo.jl:17-18); Julia runs
that with OpenBLAS and fans the runs out over tasks (src/multipath.jl:190-208).  Here: SciPy's `dgeqrf` / `dormqr` / `dtrmm` (the same
Householder convention; `dormqr` applies the block reflector exactly like `gemqrt` does) on d x N blocks, NumPy reductions, a vectorised
`logp`, NumPy's ziggurat `standard_normal` (Julia's `randn!` is a ziggurat too), threads over PATHS with BLAS pinned to one thread per
path.  The history walk and the Byrd compact form (O(d j^2) per fit, < 1 % of a fit) come from the oracle.

`kind: "lapack"` in the bench line.  VERDICT r4 next #6: the scalar port applies Q reflector by reflector (`dorm2r`), which a reader
cannot take for "what Julia would do on these cores"; this leg can.
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

LOG2PI = 1.8378770664093454835606594728112352797227949472755668


def usable_cores():
    """(cores this process may run on, os.cpu_count(), cgroup CPU quota in cores or None)"""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    return aff, os.cpu_count() or 1, quota


def factor(alpha, B, D):
    """pdfactorize(A::Diagonal, B, D)  (src/woodbury.jl:201-207) through LAPACK: U = sqrt(alpha), qr(U' \\ B), V = chol(I + R D R')"""
    from scipy.linalg import lapack
    d, m = B.shape
    sa = np.sqrt(alpha)
    k = min(d, m)
    if m == 0:
        return dict(sa=sa, qr=None, tau=None, V=None, k=0, logdet=float(np.sum(np.log(alpha))))
    qr, tau, _, info = lapack.dgeqrf(np.asfortranarray(B / sa[:, None]))
    assert info == 0
    R = np.triu(qr[:k, :])
    V = np.linalg.cholesky(np.eye(k) + R @ D @ R.T).T           # upper factor (throws LinAlgError like PosDefException)
    return dict(sa=sa, qr=qr, tau=tau, V=np.asfortranarray(V), k=k, logdet=float(np.sum(np.log(alpha)) + 2.0 * np.sum(np.log(np.diag(V)))))


def _q_apply(F, X, trans):
    """X <- Q X (trans = 'N') or Q' X ('T'), in place semantics of lmul!(Q, X) (src/woodbury.jl:132,140)"""
    from scipy.linalg import lapack
    k = F["k"]
    qr = F["qr"][:, :k]
    lw = F.get("_lwork")
    if lw is None:
        _, w, _ = lapack.dormqr("L", trans, qr, F["tau"][:k], X, -1)
        lw = F["_lwork"] = int(w[0])
    out, _, info = lapack.dormqr("L", trans, qr, F["tau"][:k], X, lw, overwrite_c=1)
    assert info == 0
    return out


def unwhiten(F, X):
    """lmul!(::WoodburyPDLeftFactor, x)  (src/woodbury.jl:136-143): x[1:k] <- V' x[1:k]; x <- Q x; x <- U' x"""
    from scipy.linalg import blas
    k = F["k"]
    if k:
        X[:k] = blas.dtrmm(1.0, F["V"], X[:k], side=0, lower=0, trans_a=1)
        X = _q_apply(F, X, "N")
    X *= F["sa"][:, None]
    return X


def fit_mean(F, theta, grad):
    """mu = theta + Sigma grad through the factor (src/mvnormal.jl:14-21, src/woodbury.jl:64-68, 129-143)"""
    g = np.asfortranarray((F["sa"] * grad)[:, None])
    k = F["k"]
    if k:
        g = _q_apply(F, g, "T")
        g[:k] = F["V"].T @ (F["V"] @ g[:k])
        g = _q_apply(F, g, "N")
    return theta + F["sa"] * g[:, 0]


def rand_and_logpdf(F, mu, U):
    """src/mvnormal.jl:24-39 on given standard normals U (d, N), overwritten with the draws"""
    d = U.shape[0]
    unormsq = np.einsum("ij,ij->j", U, U)
    X = unwhiten(F, U)
    X += mu[:, None]
    return X, (d * LOG2PI + F["logdet"] + unormsq) / -2.0


def path_elbo(theta, grad, J, target, N, seed, nfits=None):
    """every fit of one path: history walk + compact form (oracle), factor / draws / logp / ELBO through LAPACK.  Returns
    (elbo[L+1], draws made)."""
    from oracle import pf_oracle as po
    P, d = theta.shape
    L = P - 1 if nfits is None else min(P - 1, nfits)
    alpha_all, hl, hs, _ = po.lbfgs_history(theta[:L + 1], grad[:L + 1], J)
    rng = np.random.default_rng(seed)
    elbo = np.full(L + 1, np.nan)
    U = np.empty((d, N), order="F")
    ndraws = 0
    for l in range(1, L + 1):
        j = int(hl[l])
        if j:
            S = np.stack([theta[s + 1] - theta[s] for s in hs[l, :j]], axis=1)
            Y = np.stack([grad[s] - grad[s + 1] for s in hs[l, :j]], axis=1)
            B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
        else:
            B, D = np.zeros((d, 0)), np.zeros((0, 0))
        try:
            F = factor(alpha_all[l], B, D)
        except np.linalg.LinAlgError:
            continue
        mu = fit_mean(F, theta[l], grad[l])
        rng.standard_normal((N, d), out=U.T)                      # randn!(d, N): U is column-major, U.T its C-ordered view
        X, logq = rand_and_logpdf(F, mu, U)
        logr = target.logp(X) - logq
        elbo[l] = logr.mean()                                      # (the std-err is one more O(N) pass: var(logr) / N)
        _ = np.sqrt(logr.var(ddof=1) / N)
        ndraws += N
    return elbo, ndraws


def timed_run(traces, J, target, N, nfits, nthreads):
    """`nthreads` paths at a time (traces re-used cyclically, one path per thread), BLAS pinned to one thread per path.
    Returns (draws, seconds)."""
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:                                              # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    sel = [traces[i % len(traces)] for i in range(nthreads)]
    with ctx:
        t0 = time.perf_counter()
        if nthreads == 1:
            res = [path_elbo(sel[0].points, sel[0].gradients, J, target, N, 1, nfits)]
        else:
            with ThreadPoolExecutor(nthreads) as ex:
                res = list(ex.map(lambda it: path_elbo(it[1].points, it[1].gradients, J, target, N, 1 + it[0], nfits), enumerate(sel)))
        dt = time.perf_counter() - t0
    return sum(r[1] for r in res), dt
