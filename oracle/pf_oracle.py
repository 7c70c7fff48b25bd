"""ctypes/numpy binding of the CPU oracle (oracle/pf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py -- never by the product package (pathfinder.jl_amd/pfmi).  See the header of pf_oracle.c
for what it restates (reference file:line) and how it is pinned.

All matrices are column-major Float64 like the Julia reference; numpy arrays passed in/out use
``order='F'`` with shape (rows, cols).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_lp = C.POINTER(C.c_long)
c_u64p = C.POINTER(C.c_uint64)
c_i64p = C.POINTER(C.c_int64)


def build(force=False):
    so = os.path.join(_HERE, "libpf_oracle.so")
    src = os.path.join(_HERE, "pf_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpf_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.pfo_logdet.restype = C.c_double
        _LIB.pfo_logp_grad.restype = C.c_double
        _LIB.pfo_findmax_skipnan.restype = C.c_long
        _LIB.pfo_psis.restype = C.c_long
        _LIB.pfo_psis_tail_length.restype = C.c_long
        _LIB.pfo_rand_u64.restype = C.c_uint64
        _LIB.pfo_weight_to_fixed.restype = C.c_uint64
        _LIB.pfo_multipath_fit_elbo.restype = C.c_long
    return _LIB


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(c_dp)


# ---- inverse_hessian.jl -------------------------------------------------------------------------
def gilbert_init(alpha, s, y):
    alpha, s, y = _f(alpha), _f(s), _f(y)
    out = np.empty_like(alpha)
    lib().pfo_gilbert_init(len(alpha), _p(alpha), _p(s), _p(y), _p(out))
    return out


def set_hinit(hinit):
    """Hinit of the history walk: 0 / "gilbert" (default) or 1 / "nocedal_wright" (test/inverse_hessian.jl:49).  Process-wide; reset it."""
    lib().pfo_set_hinit({"gilbert": 0, "nocedal_wright": 1, 0: 0, 1: 1}[hinit])


def lbfgs_inverse_hessian(alpha, S, Y):
    """S, Y ordered oldest->newest, shape (d, j).  Returns (B (d,2j), D (2j,2j))."""
    alpha, S, Y = _f(alpha), _f(S), _f(Y)
    d = len(alpha)
    j = S.shape[1] if S.ndim == 2 else 0
    B = np.zeros((d, 2 * j), order="F")
    D = np.zeros((2 * j, 2 * j), order="F")
    if j:
        lib().pfo_lbfgs_inverse_hessian(d, j, _p(alpha), _p(S), _p(Y), _p(B), _p(D))
    return B, D


def lbfgs_history(theta, grad, J, eps=1e-12):
    """theta, grad: (L+1, d) C-order (point-major).  Returns alpha_all (L+1,d), hist_len (L+1),
    hist_src (L+1,J), n_rejected."""
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    P, d = theta.shape
    L = P - 1
    alpha_all = np.empty((P, d))
    hist_len = np.zeros(P, dtype=np.int32)
    hist_src = np.full((P, J), -1, dtype=np.int32)
    rej = lib().pfo_lbfgs_history(d, L, _p(theta), _p(grad), J, C.c_double(eps), _p(alpha_all),
                                  hist_len.ctypes.data_as(c_ip), hist_src.ctypes.data_as(c_ip))
    return alpha_all, hist_len, hist_src, rej


# ---- woodbury.jl ----------------------------------------------------------------------------------
class Factor:
    """WoodburyPDFactorization with diagonal A (src/woodbury.jl:12-21, 201-207)."""

    def __init__(self, alpha, B, D):
        alpha, B, D = _f(alpha), _f(B), _f(D)
        self.d = d = len(alpha)
        self.m = m = B.shape[1] if B.ndim == 2 else 0
        self.k = k = min(d, m)
        self.alpha, self.B, self.D = alpha, B.reshape(d, m, order="F"), D.reshape(m, m, order="F")
        self.sqrt_alpha = np.empty(d)
        self.QR = np.zeros((d, max(m, 1)), order="F")
        self.tau = np.zeros(max(k, 1))
        self.V = np.zeros((max(k, 1), max(k, 1)), order="F")
        self.status = lib().pfo_pdfactorize(d, m, _p(alpha), _p(self.B), _p(self.D), _p(self.sqrt_alpha),
                                            _p(self.QR), _p(self.tau), _p(self.V))
        self.logdet = (lib().pfo_logdet(d, k, _p(self.sqrt_alpha), _p(self.V))
                       if self.status == 0 else float("nan"))

    def _apply(self, fn, X):
        X = np.array(X, dtype=np.float64, order="F", copy=True)
        vec = X.ndim == 1
        X2 = X.reshape(self.d, -1, order="F")
        fn(self.d, self.m, _p(self.sqrt_alpha), _p(self.QR), _p(self.tau), _p(self.V), _p(X2),
           C.c_long(X2.shape[1]))
        return X2[:, 0].copy() if vec else X2

    def lmul_R(self, X): return self._apply(lib().pfo_lmul_R, X)
    def lmul_L(self, X): return self._apply(lib().pfo_lmul_L, X)
    def ldiv_R(self, X): return self._apply(lib().pfo_ldiv_R, X)
    def ldiv_L(self, X): return self._apply(lib().pfo_ldiv_L, X)
    def mul_W(self, X): return self._apply(lib().pfo_mul_W, X)

    def dense(self):
        return np.diag(self.alpha) + self.B @ self.D @ self.B.T

    def fit_mean(self, theta, grad):
        theta, grad = _f(theta), _f(grad)
        mu = np.empty(self.d)
        lib().pfo_fit_mean(self.d, self.m, _p(self.sqrt_alpha), _p(self.QR), _p(self.tau), _p(self.V),
                           _p(theta), _p(grad), _p(mu))
        return mu

    def rand_and_logpdf(self, mu, U):
        """U (d,N) standard normals -> (x (d,N), logq (N))  -- src/mvnormal.jl:24-39"""
        mu = _f(mu)
        X = np.array(U, dtype=np.float64, order="F", copy=True)
        N = X.shape[1]
        logq = np.empty(N)
        lib().pfo_rand_and_logpdf(self.d, self.m, _p(self.sqrt_alpha), _p(self.QR), _p(self.tau),
                                  _p(self.V), _p(mu), C.c_double(self.logdet), C.c_long(N), _p(X), _p(logq))
        return X, logq

    def logpdf(self, mu, X):
        mu, X = _f(mu), _f(X)
        N = X.shape[1]
        out = np.empty(N)
        lib().pfo_logpdf_mvnormal(self.d, self.m, _p(self.sqrt_alpha), _p(self.QR), _p(self.tau),
                                  _p(self.V), _p(mu), C.c_double(self.logdet), C.c_long(N), _p(X), _p(out))
        return out


def householder_qr(A):
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    n, m = A.shape
    tau = np.zeros(min(n, m))
    lib().pfo_householder_qr(n, m, _p(A), _p(tau))
    return A, tau


# ---- elbo.jl / utils.jl ---------------------------------------------------------------------------
def elbo_stats(logp, logq):
    logp, logq = _f(logp), _f(logq)
    N = len(logp)
    logr = np.empty(N)
    v, se = C.c_double(), C.c_double()
    lib().pfo_elbo_stats(C.c_long(N), _p(logp), _p(logq), _p(logr), C.byref(v), C.byref(se))
    return v.value, se.value, logr


def findmax_skipnan(x):
    x = _f(x)
    v = C.c_double()
    i = lib().pfo_findmax_skipnan(C.c_long(len(x)), _p(x), C.byref(v))
    return v.value, int(i)


# ---- targets ----------------------------------------------------------------------------------------
class GaussTarget:
    """logp(x) = offset - 1/2 [ sum a_i e_i^2 - ||G Wd' e||^2 ], e = x - mean (kind 0)."""
    kind = 0

    def __init__(self, mean, a, Wd=None, G=None, offset=0.0):
        self.mean, self.a = _f(mean), _f(a)
        self.d = len(self.mean)
        self.r = 0 if Wd is None else np.asarray(Wd).shape[1]
        self.Wd = _f(Wd) if self.r else np.zeros((self.d, 1), order="F")
        self.G = _f(G) if self.r else np.zeros((1, 1), order="F")
        self.offset = float(offset)

    def logp(self, X):
        X = _f(X)
        X2 = X.reshape(self.d, -1, order="F")
        out = np.empty(X2.shape[1])
        lib().pfo_logp_gauss(self.d, self.r, _p(self.mean), _p(self.a), _p(self.Wd), _p(self.G),
                             C.c_double(self.offset), C.c_long(X2.shape[1]), _p(X2), _p(out))
        return out


class FunnelTarget:
    kind = 1

    def __init__(self, d):
        self.d, self.r = d, 0
        self.mean = self.a = np.zeros(1)
        self.Wd = self.G = np.zeros((1, 1), order="F")
        self.offset = 0.0

    def logp(self, X):
        X = _f(X)
        X2 = X.reshape(self.d, -1, order="F")
        out = np.empty(X2.shape[1])
        lib().pfo_logp_funnel(self.d, C.c_long(X2.shape[1]), _p(X2), _p(out))
        return out


def logp_grad(tg, x):
    """(logp, grad logp) of a built-in target at one point."""
    x = _f(x)
    g = np.empty(len(x))
    lp = lib().pfo_logp_grad(tg.kind, len(x), tg.r, _p(tg.mean), _p(tg.a), _p(tg.Wd), _p(tg.G), C.c_double(tg.offset), _p(x), _p(g))
    return lp, g


def optimize_trace(tg, x0, history_length=6, maxiters=1000, g_tol=1e-8):
    """C restatement of this repo's L-BFGS trace driver (see pfo_optimize_trace).  Returns
    (points (L+1,d), log_densities (L+1,), gradients of logp (L+1,d))."""
    x0 = _f(x0)
    d = len(x0)
    pts = np.empty((maxiters + 1, d)); grads = np.empty((maxiters + 1, d)); lps = np.empty(maxiters + 1)
    n = lib().pfo_optimize_trace(tg.kind, d, tg.r, _p(tg.mean), _p(tg.a), _p(tg.Wd), _p(tg.G), C.c_double(tg.offset),
                                 _p(x0), history_length, maxiters, C.c_double(g_tol), _p(pts), _p(lps), _p(grads))
    return pts[:n].copy(), lps[:n].copy(), grads[:n].copy()


# ---- RNG -----------------------------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    lib().pfo_philox4x32_10(ctr.ctypes.data_as(u32p), key.ctypes.data_as(u32p), out.ctypes.data_as(u32p))
    return out


def philox4x32(ctr, key, rounds):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    lib().pfo_philox4x32_r(ctr.ctypes.data_as(u32p), key.ctypes.data_as(u32p), C.c_int(rounds), out.ctypes.data_as(u32p))
    return out


NORMAL_ROUNDS = 7       # rounds of the normal-generation stream (pfo_randn4); seeds / resampling use 10


def randn_fill(seed, d, N, n0=0):
    U = np.empty((d, N), order="F")
    lib().pfo_randn_fill(C.c_uint64(int(seed)), d, C.c_long(n0), C.c_long(N), _p(U))
    return U


def icdf_words(x, x2):
    """the normals of literal Philox words x (uint32 array; x2 = the refinement words) -- pfo_icdf_normal"""
    f = lib().pfo_icdf_normal
    f.restype = C.c_double
    return np.array([f(C.c_uint32(int(a)), C.c_uint32(int(b))) for a, b in zip(x, x2)])


def rand_u64(seed, t, stream):
    return int(lib().pfo_rand_u64(C.c_uint64(int(seed)), C.c_uint64(int(t)), C.c_uint32(stream)))


# ---- PSIS / resampling -------------------------------------------------------------------------------
def psis(log_ratios):
    """Returns (log_weights normalised, weights, pareto_k, tail_len)."""
    lw = np.array(log_ratios, dtype=np.float64, copy=True)
    w = np.empty_like(lw)
    k = C.c_double()
    M = lib().pfo_psis(C.c_long(len(lw)), _p(lw), _p(w), C.byref(k))
    return lw, w, k.value, int(M)


def gpd_fit(x_sorted):
    x = _f(x_sorted)
    s, k = C.c_double(), C.c_double()
    lib().pfo_gpd_fit(C.c_long(len(x)), _p(x), C.byref(s), C.byref(k))
    return s.value, k.value


def sample_weighted(weights, ndraws, seed=0, uniforms=None):
    w = _f(weights)
    idx = np.empty(ndraws, dtype=np.int64)
    up = _p(_f(uniforms)) if uniforms is not None else None
    rc = lib().pfo_sample_weighted(C.c_long(len(w)), _p(w), C.c_long(ndraws), C.c_uint64(int(seed)), up,
                                   idx.ctypes.data_as(c_i64p))
    if rc != 0:
        raise ValueError("all weights are zero")
    return idx


def sample_direct(weights, uniforms):
    """StatsBase.direct_sample! with the given rand(rng) values (0-based indices)."""
    w, u = _f(weights), _f(uniforms)
    idx = np.empty(len(u), dtype=np.int64)
    lib().pfo_sample_direct(C.c_long(len(w)), _p(w), C.c_long(len(u)), _p(u), idx.ctypes.data_as(c_i64p))
    return idx


def sample_uniform(S, ndraws, seed=0, uniforms=None):
    idx = np.empty(ndraws, dtype=np.int64)
    up = _p(_f(uniforms)) if uniforms is not None else None
    lib().pfo_sample_uniform(C.c_long(S), C.c_long(ndraws), C.c_uint64(int(seed)), up,
                             idx.ctypes.data_as(c_i64p))
    return idx


def sample_weighted_norep(weights, ndraws, seed=0):
    w = _f(weights)
    idx = np.empty(ndraws, dtype=np.int64)
    rc = lib().pfo_sample_weighted_norep(C.c_long(len(w)), _p(w), C.c_long(ndraws), C.c_uint64(int(seed)),
                                         idx.ctypes.data_as(c_i64p))
    if rc != 0:
        raise ValueError("not enough positive weights")
    return idx


# ---- whole-path drivers ------------------------------------------------------------------------------
def path_fit_elbo(theta, grad, J, target, N, seeds, eps=1e-12, want_draws=False):
    """theta, grad (L+1, d) point-major.  seeds (L+1) uint64.  Returns dict."""
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    P, d = theta.shape
    L = P - 1
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    mu = np.empty((P, d))
    logdet = np.empty(P)
    status = np.zeros(P, dtype=np.int32)
    jeff = np.zeros(P, dtype=np.int32)
    elbo = np.empty(P)
    se = np.empty(P)
    best = C.c_long()
    rej = C.c_int()
    draws = np.zeros((d, max(N, 1)), order="F") if want_draws else None
    lpb = np.zeros(max(N, 1)) if want_draws else None
    lqb = np.zeros(max(N, 1)) if want_draws else None
    t = target
    lib().pfo_path_fit_elbo(d, L, _p(theta), _p(grad), J, C.c_double(eps), t.kind, t.r, _p(t.mean), _p(t.a),
                            _p(t.Wd), _p(t.G), C.c_double(t.offset), C.c_long(N),
                            seeds.ctypes.data_as(c_u64p), _p(mu), _p(logdet), status.ctypes.data_as(c_ip),
                            jeff.ctypes.data_as(c_ip), _p(elbo), _p(se), C.byref(best), C.byref(rej),
                            _p(draws) if want_draws else None, _p(lpb) if want_draws else None,
                            _p(lqb) if want_draws else None)
    return dict(mu=mu, logdet=logdet, status=status, j_eff=jeff, elbo=elbo, se=se, best_iter=best.value,
                n_rejected=rej.value, draws=draws, logp=lpb, logq=lqb)


def multipath_fit_elbo(offsets, theta, grad, J, target, N, seeds, nthreads=1, eps=1e-12):
    """offsets (K+1) point offsets; theta, grad (P, d); seeds (P).  Returns dict + total draws."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    grad = np.ascontiguousarray(grad, dtype=np.float64)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    K = len(offsets) - 1
    P, d = theta.shape
    elbo, se, logdet = np.empty(P), np.empty(P), np.empty(P)
    status = np.zeros(P, dtype=np.int32)
    jeff = np.zeros(P, dtype=np.int32)
    best = np.zeros(K, dtype=np.int64)
    rej = np.zeros(K, dtype=np.int32)
    t = target
    total = lib().pfo_multipath_fit_elbo(K, offsets.ctypes.data_as(c_lp), d, _p(theta), _p(grad), J,
                                         C.c_double(eps), t.kind, t.r, _p(t.mean), _p(t.a), _p(t.Wd), _p(t.G),
                                         C.c_double(t.offset), C.c_long(N), seeds.ctypes.data_as(c_u64p),
                                         _p(elbo), _p(se), best.ctypes.data_as(c_lp),
                                         status.ctypes.data_as(c_ip), jeff.ctypes.data_as(c_ip), _p(logdet),
                                         rej.ctypes.data_as(c_ip), nthreads)
    return dict(elbo=elbo, se=se, best_iter=best, status=status, j_eff=jeff, logdet=logdet, n_rejected=rej,
                total_draws=int(total))
