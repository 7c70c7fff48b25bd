"""Host L-BFGS with trace capture: produces the hot path's INPUT.

Plays the role of reference src/optimize.jl:35-121 (optimize_with_trace + OptimizationCallback +
OptimizationTrace): the trace holds every iterate theta_0..theta_L with log density and the gradient
of the LOG DENSITY (src/optimize.jl:94-100).  The reference delegates to Optim.LBFGS + HagerZhang
(src/Pathfinder.jl:29-35), which is third party and out of scope; this is this
repo's own driver (history J, gamma = s'y / y'y scaling, strong-Wolfe line search, maxiters 1000,
g_tol 1e-8).  Parity with Optim's trajectory is neither claimed nor needed: the trace is an input
shared verbatim by the CPU oracle and the GPU path.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class OptimizationTrace:
    points: np.ndarray          # (L+1, d)
    log_densities: np.ndarray   # (L+1,)
    gradients: np.ndarray       # (L+1, d)   gradient of logp

    def __len__(self):
        return len(self.log_densities)


def _zoom(phi, lo, hi, f_lo, f_hi, g_lo, f0, g0, c1, c2):
    for _ in range(30):
        a = 0.5 * (lo + hi)
        f, g, pack = phi(a)
        if (f > f0 + c1 * a * g0) or (f >= f_lo):
            hi, f_hi = a, f
        else:
            if abs(g) <= -c2 * g0:
                return a, pack
            if g * (hi - lo) >= 0:
                hi, f_hi = lo, f_lo
            lo, f_lo, g_lo = a, f, g
    return a, pack


def _line_search(fg, x, f0, g0vec, p, a_init, c1=1e-4, c2=0.9, amax=1e10):
    g0 = float(g0vec @ p)

    def phi(a):
        xn = x + a * p
        f, gv = fg(xn)
        return f, float(gv @ p), (xn, f, gv)

    a_prev, f_prev, g_prev = 0.0, f0, g0
    a = a_init
    pack = None
    for it in range(25):
        f, g, pack = phi(a)
        if not np.isfinite(f):
            a = 0.5 * (a_prev + a)
            continue
        if (f > f0 + c1 * a * g0) or (it > 0 and f >= f_prev):
            return _zoom(phi, a_prev, a, f_prev, f, g_prev, f0, g0, c1, c2)
        if abs(g) <= -c2 * g0:
            return a, pack
        if g >= 0:
            return _zoom(phi, a, a_prev, f, f_prev, g, f0, g0, c1, c2)
        a_prev, f_prev, g_prev = a, f, g
        a = min(2 * a, amax)
    return a, pack


def optimize_with_trace(target, x0, history_length=6, maxiters=1000, g_tol=1e-8, fail_on_nonfinite=True, _reject_every=0):
    """Minimise f = -logp from x0.  Returns OptimizationTrace (iterate 0 included).  `_reject_every` (tests): every n-th (s, y) pair
    is treated as failing the curvature test, like the device kernel under PFMI_LBFGS_REJECT_EVERY."""
    def fg(x):
        lp, g = target.logp_and_grad(x)
        return -lp, -g

    x = np.array(x0, dtype=np.float64)
    f, g = fg(x)
    pts, lps, grads = [x.copy()], [-f], [-g.copy()]
    S, Y = [], []
    for it in range(maxiters):
        if not np.isfinite(f) or not np.all(np.isfinite(g)):
            if fail_on_nonfinite:      # src/optimize.jl:103-105
                break
        if np.max(np.abs(g)) <= g_tol:
            break
        q = g.copy()
        alphas = []
        for s, y in zip(reversed(S), reversed(Y)):
            rho = 1.0 / (y @ s)
            a = rho * (s @ q)
            alphas.append(a)
            q -= a * y
        if S:
            q *= (S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])
        for (s, y), a in zip(zip(S, Y), reversed(alphas)):
            rho = 1.0 / (y @ s)
            b = rho * (y @ q)
            q += (a - b) * s
        p = -q
        if g @ p >= 0:                 # not a descent direction: restart
            S, Y = [], []
            p = -g
        a0 = 1.0 if S else min(1.0, 1.0 / max(np.linalg.norm(g), 1e-300))
        a, pack = _line_search(fg, x, f, g, p, a0)
        if pack is None:
            break
        xn, fn, gn = pack
        if not (np.isfinite(fn) and np.all(np.isfinite(gn))):
            # the reference's callback RECORDS the offending iterate and then stops (src/optimize.jl:96-105: NaN / +Inf log density
            # or a non-finite gradient; this driver also stops on logp = -Inf, which no line search accepts anyway)
            pts.append(np.array(xn, dtype=np.float64)); lps.append(-fn); grads.append(-np.array(gn, dtype=np.float64))
            break
        s, y = xn - x, gn - g
        if y @ s > 1e-10 * (y @ y) and not (_reject_every and (it + 1) % _reject_every == 0):
            S.append(s)
            Y.append(y)
            if len(S) > history_length:
                S.pop(0)
                Y.pop(0)
        moved = np.any(xn != x)
        x, f, g = xn, fn, gn
        pts.append(x.copy())
        lps.append(-f)
        grads.append(-g.copy())
        if not moved:
            break
    return OptimizationTrace(np.array(pts), np.array(lps), np.array(grads))
