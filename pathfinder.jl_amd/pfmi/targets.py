"""Target log densities.  The hot path only sees ``logp(x) = -f(x)`` (reference
src/singlepath.jl:186, src/multipath.jl:159); the host optimiser additionally needs the gradient
(LogDensityProblems.logdensity_and_gradient, src/optimize.jl:1-29).

Built-in targets are evaluated on the device; ``CallbackTarget`` wraps an arbitrary host closure
(the reference's general case) and is evaluated on a host copy of the draws.
"""
import ctypes as C

import numpy as np

from . import _lib
from .hostrng import HostRNG

KIND_GAUSS, KIND_FUNNEL, KIND_CALLBACK, KIND_DEVICE_CALLBACK = 0, 1, 2, 3


class GaussTarget:
    """logp(x) = offset - 1/2 (x-m)' Sigma*^-1 (x-m),  Sigma* = diag(sigma2) + W W'  (W optional)."""
    kind = KIND_GAUSS

    def __init__(self, mean, sigma2, W=None, offset=0.0):
        self.mean = np.ascontiguousarray(mean, dtype=np.float64)
        self.d = len(self.mean)
        sigma2 = np.broadcast_to(np.asarray(sigma2, dtype=np.float64), (self.d,))
        self.a = np.ascontiguousarray(1.0 / sigma2)
        self.offset = float(offset)
        if W is None or np.asarray(W).size == 0:
            self.r = 0
            self.Wd = np.zeros((self.d, 0), order="F")
            self.G = np.zeros((0, 0), order="F")
        else:
            W = np.asarray(W, dtype=np.float64).reshape(self.d, -1)
            self.r = W.shape[1]
            self.Wd = np.asfortranarray(W * self.a[:, None])
            cap = np.eye(self.r) + W.T @ self.Wd                  # capacitance matrix
            self.G = np.asfortranarray(np.linalg.inv(np.linalg.cholesky(cap)))   # lower triangular

    def logp(self, x):
        x = np.asarray(x, dtype=np.float64)
        e = (x.T - self.mean).T if x.ndim == 2 else x - self.mean
        q = np.einsum("i...,i...->...", e * (self.a[:, None] if x.ndim == 2 else self.a), e)
        if self.r:
            g = self.G @ (self.Wd.T @ e)
            q = q - np.sum(g * g, axis=0)
        return self.offset - 0.5 * q

    def grad(self, x):
        e = x - self.mean
        g = self.a * e
        if self.r:
            g = g - self.Wd @ (self.G.T @ (self.G @ (self.Wd.T @ e)))
        return -g

    def logp_and_grad(self, x):
        return float(self.logp(x)), self.grad(x)

    def descriptor(self):
        t = _lib.pfmi_target()
        t.kind, t.d, t.r, t.offset = KIND_GAUSS, self.d, self.r, self.offset
        t.mean = self.mean.ctypes.data_as(C.POINTER(C.c_double))
        t.a = self.a.ctypes.data_as(C.POINTER(C.c_double))
        if self.r:
            t.Wd = self.Wd.ctypes.data_as(C.POINTER(C.c_double))
            t.G = self.G.ctypes.data_as(C.POINTER(C.c_double))
        return t


class FunnelTarget:
    """reference docs/src/examples/quickstart.md:229-234"""
    kind = KIND_FUNNEL

    def __init__(self, d):
        self.d = d

    def logp(self, x):
        x = np.asarray(x, dtype=np.float64)
        tau = x[0]
        ss = np.sum((x[1:] * np.exp(-tau / 2)) ** 2, axis=0)
        return ((tau / 3) ** 2 + (self.d - 1) * tau + ss) / -2

    def grad(self, x):
        tau = x[0]
        e = np.exp(-tau)
        g = np.empty_like(x)
        g[0] = -0.5 * (2 * tau / 9 + (self.d - 1) - e * np.sum(x[1:] ** 2))
        g[1:] = -e * x[1:]
        return g

    def logp_and_grad(self, x):
        return float(self.logp(x)), self.grad(x)

    def descriptor(self):
        t = _lib.pfmi_target()
        t.kind, t.d = KIND_FUNNEL, self.d
        return t


class CallbackTarget:
    """Arbitrary host closure, called one column at a time like the reference (src/elbo.jl:15).
    ``logp(x) -> float``; ``grad(x)`` optional (finite differences otherwise)."""
    kind = KIND_CALLBACK

    def __init__(self, d, logp, grad=None, logp_batch=None):
        """logp(x) -> float for one point (the reference's contract, src/elbo.jl:15); optional
        logp_batch(X) -> (n,) for X of shape (d, n) evaluates a whole block of draws at once (the C ABI hands the
        callback all draws of a block, so a vectorised target avoids n Python calls)."""
        self.d = d
        self._logp = logp
        self._grad = grad
        self._logp_batch = logp_batch

        self.pending_error = None       # an exception raised inside the ctypes callback (ctypes would print and swallow it)

        def _cb(Xp, d_, n, outp, _user):
            out = np.ctypeslib.as_array(outp, shape=(n,))
            try:
                X = np.ctypeslib.as_array(Xp, shape=(n, d_))
                if self._logp_batch is not None:
                    out[:] = np.asarray(self._logp_batch(X.T), dtype=np.float64)
                    return
                for i in range(n):
                    out[i] = self._logp(X[i])
            except BaseException as ex:  # noqa: BLE001 -- recorded, re-raised by the Engine once the C call has returned
                out[:] = np.nan           # never leave stale memory to be reduced into an ELBO
                if self.pending_error is None:
                    self.pending_error = ex

        self._cfn = _lib.LOGP_FN(_cb)   # keep alive

    def logp(self, x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            return float(self._logp(x))
        if self._logp_batch is not None:
            return np.asarray(self._logp_batch(x), dtype=np.float64)
        return np.array([self._logp(x[:, i]) for i in range(x.shape[1])])

    def grad(self, x):
        if self._grad is not None:
            return np.asarray(self._grad(x), dtype=np.float64)
        g = np.empty_like(x)
        for i in range(len(x)):
            h = 1e-6 * max(1.0, abs(x[i]))
            xp, xm = x.copy(), x.copy()
            xp[i] += h
            xm[i] -= h
            g[i] = (self._logp(xp) - self._logp(xm)) / (2 * h)
        return g

    def logp_and_grad(self, x):
        return float(self._logp(x)), self.grad(x)

    def descriptor(self):
        t = _lib.pfmi_target()
        t.kind, t.d = KIND_CALLBACK, self.d
        t.fn = C.cast(self._cfn, C.c_void_p)
        return t


class HostFnTarget:
    """A COMPILED host closure (PFMI_TARGET_HOST_CALLBACK with a raw C function pointer: what a C or Julia caller's `logp` is to the
    library -- `fn(X, d, n, out, user)`, include/pfmi.h: pfmi_logp_fn).  Unlike CallbackTarget there is no Python in the call, so
    Engine.set_callback_threads (the reference's ntasks) really runs it on several cores.  `host`: an object with logp / grad /
    logp_and_grad for the host optimiser."""
    kind = KIND_CALLBACK
    pending_error = None

    def __init__(self, d, fn, user=None, host=None, keepalive=None):
        self.d, self._fn, self._user, self.host, self._keep = d, fn, user, host, keepalive

    def _h(self):
        if self.host is None:
            raise ValueError("this HostFnTarget has no Python twin (logp / grad)")
        return self.host

    def logp(self, x): return self._h().logp(x)
    def grad(self, x): return self._h().grad(x)
    def logp_and_grad(self, x): return self._h().logp_and_grad(x)

    def descriptor(self):
        t = _lib.pfmi_target()
        t.kind, t.d = KIND_CALLBACK, self.d
        t.fn = self._fn if isinstance(self._fn, int) else C.cast(self._fn, C.c_void_p)
        t.user = self._user
        return t


class DeviceCallbackTarget:
    """Arbitrary DEVICE closure (PFMI_TARGET_DEVICE_CALLBACK): the draws stay in HBM, `dev_fn` launches the user's kernel on the
    engine's stream.  dev_fn: a C function pointer (int address or ctypes function) with the pfmi_logp_dev_fn signature
    (X_dev, d, n, out_dev, stream, user) -- e.g. from a HIP library, or an AMDGPU.jl launcher on the Julia side; `user` is passed
    through.  `host` (optional): an object with logp / grad / logp_and_grad for the host optimiser and for host-side checks (the
    reference evaluates the same closure in the optimiser and in the ELBO; a device closure needs its host twin for the former)."""
    kind = KIND_DEVICE_CALLBACK

    def __init__(self, d, dev_fn, user=None, host=None, keepalive=None):
        self.d = d
        self._fn = dev_fn
        self._user = user
        self.host = host
        self._keep = keepalive

    def _h(self):
        if self.host is None:
            raise ValueError("this DeviceCallbackTarget has no host twin (logp / grad on the host)")
        return self.host

    def logp(self, x): return self._h().logp(x)
    def grad(self, x): return self._h().grad(x)
    def logp_and_grad(self, x): return self._h().logp_and_grad(x)

    def descriptor(self):
        t = _lib.pfmi_target()
        t.kind, t.d = KIND_DEVICE_CALLBACK, self.d
        fn = self._fn
        t.dev_fn = fn if isinstance(fn, int) else C.cast(fn, C.c_void_p)
        t.user = self._user
        return t


class TorchDeviceTarget(DeviceCallbackTarget):
    """Device closure written with torch ops: fn(X) -> (n,) tensor for X of shape (n, d) (rows = draws; a zero-copy view of the
    draws in HBM), evaluated on the engine's own stream.  The Python host's answer to the reference's arbitrary `logp` closure
    (src/elbo.jl:15) without the PCIe round trip of CallbackTarget.  `host`: see DeviceCallbackTarget."""

    def __init__(self, d, fn, host=None, device=0):
        import torch

        class _View:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr, False), "version": 2}

        self.pending_error = None       # an exception raised inside the ctypes callback (ctypes would print and swallow it)

        def _cb(xp, d_, n, outp, stream, _user):
            try:
                dev = torch.device("cuda", device)
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    out = torch.as_tensor(_View(outp, (n,)), device=dev)
                    try:
                        X = torch.as_tensor(_View(xp, (n, d_)), device=dev)
                        out.copy_(fn(X).to(torch.float64).reshape(n))
                    except BaseException:
                        out.fill_(float("nan"))   # never leave stale memory to be reduced into an ELBO
                        raise
            except BaseException as ex:  # noqa: BLE001 -- recorded, re-raised by the Engine once the C call has returned
                if self.pending_error is None:
                    self.pending_error = ex

        self._cfn = _lib.LOGP_DEV_FN(_cb)
        super().__init__(d, self._cfn, None, host)


# ---- the synthetic targets of SURVEY.md 8(d) ------------------------------------------------------------
def t_iso(d):
    """logp = -|x|^2/2  (reference test/singlepath.jl:15)"""
    return GaussTarget(np.zeros(d), np.ones(d))


def t_diag(d, seed=1):
    rng = HostRNG(seed)
    m = rng.randn(d)
    logsig = -1.5 + 3.0 * rng.rand(d)
    return GaussTarget(m, np.exp(2 * logsig))


def t_lowrank(d, r=8, seed=2, wscale=1.0):
    """T_lr of SURVEY.md 8(d): Sigma* = diag(sigma^2) + W W', W_ij ~ N(0, wscale^2).  wscale = 1 is the headline definition: the r
    low-rank directions then carry variance ~ d (1000 x the diagonal's), more than an L-BFGS history of 6 pairs can represent, and the
    pooled importance weights are degenerate (Pareto k ~ 3.8).  wscale ~ 2 / sqrt(d) gives the same structure at a scale Pathfinder
    fits (bench.py's `fitted_variant` line)."""
    rng = HostRNG(seed)
    logsig = -0.5 + 1.0 * rng.rand(d)
    W = rng.randn(d * r).reshape(d, r) * wscale
    m = rng.randn(d)
    return GaussTarget(m, np.exp(2 * logsig), W)


def t_funnel(d):
    return FunnelTarget(d)
