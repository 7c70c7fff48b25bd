"""Shared helpers for the parity tests: synthetic traces + oracle-side target objects."""
import numpy as np

from oracle import pf_oracle as po


def oracle_target(t):
    """pfmi target -> oracle target (same parameters)."""
    if t.kind == 1:
        return po.FunnelTarget(t.d)
    return po.GaussTarget(t.mean, t.a, t.Wd if t.r else None, t.G if t.r else None, t.offset)


def make_traces(target, K, seed, scale=2.0, history_length=6, maxiters=1000):
    import pfmi
    rng = pfmi.HostRNG(seed)
    traces = []
    for k in range(K):
        x0 = rng.rand(target.d) * 2 * scale - scale
        traces.append(pfmi.optimize_with_trace(target, x0, history_length=history_length, maxiters=maxiters))
    return traces


def fit_seeds(P, seed):
    import pfmi
    return pfmi.hostrng.rand_u64(seed, np.arange(P, dtype=np.uint64), 9)


ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
STANDIN_LIB = __import__("os").path.join(ROOT, "tests", "rccl_standin", "librccl_standin.so")
DEMO_LIB = __import__("os").environ.get("PFMI_DEMO_CLOSURE_LIB") or __import__("os").path.join(ROOT, "examples", "device_logp", "liblogp_demo.so")   # (the override: experiment builds of the example closure)


def demo_device_target(tg):
    """The user-side HIP closure of examples/device_logp for a built-in target `tg` (same parameters): a
    PFMI_TARGET_DEVICE_CALLBACK target whose kernel reads the materialised draws from HBM."""
    import ctypes as C
    import pfmi
    pfmi.lib()                                            # one HIP runtime per process: libpfmi (and torch) first
    L = C.CDLL(DEMO_LIB)
    dp = C.POINTER(C.c_double)
    if tg.kind == 1:
        fn = C.cast(L.pfx_funnel_logp, C.c_void_p).value
        return pfmi.DeviceCallbackTarget(tg.d, fn, None, host=tg, keepalive=L)
    L.pfx_gauss_create.restype = C.c_void_p
    L.pfx_gauss_create.argtypes = [C.c_int32, C.c_int32, dp, dp, dp, dp, C.c_double]
    h = L.pfx_gauss_create(tg.d, tg.r, tg.mean.ctypes.data_as(dp), tg.a.ctypes.data_as(dp),
                           tg.Wd.ctypes.data_as(dp) if tg.r else None, tg.G.ctypes.data_as(dp) if tg.r else None, tg.offset)
    assert h, "pfx_gauss_create failed"
    fn = C.cast(L.pfx_gauss_logp, C.c_void_p).value
    return pfmi.DeviceCallbackTarget(tg.d, fn, C.c_void_p(h), host=tg, keepalive=(L, h))


def demo_host_target(tg):
    """The compiled HOST closure of examples/device_logp for a built-in Gaussian target `tg` (same parameters): a
    PFMI_TARGET_HOST_CALLBACK target whose function is plain C -- what a C / Julia caller hands to the library."""
    import ctypes as C
    import pfmi
    pfmi.lib()
    L = C.CDLL(DEMO_LIB)
    dp = C.POINTER(C.c_double)
    L.pfx_host_gauss_create.restype = C.c_void_p
    L.pfx_host_gauss_create.argtypes = [C.c_int32, C.c_int32, dp, dp, dp, dp, C.c_double]
    Gc = np.asfortranarray(tg.G) if tg.r else None
    h = L.pfx_host_gauss_create(tg.d, tg.r, tg.mean.ctypes.data_as(dp), tg.a.ctypes.data_as(dp),
                                tg.Wd.ctypes.data_as(dp) if tg.r else None, Gc.ctypes.data_as(dp) if tg.r else None, tg.offset)
    assert h, "pfx_host_gauss_create failed"
    fn = C.cast(L.pfx_host_gauss_logp, C.c_void_p).value
    return pfmi.HostFnTarget(tg.d, fn, C.c_void_p(h), host=tg, keepalive=(L, h, Gc))


def oracle_factor_from_gpu(f):
    """An oracle Factor whose arrays ARE the GPU's factor of one fit (pfmi_get_fit: U = sqrt(alpha), the Householder vectors, tau =
    diag(T), V).  x(u) = mu + U'Q[V'u_1; u_2] is a function of exactly these arrays, so the oracle's reflector-by-reflector apply
    (LAPACK dorm2r, oracle/pf_oracle.c:pfo_apply_q) on them pins the DRAW kernels' arithmetic for every fit -- also where the
    Householder block is numerically rank deficient and the factor itself is only defined up to roundoff (the factor is
    then pinned through the quantities that are functions of Sigma: W, logdet, mu, logq)."""
    F = po.Factor.__new__(po.Factor)
    d = len(f["alpha"])
    m = f["qr_factors"].shape[1]
    k = min(d, m)
    F.d, F.m, F.k = d, m, k
    F.alpha, F.B, F.D = np.asfortranarray(f["alpha"]), f["B"], f["D"]
    F.sqrt_alpha = np.sqrt(F.alpha)
    F.QR = np.asfortranarray(f["qr_factors"]) if m else np.zeros((d, 1), order="F")
    F.tau = np.ascontiguousarray(np.diag(f["T"])) if k else np.zeros(1)
    F.V = np.asfortranarray(f["V"]) if k else np.zeros((1, 1), order="F")
    F.status, F.logdet = 0, float(f["logdet"])
    return F


class Banana:
    """the reference's banana density with its gradient (test/test_utils.jl:29-36: y = [x1; x2 + b (x1^2 - 100); x3..], Sigma =
    diag(100, 1, ..), logp = -y' Sigma^-1 y / 2, b = 0.03) -- the target of test/inverse_hessian.jl:46-76"""
    def __init__(self, n=10, b=0.03):
        self.d, self.b = n, b
        self.sig = np.r_[100.0, np.ones(n - 1)]

    def logp_and_grad(self, x):
        y = x.copy()
        y[1] = x[1] + self.b * (x[0] ** 2 - 100.0)
        w = y / self.sig
        g = -w.copy()
        g[0] += -w[1] * 2 * self.b * x[0]
        return float(-0.5 * (y @ w)), g
