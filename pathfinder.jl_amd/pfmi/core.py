"""Engine: thin Python handle over a pfmi_ctx (one GPU, one stream).  All numerics run in
libpfmi.so on the MI355X; this file only marshals arrays across the C ABI."""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from ._lib import check

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


# ---- large result arrays live in page-locked memory (include/pfmi.h: pfmi_host_alloc) --------------------------------------------
# A result of tens of megabytes downloaded into an ordinary numpy array is copied by the runtime in 32 MB pieces, DMA and host copy one
# after the other (16 GB/s); into page-locked memory it is one DMA transfer (54 GB/s).  Allocating such memory costs milliseconds, so
# blocks are recycled: when the last array that views a block is garbage collected the block goes back to a small pool.
_PIN_MIN_BYTES = int(float(os.environ.get("PFMI_PIN_MIN_MB", "16")) * (1 << 20))      # (override: A/B runs of tests/probes)
_PIN_MAX_BYTES = int(float(os.environ.get("PFMI_PIN_MAX_MB", "512")) * (1 << 20))     # larger results go to ordinary (pageable) memory
_PIN_KEEP_BYTES = int(float(os.environ.get("PFMI_PIN_KEEP_MB", "1024")) * (1 << 20))  # page-locked bytes kept for reuse, all classes together
_PIN_GRAIN = 2 << 20                     # blocks are whole multiples of 2 MB (power-of-two classes wasted up to 2x)
_pin_free = {}                           # block size (bytes) -> [addresses]
_pin_kept = [0]                          # bytes currently parked in _pin_free


def _pin_release(cls_bytes, addr):
    if _pin_kept[0] + cls_bytes <= _PIN_KEEP_BYTES:
        _pin_free.setdefault(cls_bytes, []).append(addr)
        _pin_kept[0] += cls_bytes
    else:
        _lib.lib().pfmi_host_free(C.c_void_p(addr))


def result_empty(shape, dtype=np.float64):
    """np.empty(shape, order='F') for a result the library downloads into; page-locked when large (falls back to ordinary memory
    when page-locked memory is refused)"""
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes < _PIN_MIN_BYTES or nbytes > _PIN_MAX_BYTES:     # (multi-GB page-locked blocks starve the host and take seconds to allocate)
        return np.empty(shape, dtype=dtype, order="F")
    cls_bytes = -(-nbytes // _PIN_GRAIN) * _PIN_GRAIN
    free = _pin_free.get(cls_bytes)
    if free:
        addr = free.pop()
        _pin_kept[0] -= cls_bytes
    else:
        p = C.c_void_p()
        if _lib.lib().pfmi_host_alloc(C.c_int64(cls_bytes), C.byref(p)) != 0 or not p.value:
            return np.empty(shape, dtype=dtype, order="F")
        addr = p.value
    buf = (C.c_char * nbytes).from_address(addr)
    weakref.finalize(buf, _pin_release, cls_bytes, addr).atexit = False     # (at exit the process's memory goes anyway)
    return np.frombuffer(buf, dtype=dtype).reshape(shape, order="F")


_LIVE_COMMS = weakref.WeakSet()          # a pfmi_comm holds raw pointers to its contexts: it must go before any of them does


class StaleHandleError(RuntimeError):
    """A lazy result handle (MvNormal, ELBOEstimate, trace, per-run draws) was used after the engine it points into was
    given new traces / refitted: the reference's results own their data, these handles only index device buffers."""


# Hinit of lbfgs_inverse_hessians (src/inverse_hessian.jl:25): the reference's default and the scaling its own test passes (test/inverse_hessian.jl:49)
HINIT = {"gilbert": 0, "gilbert_init": 0, 0: 0, "nocedal_wright": 1, "nocedal_wright_scaling": 1, 1: 1}


class Engine:
    def __init__(self, device=0):
        self.gen_traces = self.gen_fit = self.gen_pool = 0   # bumped by set_traces/optimize_batch, fit_batch, pool_build
        self.L = _lib.lib()
        self.ctx = C.c_void_p()
        check(self.L.pfmi_create(C.c_int32(device), C.byref(self.ctx)))
        self.device = device
        self.target = None
        self.K = self.P = self.d = 0
        self.offsets = None
        self.npoints = None
        self.J = 0

    def close(self):
        for cm in list(_LIVE_COMMS):                                    # every group this engine is a member of goes first
            if any(e is self for e in cm.engines):
                try:
                    cm.close()
                except Exception:
                    pass
        self.__dict__.pop("_comm_cache", None)
        if getattr(self, "ctx", None) is not None and self.ctx:
            self.L.pfmi_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise_target_error(self):
        """An exception raised by a Python `logp` closure inside the library's callback cannot cross the C frames: the closure's
        wrapper records it (and fills its output block with NaN); it is re-raised here, after the C call that invoked the closure."""
        ex = getattr(self.target, "pending_error", None)
        if ex is not None:
            self.target.pending_error = None
            raise ex

    # ---- generations of the device state lazy handles point into ------------------------------------
    def trace_token(self):
        return (self.gen_traces,)

    def fit_token(self):
        return (self.gen_traces, self.gen_fit)

    def check_token(self, token, what):
        cur = self.fit_token() if len(token) == 2 else self.trace_token()
        if self.ctx is None or token != cur:
            raise StaleHandleError(f"{what}: the engine holds a newer batch of traces / fits than this handle was made for "
                                   "(materialise results before reusing the engine, or use a separate Engine)")

    # ---- instrumentation -------------------------------------------------------------------------
    def sync(self):
        check(self.L.pfmi_sync(self.ctx))

    def timer_start(self):
        check(self.L.pfmi_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_double()
        check(self.L.pfmi_timer_stop(self.ctx, C.byref(ms)))
        return ms.value

    def profile(self, enable=True):
        """True / 1: every stage host-synchronised; 2: event pairs left in the stream (read by kernel_time); False / 0: off"""
        check(self.L.pfmi_profile(self.ctx, C.c_int32(int(enable))))

    def kernel_time(self, name):
        ms, n = C.c_double(), C.c_int64()
        check(self.L.pfmi_kernel_time(self.ctx, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def defer(self, mode):
        """1: fit_status / elbo_batch_wait / psis_weights only QUEUE their downloads -- the arrays they return are filled by the next call
        that waits on this engine (sync, Comm.psis_resample_wait); 0: back to normal; -1: drop what is queued (after an error)"""
        check(self.L.pfmi_defer_downloads(self.ctx, C.c_int32(int(mode))))

    # ---- inputs -------------------------------------------------------------------------------------
    def set_target(self, target):
        self.target = target                      # keep parameter arrays / callbacks alive
        desc = target.descriptor()
        check(self.L.pfmi_set_target(self.ctx, C.byref(desc)))

    def set_traces(self, thetas, grads):
        """thetas/grads: lists (one per path) of (L_k+1, d) arrays (point-major)."""
        npts = np.array([len(t) for t in thetas], dtype=np.int64)
        theta = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.float64) for t in thetas], axis=0))
        grad = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.float64) for g in grads], axis=0))
        assert theta.shape == grad.shape and theta.ndim == 2
        self.K, self.P, self.d = len(npts), int(npts.sum()), theta.shape[1]
        self.offsets = np.concatenate([[0], np.cumsum(npts)]).astype(np.int64)
        self.npoints = None
        self.gen_traces += 1
        check(self.L.pfmi_set_traces(self.ctx, C.c_int32(self.K), npts.ctypes.data_as(_i64p), C.c_int32(self.d),
                                     _d(theta), _d(grad)))

    def optimize_batch(self, x0, history_length=6, maxiters=1000, g_tol=1e-8):
        """K L-BFGS optimisations on the device from x0 (K, d); the traces stay resident (as after set_traces).
        Returns npoints (K,)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        K, d = x0.shape
        assert self.target is not None and d == self.target.d
        npts = np.empty(K, dtype=np.int64)
        self.gen_traces += 1
        check(self.L.pfmi_optimize_batch(self.ctx, C.c_int32(K), _d(x0), C.c_int32(history_length), C.c_int32(maxiters),
                                         C.c_double(g_tol), npts.ctypes.data_as(_i64p)))
        self.K, self.P, self.d = K, int(npts.sum()), d
        self.offsets = np.concatenate([[0], np.cumsum(npts)]).astype(np.int64)
        self.npoints = None
        return npts

    def optimize_batch_enqueue(self, x0, history_length=6, maxiters=1000, g_tol=1e-8):
        """first half of optimize_batch: uploads x0 and launches the K optimisations, does not wait"""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        K, d = x0.shape
        assert self.target is not None and d == self.target.d
        self.gen_traces += 1
        self._opt_K, self.K, self.P, self.d, self.offsets = K, K, 0, d, None
        self.npoints = None
        check(self.L.pfmi_optimize_batch_enqueue(self.ctx, C.c_int32(K), _d(x0), C.c_int32(history_length), C.c_int32(maxiters),
                                                 C.c_double(g_tol)))

    def optimize_batch_wait(self):
        npts = np.empty(self._opt_K, dtype=np.int64)
        check(self.L.pfmi_optimize_batch_wait(self.ctx, npts.ctypes.data_as(_i64p)))
        self.P = int(npts.sum())
        self.offsets = np.concatenate([[0], np.cumsum(npts)]).astype(np.int64)
        return npts

    # ---- streaming pipeline (include/pfmi.h: pfmi_stream_enqueue) -------------------------------------------------------------
    def stream_enqueue(self, x0, N, seeds, history_length=6, maxiters=1000, g_tol=1e-8, eps=1e-12):
        """optimise + fit + ELBO scan of K runs as ONE enqueued dataflow (the fits and scans of the trace points a path has already
        produced run while the paths are still being optimised).  Fixed-stride layout: trace point l of path k is slot
        k * (maxiters + 1) + l of every per-point array; seeds (K * (maxiters + 1),) is laid out the same way.  Only enqueues."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        K, d = x0.shape
        cap = int(maxiters) + 1
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            assert seeds.size == K * cap
        assert self.target is not None and d == self.target.d
        self.gen_traces += 1
        self.gen_fit += 1
        self.J = history_length
        self.K, self.P, self.d = K, K * cap, d
        self.offsets = np.arange(K + 1, dtype=np.int64) * cap
        self.npoints = None
        check(self.L.pfmi_stream_enqueue(self.ctx, C.c_int32(K), _d(x0), C.c_int32(history_length), C.c_int32(maxiters),
                                         C.c_double(g_tol), C.c_double(eps), C.c_int64(N),
                                         seeds.ctypes.data_as(_u64p) if seeds is not None else None))

    def stream_seeds(self, seeds):
        """the runs' predrawn seed streams (K * (maxiters + 1),) of a stream_enqueue(..., seeds=None): drawn while the optimiser already runs"""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.size == self.P
        check(self.L.pfmi_stream_seeds(self.ctx, seeds.ctypes.data_as(_u64p)))

    def stream_pump(self):
        """one scheduling pass of the pipeline (never blocks); True once its last segment has been launched"""
        fin = C.c_int32()
        check(self.L.pfmi_stream_pump(self.ctx, C.byref(fin)))
        return bool(fin.value)

    def stream_wait(self):
        """points per path of the last stream_enqueue (the first wait of that call)"""
        npts = np.empty(self.K, dtype=np.int64)
        check(self.L.pfmi_stream_wait(self.ctx, npts.ctypes.data_as(_i64p)))
        self.npoints = npts
        return npts

    def stream_cancel(self):
        """give up an outstanding stream_enqueue (a host failure between enqueue and wait): drains what is in flight; no-op otherwise"""
        check(self.L.pfmi_stream_cancel(self.ctx))

    def path_len(self, k):
        """trace points of path k (the streaming layout reserves maxiters + 1 slots per path and fills the first npoints[k])"""
        if getattr(self, "npoints", None) is not None:
            return int(self.npoints[k])
        return int(self.offsets[k + 1] - self.offsets[k])

    def get_trace(self, k, logp=True):
        n = self.path_len(k)
        theta, grad = np.empty((n, self.d)), np.empty((n, self.d))
        lp = np.empty(n) if logp else None
        check(self.L.pfmi_get_trace(self.ctx, C.c_int32(k), _d(theta), _d(lp) if logp else None, _d(grad)))
        return theta, lp, grad

    # ---- fit -------------------------------------------------------------------------------------------
    def fit_batch(self, history_length=6, eps=1e-12, hinit=None):
        """hinit: None (the context's default, gilbert_init unless set_hinit changed it), "gilbert" or "nocedal_wright" -- the `Hinit`
        keyword of lbfgs_inverse_hessians (src/inverse_hessian.jl:25)"""
        self.J = history_length
        self.gen_fit += 1
        if hinit is None:
            check(self.L.pfmi_fit_batch(self.ctx, C.c_int32(history_length), C.c_double(eps)))
        else:
            check(self.L.pfmi_fit_batch_ex(self.ctx, C.c_int32(history_length), C.c_double(eps), C.c_int32(HINIT[hinit])))

    def set_callback_threads(self, n):
        """the reference's `ntasks` for host closures (CallbackTarget): every staged block of draws is evaluated by n host threads on
        contiguous column ranges (pfmi_set_callback_threads; the closure must be thread-safe, src/multipath.jl:104-108).  A closure
        written in Python only gains where it releases the GIL (NumPy does inside its kernels)."""
        check(self.L.pfmi_set_callback_threads(self.ctx, C.c_int32(max(1, int(n)))))

    def set_hinit(self, hinit):
        """the context's default Hinit (pfmi_fit_batch and the streaming pipeline use it)"""
        check(self.L.pfmi_set_hinit(self.ctx, C.c_int32(HINIT[hinit])))

    def fit_status(self):
        status = np.empty(self.P, dtype=np.int32)
        jeff = np.empty(self.P, dtype=np.int32)
        logdet = np.empty(self.P)
        nrej = np.empty(self.K, dtype=np.int64)
        check(self.L.pfmi_get_fit_status(self.ctx, status.ctypes.data_as(_i32p), jeff.ctypes.data_as(_i32p),
                                         _d(logdet), nrej.ctypes.data_as(_i64p)))
        return status, jeff, logdet, nrej

    def get_fit(self, p, j):
        d, m = self.d, 2 * j
        k = min(d, m)
        out = dict(alpha=np.empty(d), B=np.zeros((d, m), order="F"), D=np.zeros((m, m), order="F"),
                   qr_factors=np.zeros((d, m), order="F"), T=np.zeros((k, k), order="F"),
                   V=np.zeros((k, k), order="F"), mu=np.empty(d))
        ld = C.c_double()
        check(self.L.pfmi_get_fit(self.ctx, C.c_int64(p), _d(out["alpha"]), _d(out["B"]), _d(out["D"]),
                                  _d(out["qr_factors"]), _d(out["T"]), _d(out["V"]), _d(out["mu"]), C.byref(ld)))
        out["logdet"] = ld.value
        out["j"] = j
        return out

    # ---- ELBO ---------------------------------------------------------------------------------------------
    def elbo_batch(self, N, seeds, u=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert len(seeds) == self.P
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64)
            assert u.size == self.P * self.d * N
        elbo, se = np.empty(self.P), np.empty(self.P)
        best = np.empty(self.K, dtype=np.int64)
        check(self.L.pfmi_elbo_batch(self.ctx, C.c_int64(N), seeds.ctypes.data_as(_u64p), _d(u), _d(elbo), _d(se),
                                     best.ctypes.data_as(_i64p)))
        self._raise_target_error()
        return elbo, se, best

    def elbo_batch_enqueue(self, N, seeds, u=None):
        """first half of elbo_batch: uploads the seeds, launches scan + reduction + per-path argmax, does not wait"""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert len(seeds) == self.P
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64)
            assert u.size == self.P * self.d * N
        check(self.L.pfmi_elbo_batch_enqueue(self.ctx, C.c_int64(N), seeds.ctypes.data_as(_u64p), _d(u)))
        self._raise_target_error()

    def elbo_batch_wait(self):
        elbo, se = np.empty(self.P), np.empty(self.P)
        best = np.empty(self.K, dtype=np.int64)
        check(self.L.pfmi_elbo_batch_wait(self.ctx, _d(elbo), _d(se), best.ctypes.data_as(_i64p)))
        self._raise_target_error()
        return elbo, se, best

    def callback_stats_dev(self):
        nb = C.c_double()
        check(self.L.pfmi_callback_stats_dev(self.ctx, C.byref(nb)))
        return dict(bytes_in_hbm=nb.value)

    def callback_stats(self):
        sec, nb = C.c_double(), C.c_double()
        check(self.L.pfmi_callback_stats(self.ctx, C.byref(sec), C.byref(nb)))
        return dict(callback_seconds=sec.value, bytes_to_host=nb.value)

    def elbo_logs(self, p, N):
        lp, lq = np.empty(N), np.empty(N)
        check(self.L.pfmi_get_elbo_logs(self.ctx, C.c_int64(p), _d(lp), _d(lq)))
        return lp, lq

    def draws(self, p, seed, N, n0=0, u=None):
        """(X (d,N) column-major, logp, logq) of draws n0..n0+N-1 of fit p."""
        X = np.empty((self.d, N), order="F")
        lp, lq = np.empty(N), np.empty(N)
        if u is not None:
            u = np.asfortranarray(u, dtype=np.float64)
        check(self.L.pfmi_draws(self.ctx, C.c_int64(p), C.c_uint64(int(seed)), C.c_int64(n0), C.c_int64(N), _d(u),
                                _d(X), _d(lp), _d(lq)))
        self._raise_target_error()
        return X, lp, lq

    def logpdf(self, p, X):
        X = np.asfortranarray(X, dtype=np.float64)
        N = X.shape[1]
        out = np.empty(N)
        check(self.L.pfmi_logpdf(self.ctx, C.c_int64(p), C.c_int64(N), _d(X), _d(out)))
        return out

    # ---- remaining WoodburyPDMat operator surface -------------------------------------------------------
    OPS = dict(unwhiten=0, whiten=1, rmul=2, invunwhiten=3, mul=4, solve=5, quad=6, invquad=7)

    def woodbury_apply(self, p, op, X):
        """op in Engine.OPS applied to the columns of X (d, N); quad/invquad return N values."""
        X = np.asfortranarray(X, dtype=np.float64)
        vec = X.ndim == 1
        X2 = X.reshape(self.d, -1, order="F")
        N = X2.shape[1]
        code = self.OPS[op]
        out = np.empty(N) if code >= 6 else np.empty((self.d, N), order="F")
        check(self.L.pfmi_woodbury_apply(self.ctx, C.c_int64(p), C.c_int32(code), C.c_int64(N), _d(X2), _d(out)))
        if code >= 6:
            return out[0] if vec else out
        return out[:, 0].copy() if vec else out

    def woodbury_diag(self, p):
        out = np.empty(self.d)
        check(self.L.pfmi_woodbury_diag(self.ctx, C.c_int64(p), _d(out)))
        return out

    # ---- pool / PSIS / resample -------------------------------------------------------------------------
    def pool_build(self, N_r, points, seeds):
        points = np.ascontiguousarray(points, dtype=np.int64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        self.N_r = N_r
        self.gen_pool += 1
        check(self.L.pfmi_pool_build(self.ctx, C.c_int64(N_r), points.ctypes.data_as(_i64p),
                                     seeds.ctypes.data_as(_u64p)))
        self._raise_target_error()

    def pool_build_best(self, N_r, fail_seeds=None):
        """pool of the winners picked on the device from the last elbo_batch[_enqueue]; only enqueues"""
        if fail_seeds is not None:
            fail_seeds = np.ascontiguousarray(fail_seeds, dtype=np.uint64)
            assert len(fail_seeds) == self.K
        self.N_r = N_r
        self.gen_pool += 1
        check(self.L.pfmi_pool_build_best(self.ctx, C.c_int64(N_r),
                                          fail_seeds.ctypes.data_as(_u64p) if fail_seeds is not None else None))
        self._raise_target_error()

    def pool_winners(self):
        pts = np.empty(self.K, dtype=np.int64)
        seeds = np.empty(self.K, dtype=np.uint64)
        ok = np.empty(self.K, dtype=np.int32)
        check(self.L.pfmi_pool_winners(self.ctx, pts.ctypes.data_as(_i64p), seeds.ctypes.data_as(_u64p), ok.ctypes.data_as(_i32p)))
        return pts, seeds, ok.astype(bool)

    def psis_weights(self, S):
        w, lw = np.empty(S), np.empty(S)
        check(self.L.pfmi_psis_weights(self.ctx, C.c_int64(S), _d(w), _d(lw)))
        return w, lw

    def pool_get(self, draws=True):
        S = self.K * self.N_r
        X = result_empty((self.d, self.N_r, self.K)) if draws else None
        lr = np.empty(S)
        check(self.L.pfmi_pool_get(self.ctx, _d(X), _d(lr)))
        return X, lr

    def pool_log_ratios_dev(self):
        p, n = C.c_void_p(), C.c_int64()
        check(self.L.pfmi_pool_log_ratios_dev(self.ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def psis(self, log_ratios, want_weights=True):
        lr = np.ascontiguousarray(log_ratios, dtype=np.float64)
        S = len(lr)
        w = np.empty(S) if want_weights else None
        lw = np.empty(S) if want_weights else None
        k, M = C.c_double(), C.c_int64()
        check(self.L.pfmi_psis(self.ctx, _d(lr), C.c_int64(S), _d(w), _d(lw), C.byref(k), C.byref(M)))
        return dict(weights=w, log_weights=lw, pareto_shape=k.value, tail_length=M.value)

    def psis_dev(self, dev_ptr, S, want_weights=True):
        w = np.empty(S) if want_weights else None
        lw = np.empty(S) if want_weights else None
        k, M = C.c_double(), C.c_int64()
        check(self.L.pfmi_psis_dev(self.ctx, C.c_void_p(dev_ptr), C.c_int64(S), _d(w), _d(lw), C.byref(k),
                                   C.byref(M)))
        return dict(weights=w, log_weights=lw, pareto_shape=k.value, tail_length=M.value)

    def resample_indices(self, S, ndraws, importance=True, replace=True, seed=0, uniforms=None):
        idx = np.empty(ndraws, dtype=np.int64)
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        check(self.L.pfmi_resample_indices(self.ctx, C.c_int64(S), C.c_int64(ndraws), C.c_int32(int(importance)),
                                           C.c_int32(int(replace)), C.c_uint64(int(seed)), _d(uniforms),
                                           idx.ctypes.data_as(_i64p)))
        return idx

    def resample_indices_direct(self, S, uniforms):
        """StatsBase.direct_sample! on the current PSIS weights with host-drawn uniforms (0-based indices)."""
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        idx = np.empty(len(uniforms), dtype=np.int64)
        check(self.L.pfmi_resample_indices_direct(self.ctx, C.c_int64(S), C.c_int64(len(uniforms)), _d(uniforms),
                                                  idx.ctypes.data_as(_i64p)))
        return idx

    def pool_gather(self, idx, col_offset=0):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = result_empty((self.d, len(idx)))
        check(self.L.pfmi_pool_gather(self.ctx, C.c_int64(len(idx)), idx.ctypes.data_as(_i64p),
                                      C.c_int64(col_offset), _d(out)))
        return out

    # ---- raw device buffers (hosts that keep results on the GPU) ----------------------------------------------
    def malloc_dev(self, nbytes):
        p = C.c_void_p()
        check(self.L.pfmi_malloc_dev(self.ctx, C.c_int64(nbytes), C.byref(p)))
        return p.value

    def free_dev(self, ptr):
        check(self.L.pfmi_free_dev(self.ctx, C.c_void_p(ptr)))

    def memcpy_d2h(self, host_array, dev_ptr):
        check(self.L.pfmi_memcpy_d2h(self.ctx, host_array.ctypes.data_as(C.c_void_p), C.c_void_p(dev_ptr), C.c_int64(host_array.nbytes)))
        return host_array

    def memcpy_h2d(self, dev_ptr, host_array):
        host_array = np.ascontiguousarray(host_array)
        check(self.L.pfmi_memcpy_h2d(self.ctx, C.c_void_p(dev_ptr), host_array.ctypes.data_as(C.c_void_p), C.c_int64(host_array.nbytes)))

    def pool_gather_dev(self, idx, col_offset, dev_ptr):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        check(self.L.pfmi_pool_gather_dev(self.ctx, C.c_int64(len(idx)), idx.ctypes.data_as(_i64p),
                                          C.c_int64(col_offset), C.c_void_p(dev_ptr)))


class Comm:
    """pfmi_comm: the RCCL group behind the C ABI (csrc/comm_rccl.hip).  Comm.init_all(engines) -- one process, one Engine per
    GPU; Comm.init_rank(engine, world, rank, uid) -- one process per GPU, `uid` = Comm.unique_id() of rank 0 shipped by the
    launcher."""

    def __init__(self, handle, engines):
        self.h, self.engines, self.L = handle, list(engines), _lib.lib()
        _LIVE_COMMS.add(self)

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(_lib.lib().pfmi_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def init_all(cls, engines):
        arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        h = C.c_void_p()
        check(_lib.lib().pfmi_comm_init_all(C.c_int32(len(engines)), arr, C.byref(h)))
        return cls(h, engines)

    @classmethod
    def init_rank(cls, engine, world, rank, uid):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        h = C.c_void_p()
        check(_lib.lib().pfmi_comm_init_rank(engine.ctx, C.c_int32(world), C.c_int32(rank), buf, C.byref(h)))
        return cls(h, [engine])

    def info(self):
        w, n, v = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.pfmi_comm_info(self.h, C.byref(w), C.byref(n), C.byref(v)))
        return dict(world=w.value, nlocal=n.value, rccl_version=v.value)

    def pool_psis(self):
        k, M = C.c_double(), C.c_int64()
        check(self.L.pfmi_comm_pool_psis(self.h, C.byref(k), C.byref(M)))
        return dict(pareto_shape=k.value, tail_length=M.value)

    def resample(self, ndraws, importance=True, replace=True, seed=0, uniforms=None, want_draws=True):
        d = self.engines[0].d
        idx = np.empty(ndraws, dtype=np.int64)
        out = result_empty((d, ndraws)) if want_draws else None
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        check(self.L.pfmi_comm_resample(self.h, C.c_int64(ndraws), C.c_int32(int(importance)), C.c_int32(int(replace)),
                                        C.c_uint64(int(seed)), _d(uniforms), idx.ctypes.data_as(_i64p), _d(out)))
        return idx, out

    def psis_resample(self, ndraws, importance=True, replace=True, seed=0, uniforms=None, want_draws=True):
        """pooled PSIS + index selection + owner gather + all-reduce, enqueued on every local context, ONE synchronisation"""
        d = self.engines[0].d
        idx = np.empty(ndraws, dtype=np.int64)
        out = result_empty((d, ndraws)) if want_draws else None
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        k, M = C.c_double(), C.c_int64()
        check(self.L.pfmi_comm_psis_resample(self.h, C.c_int64(ndraws), C.c_int32(int(importance)), C.c_int32(int(replace)),
                                             C.c_uint64(int(seed)), _d(uniforms), C.byref(k), C.byref(M),
                                             idx.ctypes.data_as(_i64p), _d(out)))
        return dict(pareto_shape=k.value, tail_length=M.value), idx, out

    def psis_resample_enqueue(self, ndraws, importance=True, replace=True, seed=0, uniforms=None):
        """first half of psis_resample: everything launched on every local context, no wait"""
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        self._pr = (ndraws, uniforms)                   # (keeps the uniforms alive until the wait)
        check(self.L.pfmi_comm_psis_resample_enqueue(self.h, C.c_int64(ndraws), C.c_int32(int(importance)), C.c_int32(int(replace)),
                                                     C.c_uint64(int(seed)), _d(uniforms)))

    def psis_resample_wait(self, want_draws=True):
        """second half: the ONE host round trip (also delivers what the member engines queued under Engine.deferred())"""
        ndraws = self._pr[0]
        d = self.engines[0].d
        idx = np.empty(ndraws, dtype=np.int64)
        out = result_empty((d, ndraws)) if want_draws else None
        k, M = C.c_double(), C.c_int64()
        check(self.L.pfmi_comm_psis_resample_wait(self.h, C.byref(k), C.byref(M), idx.ctypes.data_as(_i64p), _d(out)))
        self._pr = None
        return dict(pareto_shape=k.value, tail_length=M.value), idx, out

    def close(self):
        if self.h:
            self.L.pfmi_comm_destroy(self.h)
            self.h = None
