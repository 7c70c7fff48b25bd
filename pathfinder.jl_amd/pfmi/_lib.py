"""ctypes binding of libpfmi.so (include/pfmi.h).  No CPU fallback: if the HIP library is missing
or no MI355X is visible, every entry point raises."""
import ctypes as C
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)                      # pathfinder.jl_amd/
_SO = os.environ.get("PFMI_LIB_PATH") or os.path.join(_ROOT, "lib", "libpfmi.so")   # override: ablation / experiment builds
_lib = None


class PfmiError(RuntimeError):
    pass


class PfmiRetry(PfmiError):
    """PFMI_ERR_RETRY (-7): a transient condition voided the enqueued step (the GPU is shared with another process / a profiler serialises
    dispatch and an in-kernel hand-over of the scan timed out).  The library has switched the context to the route without in-kernel
    waits; enqueue the step again."""


class pfmi_target(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("r", C.c_int32), ("reserved", C.c_int32),
                ("mean", C.POINTER(C.c_double)), ("a", C.POINTER(C.c_double)),
                ("Wd", C.POINTER(C.c_double)), ("G", C.POINTER(C.c_double)), ("offset", C.c_double),
                ("fn", C.c_void_p), ("user", C.c_void_p), ("dev_fn", C.c_void_p)]


LOGP_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int32, C.c_int64, C.POINTER(C.c_double), C.c_void_p)
# device callback: (X_dev, d, n, out_dev, stream, user) -- raw device addresses, see include/pfmi.h
LOGP_DEV_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p)

# every symbol include/pfmi.h declares (tests/test_abi.py checks the built library exports them all)
SYMBOLS = [
    "pfmi_last_error", "pfmi_version", "pfmi_device_count", "pfmi_create", "pfmi_destroy", "pfmi_sync",
    "pfmi_timer_start", "pfmi_timer_stop", "pfmi_profile", "pfmi_kernel_time", "pfmi_set_target",
    "pfmi_set_traces", "pfmi_optimize_batch", "pfmi_get_trace", "pfmi_fit_batch", "pfmi_get_fit_status", "pfmi_get_fit", "pfmi_elbo_batch",
    "pfmi_get_elbo_logs", "pfmi_callback_stats", "pfmi_draws", "pfmi_logpdf", "pfmi_woodbury_apply", "pfmi_woodbury_diag", "pfmi_pool_build", "pfmi_pool_get",
    "pfmi_pool_log_ratios_dev", "pfmi_psis_dev", "pfmi_psis", "pfmi_resample_indices", "pfmi_resample_indices_direct", "pfmi_pool_gather",
    "pfmi_pool_gather_dev", "pfmi_malloc_dev", "pfmi_free_dev", "pfmi_memcpy_h2d", "pfmi_memcpy_d2h",
    "pfmi_comm_unique_id", "pfmi_comm_init_all", "pfmi_comm_init_rank", "pfmi_comm_destroy", "pfmi_comm_info",
    "pfmi_comm_pool_psis", "pfmi_comm_resample", "pfmi_host_rand_u64", "pfmi_host_rand_u64_multi",
    "pfmi_optimize_batch_enqueue", "pfmi_optimize_batch_wait", "pfmi_elbo_batch_enqueue", "pfmi_elbo_batch_wait",
    "pfmi_callback_stats_dev", "pfmi_pool_build_best", "pfmi_pool_winners", "pfmi_psis_weights", "pfmi_comm_psis_resample", "pfmi_debug_set",
    "pfmi_host_alloc", "pfmi_host_free", "pfmi_comm_psis_resample_enqueue", "pfmi_comm_psis_resample_wait", "pfmi_defer_downloads",
    "pfmi_stream_enqueue", "pfmi_stream_seeds", "pfmi_stream_pump", "pfmi_stream_wait", "pfmi_stream_cancel",
    "pfmi_fit_batch_ex", "pfmi_set_hinit", "pfmi_set_callback_threads",
]


def build(force=False):
    """Compile libpfmi.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(_ROOT, "csrc", f) for f in os.listdir(os.path.join(_ROOT, "csrc"))]
    srcs.append(os.path.join(os.path.dirname(_ROOT), "include", "pfmi.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            raise PfmiError("libpfmi.so is missing/stale and hipcc is not available to build it")
        subprocess.check_call(["make", "-C", _ROOT, "-j8"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _SO


def lib():
    """Load libpfmi.so.  torch (when installed) is imported FIRST so that the process ends up with a
    single HIP runtime (torch bundles its own libamdhip64 with the same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    # the streaming pipeline uses four streams beside the context's own: give the runtime enough hardware queues that they (and torch's /
    # RCCL's) do not share one (only effective when HIP has not been initialised yet)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if not os.environ.get("PFMI_NO_TORCH") and "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
    if not os.path.exists(_SO):
        build()   # in-tree hipcc build; raises if hipcc is unavailable (there is no CPU fallback)
    L = C.CDLL(_SO, mode=C.RTLD_GLOBAL)
    L.pfmi_last_error.restype = C.c_char_p
    for s in SYMBOLS:
        if s != "pfmi_last_error":
            getattr(L, s).restype = C.c_int32
    _lib = L
    return L


def check(rc):
    if rc == -7:
        raise PfmiRetry(f"libpfmi: {lib().pfmi_last_error().decode()}")
    if rc != 0:
        ex = PfmiError(f"libpfmi error {rc}: {lib().pfmi_last_error().decode()}")
        ex.code = rc
        raise ex
