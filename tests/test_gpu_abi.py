"""The boundary itself on the GPU: plain-C callers built and run (examples/c_abi_demo.c, examples/julia_sequence.c), error behaviour of the C ABI,
stale result handles, the debug-hook table, page-locked result buffers."""
from concurrent.futures import ThreadPoolExecutor
import ctypes as C
import json
import os
import warnings

import numpy as np
import pytest

from helpers import fit_seeds, make_traces, oracle_factor_from_gpu, oracle_target
from oracle import pf_oracle as po
import margins as mg

pytestmark = pytest.mark.gpu


def test_c_abi_demo_program(tmp_path):
    """examples/c_abi_demo.c: the whole hot path driven from plain C through include/pfmi.h (what a Julia ccall / cgo / JNI
    binding does) -- compiled here with gcc against the in-tree libpfmi.so, no Python or torch in the process."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.join(root, "pathfinder.jl_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_demo.c"), "-o", exe,
                           "-L", libdir, "-lpfmi", f"-Wl,-rpath,{libdir}", "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1].startswith("OK ")


def test_c_abi_error_behaviour(pfmi_mod):
    """the boundary never aborts: wrong call order / bad arguments come back as negative return codes with a message
    (SURVEY.md 8b "Errors"), and the context stays usable afterwards."""
    import ctypes as C
    from pfmi import _lib
    L = _lib.lib()
    e = pfmi_mod.Engine(0)
    assert L.pfmi_fit_batch(e.ctx, C.c_int32(6), C.c_double(1e-12)) == -3                      # PFMI_ERR_STATE: no traces yet
    assert b"no traces" in L.pfmi_last_error()
    tg = pfmi_mod.t_iso(8)
    e.set_target(tg)
    x0 = np.ones((2, 8))
    e.optimize_batch(x0, 6)
    elbo = np.empty(e.P); se = np.empty(e.P); best = np.empty(2, dtype=np.int64)
    seeds = fit_seeds(e.P, 1)
    rc = L.pfmi_elbo_batch(e.ctx, C.c_int64(10), seeds.ctypes.data_as(C.POINTER(C.c_uint64)), None,
                           elbo.ctypes.data_as(C.POINTER(C.c_double)), se.ctypes.data_as(C.POINTER(C.c_double)),
                           best.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rc == -3                                                                          # fit_batch not called yet
    assert L.pfmi_fit_batch(e.ctx, C.c_int32(0), C.c_double(1e-12)) == -1                      # PFMI_ERR_ARG
    assert L.pfmi_fit_batch(e.ctx, C.c_int32(40), C.c_double(1e-12)) == -4                     # PFMI_ERR_UNSUPPORTED (J > 32)
    e.fit_batch(6)
    X = np.zeros((8, 3), order="F"); out = np.zeros((8, 3), order="F")
    dp = C.POINTER(C.c_double)
    assert L.pfmi_woodbury_apply(e.ctx, C.c_int64(0), C.c_int32(99), C.c_int64(3), X.ctypes.data_as(dp), out.ctypes.data_as(dp)) == -1
    assert L.pfmi_woodbury_apply(e.ctx, C.c_int64(10**6), C.c_int32(0), C.c_int64(3), X.ctypes.data_as(dp), out.ctypes.data_as(dp)) == -1
    assert L.pfmi_create(C.c_int32(99), C.byref(C.c_void_p())) == -1                           # no such device
    assert L.pfmi_fit_batch(None, C.c_int32(6), C.c_double(1e-12)) == -1                       # null context
    el, _, b = e.elbo_batch(32, seeds)                                                       # still usable
    assert np.isfinite(el[1]) and b[0] >= 1
    e.close()


def test_julia_call_sequence_in_c(tmp_path):
    """examples/julia_sequence.c replays, call for call, what pathfinder.jl_amd/julia/PathfinderMI355X.jl does for
    multipathfinder / resample / Comm (callback target through the trampoline, batched fit + ELBO, lazy materialisation with
    pfmi_get_fit / pfmi_draws, pooled PSIS, both index modes, fresh candidates, the RCCL group, the batched retry) and checks
    every result the Julia side relies on -- the executed stand-in for the wrapper (no Julia toolchain exists here)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "julia_sequence")
    libdir = os.path.join(root, "pathfinder.jl_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "julia_sequence.c"), "-o", exe,
                           "-L", libdir, "-lpfmi", f"-Wl,-rpath,{libdir}", "-lm", "-ldl"])
    from helpers import DEMO_LIB, STANDIN_LIB
    env = dict(os.environ, PFMI_DEMO_LIB=DEMO_LIB)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1].startswith("OK julia_sequence") and "device closure ok" in r.stdout
    assert "round-5 streaming sequence ok: G=1 engines" in r.stdout
    # round 3: multipathfinder(engines::Vector{Engine}, ...) -- the same program with the runs sharded over G engines (all on GPU 0,
    # the in-process RCCL stand-in of tests/rccl_standin) must reproduce the single-engine result bit for bit
    for G in (2, 4):
        envg = dict(env, PFMI_RCCL_LIB=STANDIN_LIB, PFMI_COMM_ALLOW_SHARED_GPU="1")
        r = subprocess.run([exe, str(G)], capture_output=True, text=True, timeout=300, env=envg)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert f"round-3 sequence ok: G={G} engines, rccl_version=99999" in r.stdout
        assert f"round-5 streaming sequence ok: G={G} engines" in r.stdout


def test_stale_result_handles_raise(pfmi_mod):
    """lazy handles index the engine's CURRENT buffers; after the engine is reused they must fail loudly."""
    tg = pfmi_mod.t_diag(10, 1)
    e = pfmi_mod.Engine(0)
    r1 = pfmi_mod.multipathfinder(tg, 100, nruns=3, ndraws_elbo=40, rng=pfmi_mod.HostRNG(1), engine=e)
    d0 = r1.pathfinder_results[0].draws.copy()                  # materialised: stays valid
    mu1 = r1.pathfinder_results[1].fit_distribution.mu.copy()
    r2 = pfmi_mod.multipathfinder(tg, 100, nruns=3, ndraws_elbo=40, rng=pfmi_mod.HostRNG(2), engine=e)
    np.testing.assert_array_equal(r1.pathfinder_results[0].draws, d0)
    np.testing.assert_array_equal(r1.pathfinder_results[1].fit_distribution.mu, mu1)
    with pytest.raises(pfmi_mod.StaleHandleError):
        r1.pathfinder_results[2].draws
    with pytest.raises(pfmi_mod.StaleHandleError):
        r1.pathfinder_results[2].fit_distribution.mu
    with pytest.raises(pfmi_mod.StaleHandleError):
        r1.pathfinder_results[0].elbo_estimates[0].draws
    with pytest.raises(pfmi_mod.StaleHandleError):
        r1.pathfinder_results[0].optim_trace.points
    with pytest.raises(pfmi_mod.StaleHandleError):
        pfmi_mod.resample(r1, 10)
    assert r2.pathfinder_results[2].draws.shape == (10, 40)     # the current result's handles work
    e.close()


def test_debug_hook_table(pfmi_mod):
    """pfmi_debug_set: the explicit form of the PFMI_* test hooks (the environment is honoured only under PFMI_DEBUG_HOOKS=1)."""
    L = pfmi_mod.lib()
    assert L.pfmi_debug_set(b"NOT_A_HOOK", b"1") == -1
    d, J = 200, 6
    tg = pfmi_mod.t_diag(d, seed=1)
    eng = pfmi_mod.Engine(0)
    try:
        eng.set_target(tg)
        x0 = pfmi_mod.HostRNG(3).rand(2 * d).reshape(2, d) * 4 - 2
        eng.optimize_batch(x0, J, 30)
        eng.fit_batch(J)
        seeds = fit_seeds(eng.P, 2)
        e0 = eng.elbo_batch(256, seeds)[0]
        assert L.pfmi_debug_set(b"PFMI_ELBO_KERNEL", b"lane") == 0     # another kernel, the same numbers to roundoff
        eng.profile(2)
        e1 = eng.elbo_batch(256, seeds)[0]
        assert L.pfmi_debug_set(b"PFMI_ELBO_KERNEL", None) == 0
        e2 = eng.elbo_batch(256, seeds)[0]
        eng.profile(0)
        f = np.isfinite(e0)
        np.testing.assert_array_equal(e0[f], e2[f])
        assert np.max(np.abs(e1[f] - e0[f]) / (1 + np.abs(e0[f]))) <= 1e-10 and not np.array_equal(e1[f], e0[f])
    finally:
        eng.close()


# ---- large results in page-locked memory (include/pfmi.h: pfmi_host_alloc) ------------------------------------------------------------
def test_large_results_arrive_in_page_locked_memory(pfmi_mod, monkeypatch):
    """The draws / pool arrays that the Python host hands to the library are page-locked above 16 MB (one DMA transfer instead of the
    runtime's staged copy): the same bytes as into ordinary memory, the block is recycled when its arrays die."""
    import gc
    from pfmi import core
    d, K, J, N_r = 600, 3, 5, 2000                                            # pool: 600 x 2000 x 3 doubles = 28.8 MB
    tg = pfmi_mod.t_lowrank(d, r=8, seed=4)
    eng = pfmi_mod.Engine(0)
    try:
        eng.set_target(tg)
        x0 = pfmi_mod.HostRNG(5).rand(K * d).reshape(K, d) * 4 - 2
        eng.optimize_batch(x0, J)
        eng.fit_batch(J)
        eng.elbo_batch(64, fit_seeds(eng.P, 3))
        pts = np.array([int(eng.offsets[k]) + 2 for k in range(K)], dtype=np.int64)
        eng.pool_build(N_r, pts, fit_seeds(K, 9))
        monkeypatch.setattr(core, "_PIN_MIN_BYTES", 1 << 62)
        X_plain, lr_plain = eng.pool_get()
        monkeypatch.setattr(core, "_PIN_MIN_BYTES", 16 << 20)
        core._pin_free.clear()
        X_pin, lr_pin = eng.pool_get()
        assert X_pin.flags["F_CONTIGUOUS"] and X_pin.shape == X_plain.shape
        np.testing.assert_array_equal(X_pin, X_plain)
        np.testing.assert_array_equal(lr_pin, lr_plain)
        # the block returns to the pool with its last view and is handed out again
        view = X_pin[:, :10, 0]
        del X_pin
        gc.collect()
        assert sum(len(v) for v in core._pin_free.values()) == 0
        del view
        gc.collect()
        assert sum(len(v) for v in core._pin_free.values()) == 1
        addr = next(v[0] for v in core._pin_free.values() if v)
        Y = core.result_empty((d, N_r, K))
        assert Y.ctypes.data == addr and sum(len(v) for v in core._pin_free.values()) == 0
    finally:
        eng.close()
