"""GPU: host closures evaluated by several host threads -- the reference's `ntasks` (src/elbo.jl:3-6, src/resample.jl:85-92 through
src/utils.jl:33-49; "the log-density function must be thread-safe", src/multipath.jl:104-108) as pfmi_set_callback_threads.  The
reference's own tests of it are invariance tests (test/singlepath.jl:173-203, test/multipath.jl:107-140: same result for ntasks = 1
and ntasks > 1); so are these, bit for bit."""
import threading

import numpy as np
import pytest

from helpers import demo_host_target, fit_seeds, make_traces

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pfmi_mod():
    import pfmi
    return pfmi


def _job(pfmi, eng, target, traces, J, N, seeds, nthreads):
    eng.set_target(target)
    eng.set_callback_threads(nthreads)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    eng.fit_batch(J)
    elbo, se, best = eng.elbo_batch(N, seeds)
    pts = [int(eng.offsets[k]) + max(int(best[k]), 1) for k in range(len(traces))]
    eng.pool_build(N + 37, pts, seeds[pts])                      # (pool / top-up draws go through the other callback route)
    pool, lr = eng.pool_get()
    X, lp, lq = eng.draws(pts[0], 99, 53, n0=11)
    return dict(elbo=elbo, se=se, best=best, lr=lr, lp=lp, stats=eng.callback_stats())


@pytest.mark.parametrize("d,r,N", [(40, 3, 96), (1000, 8, 256)])
def test_compiled_host_closure_is_invariant_under_callback_threads_and_equals_the_builtin_target(pfmi_mod, d, r, N):
    pfmi = pfmi_mod
    tg = pfmi.t_lowrank(d, r=r, seed=2)
    K, J = 3, 6
    traces = make_traces(tg, K, 7, history_length=J, maxiters=40)
    eng = pfmi.Engine(0)
    seeds = fit_seeds(sum(len(t) for t in traces), 5)
    ref = _job(pfmi, eng, tg, traces, J, N, seeds, 1)            # the same target as a built-in (no callback at all)
    ctg = demo_host_target(tg)
    outs = {n: _job(pfmi, eng, ctg, traces, J, N, seeds, n) for n in (1, 4, 16, 1000)}
    for n, o in outs.items():
        for key in ("elbo", "se", "lr", "lp"):
            np.testing.assert_array_equal(o[key], outs[1][key], err_msg=f"{key} differs between 1 and {n} callback threads")
        np.testing.assert_array_equal(o["best"], outs[1]["best"])
        assert o["stats"]["bytes_to_host"] > 0
    fin = np.isfinite(ref["elbo"])
    assert np.array_equal(fin, np.isfinite(outs[1]["elbo"]))
    # the C closure sums in another order than the device target: equal to rounding, same winners
    assert np.max(np.abs(outs[1]["elbo"][fin] - ref["elbo"][fin]) / (1 + np.abs(ref["elbo"][fin]))) < 1e-11
    np.testing.assert_array_equal(outs[1]["best"], ref["best"])
    np.testing.assert_allclose(outs[1]["lr"], ref["lr"], rtol=0, atol=1e-9 * (1 + np.abs(ref["lr"]).max()))
    with pytest.raises(pfmi.PfmiError):
        eng.L.pfmi_set_callback_threads.restype = __import__("ctypes").c_int32
        from pfmi._lib import check
        check(eng.L.pfmi_set_callback_threads(eng.ctx, 0))
    eng.close()


def test_python_closure_called_from_library_threads(pfmi_mod):
    """a Python closure under ntasks > 1: ctypes takes the GIL for every call the library's threads make, the results are those of one
    thread, the calls really come from several threads, and an exception raised in one of them is re-raised by the engine call"""
    pfmi = pfmi_mod
    d = 24
    tg = pfmi.t_diag(d, seed=1)
    seen = set()

    def batch(X):
        seen.add(threading.get_ident())
        return tg.logp(X)

    cb = pfmi.CallbackTarget(d, lambda x: float(tg.logp(x)), grad=tg.grad, logp_batch=batch)
    r1 = pfmi.multipathfinder(cb, 50, nruns=3, ndraws_elbo=64, rng=pfmi.HostRNG(3), maxiters=60, ntasks=1)
    n1 = len(seen)
    r4 = pfmi.multipathfinder(cb, 50, nruns=3, ndraws_elbo=64, rng=pfmi.HostRNG(3), maxiters=60, ntasks=2, ntasks_per_run=2)
    np.testing.assert_array_equal(r4.draws, r1.draws)            # test/multipath.jl:107-140: the result does not depend on ntasks
    np.testing.assert_array_equal(r4.draw_component_ids, r1.draw_component_ids)
    assert r4.psis_result.pareto_shape == r1.psis_result.pareto_shape
    assert n1 == 1 and len(seen) > 1, (n1, len(seen))

    def boom(X):
        if threading.get_ident() != main:
            raise RuntimeError("boom in a library thread")
        return tg.logp(X)

    main = threading.get_ident()
    cb2 = pfmi.CallbackTarget(d, lambda x: float(tg.logp(x)), grad=tg.grad, logp_batch=boom)
    with pytest.raises(RuntimeError, match="boom in a library thread"):
        pfmi.multipathfinder(cb2, 50, nruns=3, ndraws_elbo=64, rng=pfmi.HostRNG(3), maxiters=60, ntasks=4)
