"""GPU tests added in round 5:

* the in-kernel hand-over of the scan under contention (a second context + saturating GEMMs on the same GPU) and under an attached
  profiler: bit-identical to the wait-free two-launch cut, give-up counter 0 (tests/probes/handover_stress.py);
* a hand-over time-out is a RETRYABLE condition: the library discards the step, switches the context to the wait-free cut and the public
  calls re-enqueue with the same seeds -- same answer;
* a run that fails its first try is retried BEFORE the pooled stage's result counts (src/singlepath.jl:259-283): an all-NaN first try no
  longer aborts multipathfinder.
All calls go through the C ABI of libpfmi.so.
"""
import os
import shutil
import subprocess
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "probes", "handover_stress.py")


@pytest.fixture(scope="module")
def pfmi_mod():
    import pfmi
    return pfmi


def _run(cmd, timeout):
    env = dict(os.environ, PFMI_DEBUG_HOOKS="1", TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    return r.returncode, (r.stdout + r.stderr)[-3000:]


@pytest.mark.timeout(900)
def test_scan_handover_under_contention_is_bit_identical_and_never_gives_up():
    rc, out = _run([sys.executable, PROBE, "200", "--contend"], 800)
    assert rc == 0, out
    assert "give-up counter 0" in out and " 0 differ" in out, out


@pytest.mark.timeout(900)
def test_scan_handover_under_rocprofv3_kernel_trace():
    exe = shutil.which("rocprofv3")
    if not exe:
        pytest.skip("rocprofv3 not on PATH")
    rc, out = _run([exe, "--kernel-trace", "-d", "/tmp/pfmi_stress_prof", "-o", "s", "--", sys.executable, PROBE, "60", "--contend"], 800)
    shutil.rmtree("/tmp/pfmi_stress_prof", ignore_errors=True)
    assert rc == 0, out
    assert "give-up counter 0" in out and " 0 differ" in out, out


def test_handover_timeout_is_retryable_and_switches_to_the_wait_free_cut(pfmi_mod):
    """PFMI_QF_FAKE_LOST=1: the first wait of a context reports one lost piece.  pfmi_elbo_batch_wait returns PFMI_ERR_RETRY, the context
    takes the two-launch cut from then on, and multipathfinder / pathfinder re-enqueue with the same seeds: the same result as an
    undisturbed call."""
    L = pfmi_mod.lib()
    tg = pfmi_mod.t_lowrank(300, r=8, seed=2)
    kw = dict(nruns=6, ndraws_elbo=1000, history_length=6, maxiters=60)
    e0 = pfmi_mod.Engine(0)
    ref = pfmi_mod.multipathfinder(tg, 200, rng=pfmi_mod.HostRNG(5), engine=e0, **kw)
    ref1 = pfmi_mod.pathfinder(tg, ndraws=50, ndraws_elbo=1000, rng=pfmi_mod.HostRNG(6), engine=e0, maxiters=60)
    e0.close()
    assert L.pfmi_debug_set(b"PFMI_QF_FAKE_LOST", b"1") == 0
    try:
        e1 = pfmi_mod.Engine(0)
        e1.set_target(tg)
        x0 = pfmi_mod.HostRNG(3).rand(4 * 300).reshape(4, 300) * 4 - 2
        npts = e1.optimize_batch(x0, 6, 60)
        e1.fit_batch(6)
        sd = np.arange(e1.P, dtype=np.uint64) + np.uint64(7)
        with pytest.raises(pfmi_mod._lib.PfmiRetry):
            e1.elbo_batch(1000, sd)
        assert e1.kernel_time("qf_handover_lost")[1] == 1
        a = e1.elbo_batch(1000, sd)                                 # second attempt: the wait-free cut, no error
        e1.close()
        e2 = pfmi_mod.Engine(0)                                     # a fresh context trips once inside the public call ...
        got = pfmi_mod.multipathfinder(tg, 200, rng=pfmi_mod.HostRNG(5), engine=e2, **kw)
        assert e2.kernel_time("qf_handover_lost")[1] == 1           # ... which retried
        e2.close()
        e3 = pfmi_mod.Engine(0)
        got1 = pfmi_mod.pathfinder(tg, ndraws=50, ndraws_elbo=1000, rng=pfmi_mod.HostRNG(6), engine=e3, maxiters=60)
        assert e3.kernel_time("qf_handover_lost")[1] == 1
        e3.close()
    finally:
        assert L.pfmi_debug_set(b"PFMI_QF_FAKE_LOST", None) == 0
    np.testing.assert_array_equal(got.draws, ref.draws)
    np.testing.assert_array_equal(got.draw_component_ids, ref.draw_component_ids)
    assert got.psis_result.pareto_shape == ref.psis_result.pareto_shape
    np.testing.assert_array_equal(got1.draws, ref1.draws)
    assert got1.fit_iteration == ref1.fit_iteration
    e4 = pfmi_mod.Engine(0)                                         # and the undisturbed scan of the same fits equals the wait-free cut's
    e4.set_target(tg)
    e4.optimize_batch(x0, 6, 60)
    e4.fit_batch(6)
    b = e4.elbo_batch(1000, sd)
    e4.close()
    np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[2], b[2])
    assert int(npts.sum()) > 4


def test_all_runs_failing_their_first_try_are_retried_before_the_pooled_stage_counts(pfmi_mod):
    """every run's first try ends with NaN ELBOs (the closure returns NaN until the retry's init_sampler flips it), so the
    optimistic pooled stage sees only NaN log ratios and fails ("weights are all zero").  The reference retries each run up to `ntries`
    before it ever pools (src/singlepath.jl:259-283, src/multipath.jl:190-225): the call must succeed with num_tries == 2."""
    d = 8
    base = pfmi_mod.t_diag(d, seed=5)
    state = {"nan": True}

    def logp_batch(X):
        out = np.asarray(base.logp(X), dtype=np.float64)
        return np.full_like(out, np.nan) if state["nan"] else out

    class Sampler:                                                  # init_sampler(rng, point) is only called for tries >= 2
        def __call__(self, rng, point):
            state["nan"] = False
            point[:] = rng.rand(len(point)) * 4 - 2
            return point

    tgt = pfmi_mod.CallbackTarget(d, lambda x: float(base.logp(x)), grad=lambda x: base.grad(x), logp_batch=logp_batch)
    inits = [np.full(d, 0.5), np.full(d, -0.5), np.linspace(-1, 1, d)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = pfmi_mod.multipathfinder(tgt, 60, init=inits, ndraws_elbo=40, ntries=3, init_sampler=Sampler(), rng=pfmi_mod.HostRNG(2))
    assert [r.num_tries for r in res.pathfinder_results] == [2, 2, 2]
    assert all(r.success for r in res.pathfinder_results)
    assert res.draws.shape == (d, 60) and np.all(np.isfinite(res.draws))
    assert np.isfinite(res.psis_result.pareto_shape) or res.psis_result.pareto_shape == np.inf
    # with ntries = 1 the same situation is the reference's failure path: warnings, draws from fit_distributions[1], and -- all log ratios
    # NaN -- the pooled stage's error surfaces
    state["nan"] = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(Exception):
            pfmi_mod.multipathfinder(tgt, 60, init=inits, ndraws_elbo=40, ntries=1, rng=pfmi_mod.HostRNG(2))


# ---- history_length 17 .. 32 (VERDICT r4 missing #5 / next #8): the reference's keyword is unbounded (src/inverse_hessian.jl:25) -------------
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("d,J,K,maxit,N", [(50, 20, 3, 60, 256), (50, 32, 3, 80, 256), (3000, 20, 2, 45, 200), (3000, 32, 2, 50, 200)])
def test_history_length_17_to_32_against_the_oracle(pfmi_mod, d, J, K, maxit, N):
    """Column padding 64 (2 J <= 64): the memory-resident fit kernel with its small matrices in global memory and the lane-per-draw
    kernel -- slow but correct.  Walk (status, j_eff, rejections), factor (dense W, logdet, mu), ELBO / SE, per-draw logq / logp / x on the
    GPU's own factor, pool + PSIS + indices: all against the oracle within SURVEY 8(d)."""
    import margins as mg
    from helpers import fit_seeds, oracle_factor_from_gpu, oracle_target
    from oracle import pf_oracle as po
    from test_gpu_parity_r2 import _factor, _wc
    tg = pfmi_mod.t_diag(d, seed=1)
    otg = oracle_target(tg)
    rng = pfmi_mod.HostRNG(17)
    traces = [pfmi_mod.optimize_with_trace(tg, rng.rand(d) * 4 - 2, history_length=J, maxiters=maxit) for _ in range(K)]
    eng = pfmi_mod.Engine(0)
    eng.set_target(tg)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    eng.fit_batch(J)
    status, jeff, logdet, nrej = eng.fit_status()
    seeds = fit_seeds(eng.P, 4)
    elbo, se, best = eng.elbo_batch(N, seeds)
    th = np.concatenate([t.points for t in traces]); gr = np.concatenate([t.gradients for t in traces])
    ref = po.multipath_fit_elbo(eng.offsets, th, gr, J, otg, N, seeds, nthreads=K)
    np.testing.assert_array_equal(status, ref["status"])
    np.testing.assert_array_equal(jeff, ref["j_eff"])
    np.testing.assert_array_equal(nrej, ref["n_rejected"])
    assert int(jeff.max()) > 16, "the traces must fill more than 16 history pairs"
    cfg = f"J{J}:diag{d}"
    n_strict = 0
    for k in range(K):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        alpha_all, hl, hs, _ = po.lbfgs_history(th[p0:p1], gr[p0:p1], J)
        for l in sorted({1, min(18, p1 - p0 - 1), (p1 - p0) // 2, p1 - p0 - 1}):
            if ref["status"][p0 + l] != 0:
                assert np.isnan(elbo[p0 + l])
                continue
            F = _factor(th[p0:p1], gr[p0:p1], alpha_all, hl, hs, l, d)
            f = eng.get_fit(p0 + l, int(jeff[p0 + l]))
            mg.check(cfg, "logdet", mg.rel(logdet[p0 + l], ref["logdet"][p0 + l]))
            mg.check(cfg, "mu", np.max(np.abs(f["mu"] - F.fit_mean(th[p0 + l], gr[p0 + l]))) / (1 + np.abs(f["mu"]).max()))
            if d <= 200:
                Wg = np.diag(f["alpha"]) + f["B"] @ f["D"] @ f["B"].T
                Wo = np.diag(F.alpha) + F.B @ F.D @ F.B.T
                mg.check(cfg, "W", np.max(np.abs(Wg - Wo)) / np.max(np.abs(Wo)))
            if _wc(F):
                n_strict += 1
                mg.check(cfg, "elbo", mg.rel(elbo[p0 + l], ref["elbo"][p0 + l]))
                mg.check(cfg, "se", mg.rel(se[p0 + l], ref["se"][p0 + l]))
            # per-draw quantities through the oracle's reflector-by-reflector apply on the GPU's OWN factor
            Fg = oracle_factor_from_gpu(f)
            Xg, lqg = Fg.rand_and_logpdf(f["mu"], po.randn_fill(int(seeds[p0 + l]), d, N))
            lp, lq = eng.elbo_logs(p0 + l, N)
            mg.check(cfg, "logq@scan_vs_oracle_on_gpu_factor", mg.rel(lq, lqg))
            mg.check(cfg, "logp@scan_vs_oracle_on_gpu_factor", mg.rel(lp, otg.logp(Xg)))
            Xd, lpd, lqd = eng.draws(p0 + l, int(seeds[p0 + l]), 24, n0=3)
            mg.check(cfg, "draws@writer_vs_oracle_on_gpu_factor", np.abs(Xd - Xg[:, 3:27]) / (1 + np.abs(Xg[:, 3:27]).max(axis=0)))
            np.testing.assert_array_equal(lqd, lq[3:27])
            # logpdf of arbitrary points through the factor (src/resample.jl:85-89)
            mg.check(cfg, "logq@logpdf", mg.rel(eng.logpdf(p0 + l, Xd), lqg[3:27]))
    assert n_strict >= K, n_strict
    np.testing.assert_array_equal(best, ref["best_iter"])
    # pooled stage on the winners
    pts = [int(eng.offsets[k]) + int(best[k]) for k in range(K)]
    eng.pool_build(N, pts, seeds[pts])
    _, lr = eng.pool_get(draws=False)
    res = eng.psis(lr)
    lw, w, kk, M = po.psis(lr)
    mg.check(cfg, "psis_logw", np.abs(res["log_weights"] - lw) / (1 + np.abs(lw)))
    idx = eng.resample_indices(len(lr), 50, seed=3)
    np.testing.assert_array_equal(idx, po.sample_weighted(res["weights"], 50, seed=3))
    # the operator surface of a fitted covariance
    p = pts[0]
    f = eng.get_fit(p, int(jeff[p]))
    W = np.diag(f["alpha"]) + f["B"] @ f["D"] @ f["B"].T if d <= 200 else None
    x = pfmi_mod.HostRNG(2).randn(d)
    if W is not None:
        np.testing.assert_allclose(eng.woodbury_apply(p, "mul", x), W @ x, rtol=1e-9, atol=1e-9 * np.abs(W @ x).max())
        np.testing.assert_allclose(eng.woodbury_apply(p, "solve", W @ x), x, rtol=1e-7, atol=1e-8 * np.abs(x).max())
        np.testing.assert_allclose(eng.woodbury_diag(p), np.diag(W), rtol=1e-9)
    assert np.isclose(eng.woodbury_apply(p, "quad", x), x @ eng.woodbury_apply(p, "mul", x), rtol=1e-9)
    # and the public call takes the host optimiser + this route on its own
    r = pfmi_mod.multipathfinder(tg, 40, nruns=2, ndraws_elbo=64, history_length=J, rng=pfmi_mod.HostRNG(3), engine=eng, maxiters=maxit)
    assert r.draws.shape == (d, 40) and np.all(np.isfinite(r.draws))
    eng.close()
