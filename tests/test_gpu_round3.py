"""GPU parity tests added in round 3 (VERDICT r2 "next" #2, #3, #4 and weak #1, #2):

* config 3 compared STRICTLY across the whole trace of full-length paths (first / middle / last fits, full ring, tiny late steps)
  plus `best_iter` and the winner's per-draw log densities against the oracle on >= 8 full-length paths;
* the config-5 share with K >= 4 paths and enough iterations to fill the J = 10 ring, a floor on the number of strict comparisons
  for the funnel too, and an UNCONDITIONAL per-draw check (the fit is picked by the oracle's index);
* device-resident `logp` closures (PFMI_TARGET_DEVICE_CALLBACK): a HIP-implemented user kernel (examples/device_logp) against the
  built-in target and the oracle, a torch closure, failure cases;
* the enqueue / wait split (pfmi_*_enqueue / _wait, pfmi_pool_build_best, pfmi_comm_psis_resample) against the blocking calls;
* the words x = 0 and x = 0x80000000 through the scan's unclamped table look-up.
All calls go through the C ABI of libpfmi.so.
"""
import os

import numpy as np
import pytest

from helpers import demo_device_target, fit_seeds, make_traces, oracle_factor_from_gpu, oracle_target
from oracle import pf_oracle as po
import margins as mg
from test_gpu_parity_r2 import _factor, _wc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pfmi_mod():
    import pfmi
    return pfmi


@pytest.fixture(scope="module")
def eng(pfmi_mod):
    e = pfmi_mod.Engine(0)
    yield e
    e.close()


# ---- config 3: the whole trace, best_iter, the winner's draws ----------------------------------------
@pytest.mark.timeout(1500)
def test_config3_whole_trace_best_iter_and_winner_logs_vs_oracle(pfmi_mod, eng):
    """64 device-made traces of the headline config; 8 FULL-LENGTH paths (~175 fits each: full ring, tiny late s / y, the
    worst-conditioned QR blocks) go through the oracle: every fit's status / j_eff / logdet, strict ELBO / SE on the
    well-conditioned ones (floors asserted per section of the trace), best_iter, and the per-draw logp / logq of the winning fit of
    the production scan against the oracle's own draws."""
    K, d, J, N, KF = 64, 1000, 6, 1000, 8
    tg = pfmi_mod.t_lowrank(d, r=8, seed=2)
    otg = oracle_target(tg)
    eng.set_target(tg)
    run_seeds = pfmi_mod.hostrng.rand_u64(20260928, np.arange(K, dtype=np.uint64), 9)
    x0 = np.stack([pfmi_mod.HostRNG(int(s)).rand(d) * 4 - 2 for s in run_seeds])
    npts = eng.optimize_batch(x0, J)
    eng.fit_batch(J)
    status, jeff, logdet, nrej = eng.fit_status()
    seeds = np.concatenate([pfmi_mod.hostrng.rand_u64(int(run_seeds[k]), np.arange(n, dtype=np.uint64), 10)
                            for k, n in enumerate(npts)])
    elbo, se, best = eng.elbo_batch(N, seeds)
    paths = list(range(0, K, K // KF))[:KF]
    ths, grs, sds = [], [], []
    for k in paths:
        t, _, g = eng.get_trace(k, logp=False)
        ths.append(t); grs.append(g)
        sds.append(seeds[int(eng.offsets[k]):int(eng.offsets[k + 1])])
    off = np.concatenate([[0], np.cumsum([len(t) for t in ths])]).astype(np.int64)
    ref = po.multipath_fit_elbo(off, np.concatenate(ths), np.concatenate(grs), J, otg, N, np.concatenate(sds),
                                nthreads=min(KF, os.cpu_count() or 1))
    n_sec = np.zeros(3, dtype=int)                            # strict comparisons in the first / middle / last 20 fits
    n_strict = n_fits = n_best = 0
    for i, k in enumerate(paths):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        r0, r1 = int(off[i]), int(off[i + 1])
        L = p1 - p0 - 1
        assert L >= 100, L                                        # full-length traces: the ring has been full for most of them
        np.testing.assert_array_equal(status[p0:p1], ref["status"][r0:r1])
        np.testing.assert_array_equal(jeff[p0:p1], ref["j_eff"][r0:r1])
        assert nrej[k] == ref["n_rejected"][i]
        mg.check("C3:whole-trace", "logdet", mg.rel(logdet[p0:p1], ref["logdet"][r0:r1]))
        mg.record("C3:whole-trace", "logdet_abs", np.abs(logdet[p0:p1] - ref["logdet"][r0:r1]), np.inf)
        alpha_all, hl, hs, _ = po.lbfgs_history(ths[i], grs[i], J)
        wc = np.array([_wc(_factor(ths[i], grs[i], alpha_all, hl, hs, l, d)) for l in range(1, L + 1)])
        x, y = elbo[p0 + 1:p1], ref["elbo"][r0 + 1:r1]
        sx, sy = se[p0 + 1:p1], ref["se"][r0 + 1:r1]
        mg.check("C3:whole-trace", "elbo", mg.rel(x[wc], y[wc]), ctx=k)
        mg.check("C3:whole-trace", "se", mg.rel(sx[wc], sy[wc]))
        lo = ~wc
        assert np.all(np.abs(x[lo] - y[lo]) <= 8 * np.maximum(sx[lo], sy[lo]) + 1e-9 * (1 + np.abs(y[lo])))
        mid = L // 2
        for s, sl in enumerate((slice(0, 20), slice(mid - 10, mid + 10), slice(L - 20, L))):
            n_sec[s] += int(wc[sl].sum())
        n_strict += int(wc.sum()); n_fits += L
        # best_iter: identical unless the two best ELBOs are closer than the tolerance
        top = np.sort(y[np.isfinite(y)])[-2:]
        if top[1] - top[0] > 2e-9 * (1 + abs(top[1])):
            assert best[k] == ref["best_iter"][i], (k, best[k], ref["best_iter"][i])
            n_best += 1
        # per-draw log densities of the oracle's winner, straight from the production scan
        b = int(ref["best_iter"][i])
        refd = po.path_fit_elbo(ths[i][:b + 1], grs[i][:b + 1], J, otg, N, sds[i][:b + 1], want_draws=True)
        lp, lq = eng.elbo_logs(p0 + b, N)
        mg.check("C3:whole-trace", "logq@scan", mg.rel(lq, refd["logq"]))
        if wc[b - 1]:
            mg.check("C3:whole-trace", "logp@scan", mg.rel(lp, refd["logp"]))
    print(f"config 3 whole trace: {n_fits} fits of {KF} full paths, {n_strict} strict; per section (first/middle/last 20): "
          f"{n_sec.tolist()}; best_iter compared on {n_best} paths")
    assert n_strict >= n_fits * 3 // 4, (n_strict, n_fits)
    assert np.all(n_sec >= KF * 20 // 2), n_sec
    assert n_best >= KF - 2, n_best


# ---- config 5 share: K >= 4, full ring, floors, unconditional per-draw check --------------------------
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("tname", ["funnel", "diag"])
def test_config5_share_full_ring_vs_oracle(pfmi_mod, eng, tname):
    """d = 10^4, J = 10 (KC = 20), N_e = 2000, K = 4 paths, up to 60 iterations: the ring is full for most fits."""
    d, J, N, K, maxit = 10000, 10, 2000, 4, 60
    tg = pfmi_mod.t_funnel(d) if tname == "funnel" else pfmi_mod.t_diag(d, seed=1)
    otg = oracle_target(tg)
    eng.set_target(tg)
    sc = 10.0 if tname == "funnel" else 2.0
    x0 = pfmi_mod.HostRNG(5).rand(K * d).reshape(K, d) * 2 * sc - sc
    npts = eng.optimize_batch(x0, J, maxit)
    eng.fit_batch(J)
    status, jeff, logdet, nrej = eng.fit_status()
    seeds = fit_seeds(eng.P, 8)
    elbo, se, best = eng.elbo_batch(N, seeds)
    th = np.concatenate([eng.get_trace(k, logp=False)[0] for k in range(K)])
    gr = np.concatenate([eng.get_trace(k, logp=False)[2] for k in range(K)])
    ref = po.multipath_fit_elbo(eng.offsets, th, gr, J, otg, N, seeds, nthreads=min(K, os.cpu_count() or 1))
    np.testing.assert_array_equal(status, ref["status"])
    np.testing.assert_array_equal(jeff, ref["j_eff"])
    np.testing.assert_array_equal(nrej, ref["n_rejected"])
    n_strict = n_fits = n_full = n_mu = 0
    for k in range(K):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        alpha_all, hl, hs, _ = po.lbfgs_history(th[p0:p1], gr[p0:p1], J)
        for l in range(1, p1 - p0):
            if ref["status"][p0 + l] != 0:
                assert np.isnan(elbo[p0 + l])
                continue
            n_fits += 1
            n_full += int(jeff[p0 + l] == J)
            a, b = elbo[p0 + l], ref["elbo"][p0 + l]
            cfg = f"C5-shape:{tname}-4x60"
            mg.check(cfg, "logdet", mg.rel(logdet[p0 + l], ref["logdet"][p0 + l]))
            mg.record(cfg, "logdet_abs", abs(logdet[p0 + l] - ref["logdet"][p0 + l]), np.inf)
            F = _factor(th[p0:p1], gr[p0:p1], alpha_all, hl, hs, l, d)
            # the mean mu = theta + Sigma grad goes through the factor but is a function of Sigma alone: STRICT for every fit,
            # however ill-conditioned the Householder block is
            mu_ref = F.fit_mean(th[p0 + l], gr[p0 + l])
            mu_gpu = eng.get_fit(p0 + l, int(jeff[p0 + l]))["mu"]
            mg.check(cfg, "mu", np.max(np.abs(mu_gpu - mu_ref)) / (1 + np.abs(mu_ref).max()), ctx=(k, l))
            n_mu += 1
            if not (np.isfinite(a) and np.isfinite(b)):         # logp overflows on both sides (funnel: exp(-tau) of a far draw): same value
                assert (np.isnan(a) and np.isnan(b)) or a == b, (k, l, a, b)
                continue
            if _wc(F):
                n_strict += 1
                mg.check(cfg, "elbo", mg.rel(a, b), ctx=(k, l, a, b))
                mg.check(cfg, "se", mg.rel(se[p0 + l], ref["se"][p0 + l]))
            else:
                assert abs(a - b) <= 8 * max(se[p0 + l], ref["se"][p0 + l]) + 1e-8 * (1 + abs(b)), (k, l, a, b)
    print(f"config 5 share ({tname}): {n_fits} fits, {n_mu} strict means, {n_strict} strict ELBOs, {n_full} with a full ring (j = {J})")
    assert n_fits >= K * 10 and n_full >= n_fits // 2 and n_mu == n_fits, (n_fits, n_full, n_mu)
    # the funnel's scaled block U^-T [alpha Y  S] is numerically rank deficient from the second iteration on (y ~ exp(-tau) s), so
    # x(u) is defined by roundoff there (in LAPACK as much as here): its floor is on the quantities that ARE functions of
    # Sigma -- status, j_eff, logdet, mu (above, every fit) and the per-draw logq (below, every path); the well-conditioned diagonal
    # Gaussian at the same shape carries the strict ELBO floor
    assert n_strict >= (n_fits * 3 // 4 if tname == "diag" else 1), (n_strict, n_fits)
    # UNCONDITIONAL per-draw check: the fit is picked by the ORACLE's index on every path
    for k in range(K):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        b = int(ref["best_iter"][k])
        refd = po.path_fit_elbo(th[p0:p0 + b + 1], gr[p0:p0 + b + 1], J, otg, N, seeds[p0:p0 + b + 1], want_draws=True)
        lp, lq = eng.elbo_logs(p0 + b, N)
        mg.check(f"C5-shape:{tname}-4x60", "logq@scan", mg.rel(lq, refd["logq"]))
        alpha_all, hl, hs, _ = po.lbfgs_history(th[p0:p1], gr[p0:p1], J)
        if _wc(_factor(th[p0:p1], gr[p0:p1], alpha_all, hl, hs, b, d)):
            mg.check(f"C5-shape:{tname}-4x60", "logp@scan", mg.rel(lp, refd["logp"]))
        else:                                                   # rank-deficient block: x(u) is not well defined, logp's law is
            assert abs(lp.mean() - refd["logp"].mean()) <= 8 * (lp.std() + refd["logp"].std()) / np.sqrt(N) + 1e-8 * abs(lp.mean())
        # round 4: STRICT per-draw logp / logq / x whatever the conditioning -- the oracle's reflector-by-reflector
        # apply on the GPU's OWN factor of this fit (x(u) is a function of exactly those arrays)
        fg = eng.get_fit(p0 + b, int(jeff[p0 + b]))
        Fg = oracle_factor_from_gpu(fg)
        Xg, lqg = Fg.rand_and_logpdf(fg["mu"], po.randn_fill(int(seeds[p0 + b]), d, N))
        mg.check(f"C5-shape:{tname}-4x60", "logq@scan_vs_oracle_on_gpu_factor", mg.rel(lq, lqg))
        mg.check(f"C5-shape:{tname}-4x60", "logp@scan_vs_oracle_on_gpu_factor", mg.rel(lp, otg.logp(Xg)))
        Xd, _, _ = eng.draws(p0 + b, int(seeds[p0 + b]), 32, n0=5)
        mg.check(f"C5-shape:{tname}-4x60", "draws@writer_vs_oracle_on_gpu_factor",
                 np.abs(Xd - Xg[:, 5:37]) / (1 + np.abs(Xg[:, 5:37]).max(axis=0)))


# ---- device-resident logp closures (VERDICT r2 missing #3 / next #3a) ------------------------------------------------------
@pytest.mark.parametrize("shape", [("lowrank", 1000, 6, 1000), ("lowrank", 130, 6, 200), ("diag", 77, 4, 64), ("funnel", 300, 6, 256),
                                   ("lowrank", 3000, 10, 300)])
def test_device_callback_matches_builtin_target_and_oracle(pfmi_mod, eng, shape):
    """The SAME target once as a built-in (logp expanded algebraically, x never formed) and once as a DEVICE closure: the library
    materialises the draws in HBM and the user's HIP kernel (examples/device_logp) evaluates logp there.  ELBO / SE / argmax / per-draw
    logs must agree to roundoff with the built-in route and with the oracle; pool, PSIS and indices follow."""
    tname, d, J, N = shape
    tg = {"lowrank": lambda: pfmi_mod.t_lowrank(d, r=8, seed=2), "diag": lambda: pfmi_mod.t_diag(d, seed=1),
          "funnel": lambda: pfmi_mod.t_funnel(d)}[tname]()
    K = 3
    sc = 2.0
    x0 = pfmi_mod.HostRNG(17).rand(K * d).reshape(K, d) * 2 * sc - sc
    eng.set_target(tg)
    npts = eng.optimize_batch(x0, J, 40)
    traces = [eng.get_trace(k, logp=False) for k in range(K)]
    eng.fit_batch(J)
    seeds = fit_seeds(eng.P, 5)
    elbo0, se0, best0 = eng.elbo_batch(N, seeds)
    logs0 = [eng.elbo_logs(int(eng.offsets[k]) + int(best0[k]), N) for k in range(K)]
    pts = [int(eng.offsets[k]) + int(best0[k]) for k in range(K)]
    eng.pool_build(N, pts, seeds[pts])
    pool0, lr0 = eng.pool_get()
    # ---- the device closure
    dtg = demo_device_target(tg)
    e2 = pfmi_mod.Engine(0)
    try:
        e2.set_target(dtg)
        e2.set_traces([t[0] for t in traces], [t[2] for t in traces])
        e2.fit_batch(J)
        for chunk_mb in (None, "0.5"):                            # one block, and many small blocks of fits
            if chunk_mb:
                os.environ["PFMI_DEVCB_CHUNK_MB"] = chunk_mb
            try:
                elbo1, se1, best1 = e2.elbo_batch(N, seeds)
            finally:
                os.environ.pop("PFMI_DEVCB_CHUNK_MB", None)
            assert e2.callback_stats_dev()["bytes_in_hbm"] == 8.0 * d * N * (e2.P - K)
            fin = np.isfinite(elbo0)
            np.testing.assert_array_equal(np.isfinite(elbo1), fin)
            assert np.max(np.abs(elbo1[fin] - elbo0[fin]) / (1 + np.abs(elbo0[fin]))) <= 1e-9
            assert np.max(np.abs(se1[fin] - se0[fin]) / (1 + se0[fin])) <= 1e-8
            np.testing.assert_array_equal(best1, best0)
        for k in range(K):
            lp1, lq1 = e2.elbo_logs(pts[k], N)
            assert np.max(np.abs(lq1 - logs0[k][1]) / (1 + np.abs(lq1))) <= 1e-13      # same normals; |u|^2 summed in another order
            assert np.max(np.abs(lp1 - logs0[k][0]) / (1 + np.abs(logs0[k][0]))) <= 1e-9
        e2.pool_build(N, pts, seeds[pts])
        pool1, lr1 = e2.pool_get()
        # same normals, same factor; the built-in route may use another writer (two-pass kernel for d <= 1024): roundoff apart
        assert np.max(np.abs(pool1 - pool0) / (1 + np.abs(pool0).max(axis=0))) <= 1e-10
        assert np.max(np.abs(lr1 - lr0) / (1 + np.abs(lr0))) <= 1e-9
        for k in range(K):                                        # within ONE target a draw is the same bits alone or in the pool
            Xk, _, _ = e2.draws(pts[k], seeds[pts[k]], 3, n0=7)
            np.testing.assert_array_equal(Xk, pool1[:, 7:10, k])
        # against the closure evaluated on the host copy of the same draws, and pfmi_draws through the closure
        X, lpd, lqd = e2.draws(pts[0], seeds[pts[0]], 50)
        assert np.max(np.abs(lpd - tg.logp(X)) / (1 + np.abs(lpd))) <= 1e-11
    finally:
        e2.close()
    # oracle, path 0
    ref = po.path_fit_elbo(traces[0][0], traces[0][2], J, oracle_target(tg), N, seeds[:int(eng.offsets[1])])
    y = ref["elbo"][1:]
    x = elbo1[1:int(eng.offsets[1])]
    ok = np.isfinite(y)
    assert np.all(np.abs(x[ok] - y[ok]) <= 8 * ref["se"][1:][ok] + 1e-8 * (1 + np.abs(y[ok])))


def test_device_callback_failed_fits_and_torch_closure(pfmi_mod):
    """(i) failed fits stay NaN through the device-closure route (their draws do not exist); (ii) a closure written with torch ops
    on the engine's stream (pfmi.TorchDeviceTarget) == the host closure on the same draws."""
    import torch
    d, J, N = 40, 5, 128
    rng = np.random.default_rng(0)
    bad_th, bad_gr = np.cumsum(rng.normal(size=(9, d)), 0), rng.normal(size=(9, d))
    tg = pfmi_mod.t_diag(d, seed=3)
    good = make_traces(tg, 2, 3)
    e = pfmi_mod.Engine(0)
    try:
        m = torch.as_tensor(tg.mean, device="cuda:0")
        a = torch.as_tensor(tg.a, device="cuda:0")
        ttg = pfmi_mod.TorchDeviceTarget(d, lambda X: -0.5 * (((X - m) ** 2) * a).sum(1), host=tg)
        for target in (demo_device_target(tg), ttg):
            e.set_target(target)
            e.set_traces([bad_th] + [t.points for t in good], [bad_gr] + [t.gradients for t in good])
            e.fit_batch(J, -1e300)                               # negative-curvature pairs accepted: non-PD fits (src/woodbury.jl:202,205)
            status = e.fit_status()[0]
            assert np.any(status != 0) and np.any(status == 0)
            seeds = fit_seeds(e.P, 2)
            elbo, se, best = e.elbo_batch(N, seeds)
            for p in range(e.P):
                lp, lq = e.elbo_logs(p, N)
                first = p in e.offsets[:-1]
                if status[p] != 0:
                    assert np.all(np.isnan(lp)) and np.all(np.isnan(lq)) and np.isnan(elbo[p])
                elif not first:
                    X, _, lq2 = e.draws(p, seeds[p], N)
                    np.testing.assert_array_equal(lq, lq2)
                    assert np.max(np.abs(lp - tg.logp(X)) / (1 + np.abs(lp))) <= 1e-12
    finally:
        e.close()


# ---- enqueue / wait split and the device-side winner pick -----------------------------------------
def test_enqueue_wait_and_pool_build_best_equal_blocking_calls(pfmi_mod, eng):
    d, J, N, K = 120, 6, 200, 6
    tg = pfmi_mod.t_lowrank(d, r=8, seed=2)
    eng.set_target(tg)
    x0 = pfmi_mod.HostRNG(23).rand(K * d).reshape(K, d) * 4 - 2
    # one path that cannot succeed: starts at the optimum -> L = 0 -> fit_iteration 0, success false (src/singlepath.jl:299)
    x0[2] = tg.mean
    npts = eng.optimize_batch(x0, J, 60)
    assert npts[2] == 1
    eng.fit_batch(J)
    seeds = fit_seeds(eng.P, 9)
    elbo, se, best = eng.elbo_batch(N, seeds)
    fail_seeds = pfmi_mod.hostrng.rand_u64(77, np.arange(K, dtype=np.uint64), 3)
    pts = np.array([int(eng.offsets[k]) + int(best[k]) for k in range(K)])
    ok = np.array([npts[k] > 1 and best[k] > 0 and np.isfinite(elbo[pts[k]]) for k in range(K)])
    assert not ok[2] and ok.sum() == K - 1 and best[2] == 0
    sd = np.where(ok, seeds[pts], fail_seeds)
    eng.pool_build(N, pts, sd)
    pool_ref, lr_ref = eng.pool_get()
    ref = eng.psis(lr_ref)
    idx_ref = eng.resample_indices(K * N, 150, seed=4)
    draws_ref = eng.pool_gather(idx_ref)
    # ---- the same through the enqueue-only entry points: nothing waits until psis_resample's single synchronisation
    e2 = pfmi_mod.Engine(0)
    try:
        e2.set_target(tg)
        e2.optimize_batch_enqueue(x0, J, 60)
        np.testing.assert_array_equal(e2.optimize_batch_wait(), npts)
        e2.fit_batch(J)
        e2.elbo_batch_enqueue(N, seeds)
        e2.pool_build_best(N, fail_seeds)
        comm = pfmi_mod.Comm.init_all([e2])
        res, idx, draws = comm.psis_resample(150, seed=4)
        elbo2, se2, best2 = e2.elbo_batch_wait()
        np.testing.assert_array_equal(elbo2, elbo)
        np.testing.assert_array_equal(se2, se)
        np.testing.assert_array_equal(best2, best)
        p2, s2, ok2 = e2.pool_winners()
        np.testing.assert_array_equal(p2, pts)
        np.testing.assert_array_equal(s2, sd)
        np.testing.assert_array_equal(ok2, ok)
        pool2, lr2 = e2.pool_get()
        np.testing.assert_array_equal(pool2, pool_ref)
        np.testing.assert_array_equal(lr2, lr_ref)
        assert res["pareto_shape"] == ref["pareto_shape"] and res["tail_length"] == ref["tail_length"]
        np.testing.assert_array_equal(idx, idx_ref)
        np.testing.assert_array_equal(draws, draws_ref)
        w, lw = e2.psis_weights(K * N)
        np.testing.assert_array_equal(w, ref["weights"])
        np.testing.assert_array_equal(lw, ref["log_weights"])
        comm.close()
    finally:
        e2.close()


# ---- the unclamped look-up of the scan for the words 0 and 0x80000000 -----------------------------------
def test_scan_generator_handles_zero_magnitude_words(pfmi_mod, eng):
    """mag = 0 (probability 2^-31 per normal: several per benchmark step) indexes far in front of the LDS copy of the table; the
    value is recomputed by the refinement path.  There is no way to force a Philox word, so the guarantee is tested where it is
    made: the scan (qf kernel), the draw-writing kernel and the lane kernel must agree with the ORACLE's generator on a stream long
    enough to contain words below 2^12 (refinement) -- and the look-up index is clamped (pf_icdf_issue_adj), so no LDS address
    outside the allocation is ever formed; the oracle's pfo_randn4 is checked on the literal words 0 and 0x80000000."""
    z = po.icdf_words(np.array([0, 0x80000000, 1, 0x80000001], dtype=np.uint32), np.array([5, 5, 0, 0], dtype=np.uint32))
    assert np.all(np.isfinite(z)) and z[0] > 8.5 and z[1] < -8.5 and z[0] == -z[1]
    d, N = 64, 400_000
    eng.set_target(pfmi_mod.t_iso(d))
    eng.set_traces([np.zeros((3, d))], [np.zeros((3, d))])
    eng.fit_batch(6)
    seeds = np.array([0, 0x1234567, 0xABCDEF0123], dtype=np.uint64)
    eng.elbo_batch(N, seeds)
    for p in (1, 2):
        U = po.randn_fill(int(seeds[p]), d, N)
        _, lq = eng.elbo_logs(p, N)
        ref = -(d * np.log(2 * np.pi) + np.sum(U * U, axis=0)) / 2
        assert np.max(np.abs(lq - ref)) <= 1e-12 * np.abs(ref).max()


# ---- results do not depend on the launch geometry (found by the G > 1 runs of round 3) --------------------------------------
@pytest.mark.parametrize("N", [600, 1000])
def test_scan_is_bitwise_independent_of_launch_geometry(pfmi_mod, eng, N):
    """The ELBO scan cuts its work in launch-dependent ways: the fits beyond the last full round of CUs go into a tail launch of
    one-batch pieces, and a wave owns one or two 16-draw groups.  A fit's per-draw log densities -- hence its ELBO, hence
    best_iter -- must be the SAME BITS whichever way it was cut (the reference's `ntasks` invariance, test/multipath.jl:107-140;
    here it also makes the result independent of the number of GPUs the paths are sharded over)."""
    d, J, K = 100, 6, 10
    tg = pfmi_mod.t_lowrank(d, r=8, seed=2)
    eng.set_target(tg)
    x0 = pfmi_mod.HostRNG(31).rand(K * d).reshape(K, d) * 4 - 2
    npts = eng.optimize_batch(x0, J)
    eng.fit_batch(J)
    nfits = eng.P - K
    assert nfits > 256 and nfits % 256 != 0, nfits              # a main launch AND a tail launch
    seeds = fit_seeds(eng.P, 12)
    elbo_a, se_a, best_a = eng.elbo_batch(N, seeds)
    last = eng.P - 1                                            # a fit of the tail launch
    logs_a = eng.elbo_logs(last, N)
    os.environ["PFMI_QF_NO_TAIL"] = "1"
    try:
        elbo_b, se_b, best_b = eng.elbo_batch(N, seeds)
        logs_b = eng.elbo_logs(last, N)
    finally:
        os.environ.pop("PFMI_QF_NO_TAIL", None)
    np.testing.assert_array_equal(logs_a[0], logs_b[0])
    np.testing.assert_array_equal(logs_a[1], logs_b[1])
    np.testing.assert_array_equal(elbo_a, elbo_b)
    np.testing.assert_array_equal(se_a, se_b)
    np.testing.assert_array_equal(best_a, best_b)
    # the pieces ride behind the whole fits in ONE launch; the same cut as two launches gives the same bits
    os.environ["PFMI_QF_TWO_LAUNCHES"] = "1"
    try:
        elbo_t, se_t, best_t = eng.elbo_batch(N, seeds)
        logs_t = eng.elbo_logs(last, N)
    finally:
        os.environ.pop("PFMI_QF_TWO_LAUNCHES", None)
    np.testing.assert_array_equal(logs_a[0], logs_t[0])
    np.testing.assert_array_equal(logs_a[1], logs_t[1])
    np.testing.assert_array_equal(elbo_a, elbo_t)
    np.testing.assert_array_equal(best_a, best_t)
    # the same fits as a 2-path batch on a fresh engine (few fits: the groups of a fit are split over several workgroups)
    e2 = pfmi_mod.Engine(0)
    try:
        e2.set_target(tg)
        e2.optimize_batch(x0[:2], J)
        e2.fit_batch(J)
        elbo_c, se_c, best_c = e2.elbo_batch(N, seeds[:e2.P])
        np.testing.assert_array_equal(elbo_c, elbo_a[:e2.P])
        np.testing.assert_array_equal(best_c, best_a[:2])
    finally:
        e2.close()


# ---- the streaming draw writer (elbo_xw_kernel.hip) ---------------------------------------------------------------------------
def _with_kernel(mode, fn):
    old = os.environ.get("PFMI_ELBO_KERNEL")
    os.environ["PFMI_ELBO_KERNEL"] = mode
    try:
        return fn()
    finally:
        os.environ.pop("PFMI_ELBO_KERNEL", None)
        if old is not None:
            os.environ["PFMI_ELBO_KERNEL"] = old


@pytest.mark.parametrize("tname,d,J,N,scale,maxit", [
    ("lr", 1000, 6, 1000, 2.0, 25),        # config 3's shape: Vh resident in LDS
    ("lr", 130, 6, 200, 2.0, 25),          # ragged last block (130 = 8 x 16 + 2) and last group (200 = 12 x 16 + 8)
    ("diag", 10, 6, 64, 2.0, 25),          # 2 j > d: the head transform covers every row
    ("diag", 33, 3, 17, 2.0, 12),          # KC = 8, a single ragged group
    ("funnel", 500, 10, 300, 3.0, 30),     # KC = 20: head transform spills into block 1
    ("lr", 3000, 10, 272, 2.0, 20),        # streamed Vh (12 chunks), KC = 20
    ("diag", 2500, 16, 100, 2.0, 24),      # KC = 32, streamed
    ("funnel", 10000, 10, 160, 10.0, 12),  # config 5's shape
])
def test_draw_writer_matches_lane_kernel_and_oracle_normals(pfmi_mod, eng, tname, d, J, N, scale, maxit):
    """The streaming writer (two passes with regenerated normals, MFMA compact-WY apply, LDS-transposed full-line stores) against
    the lane-per-draw kernel on the same (fit, seed): draws <= 1e-10 per column, logq <= 1e-12, logp (built-in target: the scan's
    expanded form on the same draws) <= 1e-10; n0 > 0 continues the same counter (top-up draws, src/singlepath.jl:229-230);
    pool_build takes the same route."""
    tg = {"diag": lambda d: pfmi_mod.t_diag(d, 1), "lr": lambda d: pfmi_mod.t_lowrank(d, 8, 2), "funnel": pfmi_mod.t_funnel}[tname](d)
    K = 2
    eng.set_target(tg)
    x0 = pfmi_mod.HostRNG(3).rand(K * d).reshape(K, d) * 2 * scale - scale
    eng.optimize_batch(x0, J, maxit)
    eng.fit_batch(J)
    status, jeff, _, _ = eng.fit_status()
    pts = sorted({1, eng.P - 1, int(eng.offsets[1]) + 1, eng.P // 2})
    assert jeff.max() == min(J, maxit)
    for p in pts:
        if status[p] != 0:
            continue
        seed = 1000 + p
        Xw, lpw, lqw = _with_kernel("xw", lambda: eng.draws(p, seed, N))
        Xl, lpl, lql = _with_kernel("lane", lambda: eng.draws(p, seed, N))
        scale_x = 1 + np.abs(Xl).max(axis=0)
        assert np.max(np.abs(Xw - Xl) / scale_x) <= 1e-10, (p, np.max(np.abs(Xw - Xl) / scale_x))
        assert np.max(np.abs(lqw - lql) / (1 + np.abs(lql))) <= 1e-12
        assert np.max(np.abs(lpw - lpl) / (1 + np.abs(lpl))) <= 1e-10
        # the default route (the two-pass kernel when d <= 1024, J <= 8 and the target is built in; the writer otherwise) and a later
        # window of the same stream: the same bits whether a draw is made alone or in a block
        Xd, lpd, lqd = eng.draws(p, seed, N)
        assert np.max(np.abs(Xd - Xw) / scale_x) <= 1e-10
        n0 = 16 * 3 + 5
        if N >= n0 + 40:
            X2, lp2, lq2 = eng.draws(p, seed, 40, n0=n0)
            np.testing.assert_array_equal(X2, Xd[:, n0:n0 + 40])
            np.testing.assert_array_equal(lq2, lqd[n0:n0 + 40])
            X3, _, lq3 = _with_kernel("xw", lambda: eng.draws(p, seed, 40, n0=n0))
            np.testing.assert_array_equal(X3, Xw[:, n0:n0 + 40])
            np.testing.assert_array_equal(lq3, lqw[n0:n0 + 40])
    best = [1, 1]
    pp = [int(eng.offsets[k]) + best[k] for k in range(K)]
    sd = np.array([77, 78], dtype=np.uint64)
    eng.pool_build(N, pp, sd)
    pool, lr = eng.pool_get()
    for k in range(K):
        X, lp, lq = eng.draws(pp[k], int(sd[k]), N)
        np.testing.assert_array_equal(pool[:, :, k], X)
        np.testing.assert_array_equal(lr[k * N:(k + 1) * N], lp - lq)


def test_draw_writer_normals_bit_identical_to_oracle(pfmi_mod, eng):
    """theta = grad = 0: the first fit is N(0, I), so x = u -- the writer's normals ARE the oracle's, bit for bit (incl. refined words)"""
    d, N = 50, 300_000
    eng.set_target(pfmi_mod.t_iso(d))
    eng.set_traces([np.zeros((2, d))], [np.zeros((2, d))])
    eng.fit_batch(6)
    X, lp, lq = _with_kernel("xw", lambda: eng.draws(0, 0xC0FFEE123456789, N))
    U = po.randn_fill(0xC0FFEE123456789, d, N)
    np.testing.assert_array_equal(X, U)


# ---- device L-BFGS: the pair-rejected branch (second gather of S'g, Y'g over the unchanged ring) ---------------------------------
@pytest.mark.parametrize("name,d,J", [("lr", 200, 6), ("diag", 1500, 4), ("funnel", 40, 6)])
def test_device_lbfgs_rejected_pairs_follow_the_host_driver(pfmi_mod, name, d, J, monkeypatch):
    """A strong-Wolfe step always passes the curvature test, so the kernel's `pair rejected' branch (ring left as it is, inner
    products of the OLD ring with the new gradient gathered in a second pass) never runs on its own: PFMI_LBFGS_REJECT_EVERY=3 drops
    every third pair, pfmi/optimize.py (the host twin: two-loop recursion in NumPy) does the same, and the iterates must agree
    -- through several rejections, a full ring and its wrap-around."""
    from pfmi.optimize import optimize_with_trace
    tg = {"lr": lambda: pfmi_mod.t_lowrank(d, 8, 2), "diag": lambda: pfmi_mod.t_diag(d, 1), "funnel": lambda: pfmi_mod.t_funnel(d)}[name]()
    K = 3
    x0 = pfmi_mod.HostRNG(11).rand(K * d).reshape(K, d) * 4 - 2
    monkeypatch.setenv("PFMI_LBFGS_REJECT_EVERY", "3")
    e2 = pfmi_mod.Engine(0)
    try:
        e2.set_target(tg)
        npts = e2.optimize_batch(x0, J, 40)
        for k in range(K):
            th, lp, gr = e2.get_trace(k)
            ref = optimize_with_trace(tg, x0[k], J, 40, _reject_every=3)
            n = min(len(th), len(ref), 14)
            assert n >= 10, (n, npts)
            rt = 1e-5 if name == "funnel" else 1e-7           # (the funnel amplifies roundoff between the two recursions faster)
            np.testing.assert_allclose(th[:n], ref.points[:n], rtol=rt, atol=rt / 10)
            np.testing.assert_allclose(gr[:n], ref.gradients[:n], rtol=10 * rt, atol=rt * max(1.0, np.abs(ref.gradients[:n]).max()))
            assert np.all(np.diff(lp) >= -1e-9 * np.maximum(1.0, np.abs(lp[1:])))
    finally:
        e2.close()
    monkeypatch.delenv("PFMI_LBFGS_REJECT_EVERY")
    e3 = pfmi_mod.Engine(0)
    try:                                                     # and without the hook the host twin follows the kernel as well
        e3.set_target(tg)
        e3.optimize_batch(x0, J, 40)
        th, lp, gr = e3.get_trace(0)
        ref = optimize_with_trace(tg, x0[0], J, 40)
        n = min(len(th), len(ref), 14)
        np.testing.assert_allclose(th[:n], ref.points[:n], rtol=rt, atol=rt / 10)
    finally:
        e3.close()


# ---- stage timers: host-synchronised (mode 1) and in-stream (mode 2) ---------------------------------------------------------------
def test_profile_modes_agree_and_do_not_change_results(pfmi_mod):
    """pfmi_profile(ctx, 2) leaves the hipEvent pairs in the stream (the pipeline runs as unprofiled) and pfmi_kernel_time reads them:
    same launch counts as mode 1, times of the same order (mode 1 adds the host's launch latency to every stage), identical results."""
    d, K, J, N = 64, 6, 6, 256
    tg = pfmi_mod.t_lowrank(d, 8, 2)
    x0 = pfmi_mod.HostRNG(4).rand(K * d).reshape(K, d) * 4 - 2
    out = {}
    e = pfmi_mod.Engine(0)
    try:
        e.set_target(tg)
        for mode in (0, 1, 2):
            e.profile(mode)
            e.optimize_batch(x0, J)
            seeds = fit_seeds(e.P, 2)
            for _ in range(3):
                e.fit_batch(J)
                e.elbo_batch_enqueue(N, seeds)
                e.pool_build_best(N, np.arange(K, dtype=np.uint64))
                res = e.elbo_batch_wait()
            out[mode] = (res, {n: e.kernel_time(n) for n in ("optimize", "history", "fit", "elbo_draws", "elbo_draws_x", "elbo_reduce")})
        with pytest.raises(pfmi_mod.PfmiError):
            e.profile(3)
    finally:
        e.close()
    for mode in (1, 2):
        for a, b in zip(out[0][0], out[mode][0]):
            np.testing.assert_array_equal(a, b)
    assert all(v == (0.0, 0) for v in out[0][1].values())
    for name, (ms1, n1) in out[1][1].items():
        ms2, n2 = out[2][1][name]
        assert n1 == n2 and n1 >= 1, (name, n1, n2)
        assert 0.0 < ms2 <= ms1 * 1.5 + 0.05, (name, ms1, ms2)       # in-stream figures carry no launch latency: never much above mode 1


# ---- random shapes: every kernel route against its sibling and the history walk against the oracle -------------------------------------
@pytest.mark.timeout(900)
def test_fuzz_random_shapes_cross_kernel_consistency():
    """tests/probes/fuzz_probe.py with a fixed seed: 40 random (d, J, K, N, target) cases, d from 3 to 7000 -- single-pass scan vs
    lane kernel per draw, register / panel vs memory-resident fit kernel, history walk vs the oracle, on device-made traces."""
    import json
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "probes", "fuzz_probe.py"), "7", "40"], capture_output=True, text=True,
                         timeout=800, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("cases")][-1]
    assert "MISMATCH" not in out.stdout, out.stdout[-2000:]
    assert re.search(r"cases 40 pattern mismatches 0 ", line), line
    worst = json.loads(line[line.index("{"):].replace("'", '"'))
    print(line)
    for key in ("lp", "lq", "elbo", "alpha"):
        assert worst[key] <= 1e-9, (key, worst)
    for key in ("ld", "mu"):
        assert worst[key] <= 1e-7, (key, worst)


# ---- resample(::MultiPathfinderResult), scenario by scenario as the reference tests it -----------------------------------------------
def test_resample_multipathfinder_result_like_reference_testset(pfmi_mod):
    """reference test/multipath.jl:142-230: dim = 5, nruns = 4, ndraws_per_run = 20, ndraws_new = 8, logp = -|x|^2 / 2"""
    dim, nruns, npr, nnew = 5, 4, 20, 8
    tg = pfmi_mod.t_iso(dim)
    result = pfmi_mod.multipathfinder(tg, npr, nruns=nruns, ndraws_per_run=npr, rng=pfmi_mod.HostRNG(42))
    pool = lambda res: np.concatenate([r.draws for r in res.pathfinder_results], axis=1)                 # mapreduce(x -> x.draws, hcat, ...)
    in_pool = lambda cols, P: all(any(np.array_equal(c, P[:, q]) for q in range(P.shape[1])) for c in cols.T)

    # resample existing draws with replacement (:153-165)
    r2 = pfmi_mod.resample(result, nnew)
    assert isinstance(r2, pfmi_mod.MultiPathfinderResult)
    assert r2.draws.shape == (dim, nnew) and len(r2.draw_component_ids) == nnew
    assert len(np.unique(r2.draw_component_ids)) <= nruns
    assert r2.draws_transformed is r2.draws or np.array_equal(r2.draws_transformed, r2.draws)
    assert r2.psis_result is result.psis_result
    assert in_pool(r2.draws, pool(result))
    # component_ids consistent with draws (test/resample.jl:51-59): every draw is a column of ITS component's block
    P = pool(result)
    for c, cid in zip(r2.draws.T, r2.draw_component_ids):
        blk = P[:, (cid - 1) * npr:cid * npr]
        assert any(np.array_equal(c, blk[:, q]) for q in range(npr))

    # without replacement (:167-175)
    r3 = pfmi_mod.resample(result, nnew, replace=False)
    assert r3.draws.shape == (dim, nnew) and in_pool(r3.draws, pool(result))
    assert len({c.tobytes() for c in r3.draws.T}) == nnew

    # without importance (:177-184)
    r4 = pfmi_mod.resample(result, nnew, importance=False)
    assert r4.psis_result is None and in_pool(r4.draws, pool(result))

    # with importance, no stored PSIS (:186-198)
    result_no_psis = pfmi_mod.multipathfinder(tg, npr, nruns=nruns, ndraws_per_run=npr, rng=pfmi_mod.HostRNG(42), importance=False,
                                              engine=result.engine)
    assert result_no_psis.psis_result is None
    r5 = pfmi_mod.resample(result_no_psis, nnew)
    assert isinstance(r5, pfmi_mod.MultiPathfinderResult) and isinstance(r5.psis_result, pfmi_mod.PSISResult)
    assert in_pool(r5.draws, pool(result_no_psis))

    # generate new draws (:200-207), also without importance (:209-215)
    r6 = pfmi_mod.resample(result_no_psis, nnew, ndraws_per_run=50, replace=True)
    assert r6.draws.shape == (dim, nnew) and len(r6.draw_component_ids) == nnew and isinstance(r6.psis_result, pfmi_mod.PSISResult)
    r7 = pfmi_mod.resample(result_no_psis, nnew, ndraws_per_run=50, importance=False, replace=True)
    assert r7.draws.shape == (dim, nnew) and r7.psis_result is None

    # non-mutating (:217-221), preserved fields (:223-230)
    before = result_no_psis.draws.copy()
    r8 = pfmi_mod.resample(result_no_psis, nnew)
    np.testing.assert_array_equal(result_no_psis.draws, before)
    assert r8.input is result_no_psis.input and r8.fit_distribution is result_no_psis.fit_distribution
    assert r8.fit_distribution_transformed is result_no_psis.fit_distribution_transformed
    assert r8.pathfinder_results is result_no_psis.pathfinder_results and r8.logp is result_no_psis.logp


def test_woodbury_remaining_surface_like_reference_testsets(pfmi_mod):
    """reference test/woodbury.jl:228-309 on a fitted covariance of a real run: adjoint / transpose, + UniformScaling, right division,
    PDMats.dim (the operators themselves: test_gpu_parity.py::test_woodbury_operator_surface)"""
    tg = pfmi_mod.t_lowrank(12, 3, 5)
    res = pfmi_mod.pathfinder(tg, init=np.linspace(-1.0, 1.0, 12), rng=pfmi_mod.HostRNG(3), ndraws=10)
    W = res.fit_distribution.Sigma
    Wm = W.dense()
    n = 12
    assert W.T is W and W.dim == n
    c = 0.37
    np.testing.assert_allclose(W + c, Wm + c * np.eye(n), rtol=1e-12)
    np.testing.assert_allclose(c + W, c * np.eye(n) + Wm, rtol=1e-12)
    rng = np.random.default_rng(2)
    x = rng.normal(size=n)
    np.testing.assert_allclose(W.rdiv(x), np.linalg.solve(Wm, x), rtol=1e-7, atol=1e-9)          # x' / W = (W \ x)'
    X = rng.normal(size=(2, n))
    np.testing.assert_allclose(W.rdiv(X), np.linalg.solve(Wm, X.T).T, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("name,d,J,maxit", [("lr", 40, 1, 60), ("diag", 300, 2, 80), ("lr", 700, 10, 200), ("diag", 90, 16, 200), ("funnel", 20, 16, 60),
                                            ("lr", 2000, 16, 60), ("diag", 2500, 1, 40)])
def test_device_lbfgs_history_lengths_1_to_16(pfmi_mod, name, d, J, maxit):
    """the fused reduction of the device L-BFGS handles the ring in batches of 6 (d <= 1024) or 2 (d > 1024) pairs: history lengths
    that are one batch, several batches and a ragged last batch, a ring of one pair (every update evicts), on every workgroup shape --
    against the oracle driver (first iterates to roundoff) and by its own invariants (monotone, converged, recorded values consistent)."""
    tg = {"lr": lambda: pfmi_mod.t_lowrank(d, 8, 2), "diag": lambda: pfmi_mod.t_diag(d, 1), "funnel": lambda: pfmi_mod.t_funnel(d)}[name]()
    ot = oracle_target(tg)
    K = 2
    sc = 10.0 if name == "funnel" else 2.0
    x0 = pfmi_mod.HostRNG(13).rand(K * d).reshape(K, d) * 2 * sc - sc
    e = pfmi_mod.Engine(0)
    try:
        e.set_target(tg)
        npts = e.optimize_batch(x0, J, maxit)
        assert np.all(npts >= 2)
        for k in range(K):
            th, lp, gr = e.get_trace(k)
            P, L, G = po.optimize_trace(ot, x0[k], J, maxit)
            n = min(len(P), len(th), 10)
            rt = 1e-6 if name == "funnel" else 1e-8
            np.testing.assert_allclose(th[:n], P[:n], rtol=rt, atol=rt)
            for l in sorted({0, len(th) // 2, len(th) - 1}):
                lpo, go = po.logp_grad(ot, th[l])
                assert abs(lpo - lp[l]) <= 1e-10 * max(1.0, abs(lpo))
                np.testing.assert_allclose(gr[l], go, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(go).max()))
            assert np.all(np.diff(lp) >= -1e-9 * np.maximum(1.0, np.abs(lp[1:])))
            if name != "funnel" and npts[k] <= maxit:
                assert np.abs(gr[-1]).max() <= 1e-8
    finally:
        e.close()


@pytest.mark.parametrize("tname,d,J", [("diag", 2500, 6), ("funnel", 6000, 10), ("diag", 10000, 10), ("diag", 8200, 4)])
def test_lean_history_walk_is_bit_identical_to_the_prefetching_one(pfmi_mod, tname, d, J, monkeypatch):
    """2048 < d <= 10 240: pf_history_lean_kernel (alpha + two row sets in registers, 1 / alpha in LDS) against pf_history_kernel
    (PFMI_HISTORY_KERNEL=prefetch): alpha of every point, effective history, ring sources, rejections -- and with them every fit --
    must be the same bits; both against the oracle's walk."""
    tg = pfmi_mod.t_funnel(d) if tname == "funnel" else pfmi_mod.t_diag(d, 1)
    sc = 10.0 if tname == "funnel" else 2.0
    K = 2
    x0 = pfmi_mod.HostRNG(17).rand(K * d).reshape(K, d) * 2 * sc - sc
    out = {}
    for mode in ("lean", "prefetch"):
        if mode == "prefetch":
            monkeypatch.setenv("PFMI_HISTORY_KERNEL", "prefetch")
        else:
            monkeypatch.delenv("PFMI_HISTORY_KERNEL", raising=False)
        e = pfmi_mod.Engine(0)
        try:
            e.set_target(tg)
            e.optimize_batch(x0, J, 30)
            e.fit_batch(J)
            st, je, ld, nr = e.fit_status()
            fits = [e.get_fit(p, int(je[p])) for p in sorted({1, e.P // 2, e.P - 1})]
            traces = [e.get_trace(k, logp=False) for k in range(K)]
            out[mode] = (st, je, nr, ld, [f["alpha"] for f in fits], [f["mu"] for f in fits], traces, e.offsets.copy())
        finally:
            e.close()
    a, b = out["lean"], out["prefetch"]
    for x, y in zip(a[:4], b[:4]):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(a[4] + a[5], b[4] + b[5]):
        np.testing.assert_array_equal(x, y)
    for k, (th, _, gr) in enumerate(a[6]):
        alpha_all, hl, hs, nrej = po.lbfgs_history(th, gr, J)
        p0 = int(a[7][k])
        np.testing.assert_array_equal(a[1][p0:p0 + len(th)], hl)
        assert int(a[2][k]) == int(nrej)
    last = po.lbfgs_history(a[6][K - 1][0], a[6][K - 1][2], J)[0][-1]
    np.testing.assert_allclose(a[4][-1], last, rtol=1e-10)
