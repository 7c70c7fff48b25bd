"""GPU parity tests added in round 2 (VERDICT r1 "next round" #2, #3 and weak #1-#6):

* failure handling END TO END on the GPU: non-PD fits (both PosDefException sites of src/woodbury.jl:202,205), NaN ELBOs,
  the NaN-skipping argmax with NaNs first / middle / everywhere (src/utils.jl:55-72, test/utils.jl:8-12), PosDefException
  from fit_mvnormals, the retry loop of src/singlepath.jl:259-283;
* every BASELINE config at its stated size on one GPU against the oracle: C2 exactly, C3 with K = 64 device-made traces
  (first 20 fits of every path + PSIS + resample indices on the pooled log ratios), the single-GPU share of C5 at
  d = 10^4, J = 10, N_e = 2000 (multi-batch streaming of V_h);
* value-level tests of resample() (stored draws, fresh candidates, without replacement beyond 4096) and of the lazy result
  handles' staleness guard.
All calls go through the C ABI of libpfmi.so.
"""
import os
import warnings

import numpy as np
import pytest

from helpers import fit_seeds, make_traces, oracle_target
from oracle import pf_oracle as po
import margins as mg

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


def _wc(F, tol=1e-4):
    k = F.k
    if k == 0:
        return True
    dg = np.abs(np.diag(F.QR[:k, :k]))
    return dg.max() > 0 and dg.min() / dg.max() > tol


def _factor(th, gr, alpha_all, hl, hs, l, d):
    j = int(hl[l])
    S = np.stack([th[s + 1] - th[s] for s in hs[l, :j]], axis=1) if j else np.zeros((d, 0))
    Y = np.stack([gr[s] - gr[s + 1] for s in hs[l, :j]], axis=1) if j else np.zeros((d, 0))
    B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
    return po.Factor(alpha_all[l], B, D)


# ---- the generator -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kern", ["mfma", "lane"])
def test_device_normals_bit_identical_to_oracle(pfmi_mod, eng, kern):
    """The table inverse-CDF generator uses only exactly rounded IEEE operations, so the device reproduces the oracle's normals
    BIT FOR BIT.  With theta = grad = 0 the first fit is N(0, I) (alpha = 1, no history), so x = 0 + 1 * u: the draws ARE the
    normals.  1.5 x 10^7 normals contain ~29 words below 2^12, i.e. the refinement path (second Philox call, full table in
    global memory) is compared too; the single-pass scan's normals enter through sum u^2 (logq) below."""
    d, N = 50, 300_000
    eng.set_target(pfmi_mod.t_iso(d))
    eng.set_traces([np.zeros((2, d))], [np.zeros((2, d))])
    eng.fit_batch(6)
    old = os.environ.get("PFMI_ELBO_KERNEL")
    os.environ["PFMI_ELBO_KERNEL"] = kern
    try:
        X, lp, lq = eng.draws(0, 0xC0FFEE123456789, N)
    finally:
        os.environ.pop("PFMI_ELBO_KERNEL", None)
        if old is not None:
            os.environ["PFMI_ELBO_KERNEL"] = old
    U = po.randn_fill(0xC0FFEE123456789, d, N)
    np.testing.assert_array_equal(X, U)
    assert np.abs(U).max() > 5.0
    # scan kernel (no draws written): logq = -(d log 2pi + logdet + |u|^2) / 2 per draw from ITS normals, N >= 64 -> qf kernel
    eng.set_traces([np.zeros((3, d))], [np.zeros((3, d))])
    eng.fit_batch(6)
    seeds = np.array([0, 11, 0xC0FFEE123456789], dtype=np.uint64)
    eng.elbo_batch(N, seeds)
    _, lq2 = eng.elbo_logs(2, N)
    ref = -(d * np.log(2 * np.pi) + np.sum(U * U, axis=0)) / 2
    assert np.max(np.abs(lq2 - ref)) <= 1e-12 * np.abs(ref).max()


# ---- failure handling on the GPU ---------------------------------------------------------------------------
def test_failed_fits_nan_elbos_and_skipnan_argmax_on_gpu(pfmi_mod, eng):
    """The chain non-PD fit -> per-fit status -> NaN ELBO -> NaN-skipping argmax, executed on the GPU and compared with the
    oracle.  A BFGS-accepted trace is PD in exact arithmetic, so the failures are injected:
      path 0  random (theta, grad) walk with the curvature threshold eps = -1e300 (the reference's `ϵ` keyword,
              src/inverse_hessian.jl:25): negative-curvature pairs are accepted -> alpha < 0 (A not PD, :202) and
              indefinite C = I + R D R' (:205);
      path 1  NaN gradient at the FIRST fitted point  -> NaN mean -> NaN ELBO first      (test/utils.jl:10)
      path 2  NaN gradient in the MIDDLE of the trace -> NaN ELBO in the middle           (test/utils.jl:9)
      path 3  NaN gradient everywhere                 -> all NaN -> (NaN, 1)              (test/utils.jl:11)
      path 4  clean trace."""
    d, J, eps = 8, 6, -1e300
    tg = pfmi_mod.t_diag(d, seed=3)
    otg = oracle_target(tg)
    rng = np.random.default_rng(0)
    bad_th = np.cumsum(rng.normal(size=(9, d)), 0)
    bad_gr = rng.normal(size=(9, d))
    good = make_traces(tg, 4, 3)
    ths = [bad_th] + [t.points.copy() for t in good]
    grs = [bad_gr] + [t.gradients.copy() for t in good]
    grs[1][1, 2] = np.nan
    grs[2][len(grs[2]) // 2, 0] = np.nan
    grs[3][:, 1] = np.nan
    eng.set_target(tg)
    eng.set_traces(ths, grs)
    eng.fit_batch(J, eps)
    status, jeff, logdet, nrej = eng.fit_status()
    refs = []
    for k in range(5):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        ref = po.path_fit_elbo(ths[k], grs[k], J, otg, 0, np.zeros(p1 - p0, dtype=np.uint64), eps=eps)
        np.testing.assert_array_equal(status[p0:p1], ref["status"])
        np.testing.assert_array_equal(jeff[p0:p1], ref["j_eff"])
        assert nrej[k] == ref["n_rejected"]
        assert np.all(np.isnan(logdet[p0:p1][ref["status"] != 0]))
    p0 = int(eng.offsets[0])
    st0 = status[p0:int(eng.offsets[1])]
    assert set(st0.tolist()) >= {0, 1, 2}, st0                  # both PosDefException sites really fired on the GPU
    assert np.all(status[int(eng.offsets[1]):] == 0)            # a NaN gradient is rejected by the curvature test: the factor stays PD
    seeds = fit_seeds(eng.P, 2)
    for N, kern in ((32, None), (200, None), (200, "lane"), (200, "mfma")):      # two-pass (N < 64), single-pass scan, the others
        old = os.environ.get("PFMI_ELBO_KERNEL")
        if kern:
            os.environ["PFMI_ELBO_KERNEL"] = kern
        try:
            elbo, se, best = eng.elbo_batch(N, seeds)
        finally:
            if kern:
                os.environ.pop("PFMI_ELBO_KERNEL", None)
                if old is not None:
                    os.environ["PFMI_ELBO_KERNEL"] = old
        for k in range(5):
            p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
            ref = po.path_fit_elbo(ths[k], grs[k], J, otg, N, seeds[p0:p1], eps=eps)
            np.testing.assert_array_equal(np.isnan(elbo[p0:p1]), np.isnan(ref["elbo"]))
            fin = np.isfinite(ref["elbo"])
            if k != 0:                                          # path 0 is indefinite garbage: only the NaN pattern / argmax logic
                assert np.all(np.abs(elbo[p0:p1][fin] - ref["elbo"][fin]) <= 1e-9 * (1 + np.abs(ref["elbo"][fin])))
                assert best[k] == ref["best_iter"], (N, kern, k)
            else:                                               # argmax of the GPU's own values with the reference's rule
                assert best[k] == po.findmax_skipnan(elbo[p0 + 1:p1])[1]
            failed = np.flatnonzero(status[p0:p1] != 0)
            assert np.all(np.isnan(elbo[p0:p1][failed])) and np.all(np.isnan(se[p0:p1][failed]))
            for l in failed[:2]:
                lp, lq = eng.elbo_logs(p0 + int(l), N)
                assert np.all(np.isnan(lp)) and np.all(np.isnan(lq))
        # the three NaN placements of test/utils.jl:8-12
        p1_ = int(eng.offsets[1])
        assert np.isnan(elbo[p1_ + 1]) and best[1] > 1                       # NaN first: a later finite value wins
        mid = len(grs[2]) // 2
        p2_ = int(eng.offsets[2])
        assert np.isnan(elbo[p2_ + mid]) and best[2] != mid and np.isfinite(elbo[p2_ + best[2]])
        p3_ = int(eng.offsets[3])
        assert np.all(np.isnan(elbo[p3_:int(eng.offsets[4])])) and best[3] == 1   # all NaN -> (NaN, 1)
    # WoodburyPDMat's constructor throws (src/woodbury.jl:202,205): fit_mvnormals mirrors it
    with pytest.raises(pfmi_mod.PosDefException):
        pfmi_mod.fit_mvnormals(bad_th, bad_gr, history_length=J, engine=eng, eps=eps)
    dists, nrej1 = pfmi_mod.fit_mvnormals(good[0].points, good[0].gradients, history_length=J, engine=eng)
    assert len(dists) == len(good[0]) and nrej1 == 0


def test_retry_loop_resamples_the_initial_point(pfmi_mod):
    """src/singlepath.jl:259-283: a failed run is retried from a freshly sampled point, up to ntries; num_tries is reported.
    The target is NaN exactly at the supplied init, so try 1 has no iterations (L = 0 -> failure, :299) and try 2 starts
    from init_sampler(rng)."""
    d = 6
    base = pfmi_mod.t_diag(d, seed=5)
    init = np.full(d, 0.25)

    def logp(x):
        return float("nan") if np.array_equal(x, init) else float(base.logp(x))

    tgt = pfmi_mod.CallbackTarget(d, logp, grad=lambda x: base.grad(x))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = pfmi_mod.pathfinder(tgt, init=init, ndraws=20, ndraws_elbo=50, rng=pfmi_mod.HostRNG(4), ntries=5)
        assert res.success and res.num_tries == 2 and len(res.optim_trace) > 2
        assert not np.array_equal(res.optim_trace.points[0], init)
        with pytest.warns(UserWarning, match="Pathfinder failed after 1 tries"):
            r1 = pfmi_mod.pathfinder(tgt, init=init, ndraws=20, ndraws_elbo=50, rng=pfmi_mod.HostRNG(4), ntries=1)
    assert not r1.success and r1.num_tries == 1 and r1.fit_iteration == 0
    assert r1.draws.shape == (d, 20)                              # src/singlepath.jl:231-233: draws from fit_distributions[1]


# ---- BASELINE configs at their stated size -------------------------------------------------------------------
def _pool_stage_vs_oracle(eng, K, N_r, ndraws, seeds, best, cfg="pool"):
    """pool_build -> PSIS -> resample on the GPU; PSIS and the index draw re-run by the oracle on the SAME pooled log ratios."""
    pts = [int(eng.offsets[k]) + int(best[k]) for k in range(K)]
    eng.pool_build(N_r, pts, seeds[pts])
    _, lr = eng.pool_get(draws=False)
    res = eng.psis(lr)
    lw, w, khat, M = po.psis(lr)
    assert res["tail_length"] == M == min(-(-len(lr) // 5), int(np.ceil(3 * np.sqrt(len(lr)))))
    if np.isfinite(khat):
        mg.check(cfg, "pareto_k", abs(res["pareto_shape"] - khat))     # |dk| <= 1e-8 absolute
    mg.check(cfg, "psis_logw", np.max(np.abs(res["log_weights"] - lw)) / (1 + np.abs(lw).max()))
    idx = eng.resample_indices(len(lr), ndraws, seed=20260928)
    np.testing.assert_array_equal(idx, po.sample_weighted(res["weights"], ndraws, seed=20260928))
    draws = eng.pool_gather(idx)
    for t in (0, ndraws // 2, ndraws - 1):                       # draws = draws_all[:, inds], ids = cld(inds, N_r)
        k, n = divmod(int(idx[t]), N_r)
        X, _, _ = eng.draws(pts[k], seeds[pts[k]], 1, n0=n)
        np.testing.assert_array_equal(draws[:, t], X[:, 0])
    return res, idx


def test_config2_exact_size_vs_oracle(pfmi_mod, eng):
    """BASELINE config 2 exactly: multipathfinder npaths = 8, d = 100 diagonal Gaussian, ndraws_elbo = 1000, history 6,
    device-made traces -- EVERY fit of every path against the oracle, then the pooled stage."""
    K, d, J, N = 8, 100, 6, 1000
    tg = pfmi_mod.t_diag(d, seed=1)
    otg = oracle_target(tg)
    eng.set_target(tg)
    run_seeds = pfmi_mod.hostrng.rand_u64(20260928, np.arange(K, dtype=np.uint64), 9)
    x0 = np.stack([pfmi_mod.HostRNG(int(s)).rand(d) * 4 - 2 for s in run_seeds])
    npts = eng.optimize_batch(x0, J)
    eng.fit_batch(J)
    status, jeff, logdet, nrej = eng.fit_status()
    seeds = fit_seeds(eng.P, 21)
    elbo, se, best = eng.elbo_batch(N, seeds)
    th = np.concatenate([eng.get_trace(k, logp=False)[0] for k in range(K)])
    gr = np.concatenate([eng.get_trace(k, logp=False)[2] for k in range(K)])
    ref = po.multipath_fit_elbo(eng.offsets, th, gr, J, otg, N, seeds, nthreads=min(K, os.cpu_count() or 1))
    np.testing.assert_array_equal(status, ref["status"])
    np.testing.assert_array_equal(jeff, ref["j_eff"])
    np.testing.assert_array_equal(nrej, ref["n_rejected"])
    assert jeff.max() == J and eng.P - K > 200
    n_strict = 0
    for k in range(K):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        alpha_all, hl, hs, _ = po.lbfgs_history(th[p0:p1], gr[p0:p1], J)
        wc = np.array([_wc(_factor(th[p0:p1], gr[p0:p1], alpha_all, hl, hs, l, d)) for l in range(1, p1 - p0)])
        x, y = elbo[p0 + 1:p1], ref["elbo"][p0 + 1:p1]
        assert np.all(np.isfinite(x)) and np.all(np.isfinite(y))
        mg.check("C2", "logdet", mg.rel(logdet[p0:p1], ref["logdet"][p0:p1]))
        mg.record("C2", "logdet_abs", np.abs(logdet[p0:p1] - ref["logdet"][p0:p1]), np.inf)
        mg.check("C2", "elbo", mg.rel(x[wc], y[wc]))
        mg.check("C2", "se", mg.rel(se[p0 + 1:p1][wc], ref["se"][p0 + 1:p1][wc]))
        loose = ~wc
        assert np.all(np.abs(x[loose] - y[loose]) <= 8 * np.maximum(se[p0 + 1:p1][loose], ref["se"][p0 + 1:p1][loose]) + 1e-9)
        n_strict += int(wc.sum())
        if np.all(wc):
            top = np.sort(y)[-2:]
            if top[1] - top[0] > 1e-8 * (1 + abs(top[1])):
                assert best[k] == ref["best_iter"][k]
    assert n_strict >= (eng.P - K) // 2, (n_strict, eng.P)
    res, idx = _pool_stage_vs_oracle(eng, K, 1000, 1000, seeds, best, "C2")
    assert np.isfinite(res["pareto_shape"])


@pytest.mark.timeout(900)
def test_config3_exact_size_vs_oracle(pfmi_mod, eng):
    """BASELINE config 3 (the headline): npaths = 64, d = 1000 low-rank + diagonal Gaussian, history 6, ndraws_elbo = 1000,
    traces made on the device -- the first 20 fits of EVERY path against the oracle (OpenMP over paths on the host cores),
    then PSIS k-hat, smoothed weights and resample indices re-computed by the oracle on the pooled log ratios."""
    K, d, J, N, NF = 64, 1000, 6, 1000, 20
    tg = pfmi_mod.t_lowrank(d, r=8, seed=2)
    otg = oracle_target(tg)
    eng.set_target(tg)
    run_seeds = pfmi_mod.hostrng.rand_u64(20260928, np.arange(K, dtype=np.uint64), 9)
    x0 = np.stack([pfmi_mod.HostRNG(int(s)).rand(d) * 4 - 2 for s in run_seeds])
    npts = eng.optimize_batch(x0, J)
    assert npts.min() > NF + 1
    eng.fit_batch(J)
    status, jeff, logdet, nrej = eng.fit_status()
    assert np.all(status == 0)
    seeds = np.concatenate([pfmi_mod.hostrng.rand_u64(int(run_seeds[k]), np.arange(n, dtype=np.uint64), 10)
                            for k, n in enumerate(npts)])
    elbo, se, best = eng.elbo_batch(N, seeds)
    # oracle on the truncated traces (history and fits of point l only depend on points <= l)
    ths, grs, sds = [], [], []
    for k in range(K):
        t, _, g = eng.get_trace(k, logp=False)
        ths.append(t[:NF + 1]); grs.append(g[:NF + 1])
        sds.append(seeds[int(eng.offsets[k]):int(eng.offsets[k]) + NF + 1])
    off = np.arange(K + 1, dtype=np.int64) * (NF + 1)
    ref = po.multipath_fit_elbo(off, np.concatenate(ths), np.concatenate(grs), J, otg, N, np.concatenate(sds),
                                nthreads=min(K, os.cpu_count() or 1))
    n_strict = 0
    for k in range(K):
        p0 = int(eng.offsets[k])
        sl = slice(p0, p0 + NF + 1)
        rs = slice(k * (NF + 1), (k + 1) * (NF + 1))
        np.testing.assert_array_equal(jeff[sl], ref["j_eff"][rs])
        np.testing.assert_array_equal(status[sl], ref["status"][rs])
        mg.check("C3:first20", "logdet", mg.rel(logdet[sl], ref["logdet"][rs]))
        mg.record("C3:first20", "logdet_abs", np.abs(logdet[sl] - ref["logdet"][rs]), np.inf)
        x, y = elbo[sl][1:], ref["elbo"][rs][1:]
        alpha_all, hl, hs, _ = po.lbfgs_history(ths[k], grs[k], J)
        wc = np.array([_wc(_factor(ths[k], grs[k], alpha_all, hl, hs, l, d)) for l in range(1, NF + 1)])
        mg.check("C3:first20", "elbo", mg.rel(x[wc], y[wc]), ctx=k)
        mg.check("C3:first20", "se", mg.rel(se[sl][1:][wc], ref["se"][rs][1:][wc]))
        lo = ~wc
        assert np.all(np.abs(x[lo] - y[lo]) <= 8 * np.maximum(se[sl][1:][lo], ref["se"][rs][1:][lo]) + 1e-9 * (1 + np.abs(y[lo])))
        n_strict += int(wc.sum())
    assert n_strict >= K * NF * 3 // 4, n_strict
    res, idx = _pool_stage_vs_oracle(eng, K, 1000, 1000, seeds, best, "C3")
    # the headline workload's Pareto k-hat is what the ORACLE's PSIS gives on the same pool
    assert np.isfinite(res["pareto_shape"])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("tname,maxit", [("funnel", 12), ("diag", 10)])
def test_config5_share_exact_shape_vs_oracle(pfmi_mod, eng, tname, maxit):
    """The single-GPU share of BASELINE config 5 at its stated shape: d = 10^4, history_length = 10 (KC = 20), ndraws_elbo =
    2000 -- 2 paths.  N_e = 2000 is 16 batches of 128 draws per fit, so V_h (1.6 MB per fit) is re-streamed through LDS for
    every batch (only d = 2500 covered multi-batch streaming before).  The funnel is BASELINE's
    target (its scaled block is numerically rank deficient -> statistical branch for most fits); the diagonal Gaussian at the
    same shape has a well-conditioned QR, so there every fit is compared strictly."""
    d, J, N, K = 10000, 10, 2000, 2
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
    ref = po.multipath_fit_elbo(eng.offsets, th, gr, J, otg, N, seeds, nthreads=K)
    np.testing.assert_array_equal(status, ref["status"])
    np.testing.assert_array_equal(jeff, ref["j_eff"])
    n_strict = n_fits = 0
    for k in range(K):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        alpha_all, hl, hs, _ = po.lbfgs_history(th[p0:p1], gr[p0:p1], J)
        for l in range(1, p1 - p0):
            if ref["status"][p0 + l] != 0:
                assert np.isnan(elbo[p0 + l])
                continue
            n_fits += 1
            a, b = elbo[p0 + l], ref["elbo"][p0 + l]
            cfg = f"C5-shape:{tname}-2x{maxit}"
            mg.check(cfg, "logdet", mg.rel(logdet[p0 + l], ref["logdet"][p0 + l]))
            mg.record(cfg, "logdet_abs", abs(logdet[p0 + l] - ref["logdet"][p0 + l]), np.inf)
            if _wc(_factor(th[p0:p1], gr[p0:p1], alpha_all, hl, hs, l, d)):
                n_strict += 1
                mg.check(cfg, "elbo", mg.rel(a, b), ctx=(k, l, a, b))
                mg.check(cfg, "se", mg.rel(se[p0 + l], ref["se"][p0 + l]))
            else:
                assert abs(a - b) <= 8 * max(se[p0 + l], ref["se"][p0 + l]) + 1e-8 * (1 + abs(b)), (k, l, a, b)
    assert n_fits >= K * (maxit - 2)
    if tname == "diag":
        assert n_strict >= n_fits * 3 // 4 and jeff.max() >= 9, (n_strict, n_fits, jeff.max())
    # winner's per-draw log densities straight from the production scan against the oracle's draws of the same fit
    k = 0
    p0, p1 = int(eng.offsets[0]), int(eng.offsets[1])
    if best[k] == ref["best_iter"][k]:
        refd = po.path_fit_elbo(th[p0:p1], gr[p0:p1], J, otg, N, seeds[p0:p1], want_draws=True)
        lp, lq = eng.elbo_logs(p0 + int(best[k]), N)
        alpha_all, hl, hs, _ = po.lbfgs_history(th[p0:p1], gr[p0:p1], J)
        if _wc(_factor(th[p0:p1], gr[p0:p1], alpha_all, hl, hs, int(best[k]), d)):
            mg.check(f"C5-shape:{tname}-2x{maxit}", "logp@scan", mg.rel(lp, refd["logp"]))
        mg.check(f"C5-shape:{tname}-2x{maxit}", "logq@scan", mg.rel(lq, refd["logq"]))
    # pooled stage at N_r = ndraws = 2000 (config 5's resample size)
    _pool_stage_vs_oracle(eng, K, 2000, 2000, seeds, best, f"C5-shape:{tname}-2x{maxit}")


# ---- resample(): value-level -----------------------------------------------------
def test_resample_modes_value_level_vs_oracle(pfmi_mod):
    """src/resample.jl:20-46, 97-109 (test/resample.jl:111-159): (i) stored draws + stored PSIS reproduce the original
    candidates and weights, (ii) fresh candidates: per-component draws, log ratios, PSIS and the selected columns against the
    oracle, (iii) uniform / without replacement."""
    d, K, N_r = 12, 5, 400
    tg = pfmi_mod.t_lowrank(d, r=3, seed=4)
    otg = oracle_target(tg)
    res = pfmi_mod.multipathfinder(tg, 300, nruns=K, ndraws_elbo=60, ndraws_per_run=N_r, rng=pfmi_mod.HostRNG(17), optimizer="host")
    eng = res.engine
    cand = np.stack([r.draws for r in res.pathfinder_results], axis=2)              # (d, N_r, K) = stack(draws)
    # (i) stored draws: same candidates, same PSIS weights, indices = oracle sampler on those weights
    rng = pfmi_mod.HostRNG(5)
    r1 = pfmi_mod.resample(res, 250, rng=rng)
    np.testing.assert_array_equal(r1.psis_result.weights, res.psis_result.weights)
    sd = int(pfmi_mod.HostRNG(5).rand_u64(1)[0])
    idx = po.sample_weighted(res.psis_result.weights, 250, seed=sd)
    np.testing.assert_array_equal(r1.draws, cand.reshape(d, -1, order="F")[:, idx])
    np.testing.assert_array_equal(r1.draw_component_ids, idx // N_r + 1)
    # (ii) fresh candidates (ndraws_per_run = M): rand(rng, component_k, M) for every component, then PSIS again
    M = 150
    rng = pfmi_mod.HostRNG(6)
    r2 = pfmi_mod.resample(res, 200, rng=rng, ndraws_per_run=M)
    chk = pfmi_mod.HostRNG(6)
    cseeds = chk.rand_u64(K)
    sd = int(chk.rand_u64(1)[0])
    lrs, cands = [], []
    for k, pr in enumerate(res.pathfinder_results):
        tr = pr.optim_trace
        alpha_all, hl, hs, _ = po.lbfgs_history(tr.points, tr.gradients, 6)
        l = pr.fit_iteration
        F = _factor(tr.points, tr.gradients, alpha_all, hl, hs, l, d)
        assert _wc(F)
        mu = F.fit_mean(tr.points[l], tr.gradients[l])
        X, lq = F.rand_and_logpdf(mu, po.randn_fill(int(cseeds[k]), d, M))
        cands.append(X); lrs.append(otg.logp(X) - lq)
    lr = np.concatenate(lrs)
    lw, w, khat, _ = po.psis(lr)
    assert len(r2.psis_result.weights) == K * M
    np.testing.assert_allclose(r2.psis_result.log_weights, lw, rtol=0, atol=1e-8 * (1 + np.abs(lw).max()))
    assert abs(r2.psis_result.pareto_shape - khat) <= 1e-6
    idx2 = po.sample_weighted(r2.psis_result.weights, 200, seed=sd)
    allc = np.concatenate(cands, axis=1)
    assert np.max(np.abs(r2.draws - allc[:, idx2]) / (1 + np.abs(allc[:, idx2]))) <= 1e-10
    np.testing.assert_array_equal(r2.draw_component_ids, idx2 // M + 1)
    # the original per-run draws are still the ORIGINAL ones after the pool was rebuilt (stale handles)
    for k in (0, K - 1):
        np.testing.assert_array_equal(res.pathfinder_results[k].draws, cand[:, :, k])
    # and a second stored-draws resample of the fresh result goes back to the stored candidates (reference :97-101)
    r3 = pfmi_mod.resample(r2, 100, rng=pfmi_mod.HostRNG(5))
    np.testing.assert_array_equal(r3.psis_result.weights, res.psis_result.weights)
    # (iii) importance = false: uniform over the pool, psis_result === nothing; replace = false: unique columns
    r4 = pfmi_mod.resample(res, 120, rng=pfmi_mod.HostRNG(8), importance=False)
    sd = int(pfmi_mod.HostRNG(8).rand_u64(1)[0])
    idx4 = po.sample_uniform(K * N_r, 120, seed=sd)
    assert r4.psis_result is None
    np.testing.assert_array_equal(r4.draws, cand.reshape(d, -1, order="F")[:, idx4])
    r5 = pfmi_mod.resample(res, 300, rng=pfmi_mod.HostRNG(9), replace=False)
    sd = int(pfmi_mod.HostRNG(9).rand_u64(1)[0])
    idx5 = po.sample_weighted_norep(res.psis_result.weights, 300, seed=sd)
    assert len(set(idx5.tolist())) == 300
    np.testing.assert_array_equal(r5.draws, cand.reshape(d, -1, order="F")[:, idx5])
    eng.close()


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


def test_statsbase_direct_index_mode_and_large_norep(pfmi_mod, eng):
    """(a) pfmi_resample_indices_direct == StatsBase.direct_sample! (sequential fp64 running sum, `cw < t` scan) on
    host-drawn uniforms: against the oracle's literal loop and an independent NumPy restatement (np.cumsum is sequential);
    (b) replace = false beyond the 4096-draw LDS path: bit-exact against the oracle up to ndraws = S."""
    import scipy.stats as st
    for S, seed in ((64000, 3), (37, 4), (512000, 5)):
        lr = st.t(4).rvs(S, random_state=np.random.default_rng(seed))
        w = eng.psis(lr)["weights"]
        u = np.random.default_rng(seed).random(3000)
        u[:3] = [0.0, np.nextafter(1.0, 0.0), 0.5]
        idx = eng.resample_indices_direct(S, u)
        np.testing.assert_array_equal(idx, po.sample_direct(w, u))
        cw = np.cumsum(w)
        np.testing.assert_array_equal(idx, np.minimum(np.searchsorted(cw, u, side="left"), S - 1))
    with pytest.raises(pfmi_mod.PfmiError, match="not in"):
        eng.resample_indices_direct(S, np.array([1.0]))
    S = 64000
    lr = st.t(4).rvs(S, random_state=np.random.default_rng(1))
    w = eng.psis(lr)["weights"]
    for nd in (4096, 4097, 20000, S):
        idx = eng.resample_indices(S, nd, replace=False, seed=7)
        assert len(set(idx.tolist())) == nd
        np.testing.assert_array_equal(idx, po.sample_weighted_norep(w, nd, seed=7))
    idx = eng.resample_indices(S, 10000, importance=False, replace=False, seed=8)
    assert len(set(idx.tolist())) == 10000
    w0 = w.copy()
    lr2 = lr.copy(); lr2[100:] = -np.inf                              # only 100 positive weights
    eng.psis(lr2)
    with pytest.raises(pfmi_mod.PfmiError):
        eng.resample_indices(S, 5000, replace=False, seed=1)


# ---- collectives behind the C ABI -------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["init_all", "init_all_local", "init_rank"])
def test_comm_rccl_world1_equals_local_path(pfmi_mod, eng, mode):
    """pfmi_comm_* (csrc/comm_rccl.hip: ncclAllGather of the log-ratio shards, replicated PSIS / index selection, owner gather,
    ncclAllReduce) in the only RCCL world a 1-GPU box allows.  Both ways of forming the group -- ncclCommInitAll (one process,
    G contexts: a single Julia caller) and ncclCommInitRank with a shipped id (one process per GPU) -- must reproduce the
    single-GPU calls bit for bit (result invariance under the GPU count, test/multipath.jl:107-140 extended to G)."""
    tg = pfmi_mod.t_lowrank(50, r=8, seed=2)
    traces = make_traces(tg, 4, 11)
    eng.set_target(tg)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    eng.fit_batch(6)
    seeds = fit_seeds(eng.P, 4)
    elbo, se, best = eng.elbo_batch(64, seeds)
    pts = [int(eng.offsets[k]) + int(best[k]) for k in range(4)]
    eng.pool_build(96, pts, seeds[pts])
    pool, lr = eng.pool_get()
    ref = eng.psis(lr)
    ref_idx = eng.resample_indices(len(lr), 40, seed=9)
    ref_norep = eng.resample_indices(len(lr), 40, replace=False, seed=9)
    if mode == "init_all":                                           # the REAL librccl in the only world a 1-GPU box allows
        os.environ["PFMI_COMM_FORCE_RCCL"] = "1"
        try:
            comm = pfmi_mod.Comm.init_all([eng])
        finally:
            os.environ.pop("PFMI_COMM_FORCE_RCCL", None)
    elif mode == "init_all_local":                                   # round 3: a world of one context does not touch RCCL at all
        comm = pfmi_mod.Comm.init_all([eng])
    else:
        comm = pfmi_mod.Comm.init_rank(eng, 1, 0, pfmi_mod.Comm.unique_id())
    try:
        info = comm.info()
        assert info["world"] == 1 and info["nlocal"] == 1
        assert (info["rccl_version"] == 0) if mode == "init_all_local" else (info["rccl_version"] > 20000)
        r_f, idx_f, draws_f = comm.psis_resample(40, seed=9)         # the fused entry (one synchronisation)
        assert r_f["pareto_shape"] == ref["pareto_shape"] and r_f["tail_length"] == ref["tail_length"]
        np.testing.assert_array_equal(idx_f, ref_idx)
        np.testing.assert_array_equal(draws_f, pool.reshape(tg.d, -1, order="F")[:, ref_idx])
        res = comm.pool_psis()
        assert res["pareto_shape"] == ref["pareto_shape"] and res["tail_length"] == ref["tail_length"]
        idx, draws = comm.resample(40, seed=9)
        np.testing.assert_array_equal(idx, ref_idx)
        np.testing.assert_array_equal(draws, pool.reshape(tg.d, -1, order="F")[:, ref_idx])
        idx2, draws2 = comm.resample(40, replace=False, seed=9)
        np.testing.assert_array_equal(idx2, ref_norep)
        np.testing.assert_array_equal(draws2, pool.reshape(tg.d, -1, order="F")[:, ref_norep])
        u = np.random.default_rng(1).random(25)
        idx3, _ = comm.resample(25, uniforms=u, want_draws=False)
        np.testing.assert_array_equal(idx3, po.sample_weighted(ref["weights"], 25, uniforms=u))
    finally:
        comm.close()
    with pytest.raises(pfmi_mod.PfmiError, match="share GPU"):
        e2 = pfmi_mod.Engine(0)
        try:
            pfmi_mod.Comm.init_all([eng, e2])                          # one rank per GPU
        finally:
            e2.close()


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


# ---- bench.py contract on the GPU box ---------------------------------------------------------------------------
@pytest.mark.parametrize("comm_mode", ["c_abi"])
def test_bench_force_dist_counts_its_ranks(comm_mode):
    """bench.py --gpus 1 --force-dist: the N > 1 code path (pfmi_comm_init_rank: RCCL through the C ABI) in a single-rank world; the JSON
    line reports the ranks counted through the collective.  (Round 6: the torch.distributed fallback data path is gone.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "1",
                        "--npaths", "8", "--dim", "100", "--target", "diag", "--no-cpu-baseline", "--verify-sharding"], capture_output=True,
                       text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["ranks_in_collective"] == 1
    # round 4: --verify-sharding through the torch.distributed world (broadcast of rank 0's single-GPU reference, MIN all-reduce of the
    # per-rank verdicts) -- here the "sharded" run IS a world of one, so it must equal the recomputation bit for bit
    assert line["sharded_equals_single"] is True, line.get("sharded_equals_single_note")
    if comm_mode == "c_abi":
        assert line["rccl_version"] and line["rccl_version"] > 0
    assert line["config"]["collective_backend"].startswith("RCCL" if comm_mode == "c_abi" else "torch.distributed")
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_woodbury_inv_and_scaling_build_host_objects(pfmi_mod):
    """inv(W) and W * c (reference src/woodbury.jl:317-321, 357-360, test/woodbury.jl inv / * testsets): new WoodburyPDMat objects built
    on the host from the downloaded factor -- inv(F) = (U'^-1, Q, V'^-1), (A, B, D) = pdunfactorize -- against dense algebra."""
    tg = pfmi_mod.t_lowrank(40, r=8, seed=5)
    res = pfmi_mod.pathfinder(tg, ndraws=10, rng=pfmi_mod.HostRNG(3), history_length=6, ndraws_elbo=20)
    n_checked = 0
    for dist_ in (res.fit_distributions[2], res.fit_distributions[len(res.fit_distributions) - 1], res.fit_distribution):
        W = dist_.Sigma
        Wd = W.dense()
        Wi = W.inv()
        assert Wi.B.shape == W.B.shape and Wi.D.shape == W.D.shape
        np.testing.assert_allclose(Wi.dense(), np.linalg.inv(Wd), rtol=1e-8, atol=1e-10 * np.abs(np.linalg.inv(Wd)).max())
        np.testing.assert_allclose(Wi.diag(), np.diag(np.linalg.inv(Wd)), rtol=1e-8)
        assert abs(Wi.logdet + W.logdet) < 1e-12 and abs(W.logdet - np.linalg.slogdet(Wd)[1]) < 1e-8 * (1 + abs(W.logdet))
        Q1 = W.thin_Q()
        np.testing.assert_allclose(Q1.T @ Q1, np.eye(Q1.shape[1]), atol=1e-12)
        W3 = W * 3.0
        np.testing.assert_allclose(W3.dense(), 3.0 * Wd, rtol=1e-12, atol=1e-13 * np.abs(Wd).max())
        assert abs(W3.logdet - np.linalg.slogdet(3.0 * Wd)[1]) < 1e-8 * (1 + abs(W3.logdet))
        np.testing.assert_allclose((2.0 * W).dense(), 2.0 * Wd, rtol=1e-12, atol=1e-13 * np.abs(Wd).max())
        np.testing.assert_allclose(W * -1.0, -Wd, rtol=1e-13)            # c <= 0: the dense matrix (src/woodbury.jl:358)
        with pytest.raises(RuntimeError):
            Wi.mul(np.ones(40))
        n_checked += 1
    assert n_checked == 3


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["t4", "ties", "all_equal", "narrow", "with_inf", "big"])
def test_psis_multi_workgroup_equals_single_workgroup_and_oracle(pfmi_mod, case, monkeypatch):
    """S >= 8192 takes the multi-workgroup PSIS (key range, 4096-bin histogram, candidate compaction, sorted tail + GPD fit in one
    workgroup, multi-workgroup normalisation).  Same selection as the one-workgroup kernel (PFMI_PSIS_KERNEL=single) on every input,
    including the ones that overflow the candidate list (heavy ties, all values equal -> the tail kernel selects by itself), and the
    oracle's numbers."""
    rng = np.random.default_rng(11)
    S = 64000
    if case == "t4":
        import scipy.stats as st
        lr = st.t(4).rvs(S, random_state=rng) * 2.0 - 1.0
    elif case == "ties":
        lr = np.round(rng.normal(size=S), 1)                      # ~70 distinct values: thousands of ties at the cutoff
    elif case == "all_equal":
        lr = np.full(S, -3.25)
    elif case == "narrow":
        lr = -1000.0 + 1e-9 * rng.normal(size=S)                  # all keys share their leading 30+ bits
    elif case == "with_inf":
        lr = rng.normal(size=S)
        lr[rng.integers(0, S, 50)] = -np.inf                      # zero-weight draws (logp = -Inf)
    else:
        S = 300000
        lr = rng.standard_t(3, size=S) * 3.0
    eng = pfmi_mod.Engine(0)
    try:
        a = eng.psis(lr)
        monkeypatch.setenv("PFMI_PSIS_KERNEL", "single")
        b = eng.psis(lr)
        monkeypatch.delenv("PFMI_PSIS_KERNEL")
    finally:
        eng.close()
    assert a["tail_length"] == b["tail_length"]
    if np.isnan(b["pareto_shape"]):
        assert np.isnan(a["pareto_shape"])
    else:
        assert abs(a["pareto_shape"] - b["pareto_shape"]) <= 1e-13 * (1 + abs(b["pareto_shape"]))
    fin = np.isfinite(b["log_weights"])
    np.testing.assert_array_equal(np.isfinite(a["log_weights"]), fin)
    assert np.max(np.abs(a["log_weights"][fin] - b["log_weights"][fin])) <= 1e-12 * (1 + np.abs(b["log_weights"][fin]).max())
    np.testing.assert_allclose(a["weights"], b["weights"], rtol=1e-11, atol=1e-300)
    lw, w, k, M = po.psis(lr)
    assert a["tail_length"] == M
    if np.isfinite(k):
        assert abs(a["pareto_shape"] - k) <= 1e-8
    assert np.max(np.abs(a["log_weights"][fin] - lw[fin])) <= 1e-10 * (1 + np.abs(lw[fin]).max())

