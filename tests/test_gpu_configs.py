"""GPU parity, BASELINE configurations at their stated sizes and the end-to-end calls (SURVEY.md 8c / BASELINE.json configs): config 2 exactly,
config 3 (exact size; whole traces), config 5's single-GPU share (exact shape; full ring; ONE GPU'S FULL SHARE of 32 paths x 1000 iterations),
d > 16 384, history_length 17 .. 32, `multipathfinder()` / `pathfinder()` through the public mirror incl. the retry loop and the reference's literal
5 x 5 covariance.  Collected FIRST (tests/conftest.py), so that a `-x` stop still reports them.  Every oracle comparison asserts SURVEY 8(d)'s
tolerance through tests/margins.py and records its margin."""
from concurrent.futures import ThreadPoolExecutor
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import warnings

import numpy as np
import pytest

from helpers import demo_device_target, fit_seeds, make_traces, oracle_factor_from_gpu, oracle_target
from oracle import pf_oracle as po
import margins as mg
from gpu_common import _factor, _pool_stage_vs_oracle, _wc

pytestmark = pytest.mark.gpu


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


# ---- config 5, one GPU's full share --------------------------------------------------------------------------------------------
@pytest.mark.timeout(3000)
def test_config5_full_single_gpu_share(pfmi_mod):
    """reference docs/src/examples/quickstart.md:229-245 (the funnel, init_scale 10), src/optimize.jl:40 (maxiters = 1000): the
    share of BASELINE config 5 that one of 8 GPUs owns, at its stated size."""
    K, d, J, N, maxit, KO = 32, 10000, 10, 2000, 1000, 4
    cfg = "C5-full-share"
    tg = pfmi_mod.t_funnel(d)
    otg = oracle_target(tg)
    eng = pfmi_mod.Engine(0)
    try:
        eng.set_target(tg)
        run_seeds = pfmi_mod.hostrng.rand_u64(20260928, np.arange(K, dtype=np.uint64), 9)
        x0 = np.stack([pfmi_mod.HostRNG(int(s)).rand(d) * 20 - 10 for s in run_seeds])
        npts = eng.optimize_batch(x0, J, maxit)
        P = eng.P
        nfits = P - K
        # the workload is what the config says it is: (nearly) every path runs the full thousand iterations
        assert nfits >= 25000 and npts.max() == maxit + 1, (nfits, npts)
        assert P * d * 2 * J > 2 ** 32                                  # element offsets of the factor block beyond 32 bits
        eng.fit_batch(J)
        status, jeff, logdet, nrej = eng.fit_status()
        seeds = np.concatenate([pfmi_mod.hostrng.rand_u64(int(run_seeds[k]), np.arange(n, dtype=np.uint64), 10)
                                for k, n in enumerate(npts)])
        elbo, se, best = eng.elbo_batch(N, seeds)
        off = eng.offsets
        # ---- (1) full-size properties: every table entry is what its status says --------------------------------------------
        first = np.zeros(P, dtype=bool); first[off[:-1]] = True
        ok = (status == 0) & ~first
        assert ok.sum() >= nfits * 9 // 10, (int(ok.sum()), nfits)
        assert np.all(np.isfinite(logdet[status == 0]))
        failed = (status != 0) & ~first
        assert np.all(np.isnan(elbo[failed]))                           # a failed fit is a NaN ELBO, never a number
        assert np.all(jeff[ok] >= 1) and jeff.max() == J
        assert (jeff == J).sum() >= nfits * 9 // 10                     # the ring is full for ~990 of every 1000 fits
        fin = np.isfinite(elbo)
        assert np.all(se[fin & ok] >= 0)
        # a NaN / -Inf ELBO of a healthy fit is a logp overflow of the funnel (exp(-tau) of a far draw), never a NaN logq
        for k in range(K):
            b = int(best[k])
            assert 0 <= b < npts[k]
            if b > 0:
                v = elbo[off[k] + b]
                seg = elbo[off[k] + 1:off[k + 1]]
                assert not np.isnan(v) and v == np.nanmax(seg), (k, b)
        # ---- (2) the factor at the FAR END of the 51 GB block (element offsets > 2^32): W = R'R, round trips, quadratic forms ---
        rng = np.random.default_rng(4)
        X = rng.normal(size=(d, 6))
        for p in (P - 1, P - 2, int(off[K // 2]) + 500, int(off[1]) - 1):
            if status[p] != 0:
                continue
            j = int(jeff[p])
            f = eng.get_fit(p, j)
            Wx = f["alpha"][:, None] * X + f["B"] @ (f["D"] @ (f["B"].T @ X))       # the dense definition, applied
            sc = np.abs(Wx).max()
            mg.check(cfg, "W@mul_vs_A+BDB'", np.abs(eng.woodbury_apply(p, "mul", X) - Wx).max() / sc, 1e-9, contract=1e-11,
                     why="W x through the factor (R'R x) against (A + B D B') x: two different orders of O(d m) roundings, "
                         "amplified by cond(D)^(1/2) -- a consistency check of the factor, not the dense-W contract")
            Rx = eng.woodbury_apply(p, "rmul", X)
            mg.check(cfg, "W@quad_vs_|Rx|^2", mg.rel(eng.woodbury_apply(p, "quad", X), np.einsum("ij,ij->j", Rx, Rx)), 1e-10)
            back = eng.woodbury_apply(p, "whiten", eng.woodbury_apply(p, "unwhiten", X))
            mg.check(cfg, "draws@unwhiten_whiten_roundtrip", np.abs(back - X).max() / np.abs(X).max(), 1e-8, contract=1e-10,
                     why="round trip through L and L^-1 of a factor whose triangular block has condition ~1e4..1e6 (funnel)")
            assert abs(f["logdet"] - logdet[p]) == 0.0
        # ---- (3) the oracle on KO whole traces: status / j_eff / rejected / logdet of EVERY fit, the mean of sampled ones -----
        paths = [0, K // 3, 2 * K // 3, K - 1][:KO]
        tr = {}
        for k in paths:
            th, _, gr = eng.get_trace(k, logp=False)
            tr[k] = (th, gr)

        def oracle_path(k):
            th, gr = tr[k]
            return po.path_fit_elbo(th, gr, J, otg, 0, np.zeros(len(th), dtype=np.uint64))

        with ThreadPoolExecutor(KO) as ex:
            refs = dict(zip(paths, ex.map(oracle_path, paths)))
        sampled = []                                                    # (k, l) across the whole trace
        n_mu = 0
        for k in paths:
            p0, p1 = int(off[k]), int(off[k + 1])
            L = p1 - p0 - 1
            assert L >= 900, L
            ref = refs[k]
            np.testing.assert_array_equal(status[p0:p1], ref["status"])
            np.testing.assert_array_equal(jeff[p0:p1], ref["j_eff"])
            assert nrej[k] == ref["n_rejected"]
            good = ref["status"] == 0
            mg.check(cfg, "logdet", mg.rel(logdet[p0:p1][good], ref["logdet"][good]))
            mg.record(cfg, "logdet_abs", np.abs(logdet[p0:p1][good] - ref["logdet"][good]), np.inf)
            mid = L // 2
            ls = list(range(1, 21)) + list(range(mid - 10, mid + 10)) + list(range(L - 19, L + 1))
            for l in ls:
                if not good[l]:
                    continue
                sampled.append((k, l))
                mu_gpu = eng.get_fit(p0 + l, int(jeff[p0 + l]))["mu"]
                mu_ref = ref["mu"][l]
                mg.check(cfg, "mu", np.max(np.abs(mu_gpu - mu_ref)) / (1 + np.abs(mu_ref).max()), ctx=(k, l))
                n_mu += 1
        assert n_mu >= KO * 50, n_mu
        # ---- (4) ELBO / SE of fits sampled across the trace + per-draw logs of every winner, oracle in a thread pool ------------
        hist = {k: po.lbfgs_history(tr[k][0], tr[k][1], J) for k in paths}

        def oracle_fit(kl):
            k, l = kl
            th, gr = tr[k]
            alpha_all, hl, hs, _ = hist[k]
            F = _factor(th, gr, alpha_all, hl, hs, l, d)
            mu = F.fit_mean(th[l], gr[l])
            U = po.randn_fill(int(seeds[int(off[k]) + l]), d, N)
            Xd, lq = F.rand_and_logpdf(mu, U)
            lp = otg.logp(Xd)
            v, s, _ = po.elbo_stats(lp, lq)
            return kl, _wc(F), v, s, lp, lq

        some = [kl for i, kl in enumerate(sampled) if i % 5 == 0]                    # 4 per section and path
        winners = [(k, int(best[k])) for k in paths if best[k] > 0]
        with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
            outs = list(ex.map(oracle_fit, some + winners))
        n_strict = n_stat = 0
        for (k, l), wc, v, s, lp, lq in outs:
            p = int(off[k]) + l
            a, sa = elbo[p], se[p]
            if not (np.isfinite(a) and np.isfinite(v)):
                assert (np.isnan(a) and np.isnan(v)) or a == v, (k, l, a, v)
                continue
            if wc:
                n_strict += 1
                mg.check(cfg, "elbo", mg.rel(a, v), ctx=(k, l))
                mg.check(cfg, "se", mg.rel(sa, s))
            else:                                                        # rank-deficient block: x(u) is roundoff-defined
                n_stat += 1
                assert abs(a - v) <= 8 * max(sa, s) + 1e-8 * (1 + abs(v)), (k, l, a, v)
        for (k, l), wc, v, s, lp, lq in outs[len(some):]:              # the winners: per-draw logs of the production scan
            glp, glq = eng.elbo_logs(int(off[k]) + l, N)
            mg.check(cfg, "logq@scan", mg.rel(glq, lq), ctx=(k, l))
            if wc:
                mg.check(cfg, "logp@scan", mg.rel(glp, lp), ctx=(k, l))
            else:
                assert abs(glp.mean() - lp.mean()) <= 8 * (glp.std() + lp.std()) / np.sqrt(N) + 1e-8 * abs(lp.mean())
            # STRICT per-draw check at config 5's own target whatever the conditioning: the oracle's apply on the
            # GPU's OWN factor of this fit -- x(u), logq and logp(x) are functions of exactly these arrays
            fg = eng.get_fit(int(off[k]) + l, int(jeff[int(off[k]) + l]))
            Fg = oracle_factor_from_gpu(fg)
            Xg, lqg = Fg.rand_and_logpdf(fg["mu"], po.randn_fill(int(seeds[int(off[k]) + l]), d, N))
            mg.check(cfg, "logq@scan_vs_oracle_on_gpu_factor", mg.rel(glq, lqg), ctx=(k, l))
            mg.check(cfg, "logp@scan_vs_oracle_on_gpu_factor", mg.rel(glp, otg.logp(Xg)), ctx=(k, l))
            Xd, lpd, lqd = eng.draws(int(off[k]) + l, int(seeds[int(off[k]) + l]), 48, n0=N - 48)
            mg.check(cfg, "draws@writer_vs_oracle_on_gpu_factor", np.abs(Xd - Xg[:, N - 48:]) / (1 + np.abs(Xg[:, N - 48:]).max(axis=0)), ctx=(k, l))
            np.testing.assert_array_equal(lqd, glq[N - 48:])              # the writer's logq IS the scan's (same order of operations)
            # the oracle agrees that this fit beats the sampled ones of its path (best_iter, src/elbo.jl:8)
            for (k2, l2), _, v2, s2, _, _ in outs[:len(some)]:
                if k2 == k and np.isfinite(v2):
                    assert v2 <= elbo[int(off[k]) + l] + 8 * max(s2, se[int(off[k]) + l]) + 1e-8 * (1 + abs(v2)), (k, l, l2)
        print(f"config 5 full share: {nfits} fits ({int((jeff == J).sum())} with a full ring), {n_mu} means, {n_strict} strict + {n_stat} "
              f"statistical ELBOs across {KO} traces, {len(winners)} winners' per-draw logs")
        assert n_strict + n_stat >= KO * 8
        # ---- (5) the pooled stage at config 5's size: winners picked on the device, PSIS / indices against the oracle ----------
        eng.pool_build_best(N)
        pts, wseeds, succ = eng.pool_winners()
        np.testing.assert_array_equal(pts, off[:-1] + best)
        _, lr = eng.pool_get(draws=False)
        assert lr.shape == (K * N,)
        res = eng.psis(lr)
        lw, w, khat, M = po.psis(lr)
        assert res["tail_length"] == M
        if np.isfinite(khat):
            mg.check(cfg, "pareto_k", abs(res["pareto_shape"] - khat))     # |dk| <= 1e-8 absolute
        flw = np.isfinite(lw)
        np.testing.assert_array_equal(np.isfinite(res["log_weights"]), flw)
        mg.check(cfg, "psis_logw", np.max(np.abs(res["log_weights"][flw] - lw[flw])) / (1 + np.abs(lw[flw]).max()))
        idx = eng.resample_indices(len(lr), N, seed=20260928)
        np.testing.assert_array_equal(idx, po.sample_weighted(res["weights"], N, seed=20260928))
        draws = eng.pool_gather(idx)
        assert draws.shape == (d, N) and np.all(np.isfinite(draws))
        for t in (0, N // 2, N - 1):                                    # a gathered column IS draw n of its run's winner
            kk, n = divmod(int(idx[t]), N)
            Xw, _, _ = eng.draws(int(pts[kk]), int(wseeds[kk]), 1, n0=n)
            np.testing.assert_array_equal(draws[:, t], Xw[:, 0])
    finally:
        eng.close()


# ---- properties at BASELINE sizes (config 3: d = 1000, J = 6, N = 1000) ------------------------------------
def test_full_size_properties_config3(pfmi_mod, eng):
    tg = pfmi_mod.t_lowrank(1000, r=8, seed=2)
    traces = make_traces(tg, 4, 20260928)
    eng.set_target(tg)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    eng.fit_batch(6)
    status, jeff, logdet, nrej = eng.fit_status()
    assert np.all(status == 0) and jeff.max() == 6
    N = 1000
    seeds = fit_seeds(eng.P, 77)
    elbo, se, best = eng.elbo_batch(N, seeds)
    for k in range(4):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        assert np.all(np.isfinite(elbo[p0 + 1:p1]))
        p = p0 + int(best[k])
        X, lp, lq = eng.draws(p, seeds[p], N)
        # unwhiten -> whiten round trip: logpdf(x) recomputed through L \ (x - mu) equals logq from |u|^2
        assert np.max(np.abs(eng.logpdf(p, X) - lq)) <= 1e-8 * (1 + np.abs(lq).max())
        # target evaluation against the NumPy formula
        np.testing.assert_allclose(lp, tg.logp(X), rtol=1e-9, atol=1e-7)
        # the Gaussian target is fitted essentially exactly at convergence: closed-form ELBO of the last fit
        # = -1/2 [tr(P Sigma) + (mu-m)'P(mu-m)] + 1/2 logdet(2 pi e Sigma)  
        pl = p1 - 1
        f = eng.get_fit(pl, int(jeff[pl]))
        Sig = np.diag(f["alpha"]) + f["B"] @ f["D"] @ f["B"].T
        Pm = np.diag(tg.a) - tg.Wd @ (tg.G.T @ tg.G) @ tg.Wd.T
        e = f["mu"] - tg.mean
        closed = -0.5 * (np.sum(Pm * Sig) + e @ Pm @ e) + 0.5 * (f["logdet"] + 1000 * (1 + np.log(2 * np.pi)))
        assert abs(elbo[pl] - closed) <= 5 * se[pl] + 1e-6
        # oracle parity on one full-size fit (same Philox normals)
        ref = po.path_fit_elbo(traces[k].points[:12], traces[k].gradients[:12], 6, oracle_target(tg), N, seeds[p0:p0 + 12])
        x, y = elbo[p0 + 1:p0 + 12], ref["elbo"][1:]
        assert np.all(np.abs(x - y) <= 1e-9 * (1 + np.abs(y)))


# ---- end to end -----------------------------------------------------------------------------------------
def test_multipathfinder_end_to_end(pfmi_mod):
    """reference test/multipath.jl:12-85: d = 10 correlated normal, 20 runs; mean / covariance of the draws within
    Monte Carlo tolerance; reseeding reproduces draws and component ids (:63-69)."""
    rng0 = np.random.default_rng(3)
    d, nruns, ndraws = 10, 20, 20000
    A = rng0.normal(size=(d, d)); Sigma = A @ A.T / d + np.eye(d) * 0.3
    mean = rng0.normal(size=d)
    w, Vv = np.linalg.eigh(Sigma)
    tgt = pfmi_mod.CallbackTarget(d, lambda x: float(-0.5 * (x - mean) @ np.linalg.solve(Sigma, x - mean)),
                                  grad=lambda x: -np.linalg.solve(Sigma, x - mean))
    res = pfmi_mod.multipathfinder(tgt, ndraws, nruns=nruns, ndraws_elbo=25, ndraws_per_run=2000, rng=pfmi_mod.HostRNG(42))
    assert res.draws.shape == (d, ndraws) and res.draw_component_ids.shape == (ndraws,)
    assert res.draw_component_ids.min() >= 1 and res.draw_component_ids.max() <= nruns
    assert len(res.pathfinder_results) == nruns and abs(res.psis_result.weights.sum() - 1) < 1e-10
    tol = 15 / np.sqrt(ndraws)
    assert np.all(np.abs(res.draws.mean(1) - mean) < tol * np.sqrt(np.diag(Sigma)))
    C = np.cov(res.draws)
    assert np.max(np.abs(C - Sigma)) < tol * np.max(np.diag(Sigma)) * 1.5
    res2 = pfmi_mod.multipathfinder(tgt, ndraws, nruns=nruns, ndraws_elbo=25, ndraws_per_run=2000, rng=pfmi_mod.HostRNG(42))
    np.testing.assert_array_equal(res.draws, res2.draws)
    np.testing.assert_array_equal(res.draw_component_ids, res2.draw_component_ids)
    # every drawn column is a pool column of the component it claims (reference test/resample.jl:51-59)
    for t in range(0, ndraws, 997):
        k = res.draw_component_ids[t] - 1
        assert np.any(np.all(res.pathfinder_results[k].draws == res.draws[:, [t]], axis=0))
    # resample(): stored draws, fresh candidates, no importance, without replacement (src/resample.jl:20-46)
    r3 = pfmi_mod.resample(res, 500)
    assert r3.draws.shape == (d, 500)
    r4 = pfmi_mod.resample(res, 300, ndraws_per_run=100, rng=pfmi_mod.HostRNG(1))
    assert r4.draws.shape == (d, 300) and len(r4.psis_result.weights) == 100 * nruns
    assert "Multi-path Pathfinder result" in str(res) and f"runs: {nruns}" in str(res) and "Pareto shape diagnostic" in str(res)
    r5 = pfmi_mod.resample(res, 50, importance=False, replace=False)
    assert r5.psis_result is None and len({tuple(c) for c in r5.draws.T}) == 50


def test_pathfinder_single_path_plumbing(pfmi_mod):
    """BASELINE config 1 (reference test/singlepath.jl:13-66): d = 10 iso normal, history 6, ndraws = 100."""
    tg = pfmi_mod.t_iso(10)
    init = np.random.default_rng(0).normal(size=10)
    res = pfmi_mod.pathfinder(tg, init=init, ndraws=100, rng=pfmi_mod.HostRNG(42))
    assert res.success and res.draws.shape == (10, 100)
    np.testing.assert_allclose(res.fit_distribution.mu, 0, atol=1e-6)
    np.testing.assert_allclose(res.fit_distribution.Sigma.dense(), np.eye(10), atol=1e-6)
    assert len(res.fit_distributions) == len(res.optim_trace)
    vals = [e.value for e in res.elbo_estimates]
    assert res.fit_iteration == int(np.nanargmax(vals)) + 1
    np.testing.assert_array_equal(res.draws[:, :5], res.elbo_estimates[res.fit_iteration - 1].draws)  # ELBO draws reused
    res2 = pfmi_mod.pathfinder(tg, init=init, ndraws=100, rng=pfmi_mod.HostRNG(42))
    np.testing.assert_array_equal(res.draws, res2.draws)
    assert [e.value for e in res2.elbo_estimates] == vals
    assert pfmi_mod.pathfinder(tg, init=init, ndraws=2).draws.shape == (10, 2)
    txt = str(res)                                                       # Base.show (src/singlepath.jl:72-83)
    assert txt.startswith("Single-path Pathfinder result") and f"fit iteration: {res.fit_iteration} (total: {len(res.optim_trace) - 1})" in txt
    with pytest.raises(ValueError):
        pfmi_mod.pathfinder(pfmi_mod.CallbackTarget(0, lambda x: 0.0))


def test_multipathfinder_device_and_host_optimizers_agree(pfmi_mod):
    """same target, same rng: the device-optimised run and the host-optimised run find the same optimum / ELBO level
    and both recover the target moments (reference test/multipath.jl:12-85 tolerances)."""
    tg = pfmi_mod.t_diag(10, 1)
    out = {}
    for opt in ("device", "host"):
        res = pfmi_mod.multipathfinder(tg, 4000, nruns=8, ndraws_elbo=100, ndraws_per_run=1000, rng=pfmi_mod.HostRNG(9), optimizer=opt)
        best = [max(e.value for e in r.elbo_estimates) for r in res.pathfinder_results]
        out[opt] = (res, np.array(best))
        assert all(r.success for r in res.pathfinder_results)
        tr = res.pathfinder_results[0].optim_trace
        assert len(tr) == len(res.pathfinder_results[0].fit_distributions) and tr.points.shape == (len(tr), 10)
        assert res.psis_result.pareto_shape < 0.7
    np.testing.assert_allclose(out["device"][1], out["host"][1], atol=0.5)
    sd = np.sqrt(1 / tg.a)
    for res, _ in out.values():
        assert np.all(np.abs(res.draws.mean(1) - tg.mean) < 0.15 * sd)
        assert np.all(np.abs(res.draws.std(1) / sd - 1) < 0.15)
    r1 = pfmi_mod.multipathfinder(tg, 500, nruns=4, ndraws_elbo=50, rng=pfmi_mod.HostRNG(2))
    r2 = pfmi_mod.multipathfinder(tg, 500, nruns=4, ndraws_elbo=50, rng=pfmi_mod.HostRNG(2))
    np.testing.assert_array_equal(r1.draws, r2.draws)       # device optimiser is deterministic


@pytest.mark.parametrize("optimizer", ["device", "host"])
def test_reference_literal_5x5_covariance_recovered(pfmi_mod, optimizer):
    """reference test/singlepath.jl:67-100 (same matrix as docs/src/examples/quickstart.md:26-33): single-path Pathfinder on
    N(0, Sigma) with the literal 5 x 5 Sigma, history 6, ndraws_elbo = 500: fit_distribution.Sigma ~ Sigma (rtol 0.1 in the
    Frobenius norm, the reference's `isapprox`), reseeding reproduces fit, draws and ELBO values."""
    Sigma = np.array([[2.71, 0.5, 0.19, 0.07, 1.04], [0.5, 1.11, -0.08, -0.17, -0.08], [0.19, -0.08, 0.26, 0.07, -0.7],
                      [0.07, -0.17, 0.07, 0.11, -0.21], [1.04, -0.08, -0.7, -0.21, 8.65]])
    lam, V = np.linalg.eigh(Sigma)
    s2 = 0.5 * lam.min()                                              # Sigma = s2 I + W W'  (built-in Gaussian family, r = 5)
    W = V * np.sqrt(lam - s2)
    tg = pfmi_mod.GaussTarget(np.zeros(5), np.full(5, s2), W)
    np.testing.assert_allclose(np.diag(np.full(5, s2)) + W @ W.T, Sigma, atol=1e-12)
    x0 = pfmi_mod.HostRNG(38).randn(5)
    res = pfmi_mod.pathfinder(tg, init=x0, ndraws_elbo=500, history_length=6, rng=pfmi_mod.HostRNG(38), optimizer=optimizer)
    assert res.success
    S = res.fit_distribution.Sigma.dense()
    assert np.linalg.norm(S - Sigma) <= 0.1 * max(np.linalg.norm(S), np.linalg.norm(Sigma))
    res2 = pfmi_mod.pathfinder(tg, init=x0, ndraws_elbo=500, history_length=6, rng=pfmi_mod.HostRNG(38), optimizer=optimizer)
    np.testing.assert_array_equal(res2.draws, res.draws)
    assert [e.value for e in res2.elbo_estimates] == [e.value for e in res.elbo_estimates]
    np.testing.assert_array_equal(res2.fit_distribution.Sigma.dense(), S)


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


# ---- dimensions beyond the register kernels (VERDICT r3 missing #5: d > 16 384 was refused) ------------------------------------------
def test_dimension_beyond_16384_runs_the_whole_hot_path(pfmi_mod):
    """d = 20 000: the memory-resident history walk (pf_history_mem_kernel), the TSQR fit (round 6: to d = 32 768; until round 5 the column-by-column
    kernel, which the last lines re-run and compare), the streamed ELBO scan and the streaming draw writer -- every stage of the hot path -- against the
    oracle on traces from the host driver (the device L-BFGS stops
    at 16 384 coordinates; pfmi.pathfinder(optimizer="auto") therefore falls back to the host driver there)."""
    from pfmi.optimize import optimize_with_trace
    K, d, J, N = 2, 20000, 5, 128
    cfg = "d=20000"
    tg = pfmi_mod.t_diag(d, 1)
    otg = oracle_target(tg)
    eng = pfmi_mod.Engine(0)
    try:
        eng.set_target(tg)
        x0 = pfmi_mod.HostRNG(11).rand(K * d).reshape(K, d) * 4 - 2
        trs = [optimize_with_trace(tg, x0[k], history_length=J, maxiters=9) for k in range(K)]
        eng.set_traces([t.points for t in trs], [t.gradients for t in trs])
        eng.fit_batch(J)
        status, jeff, logdet, nrej = eng.fit_status()
        seeds = fit_seeds(eng.P, 5)
        elbo, se, best = eng.elbo_batch(N, seeds)
        n_cmp = 0
        for k, tr in enumerate(trs):
            p0 = int(eng.offsets[k])
            alpha_all, hl, hs, nr = po.lbfgs_history(tr.points, tr.gradients, J)
            np.testing.assert_array_equal(jeff[p0:p0 + len(hl)], hl)
            assert int(nrej[k]) == int(nr)
            for l in (1, 2, len(hl) // 2, len(hl) - 1):
                p = p0 + l
                assert status[p] == 0
                f = eng.get_fit(p, int(jeff[p]))
                mg.check(cfg, "alpha (memory-resident walk)", np.max(np.abs(f["alpha"] - alpha_all[l]) / alpha_all[l]), 1e-10)
                F = _factor(tr.points, tr.gradients, alpha_all, hl, hs, l, d)
                assert F.status == 0
                mg.check(cfg, "logdet", abs(F.logdet - logdet[p]) / (1 + abs(F.logdet)), 1e-10)
                mu_o = F.fit_mean(tr.points[l], tr.gradients[l])
                mg.check(cfg, "mu", np.max(np.abs(f["mu"] - mu_o) / (1 + np.abs(mu_o))), 1e-10)
                # the draws of the streaming writer and their log densities on the GPU's own factor, reflector by reflector
                X, lp, lq = eng.draws(p, int(seeds[p]), N)
                Fg = oracle_factor_from_gpu(f)
                U = po.randn_fill(int(seeds[p]), d, N)
                Xo, lqo = Fg.rand_and_logpdf(f["mu"], U)
                mg.check(cfg, "x per draw", np.max(np.abs(X - Xo) / (1 + np.abs(Xo).max(axis=0))), 1e-10)
                mg.check(cfg, "logq per draw", np.max(np.abs(lq - lqo) / (1 + np.abs(lqo))), 1e-9)
                lpo = otg.logp(Xo)
                mg.check(cfg, "logp per draw", np.max(np.abs(lp - lpo) / (1 + np.abs(lpo))), 1e-9)
                # the scan's ELBO of this fit = the mean of exactly these log ratios
                e_o = float(np.mean(lpo - lqo))
                mg.check(cfg, "ELBO (scan) vs oracle draws", abs(elbo[p] - e_o) / (1 + abs(e_o)), 1e-10)
                n_cmp += 1
        assert n_cmp == 8
        # the column-by-column kernel (what took d > 16 384 until round 5) on the same traces: same status / logdet / mean
        lib = pfmi_mod.lib()
        lib.pfmi_debug_set(b"PFMI_FIT_KERNEL", b"mem")
        try:
            eng.fit_batch(J)
            st2, je2, ld2, _ = eng.fit_status()
            np.testing.assert_array_equal(st2, status)
            np.testing.assert_array_equal(je2, jeff)
            ok = status == 0
            mg.check(cfg, "logdet, TSQR vs column-by-column kernel", float(np.max(np.abs(ld2[ok] - logdet[ok]) / (1 + np.abs(logdet[ok])))), 1e-10)
            p = int(eng.offsets[1]) - 1
            f2 = eng.get_fit(p, int(je2[p]))
            eng2mu = f2["mu"]
        finally:
            lib.pfmi_debug_set(b"PFMI_FIT_KERNEL", None)
        eng.fit_batch(J)
        f1 = eng.get_fit(p, int(jeff[p]))
        mg.check(cfg, "mu, TSQR vs column-by-column kernel", float(np.max(np.abs(f1["mu"] - eng2mu) / (1 + np.abs(eng2mu)))), 1e-10)
        # the public mirror picks the host driver at this size instead of failing in the device optimiser
        from pfmi.api import _use_device_optimizer
        assert not _use_device_optimizer(tg, "auto") and _use_device_optimizer(pfmi_mod.t_diag(16384, 1), "auto")
    finally:
        eng.close()


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
