"""GPU parity tests: the HIP path (through the C ABI of libpfmi.so) against the CPU oracle on the same
seeded inputs, against the reference's own fixtures / known answers, and -- at BASELINE sizes --
through size-independent properties.  fp64 tolerances are those of SURVEY.md 8(d).
"""
import json
import os

import numpy as np
import pytest

from helpers import fit_seeds, make_traces, oracle_factor_from_gpu, oracle_target
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


def _targets(pfmi):
    return {
        "iso10": pfmi.t_iso(10),                 # converges in <= 2 iterations (m <= 4 < d): the TINY-history corner, not n < m
        "lr10": pfmi.t_lowrank(10, r=3, seed=3), # d = 10 < 2J = 12 / 16 with >= 20 iterations: the real k = min(d, m) = d path
                                                 # (n < m, reference test/woodbury.jl:21-31), well-conditioned QR
        "diag30": pfmi.t_diag(30, seed=1),
        "lr50": pfmi.t_lowrank(50, r=8, seed=2),
        "lr64r3": pfmi.t_lowrank(64, r=3, seed=5),
        "funnel12": pfmi.t_funnel(12),
    }


CASES = [("iso10", 3, 6), ("diag30", 4, 6), ("lr50", 3, 6), ("lr64r3", 2, 2), ("funnel12", 3, 6), ("diag30", 2, 10),
         ("lr10", 3, 8), ("lr10", 2, 6), ("lr50", 2, 4), ("lr50", 2, 16)]

# minimum number of STRICT (well-conditioned QR => same u -> same x) comparisons a case must reach, so that a gated loop can
# never go vacuous.  iso: y == s makes U'\B rank deficient for every fit and funnel12's scaled block is
# numerically rank deficient too (measured on the oracle: 0 / 6 and 3 / 75 fits pass the gate) -- those two cases are pinned
# through the dense W / logdet / mu and the statistical ELBO branch, and say so here instead of silently skipping.
# iso10 / funnel12: every Householder block is numerically rank deficient (y = s on the iso target), so the strict same-u / ELBO
# branches of the gated loops see no fit there (their margins rows read "0 comparisons"); those cases are covered instead by the
# `*_gpu_factor_*` tests, where the oracle applies the GPU's OWN factor reflector by reflector (strict per-draw parity whatever the
# conditioning), and by the dense W / logdet / mu comparisons above, which have no gate.
MIN_STRICT = {"iso10": 0, "funnel12": 0}


def _qr_ratio(F):
    """min/max |diag R| of the QR of U'\\B: roundoff in the Householder vectors is amplified by 1/ratio."""
    k = F.k
    if k == 0:
        return 1.0
    dg = np.abs(np.diag(F.QR[:k, :k]))
    return float(dg.min() / dg.max()) if dg.max() > 0 else 0.0


def _well_conditioned(F, tol=1e-4):
    """QR of U'\\B has no (numerically) dependent column -> Householder vectors are well defined and
    draw-level parity (same u -> same x) is meaningful; otherwise Q is roundoff-defined (also in LAPACK)
    and only the distribution N(mu, W) is pinned."""
    return _qr_ratio(F) > tol


def _oracle_factor(tr, alpha_all, hl, hs, l, d):
    j = int(hl[l])
    S = np.stack([tr.points[s + 1] - tr.points[s] for s in hs[l, :j]], axis=1) if j else np.zeros((d, 0))
    Y = np.stack([tr.gradients[s] - tr.gradients[s + 1] for s in hs[l, :j]], axis=1) if j else np.zeros((d, 0))
    B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
    return po.Factor(alpha_all[l], B, D)


def _setup(pfmi, eng, name, K, J, seed=11):
    tg = _targets(pfmi)[name]
    maxit = 25 if name.startswith("funnel") else 1000
    scale = 2.0
    traces = make_traces(tg, K, seed, scale=scale, history_length=J, maxiters=maxit)
    eng.set_target(tg)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    eng.fit_batch(J)
    return tg, traces


@pytest.mark.parametrize("name,K,J", CASES)
def test_fit_batch_matches_oracle(pfmi_mod, eng, name, K, J):
    """fit_mvnormals / lbfgs_inverse_hessians / pdfactorize (src/mvnormal.jl:14-21, src/inverse_hessian.jl:25-133,
    src/woodbury.jl:201-207): status, effective history, rejected updates, logdet, mu, and the dense
    W = A + B D B' rebuilt from the GPU factors."""
    tg, traces = _setup(pfmi_mod, eng, name, K, J)
    status, jeff, logdet, nrej = eng.fit_status()
    otg = oracle_target(tg)
    cfg = f"small:{name}"
    n_strict = n_wide = 0
    for k, tr in enumerate(traces):
        p0 = int(eng.offsets[k])
        P = len(tr)
        ref = po.path_fit_elbo(tr.points, tr.gradients, J, otg, 0, np.zeros(P, dtype=np.uint64))
        np.testing.assert_array_equal(status[p0:p0 + P], ref["status"])
        np.testing.assert_array_equal(jeff[p0:p0 + P], ref["j_eff"])
        assert nrej[k] == ref["n_rejected"]
        ok = ref["status"] == 0
        mg.check(cfg, "logdet", mg.rel(logdet[p0:p0 + P][ok], ref["logdet"][ok]))
        mg.record(cfg, "logdet_abs", np.abs(logdet[p0:p0 + P][ok] - ref["logdet"][ok]), np.inf)
        alpha_all, hl, hs, _ = po.lbfgs_history(tr.points, tr.gradients, J)
        for l in sorted(set(list(range(min(P, 9))) + [P // 2, min(P - 1, 2 * J + 3), P - 1])):
            if not ok[l]:
                continue
            f = eng.get_fit(p0 + l, int(jeff[p0 + l]))
            n_wide += int(2 * int(jeff[p0 + l]) > tg.d)
            mu_ref = ref["mu"][l]
            mg.check(cfg, "mu", np.max(np.abs(f["mu"] - mu_ref)) / (1 + np.abs(mu_ref).max()))
            np.testing.assert_allclose(f["alpha"], alpha_all[l], rtol=1e-12)
            j = int(hl[l])
            S = np.stack([tr.points[s + 1] - tr.points[s] for s in hs[l, :j]], axis=1) if j else np.zeros((tg.d, 0))
            Y = np.stack([tr.gradients[s] - tr.gradients[s + 1] for s in hs[l, :j]], axis=1) if j else np.zeros((tg.d, 0))
            B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
            Wref = np.diag(alpha_all[l]) + B @ D @ B.T
            Wgpu = np.diag(f["alpha"]) + f["B"] @ f["D"] @ f["B"].T
            # SURVEY 8(d) as written: max |dW| <= 1e-11 max |W|, no conditioning allowance (until round 4 the
            # deviation was divided by cond(D)^(1/2) before the comparison; the recorded margins never needed it)
            mg.check(cfg, "W", np.max(np.abs(Wgpu - Wref)) / np.abs(Wref).max())
            assert f["B"].shape == (tg.d, 2 * j)                       # size(Σ.B) == (d, 2j), test/singlepath.jl:41
            # the factor itself: R = [V 0;0 I] Q' U,  W = R'R   (src/woodbury.jl:178-187)
            F = po.Factor(alpha_all[l], B, D)
            kk = min(tg.d, 2 * j)
            if kk:
                # (a) self-consistency of the GPU factor: R = [V 0;0 I] Q' U rebuilt from (U, Vh, T, V) gives W = R'R
                Vh = np.tril(f["qr_factors"][:, :kk], -1) + np.eye(tg.d, kk)
                Q = np.eye(tg.d) - Vh @ f["T"] @ Vh.T
                np.testing.assert_allclose(Q.T @ Q, np.eye(tg.d), atol=1e-12)
                blk = np.eye(tg.d); blk[:kk, :kk] = f["V"]
                Rm = blk @ Q.T @ np.diag(np.sqrt(f["alpha"]))
                assert np.max(np.abs(Rm.T @ Rm - Wref)) <= 1e-10 * np.abs(Wref).max() * max(1.0, np.linalg.cond(D) ** 0.5)
                assert abs(f["logdet"] - np.linalg.slogdet(Wref)[1]) <= 1e-8 * (1 + abs(f["logdet"]))
                # (b) reflector-level parity with the oracle (LAPACK convention) whenever the QR is well
                #     conditioned; for rank-deficient B~ (e.g. iso: y == s) later reflectors are roundoff-defined
                if _well_conditioned(F):
                    n_strict += 1
                    amp = 1e-13 / _qr_ratio(F)            # roundoff amplification of the reflectors
                    np.testing.assert_allclose(f["V"], F.V[:kk, :kk], rtol=1e-8, atol=max(1e-9, amp) * np.abs(F.V).max())
                    np.testing.assert_allclose(f["qr_factors"], F.QR[:, :2 * j], rtol=1e-8,
                                               atol=max(1e-9, amp) * np.abs(F.QR).max())
                    z = np.eye(tg.d, order="F").copy(order="F")
                    po.lib().pfo_apply_q(tg.d, kk, po._p(F.QR), po._p(F.tau), 0, po._p(z), tg.d)
                    np.testing.assert_allclose(Q, z, atol=max(1e-10, amp))
    assert n_strict >= MIN_STRICT.get(name, 3 * K), (name, n_strict)
    if name == "lr10":
        assert n_wide >= 3 * K, n_wide                     # fits with 2j > d really ran (k = d, R is d x 2j upper trapezoidal)


@pytest.mark.parametrize("name,K,J", CASES[:8])
def test_draws_and_logq_match_oracle_same_u_and_rng(pfmi_mod, eng, name, K, J):
    """rand_and_logpdf (src/mvnormal.jl:24-39) + target: identical host-supplied u (parity mode) and
    the in-kernel Philox normals (production mode) against the oracle."""
    tg, traces = _setup(pfmi_mod, eng, name, K, J)
    otg = oracle_target(tg)
    status, jeff, logdet, _ = eng.fit_status()
    N = 130
    rng = np.random.default_rng(0)
    n_strict = n_wide = n_loose = 0
    for k, tr in enumerate(traces):
        p0 = int(eng.offsets[k])
        alpha_all, hl, hs, _ = po.lbfgs_history(tr.points, tr.gradients, J)
        for l in sorted({1, min(3, len(tr) - 1), len(tr) // 2, min(len(tr) - 1, 2 * J + 3), len(tr) - 1}):
            if status[p0 + l] != 0:
                continue
            j = int(hl[l])
            S = np.stack([tr.points[s + 1] - tr.points[s] for s in hs[l, :j]], axis=1)
            Y = np.stack([tr.gradients[s] - tr.gradients[s + 1] for s in hs[l, :j]], axis=1)
            B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
            F = po.Factor(alpha_all[l], B, D)
            if not _well_conditioned(F):
                # x(u) of the ORACLE's factor is roundoff-defined here.  Round 4: the draw kernels are still pinned strictly,
                # against the oracle's reflector-by-reflector apply on the GPU's own factor of this fit
                fg = eng.get_fit(p0 + l, j)
                Fg = oracle_factor_from_gpu(fg)
                for mode in ("mem", "rng"):
                    U = rng.normal(size=(tg.d, N)) if mode == "mem" else po.randn_fill(1000 + 17 * l + k, tg.d, N)
                    Xr, lqr = Fg.rand_and_logpdf(fg["mu"], U)
                    X, lp, lq = eng.draws(p0 + l, 1000 + 17 * l + k, N, u=U if mode == "mem" else None)
                    mg.check(f"small:{name}", "draws@gpu_factor_" + mode, np.abs(X - Xr) / (1 + np.abs(Xr).max(axis=0)), ctx=(l, mode))
                    mg.check(f"small:{name}", "logq@gpu_factor_" + mode, mg.rel(lq, lqr))
                    mg.check(f"small:{name}", "logp@gpu_factor_" + mode, mg.rel(lp, otg.logp(Xr)))
                n_loose += 1
                continue
            n_strict += 1
            n_wide += int(2 * j > tg.d)
            mu = F.fit_mean(tr.points[l], tr.gradients[l])
            seed = 1000 + 17 * l + k
            for mode in ("mem", "rng"):
                U = rng.normal(size=(tg.d, N)) if mode == "mem" else po.randn_fill(seed, tg.d, N)
                Xr, lqr = F.rand_and_logpdf(mu, U)
                lpr = otg.logp(Xr)
                X, lp, lq = eng.draws(p0 + l, seed, N, u=U if mode == "mem" else None)
                scale = 1 + np.abs(Xr).max(axis=0)
                mg.check(f"small:{name}", "draws@" + mode, np.abs(X - Xr) / scale, ctx=(l, mode))
                mg.check(f"small:{name}", "logq@" + mode, mg.rel(lq, lqr))
                mg.check(f"small:{name}", "logp@" + mode, mg.rel(lp, lpr))
            # counter-based: draws n0.. are a pure function of (seed, n)
            X2, _, _ = eng.draws(p0 + l, seed, 40, n0=90)
            np.testing.assert_array_equal(X2, X[:, 90:130])
            # Distributions.logpdf through the factor (src/resample.jl:85-89) == logq from u
            lpdf = eng.logpdf(p0 + l, X)
            mg.check(f"small:{name}", "logq@logpdf_vs_logq", mg.rel(lpdf, lq))
            np.testing.assert_allclose(lpdf, F.logpdf(mu, X), rtol=1e-9, atol=1e-9)
    assert n_strict >= MIN_STRICT.get(name, 2 * K), (name, n_strict)
    if name == "lr10":
        assert n_wide >= K, n_wide                         # same-u draw parity on fits with 2j > d


@pytest.mark.parametrize("name,K,J", CASES)
def test_elbo_batch_matches_oracle(pfmi_mod, eng, name, K, J):
    """maximize_elbo (src/elbo.jl:1-20) over every path: ELBO, SE, NaN-skipping argmax."""
    tg, traces = _setup(pfmi_mod, eng, name, K, J)
    otg = oracle_target(tg)
    N = 200
    seeds = fit_seeds(eng.P, 5)
    elbo, se, best = eng.elbo_batch(N, seeds)
    # parity mode with uploaded normals gives the same answers as the in-kernel generator
    U = np.concatenate([po.randn_fill(int(seeds[p]), tg.d, N).T.ravel() for p in range(eng.P)])
    elbo_m, se_m, best_m = eng.elbo_batch(N, seeds, u=U)
    n_strict = 0
    for k, tr in enumerate(traces):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        ref = po.path_fit_elbo(tr.points, tr.gradients, J, otg, N, seeds[p0:p1])
        assert np.isnan(elbo[p0]) and np.isnan(se[p0])
        alpha_all, hl, hs, _ = po.lbfgs_history(tr.points, tr.gradients, J)
        wc = np.array([ref["status"][l] == 0 and _well_conditioned(_oracle_factor(tr, alpha_all, hl, hs, l, tg.d))
                       for l in range(1, p1 - p0)])
        for a, b, sa, sb in ((elbo, ref["elbo"], se, ref["se"]), (elbo_m, ref["elbo"], se_m, ref["se"])):
            x, y = a[p0 + 1:p1], b[1:]
            fin = np.isfinite(y)
            np.testing.assert_array_equal(np.isfinite(x), fin)
            strict = fin & wc
            n_strict += int(strict.sum())
            mg.check(f"small:{name}", "elbo", mg.rel(x[strict], y[strict]))
            mg.check(f"small:{name}", "se", mg.rel(sa[p0 + 1:p1][strict], sb[1:][strict]))
            loose = fin & ~wc     # rank-deficient QR: same distribution, roundoff-defined draws -> statistical agreement
            tol = 8 * np.maximum(sa[p0 + 1:p1][loose], sb[1:][loose]) + 1e-9 * (1 + np.abs(y[loose]))
            assert np.all(np.abs(x[loose] - y[loose]) <= tol)
        vals = ref["elbo"][1:]
        top = np.sort(vals[np.isfinite(vals)])[-2:] if np.sum(np.isfinite(vals)) >= 2 else None
        if np.all(wc) and (top is None or top[1] - top[0] > 1e-8 * (1 + abs(top[1]))):
            assert best[k] == ref["best_iter"] == best_m[k]
        lp, lq = eng.elbo_logs(p0 + int(best[k]), N)
        v, s, _ = po.elbo_stats(lp, lq)
        assert abs(v - elbo[p0 + int(best[k])]) <= 1e-10 * (1 + abs(v))
        # per-draw log densities of the production launch against the oracle's own draws of the same fit
        if wc[int(best[k]) - 1] and best[k] == ref["best_iter"]:
            refd = po.path_fit_elbo(tr.points, tr.gradients, J, otg, N, seeds[p0:p1], want_draws=True)
            mg.check(f"small:{name}", "logp@scan", mg.rel(lp, refd["logp"]))
            mg.check(f"small:{name}", "logq@scan", mg.rel(lq, refd["logq"]))
    assert n_strict >= MIN_STRICT.get(name, 8 * K), (name, n_strict)


def test_reference_fixture_S0Y0_through_gpu(pfmi_mod, eng, golden_dir):
    """reference test/inverse_hessian.jl:19-44 on the GPU: the literal S0/Y0 history (as a trace whose
    steps are the fixture columns) reproduces the explicit dense Byrd formula, incl. ring rotation."""
    g = json.load(open(os.path.join(golden_dir, "lbfgs_S0Y0.json")))
    S = np.array(g["S0_columns"]).T
    Y = np.array(g["Y0_columns"]).T
    n, nh = S.shape
    theta = np.zeros((nh + 1, n)); grad = np.zeros((nh + 1, n))
    for l in range(nh):
        theta[l + 1] = theta[l] + S[:, l]
        grad[l + 1] = grad[l] - Y[:, l]
    for J in (3, 5):
        eng.set_traces([theta], [grad])
        eng.fit_batch(J)
        status, jeff, _, nrej = eng.fit_status()
        assert nrej[0] == 0 and list(jeff) == [min(l, J) for l in range(nh + 1)]
        for l in range(nh + 1):
            f = eng.get_fit(l, int(jeff[l]))
            j = int(jeff[l])
            if j == 0:
                np.testing.assert_allclose(f["alpha"], 1.0)
                continue
            Sl, Yl = S[:, l - j:l], Y[:, l - j:l]
            H0 = np.diag(f["alpha"])
            R = np.triu(Sl.T @ Yl); Rinv = np.linalg.inv(R)
            Bx = np.hstack([H0 @ Yl, Sl])
            Dx = np.block([[np.zeros((j, j)), -Rinv], [-Rinv.T, Rinv.T @ (np.diag(np.diag(R)) + Yl.T @ H0 @ Yl) @ Rinv]])
            Hexp = H0 + Bx @ Dx @ Bx.T
            Hgpu = H0 + f["B"] @ f["D"] @ f["B"].T
            np.testing.assert_allclose(Hgpu, Hexp, rtol=1e-9, atol=1e-10 * np.abs(Hexp).max())


@pytest.mark.parametrize("sigma", [1e-3, 0.05, 0.8, 1.0, 1.1, 1.2, 5.0, 10.0])
def test_analytic_elbo_known_answer_on_gpu(pfmi_mod, eng, sigma):
    """reference test/elbo.jl:7-28 on the GPU: 1-D, ELBO = (1 - r^2)/2 + log r within 3 SE.  A 1-D Normal(0, sigma)
    is obtained as the fit of a one-step trace on the quadratic with curvature 1/sigma^2."""
    sigma_t = 0.08
    tgt = pfmi_mod.GaussTarget(np.zeros(1), np.array([sigma_t**2]), offset=-0.5 * np.log(2 * np.pi) - np.log(sigma_t))
    th0 = 0.3
    theta = np.array([[th0], [0.0]])
    grad = np.array([[-th0 / sigma**2], [0.0]])      # gradient of -x^2/(2 sigma^2): one exact Newton step
    eng.set_target(tgt)
    eng.set_traces([theta], [grad])
    eng.fit_batch(6)
    f = eng.get_fit(1, 1)
    Sig = f["alpha"][0] + (f["B"] @ f["D"] @ f["B"].T)[0, 0]
    assert abs(Sig - sigma**2) < 1e-9 * sigma**2 and abs(f["mu"][0]) < 1e-12
    N = 400_000
    elbo, se, best = eng.elbo_batch(N, np.array([0, 4242], dtype=np.uint64))
    r = sigma / sigma_t
    assert abs(elbo[1] - ((1 - r * r) / 2 + np.log(r))) <= 3 * se[1] + 1e-12
    assert best[0] == 1


def test_isonormal_exact_after_one_iteration_on_gpu(pfmi_mod, eng):
    """reference test/singlepath.jl:13-41 (BASELINE config 1 numerics): mu ~ 0, Sigma ~ I, size(B) = (d, 2)."""
    rng = np.random.default_rng(1)
    for d in (1, 5, 10, 100):
        th0 = rng.normal(size=d)
        eng.set_target(pfmi_mod.t_iso(d))
        eng.set_traces([np.stack([th0, np.zeros(d)])], [np.stack([-th0, np.zeros(d)])])
        eng.fit_batch(6)
        f = eng.get_fit(1, 1)
        assert f["B"].shape == (d, 2)
        np.testing.assert_allclose(f["mu"], 0, atol=1e-6)
        np.testing.assert_allclose(np.diag(f["alpha"]) + f["B"] @ f["D"] @ f["B"].T, np.eye(d), atol=1e-6)
        elbo, se, best = eng.elbo_batch(100, np.array([1, 2], dtype=np.uint64))
        assert best[0] == 1 and abs(elbo[1] - d / 2 * np.log(2 * np.pi)) < 1e-9 * d + 1e-9


def test_callback_target_equals_builtin(pfmi_mod, eng):
    """the host-closure target (reference's general logp, src/elbo.jl:15) gives the built-in target's numbers"""
    tg = pfmi_mod.t_diag(20, seed=3)
    traces = make_traces(tg, 2, 5)
    seeds = None
    out = []
    for target in (tg, pfmi_mod.CallbackTarget(20, lambda x: float(tg.logp(x))),
                   pfmi_mod.CallbackTarget(20, lambda x: float(tg.logp(x)), logp_batch=lambda X: tg.logp(X))):   # vectorised closure
        eng.set_target(target)
        eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
        eng.fit_batch(6)
        seeds = fit_seeds(eng.P, 3)
        out.append(eng.elbo_batch(64, seeds))
        eng.pool_build(70, [int(eng.offsets[k]) + int(out[-1][2][k]) for k in range(2)], [1, 2])
        out[-1] = out[-1] + eng.pool_get()
    for o in out[1:]:
        for a, b in zip(out[0], o):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12, equal_nan=True)


# ---- PSIS / resampling ----------------------------------------------------------------------------------
@pytest.mark.parametrize("S,df", [(1000, 3.0), (64000, 5.0), (200, 1.5), (30, 2.0), (512000, 4.0)])
def test_psis_matches_oracle(pfmi_mod, eng, S, df):
    import scipy.stats as st
    lr = st.t(df).rvs(S, random_state=np.random.default_rng(S)) * 1.5 - 3.0
    res = eng.psis(lr)
    lw, w, k, M = po.psis(lr)
    assert res["tail_length"] == M
    mg.check("psis", "pareto_k", abs(res["pareto_shape"] - k))
    mg.check("psis", "psis_logw", np.max(np.abs(res["log_weights"] - lw)) / (1 + np.abs(lw).max()))
    mg.check("psis", "psis_w", np.max(np.abs(res["weights"] - w) / np.maximum(w, 1e-300)), 1e-9, why="w = exp(log w): a log-weight of "
             "magnitude ~50 carries 50 eps of absolute error, i.e. ~1e-14 relative in w; 1e-9 is the historical bound, see the margin")
    assert abs(res["weights"].sum() - 1) < 1e-12                       # reference test/resample.jl:108


def test_psis_ties_small_and_degenerate(pfmi_mod, eng):
    lr = np.array([0.1, -0.3, 0.5, 0.0, 1.0, -2.0])                    # M < 5: normalise only
    res = eng.psis(lr)
    assert np.isnan(res["pareto_shape"])
    np.testing.assert_allclose(res["log_weights"], lr - np.logaddexp.reduce(lr), rtol=1e-13)
    lwm = np.full((10, 4), -1000.0); lwm[:, 0] = 0.0                   # reference test/resample.jl:36-49
    lr = lwm.T.ravel()
    res = eng.psis(lr)
    _, w, k, M = po.psis(lr)
    np.testing.assert_allclose(res["weights"], w, rtol=1e-12, atol=1e-300)
    idx = eng.resample_indices(40, 20, seed=3)
    assert np.all(idx < 10)                                            # all(==(1), component_ids)
    # heavy ties at the cutoff: (value, index) order must match the oracle's stable sort
    rng = np.random.default_rng(0)
    lr = np.round(rng.normal(size=5000), 1)
    res = eng.psis(lr)
    lw, w, k, M = po.psis(lr)
    mg.check("psis", "pareto_k", abs(res["pareto_shape"] - k))
    mg.check("psis", "psis_logw", np.max(np.abs(res["log_weights"] - lw)) / (1 + np.abs(lw).max()))


def test_resample_indices_bit_exact(pfmi_mod, eng):
    """index selection is bit-exact against the oracle on identical (weights, uniforms) -- SURVEY.md H4"""
    import scipy.stats as st
    S = 64000
    lr = st.t(4).rvs(S, random_state=np.random.default_rng(1))
    res = eng.psis(lr)
    w = res["weights"]
    for nd in (1, 1000, 5000):
        idx = eng.resample_indices(S, nd, seed=99)
        np.testing.assert_array_equal(idx, po.sample_weighted(w, nd, seed=99))
        u = np.random.default_rng(nd).random(nd)
        np.testing.assert_array_equal(eng.resample_indices(S, nd, uniforms=u), po.sample_weighted(w, nd, uniforms=u))
    np.testing.assert_array_equal(eng.resample_indices(S, 300, importance=False, seed=5), po.sample_uniform(S, 300, seed=5))
    # without replacement (reference test/resample.jl:31-34): unique, and equal to the oracle's Efraimidis-Spirakis
    idx = eng.resample_indices(S, 500, replace=False, seed=7)
    assert len(set(idx.tolist())) == 500
    np.testing.assert_array_equal(idx, po.sample_weighted_norep(w, 500, seed=7))
    idx = eng.resample_indices(S, 50, importance=False, replace=False, seed=8)
    assert len(set(idx.tolist())) == 50


def test_pool_log_ratio_ordering_and_gather(pfmi_mod, eng):
    """reference test/resample.jl:62-89: ratios[(k-1)N + n] = logp(x_nk) - logpdf(comp_k, x_nk); and
    draws = draws_all[:, inds] (src/resample.jl:68)"""
    tg, traces = _setup(pfmi_mod, eng, "lr50", 3, 6)
    seeds = fit_seeds(eng.P, 1)
    elbo, se, best = eng.elbo_batch(50, seeds)
    pts = [int(eng.offsets[k]) + int(best[k]) for k in range(3)]
    N_r = 80                                                           # > N_e: top-up draws (src/singlepath.jl:229-230)
    eng.pool_build(N_r, pts, seeds[pts])
    pool, lr = eng.pool_get()
    assert pool.shape == (tg.d, N_r, 3)
    for k in range(3):
        X, lp, lq = eng.draws(pts[k], seeds[pts[k]], N_r)
        np.testing.assert_array_equal(pool[:, :, k], X)
        np.testing.assert_array_equal(lr[k * N_r:(k + 1) * N_r], lp - lq)
        np.testing.assert_allclose(lr[k * N_r:(k + 1) * N_r], tg.logp(X) - eng.logpdf(pts[k], X), rtol=1e-9, atol=1e-9)
        Xe, _, _ = eng.draws(pts[k], seeds[pts[k]], 50)                # the first N_e columns ARE the ELBO draws
        np.testing.assert_array_equal(pool[:, :50, k], Xe)
    idx = np.array([0, 79, 80, 239, 100, 100])
    g = eng.pool_gather(idx)
    np.testing.assert_array_equal(g, pool.reshape(tg.d, -1, order="F")[:, idx])
    g2 = eng.pool_gather(idx + 1000, col_offset=1000)
    np.testing.assert_array_equal(g2, g)
    # the host variant never zero-fills: an index outside this ctx's window is an error
    for bad, off in ((np.array([0, 240]), 0), (np.array([-1]), 0), (idx, 100)):
        with pytest.raises(pfmi_mod.PfmiError, match="outside this pool"):
            eng.pool_gather(bad, col_offset=off)
    # ownership window (multi-GPU, device variant): columns outside [col_offset, col_offset + K*N_r) come back as zeros
    buf = eng.malloc_dev(8 * tg.d * len(idx))
    eng.pool_gather_dev(idx, 100, buf)
    g3 = eng.memcpy_d2h(np.empty((tg.d, len(idx)), order="F"), buf)
    eng.free_dev(buf)
    assert np.all(g3[:, :3] == 0) and np.array_equal(g3[:, 3], pool.reshape(tg.d, -1, order="F")[:, 139])


def test_not_pd_fit_reports_status_and_nan_elbo(pfmi_mod, eng):
    """src/woodbury.jl:202,205: a non-PD fit is a per-fit status + NaN ELBO, never an abort; other paths proceed."""
    d = 8
    tg = pfmi_mod.t_iso(d)
    good = make_traces(tg, 1, 3)[0]
    theta = np.array([np.ones(d), np.zeros(d), -np.ones(d) * 0.5])
    grad = -theta.copy(); grad[2] = grad[1] * 0 + 1e-3 * np.arange(1, d + 1)   # inconsistent curvature on step 2
    eng.set_target(tg)
    eng.set_traces([theta, good.points], [grad, good.gradients])
    eng.fit_batch(6)
    status, jeff, logdet, nrej = eng.fit_status()
    ref = po.path_fit_elbo(theta, grad, 6, oracle_target(tg), 0, np.zeros(3, dtype=np.uint64))
    np.testing.assert_array_equal(status[:3], ref["status"])
    assert nrej[0] == ref["n_rejected"]
    for N in (32, 200):                                    # two-pass kernel (N < 64) and single-pass scan (N >= 64)
        elbo, se, best = eng.elbo_batch(N, fit_seeds(eng.P, 2))
        assert np.all(np.isfinite(elbo[4:]))
        if np.any(status[:3] != 0):
            assert np.all(np.isnan(elbo[:3][status[:3] != 0]))
            lp, lq = eng.elbo_logs(int(np.flatnonzero(status[:3] != 0)[0]), N)
            assert np.all(np.isnan(lp)) and np.all(np.isnan(lq))
    if np.any(status[:3] != 0):
        with pytest.raises(pfmi_mod.PosDefException):
            pfmi_mod.fit_mvnormals(theta, grad, history_length=6, engine=eng)


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


def test_torch_interop_for_the_collective_path(pfmi_mod, eng):
    """the `_dev` entry points of the pooled stage for hosts that keep buffers on the GPU (pfmi_pool_log_ratios_dev, pfmi_psis_dev,
    pfmi_pool_gather_dev), on one GPU: zero-copy torch view of the engine's log-ratio shard (CUDA array interface), PSIS on a
    torch-owned device buffer, owner-gather into a torch tensor."""
    import torch

    class DevArray:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    tg, traces = _setup(pfmi_mod, eng, "lr50", 4, 6)
    seeds = fit_seeds(eng.P, 4)
    elbo, se, best = eng.elbo_batch(64, seeds)
    pts = [int(eng.offsets[k]) + int(best[k]) for k in range(4)]
    eng.pool_build(64, pts, seeds[pts])
    pool, lr = eng.pool_get()
    ptr, cnt = eng.pool_log_ratios_dev()
    shard = torch.as_tensor(DevArray(ptr, cnt), device="cuda:0")
    np.testing.assert_array_equal(shard.cpu().numpy(), lr)
    lr_all = shard.clone()                                            # torch-owned device memory
    out = torch.zeros(tg.d * 32, dtype=torch.float64, device="cuda:0")
    res = eng.psis_dev(lr_all.data_ptr(), lr_all.numel(), want_weights=True)
    idx = eng.resample_indices(lr_all.numel(), 32, seed=9)
    eng.pool_gather_dev(idx, 0, out.data_ptr())
    torch.cuda.synchronize()
    ref = eng.psis(lr)
    np.testing.assert_array_equal(res["weights"], ref["weights"])
    np.testing.assert_array_equal(idx, po.sample_weighted(ref["weights"], 32, seed=9))
    np.testing.assert_array_equal(out.cpu().numpy().reshape(32, tg.d).T, pool.reshape(tg.d, -1, order="F")[:, idx])


@pytest.mark.parametrize("d,maxit", [(2500, 14), (10000, 8)])
def test_large_d_general_paths(pfmi_mod, eng, d, maxit):
    """d beyond the resident-LDS / register kernels (config-5 style: funnel, history_length = 10 -> KC = 20, d = 2500 and the
    full d = 10^4): the streamed single-pass ELBO scan (V_h through LDS in 256-row chunks, head transform across two blocks)
    and the memory-resident fit kernel against the oracle; draw-writing launches take the lane-per-draw kernel."""
    J = 10
    tg = pfmi_mod.t_funnel(d)
    rng = pfmi_mod.HostRNG(5)
    traces = [pfmi_mod.optimize_with_trace(tg, rng.rand(d) * 2 - 1, history_length=J, maxiters=maxit) for _ in range(2)]
    eng.set_target(tg)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    eng.fit_batch(J)
    status, jeff, logdet, nrej = eng.fit_status()
    seeds = fit_seeds(eng.P, 8)
    N = 64
    elbo, se, best = eng.elbo_batch(N, seeds)
    otg = oracle_target(tg)
    for k, tr in enumerate(traces):
        p0, p1 = int(eng.offsets[k]), int(eng.offsets[k + 1])
        ref = po.path_fit_elbo(tr.points, tr.gradients, J, otg, N, seeds[p0:p1])
        np.testing.assert_array_equal(status[p0:p1], ref["status"])
        np.testing.assert_array_equal(jeff[p0:p1], ref["j_eff"])
        ok = ref["status"] == 0
        mg.check("c5-shape:funnel-2x25", "logdet", mg.rel(logdet[p0:p1][ok], ref["logdet"][ok]))
        alpha_all, hl, hs, _ = po.lbfgs_history(tr.points, tr.gradients, J)
        for l in range(1, p1 - p0):
            if not ok[l]:
                continue
            F = _oracle_factor(tr, alpha_all, hl, hs, l, d)
            if _well_conditioned(F):
                mg.check("c5-shape:funnel-2x25", "elbo", mg.rel(elbo[p0 + l], ref["elbo"][l]), ctx=(k, l))
            else:
                assert abs(elbo[p0 + l] - ref["elbo"][l]) <= 8 * max(se[p0 + l], ref["se"][l]) + 1e-8 * (1 + abs(ref["elbo"][l]))
    p = int(eng.offsets[0]) + 2
    X, lp, lq = eng.draws(p, seeds[p], 32)
    np.testing.assert_allclose(lp, tg.logp(X), rtol=1e-9, atol=1e-6)
    assert np.max(np.abs(eng.logpdf(p, X) - lq)) <= 1e-8 * (1 + np.abs(lq).max())


def test_mfma_and_lane_kernels_agree(pfmi_mod):
    """the two ELBO kernels are interchangeable: same seeds -> same log densities to fp64 roundoff"""
    import subprocess, sys, json
    code = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "pathfinder.jl_amd"); sys.path.insert(0, ".")
import pfmi
from helpers import make_traces, fit_seeds
tg = pfmi.t_lowrank(300, r=8, seed=2)
traces = make_traces(tg, 2, 3)
e = pfmi.Engine(0); e.set_target(tg); e.set_traces([t.points for t in traces], [t.gradients for t in traces]); e.fit_batch(6)
seeds = fit_seeds(e.P, 1)
elbo, se, best = e.elbo_batch(100, seeds)
p = int(e.offsets[0]) + int(best[0])
X, lp, lq = e.draws(p, seeds[p], 100)
print(json.dumps(dict(elbo=np.nan_to_num(elbo).tolist(), best=best.tolist(), x=X[:, :3].ravel().tolist(), lp=lp.tolist(), lq=lq.tolist())))
'''
    outs = []
    for mode in ("mfma", "lane"):
        env = dict(os.environ, PFMI_ELBO_KERNEL=mode)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b = outs
    assert a["best"] == b["best"]
    for key in ("elbo", "x", "lp", "lq"):
        x, y = np.array(a[key]), np.array(b[key])
        assert np.max(np.abs(x - y) / (1 + np.abs(y))) <= 1e-10, key


@pytest.mark.parametrize("name,K,J", [("iso10", 1, 6), ("lr50", 2, 6), ("diag30", 1, 10), ("lr10", 1, 8), ("lr10", 1, 6)])
def test_woodbury_operator_surface(pfmi_mod, eng, name, K, J):
    """remaining PDMats surface on the device vs the oracle and dense algebra (reference test/woodbury.jl:239-402):
    unwhiten / whiten / invunwhiten / R*x / W*x / W\\x / quad / invquad / diag, matrices and vectors; the lr10 cases are the
    n < m ones (d = 10, 2j = 16 / 12: test/woodbury.jl:21-31 has n = 5, m = 8)."""
    tg, traces = _setup(pfmi_mod, eng, name, K, J)
    status, jeff, logdet, _ = eng.fit_status()
    rng = np.random.default_rng(5)
    tr = traces[0]
    alpha_all, hl, hs, _ = po.lbfgs_history(tr.points, tr.gradients, J)
    n_strict = n_wide = 0
    for l in sorted({0, 1, min(4, len(tr) - 1), min(len(tr) - 1, J + 2), min(len(tr) - 1, 2 * J + 1), len(tr) - 1}):
        if status[l] != 0:
            continue
        F = _oracle_factor(tr, alpha_all, hl, hs, l, tg.d)
        n_wide += int(2 * int(hl[l]) > tg.d)
        W = F.dense()
        X = rng.normal(size=(tg.d, 9))
        tol = dict(rtol=1e-8, atol=1e-9 * max(1.0, np.abs(W).max()))
        np.testing.assert_allclose(eng.woodbury_apply(l, "mul", X), W @ X, **tol)
        np.testing.assert_allclose(eng.woodbury_apply(l, "solve", X), np.linalg.solve(W, X), rtol=1e-6, atol=1e-7 * np.abs(np.linalg.solve(W, X)).max())
        np.testing.assert_allclose(eng.woodbury_apply(l, "quad", X), np.einsum("ij,ij->j", X, W @ X), rtol=1e-8)
        np.testing.assert_allclose(eng.woodbury_apply(l, "invquad", X), np.einsum("ij,ij->j", X, np.linalg.solve(W, X)), rtol=1e-6)
        np.testing.assert_allclose(eng.woodbury_diag(l), np.diag(W), rtol=1e-9, atol=1e-12)
        x = X[:, 0].copy()
        np.testing.assert_allclose(eng.woodbury_apply(l, "mul", x), W @ x, **tol)
        # L and R themselves agree with the oracle's factor when the QR is well conditioned, and always satisfy
        # L (L \ x) = x, R \ (R x) = x, unwhiten(whiten(x)) = x
        if _well_conditioned(F):
            n_strict += 1
            np.testing.assert_allclose(eng.woodbury_apply(l, "unwhiten", X), F.lmul_L(X), rtol=1e-7, atol=1e-8 * np.abs(X).max() * np.sqrt(np.abs(W).max()))
            np.testing.assert_allclose(eng.woodbury_apply(l, "rmul", X), F.lmul_R(X), rtol=1e-7, atol=1e-8 * np.abs(X).max() * np.sqrt(np.abs(W).max()))
            np.testing.assert_allclose(eng.woodbury_apply(l, "whiten", X), F.ldiv_L(X), rtol=1e-6, atol=1e-7 * np.abs(F.ldiv_L(X)).max())
            np.testing.assert_allclose(eng.woodbury_apply(l, "invunwhiten", X), F.ldiv_R(X), rtol=1e-6, atol=1e-7 * np.abs(F.ldiv_R(X)).max())
        back = eng.woodbury_apply(l, "unwhiten", eng.woodbury_apply(l, "whiten", X))
        np.testing.assert_allclose(back, X, rtol=1e-6, atol=1e-7 * np.abs(X).max())
        back = eng.woodbury_apply(l, "invunwhiten", eng.woodbury_apply(l, "rmul", X))
        np.testing.assert_allclose(back, X, rtol=1e-6, atol=1e-7 * np.abs(X).max())
    assert n_strict >= MIN_STRICT.get(name, 2), (name, n_strict)
    if name == "lr10":
        assert n_wide >= 2, n_wide


# ---- device trajectory generation -----------------------------------------------------
@pytest.mark.parametrize("name,d,scale,maxit", [("iso", 10, 2, 1000), ("diag", 30, 2, 1000), ("lr", 50, 2, 1000), ("funnel", 12, 10, 60),
                                               ("lr", 1000, 2, 1000), ("diag", 3000, 2, 200), ("lr16", 600, 2, 1000), ("lr11", 200, 2, 1000),
                                               ("lr", 1500, 2, 300)])
def test_device_lbfgs_traces_match_oracle_driver(pfmi_mod, eng, name, d, scale, maxit):
    """pfmi_optimize_batch vs oracle pfo_optimize_trace (same algorithm, scalar C): early iterates agree to roundoff
    (later ones drift apart through line-search branches, as between any two L-BFGS implementations), every recorded
    (logp, grad) belongs to its recorded point, the objective never increases, Gaussian targets converge to g_tol.
    d = 1000 exercises the LDS ring, d = 3000 the global ring and the 1024-thread variant; rank 16 / 11 the 16-column padding of the
    low-rank factor (its rows cached in registers), d = 1500 the low-rank target with rows re-read from memory."""
    tg = {"iso": pfmi_mod.t_iso, "diag": lambda d: pfmi_mod.t_diag(d, 1), "lr": lambda d: pfmi_mod.t_lowrank(d, 8, 2),
          "lr16": lambda d: pfmi_mod.t_lowrank(d, 16, 3), "lr11": lambda d: pfmi_mod.t_lowrank(d, 11, 4),
          "funnel": pfmi_mod.t_funnel}[name](d)
    ot = oracle_target(tg)
    K = 3
    x0 = pfmi_mod.HostRNG(3).rand(K * d).reshape(K, d) * 2 * scale - scale
    eng.set_target(tg)
    npts = eng.optimize_batch(x0, 6, maxit)
    assert np.all(npts >= 2) and np.all(npts <= maxit + 1)
    for k in range(K):
        th, lp, gr = eng.get_trace(k)
        assert th.shape == (npts[k], d) and np.array_equal(th[0], x0[k])
        P, L, G = po.optimize_trace(ot, x0[k], 6, maxit)
        n = min(len(P), len(th), 8)
        np.testing.assert_allclose(th[:n], P[:n], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(gr[:n], G[:n], rtol=1e-8, atol=1e-9 * max(1.0, np.abs(G[:n]).max()))
        for l in sorted({0, 1, len(th) // 2, len(th) - 1}):
            lpo, go = po.logp_grad(ot, th[l])
            assert abs(lpo - lp[l]) <= 1e-11 * max(1.0, abs(lpo))
            np.testing.assert_allclose(gr[l], go, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(go).max()))
        assert np.all(np.diff(lp) >= -1e-9 * np.maximum(1.0, np.abs(lp[1:])))
        if name != "funnel" and npts[k] <= maxit:
            assert np.abs(gr[-1]).max() <= 1e-8
            np.testing.assert_allclose(th[-1], P[-1], atol=1e-5)


def test_device_traces_feed_fit_batch_like_uploaded_ones(pfmi_mod, eng):
    """the traces pfmi_optimize_batch leaves in HBM are the same input pfmi_set_traces would upload: refitting from
    the downloaded copy gives bit-identical ELBOs."""
    tg = pfmi_mod.t_lowrank(64, 8, 2)
    eng.set_target(tg)
    x0 = pfmi_mod.HostRNG(7).rand(5 * 64).reshape(5, 64) * 4 - 2
    npts = eng.optimize_batch(x0, 6)
    seeds = fit_seeds(int(npts.sum()), 4)
    eng.fit_batch(6)
    e1, s1, b1 = eng.elbo_batch(200, seeds)
    traces = [eng.get_trace(k) for k in range(5)]
    eng.set_traces([t[0] for t in traces], [t[2] for t in traces])
    eng.fit_batch(6)
    e2, s2, b2 = eng.elbo_batch(200, seeds)
    np.testing.assert_array_equal(e1, e2)
    np.testing.assert_array_equal(b1, b2)
    with pytest.raises(pfmi_mod.PfmiError):
        eng.get_trace(0)                                   # log densities only exist for device-made traces
    eng.set_target(pfmi_mod.CallbackTarget(64, lambda x: 0.0))
    with pytest.raises(pfmi_mod.PfmiError):
        eng.optimize_batch(x0, 6)


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


@pytest.mark.parametrize("tname,d,K,J,N,scale,maxit", [
    ("iso", 10, 2, 6, 100, 2, 1000), ("diag", 30, 2, 6, 200, 2, 1000), ("lr", 50, 2, 6, 200, 2, 1000), ("lr", 300, 2, 6, 500, 2, 1000),
    ("funnel", 12, 2, 6, 100, 10, 40), ("diag", 30, 2, 10, 200, 2, 1000), ("lr", 50, 2, 16, 200, 2, 1000),
    ("diag", 3000, 2, 6, 200, 2, 30), ("funnel", 2500, 2, 10, 300, 10, 30), ("lr", 1100, 2, 8, 130, 2, 40),
    ("lr", 64, 8, 6, 1000, 2, 45), ("diag", 48, 7, 4, 500, 2, 50),
    # round 3: two groups per wave at KC = 16 / 20 (N >= 768), resident and streamed, every target family, ragged tails
    ("funnel", 2000, 2, 10, 800, 10, 24), ("lr", 600, 2, 8, 1000, 2, 30), ("diag", 100, 2, 10, 784, 2, 40), ("lr", 1500, 2, 10, 770, 2, 24),
    ("lr", 3000, 2, 16, 200, 2, 20)])
def test_single_pass_scan_matches_lane_kernel(pfmi_mod, eng, tname, d, K, J, N, scale, maxit):
    """the single-pass quadratic-form scan (elbo_qf_kernel.hip: logp from per-draw contractions, x never formed; Vh resident or
    streamed through LDS; KC up to 32) against the lane-per-draw kernel that evaluates logp(x) on the materialised draw, same
    seeds: per-draw logp / logq and the per-fit ELBO agree to fp64 roundoff.  Covers the head transform spilling into
    block 1 (J = 10, 16), chunked streaming (d = 2500, 3000), ragged last block / last group and the low-rank target; the two
    K = 8 / 7 cases have more fits than the GPU has CUs and not a multiple of them, so the fits of the last partial round take the
    second, one-batch-per-workgroup launch (two groups per wave at N = 1000, one at N = 500)."""
    tg = {"iso": pfmi_mod.t_iso, "diag": lambda d: pfmi_mod.t_diag(d, 1), "lr": lambda d: pfmi_mod.t_lowrank(d, 8, 2),
          "funnel": pfmi_mod.t_funnel}[tname](d)
    eng.set_target(tg)
    x0 = pfmi_mod.HostRNG(3).rand(K * d).reshape(K, d) * 2 * scale - scale
    eng.optimize_batch(x0, J, maxit)
    eng.fit_batch(J)
    if K >= 7:
        nfits = eng.P - K
        assert nfits > 256 and nfits % 256 != 0, nfits               # the tail launch really runs
    seeds = fit_seeds(eng.P, 1)
    out = {}
    old = os.environ.get("PFMI_ELBO_KERNEL")
    try:
        for mode in ("lane", "qf"):
            os.environ["PFMI_ELBO_KERNEL"] = mode
            elbo, se, best = eng.elbo_batch(N, seeds)
            pts = sorted({1, min(3, eng.P - 1), eng.P // 2, eng.P - 1, max(eng.P - 7, 1), max(eng.P - 20, 1)})
            out[mode] = (elbo, se, best, [eng.elbo_logs(p, N) for p in pts])
    finally:
        if old is None:
            os.environ.pop("PFMI_ELBO_KERNEL", None)
        else:
            os.environ["PFMI_ELBO_KERNEL"] = old
    a, b = out["qf"], out["lane"]
    assert np.array_equal(np.isnan(a[0]), np.isnan(b[0]))
    ok = np.isfinite(b[0])
    assert np.max(np.abs(a[0][ok] - b[0][ok]) / (1 + np.abs(b[0][ok]))) <= 1e-10
    oks = ok & np.isfinite(b[1])
    assert np.array_equal(np.isfinite(a[1][ok]), np.isfinite(b[1][ok]))
    assert np.max(np.abs(a[1][oks] - b[1][oks]) / (1 + np.abs(b[1][oks]))) <= 1e-8
    for (lpa, lqa), (lpb, lqb) in zip(a[3], b[3]):
        assert np.max(np.abs(lpa - lpb) / (1 + np.abs(lpb))) <= 1e-10
        assert np.max(np.abs(lqa - lqb) / (1 + np.abs(lqb))) <= 1e-12


def test_rccl_collectives_on_engine_memory_world1(pfmi_mod, eng):
    """torch's `nccl` (= RCCL) collectives on ENGINE-OWNED device memory at world_size 1 -- the only RCCL configuration a 1-GPU box
    allows (the product's own collectives are pfmi_comm_*, csrc/comm_rccl.hip; this checks the interop a torch host relies on when it
    passes engine buffers to its own collectives): the collectives run directly on
    device memory owned by libpfmi (zero-copy view) and on torch tensors the engine writes through raw pointers, and the
    stream hand-over (engine stream -> torch stream -> engine stream) leaves the data intact.  world_size 2 is covered on
    CPU by tests/test_distributed_cpu.py (gloo)."""
    import torch
    import torch.distributed as dist

    class DevArray:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    except Exception as e:  # pragma: no cover
        pytest.skip(f"RCCL process group could not be created on this box: {e!r}")
    try:
        tg, traces = _setup(pfmi_mod, eng, "lr50", 4, 6)
        seeds = fit_seeds(eng.P, 4)
        elbo, se, best = eng.elbo_batch(64, seeds)
        pts = [int(eng.offsets[k]) + int(best[k]) for k in range(4)]
        eng.pool_build(64, pts, seeds[pts])
        pool, lr = eng.pool_get()
        ptr, cnt = eng.pool_log_ratios_dev()
        shard = torch.as_tensor(DevArray(ptr, cnt), device="cuda:0")
        lr_all = torch.empty(cnt, dtype=torch.float64, device="cuda:0")
        dist.all_gather_into_tensor(lr_all, shard)                    # RCCL reads libpfmi's buffer
        torch.cuda.synchronize()
        np.testing.assert_array_equal(lr_all.cpu().numpy(), lr)
        res = eng.psis_dev(lr_all.data_ptr(), lr_all.numel())
        idx = eng.resample_indices(cnt, 32, seed=9)
        out = torch.zeros(tg.d * 32, dtype=torch.float64, device="cuda:0")
        eng.pool_gather_dev(idx, 0, out.data_ptr())                 # engine stream writes a torch tensor ...
        eng.sync()
        dist.all_reduce(out)                                          # ... RCCL reduces it in place
        torch.cuda.synchronize()
        ref = eng.psis(lr)
        np.testing.assert_array_equal(res["weights"], ref["weights"])
        np.testing.assert_array_equal(out.cpu().numpy().reshape(32, tg.d).T, pool.reshape(tg.d, -1, order="F")[:, idx])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("tname,d,J", [("diag", 2100, 6), ("lr", 1500, 6), ("funnel", 2500, 10)])
def test_memory_resident_fit_kernel_at_large_d(pfmi_mod, eng, tname, d, J, monkeypatch):
    """the column-by-column memory-resident kernel (fused reflector-apply + next-column dots, two Gram rows per sweep; the large-d
    default of round 1, now behind PFMI_FIT_KERNEL=mem and for d > 16384): dense W / logdet / mu against the oracle at four fits
    of the first path."""
    monkeypatch.setenv("PFMI_FIT_KERNEL", "mem")
    tg = {"diag": lambda d: pfmi_mod.t_diag(d, 1), "lr": lambda d: pfmi_mod.t_lowrank(d, 8, 2), "funnel": pfmi_mod.t_funnel}[tname](d)
    eng.set_target(tg)
    x0 = pfmi_mod.HostRNG(3).rand(2 * d).reshape(2, d) * (20 if tname == "funnel" else 4) - (10 if tname == "funnel" else 2)
    eng.optimize_batch(x0, J, 25)
    eng.fit_batch(J)
    st, je, ld, nr = eng.fit_status()
    th, _, gr = eng.get_trace(0, logp=False)
    alpha_all, hl, hs, _ = po.lbfgs_history(th, gr, J)
    n_checked = 0
    for p in sorted({1, 2, len(th) // 2, len(th) - 1}):
        if st[p] != 0:
            continue
        fa = eng.get_fit(p, int(je[p]))
        F = _oracle_factor(type("T", (), {"points": th, "gradients": gr})(), alpha_all, hl, hs, p, d)
        assert int(je[p]) == int(hl[p])
        assert abs(F.logdet - fa["logdet"]) <= 1e-9 * (1 + abs(F.logdet))
        mu_o = F.fit_mean(th[p], gr[p])
        np.testing.assert_allclose(fa["mu"], mu_o, rtol=1e-7, atol=1e-8 * (1 + np.abs(mu_o).max()))
        Wa = np.diag(fa["alpha"]) + fa["B"] @ fa["D"] @ fa["B"].T if fa["B"].size else np.diag(fa["alpha"])
        Wo = F.dense()
        assert np.max(np.abs(Wa - Wo)) <= 1e-10 * np.abs(Wo).max() * max(1.0, np.linalg.cond(fa["D"]) ** 0.5 if fa["D"].size else 1.0)
        n_checked += 1
    assert n_checked >= 3


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tname,d,J,maxit", [("diag", 1500, 4, 12), ("lr", 3000, 6, 14), ("funnel", 6000, 10, 16), ("diag", 12000, 10, 14),
                                               ("diag", 2000, 16, 22), ("funnel", 10000, 10, 14)])
def test_panel_fit_kernel_variants(pfmi_mod, eng, tname, d, J, maxit):
    """The panel-blocked fit kernel (the default for 1024 < d <= 16384: register panels of 4 / 2 columns, MFMA cross products,
    G = R'R, mean without a sweep) in every instantiation: rows per thread 5 / 10 / 20 / 32, one and two MFMA column tiles
    (KPAD 8, 12, 20, 32), panels with 2 valid columns (odd history lengths at the start of a path).  Against the oracle:
    status, logdet, mu, reflector-level QR / T / V where the QR is well conditioned, W x through the factor; and against the
    column-by-column memory-resident kernel (PFMI_FIT_KERNEL=mem) on every fit."""
    tg = {"diag": lambda d: pfmi_mod.t_diag(d, 1), "lr": lambda d: pfmi_mod.t_lowrank(d, 8, 2), "funnel": pfmi_mod.t_funnel}[tname](d)
    eng.set_target(tg)
    x0 = pfmi_mod.HostRNG(4).rand(2 * d).reshape(2, d) * (20 if tname == "funnel" else 4) - (10 if tname == "funnel" else 2)
    eng.optimize_batch(x0, J, maxit)
    eng.fit_batch(J)
    st, je, ld, nr = eng.fit_status()
    fits = {p: eng.get_fit(p, int(je[p])) for p in range(eng.P)}
    old = os.environ.get("PFMI_FIT_KERNEL")
    os.environ["PFMI_FIT_KERNEL"] = "mem"
    try:
        eng.fit_batch(J)
        st2, je2, ld2, _ = eng.fit_status()
        fits2 = {p: eng.get_fit(p, int(je2[p])) for p in range(eng.P)}
    finally:
        if old is None:
            os.environ.pop("PFMI_FIT_KERNEL", None)
        else:
            os.environ["PFMI_FIT_KERNEL"] = old
    np.testing.assert_array_equal(st, st2)
    np.testing.assert_array_equal(je, je2)
    seen_m = set()
    n_strict = 0
    rng = np.random.default_rng(0)
    for k in range(2):
        th, _, gr = eng.get_trace(k, logp=False)
        p0 = int(eng.offsets[k])
        alpha_all, hl, hs, _ = po.lbfgs_history(th, gr, J)
        tr = type("T", (), {"points": th, "gradients": gr})()
        for l in range(len(th)):
            p = p0 + l
            if st[p] != 0:
                continue
            fa, fb = fits[p], fits2[p]
            j = int(je[p])
            seen_m.add(2 * j)
            assert abs(ld[p] - ld2[p]) <= 1e-9 * (1 + abs(ld2[p]))
            if l in (1, 2, 3, len(th) // 2, len(th) - 1):
                F = _oracle_factor(tr, alpha_all, hl, hs, l, d)
                assert j == int(hl[l]) and F.status == 0
                assert abs(F.logdet - fa["logdet"]) <= 1e-9 * (1 + abs(F.logdet))
                mu_o = F.fit_mean(th[l], gr[l])
                np.testing.assert_allclose(fa["mu"], mu_o, rtol=1e-7, atol=1e-8 * (1 + np.abs(mu_o).max()))
                np.testing.assert_allclose(fa["D"], F.D, rtol=1e-6, atol=1e-9 * np.abs(F.D).max())
                if j:
                    X = rng.normal(size=(d, 3))
                    Wx = fa["alpha"][:, None] * X + fa["B"] @ (fa["D"] @ (fa["B"].T @ X))
                    np.testing.assert_allclose(Wx, F.mul_W(X), rtol=1e-8, atol=1e-9 * np.abs(Wx).max())
                    Vh = np.tril(fa["qr_factors"], -1) + np.eye(d, 2 * j)
                    G = Vh.T @ Vh                                        # Q'Q = I  <=>  T^-1 + T^-T = Vh'Vh
                    assert np.all(np.diag(fa["T"]) > 0)
                    np.testing.assert_allclose(np.linalg.inv(fa["T"]) + np.linalg.inv(fa["T"]).T, G, rtol=1e-9, atol=1e-10 * np.abs(G).max())
                    if _well_conditioned(F):
                        n_strict += 1
                        amp = 1e-13 / _qr_ratio(F)
                        np.testing.assert_allclose(fa["V"], F.V[:2 * j, :2 * j], rtol=1e-8, atol=max(1e-9, amp) * np.abs(F.V).max())
                        np.testing.assert_allclose(fa["qr_factors"], F.QR[:, :2 * j], rtol=1e-8, atol=max(1e-9, amp) * np.abs(F.QR).max())
                        np.testing.assert_allclose(np.diag(fa["T"]), F.tau[:2 * j], rtol=1e-9, atol=1e-12)
            # panel kernel vs column-by-column kernel: same reflectors up to the conditioning of the block
            scale = max(np.abs(fb["qr_factors"]).max(), 1e-300) if j else 1.0
            Rd = np.abs(np.diag(fb["qr_factors"][:2 * j, :2 * j])) if j else np.ones(1)
            amp = 1e-12 * (Rd.max() / max(Rd.min(), 1e-300)) if j else 0.0
            if j and amp < 1e-6:
                np.testing.assert_allclose(fa["qr_factors"], fb["qr_factors"], rtol=1e-7, atol=max(1e-10, amp) * scale)
                np.testing.assert_allclose(fa["T"], fb["T"], rtol=1e-7, atol=max(1e-10, amp))
            np.testing.assert_allclose(fa["mu"], fb["mu"], rtol=1e-7, atol=1e-8 * (1 + np.abs(fb["mu"]).max()))
    assert n_strict >= 2, n_strict
    assert any(mm % 4 == 2 for mm in seen_m) and max(seen_m) == 2 * J, seen_m      # ragged last panel and the full history both ran


@pytest.mark.parametrize("name,K,J", [("iso10", 2, 6), ("lr50", 2, 6), ("diag30", 2, 10), ("funnel12", 2, 6)])
def test_memory_resident_fit_kernel_matches_oracle(pfmi_mod, eng, name, K, J):
    """the general fit kernel (the default only for d > 1024 or J > 8) forced onto the small oracle cases:
    same dense W / logdet / mu / reflector-level checks as test_fit_batch_matches_oracle."""
    old = os.environ.get("PFMI_FIT_KERNEL")
    os.environ["PFMI_FIT_KERNEL"] = "mem"
    try:
        test_fit_batch_matches_oracle(pfmi_mod, eng, name, K, J)
    finally:
        if old is None:
            os.environ.pop("PFMI_FIT_KERNEL", None)
        else:
            os.environ["PFMI_FIT_KERNEL"] = old


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


def test_consistency_of_rand_300k_draws(pfmi_mod, eng):
    """reference test/mvnormal.jl:66-109 ("consistency of rand"): 300 000 draws of a fitted MvNormal{WoodburyPDMat} -- sample
    means, variances and (variance-stabilised) correlations against mu / Sigma with the reference's Bonferroni-corrected
    normal tolerances.  Pins the device generator + transform statistically (d = 50, history 4)."""
    from scipy.stats import norm
    tg, traces = _setup(pfmi_mod, eng, "lr50", 1, 4)
    status, jeff, _, _ = eng.fit_status()
    p = int(np.flatnonzero((status == 0) & (jeff == 4))[3])
    f = eng.get_fit(p, 4)
    d = tg.d
    Sig = np.diag(f["alpha"]) + f["B"] @ f["D"] @ f["B"].T
    nd = 300_000
    X = eng.draws(p, 123456789, nd)[0]
    v = np.diag(Sig)
    R = Sig / np.sqrt(v) / np.sqrt(v)[:, None]
    mu_est, v_est, R_est = X.mean(1), X.var(1), np.corrcoef(X)
    nchecks = 2 * d + d * (d - 1) // 2
    tol = norm.ppf(1 - (0.01 / nchecks) / 2) / np.sqrt(nd)
    assert np.all(np.abs(mu_est - f["mu"]) <= tol * np.sqrt(v))
    assert np.all(np.abs(v_est - v) <= tol * np.sqrt(2) * v)
    iu = np.triu_indices(d, 1)
    assert np.all(np.abs(np.arctanh(R_est[iu]) - np.arctanh(R[iu])) <= tol)


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
