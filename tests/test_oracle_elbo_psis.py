"""Pins the CPU oracle's ELBO / MvNormal / RNG / PSIS / resampling pieces.

Reference tests mirrored: test/elbo.jl:7-54 (analytic ELBO, argmax = 3, reseed reproducibility,
empty input), test/utils.jl:8-12 (_findmax_skipnan), test/mvnormal.jl:31-107 (rand_and_logpdf ==
rand + logpdf, statistical consistency), test/resample.jl:8-109 (membership, degenerate weights,
log-ratio ordering, sum(weights) ~ 1), test/singlepath.jl:13-41 (iso-normal exactness).
"""
import numpy as np
import pytest
import scipy.stats as st

from oracle import pf_oracle as po


# ---- utils.jl --------------------------------------------------------------------------------------
def test_findmax_skipnan_reference_cases():
    rng = np.random.default_rng(0)
    x = rng.normal(size=100)
    v, i = po.findmax_skipnan(x)
    assert v == x.max() and i == int(np.argmax(x)) + 1
    assert po.findmax_skipnan([np.nan, 3.0, 1.0]) == (3.0, 2)       # test/utils.jl:10
    v, i = po.findmax_skipnan([np.nan, np.nan, np.nan])             # :11
    assert np.isnan(v) and i == 1
    assert po.findmax_skipnan([2.0, np.nan, 4.0]) == (4.0, 3)       # :12
    assert po.findmax_skipnan([1.0, 5.0, 5.0])[1] == 2              # first maximum wins
    assert po.findmax_skipnan([])[1] == 0                           # maximize_elbo empty -> 0 (src/elbo.jl:7)


# ---- RNG ----------------------------------------------------------------------------------------------
def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, exp in kat:
        assert [int(v) for v in po.philox4x32_10(ctr, key)] == exp


def test_randn_fill_is_counter_based_and_standard_normal():
    U = po.randn_fill(12345, 10, 5000)
    U2 = po.randn_fill(12345, 10, 100, n0=4900)
    np.testing.assert_array_equal(U[:, 4900:], U2)       # draw n is a pure function of (seed, n)
    V = po.randn_fill(12345, 7, 50)                      # a different d shares rows 0..6 of each group
    np.testing.assert_array_equal(V[:4], U[:4, :50])
    Z = po.randn_fill(99, 64, 40000).ravel()
    assert abs(Z.mean()) < 4 / np.sqrt(Z.size)
    assert abs(Z.var() - 1) < 4 * np.sqrt(2 / Z.size)
    assert st.kstest(Z[:200000], "norm").pvalue > 1e-3
    # Box-Muller pairs are uncorrelated across rows
    C = np.corrcoef(po.randn_fill(7, 8, 50000))
    assert np.abs(C - np.eye(8)).max() < 0.03


# ---- elbo.jl ------------------------------------------------------------------------------------------
def _normal_1d_factor(sigma):
    """1-D Normal(0, sigma) as a rank-0 Woodbury MvNormal (alpha = sigma^2)."""
    return po.Factor(np.array([sigma**2]), np.zeros((1, 0)), np.zeros((0, 0)))


@pytest.mark.parametrize("sigma", [1e-3, 0.05, 0.8, 1.0, 1.1, 1.2, 5.0, 10.0])
def test_elbo_analytic_known_answer(sigma):
    """test/elbo.jl:7-28: ELBO = (1 - r^2)/2 + log r, r = sigma / sigma_target, atol 3 SE."""
    sigma_t = 0.08
    N = 200_000
    tgt = po.GaussTarget(np.zeros(1), np.array([1 / sigma_t**2]),
                         offset=-0.5 * np.log(2 * np.pi) - np.log(sigma_t))
    F = _normal_1d_factor(sigma)
    U = po.randn_fill(42 + int(sigma * 1000), 1, N)
    X, logq = F.rand_and_logpdf(np.zeros(1), U)
    logp = tgt.logp(X)
    value, se, logr = po.elbo_stats(logp, logq)
    r = sigma / sigma_t
    assert abs(value - ((1 - r * r) / 2 + np.log(r))) <= 3 * se + 1e-12
    np.testing.assert_allclose(logq, st.norm(0, sigma).logpdf(X[0]), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(logp, st.norm(0, sigma_t).logpdf(X[0]), rtol=1e-10, atol=1e-9)
    np.testing.assert_array_equal(logr, logp - logq)
    assert value == pytest.approx(np.mean(logr), rel=1e-13)
    assert se == pytest.approx(np.std(logr, ddof=1) / np.sqrt(N), rel=1e-12)


def test_maximize_elbo_argmax_is_exact_fit():
    """test/elbo.jl:30-54: with sigma[2] == sigma_target the ELBO is ~0 and wins."""
    sigma_t = 0.08
    sigmas = [1e-3, 0.05, sigma_t, 1.0, 1.1, 1.2, 5.0, 10.0]
    tgt = po.GaussTarget(np.zeros(1), np.array([1 / sigma_t**2]),
                         offset=-0.5 * np.log(2 * np.pi) - np.log(sigma_t))
    vals = []
    for i, s in enumerate(sigmas):
        F = _normal_1d_factor(s)
        X, logq = F.rand_and_logpdf(np.zeros(1), po.randn_fill(1000 + i, 1, 100))
        vals.append(po.elbo_stats(tgt.logp(X), logq)[0])
    v, i = po.findmax_skipnan(vals)
    assert i == 3 and abs(v) < 1e-12


# ---- mvnormal.jl -----------------------------------------------------------------------------------
def rand_pd(rng, n):
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    return (Q * rng.uniform(0.05, 1.0, n)) @ Q.T


def test_rand_and_logpdf_equals_rand_plus_logpdf():
    """test/mvnormal.jl:31-69: fused path == (mu + L u, logpdf) for a Woodbury covariance."""
    rng = np.random.default_rng(42)
    n, nhist, N = 10, 4, 20
    alpha = rng.uniform(0.1, 1, n); B = rng.normal(size=(n, 2 * nhist)); D = rand_pd(rng, 2 * nhist)
    mu = rng.normal(size=n)
    F = po.Factor(alpha, B, D)
    U = po.randn_fill(42, n, N)
    X, logq = F.rand_and_logpdf(mu, U)
    np.testing.assert_allclose(X, mu[:, None] + F.lmul_L(U), rtol=1e-13, atol=1e-13)
    ref = st.multivariate_normal(mu, F.dense()).logpdf(X.T)
    np.testing.assert_allclose(logq, ref, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(F.logpdf(mu, X), ref, rtol=1e-9, atol=1e-9)   # Distributions.logpdf path


def test_rand_consistency_statistical():
    """test/mvnormal.jl:71-107 (mean / variance / atanh-correlation with Bonferroni tolerance)."""
    rng = np.random.default_rng(7)
    n, nhist, N = 10, 4, 300_000
    alpha = rng.uniform(0.1, 1, n); B = rng.normal(size=(n, 2 * nhist)); D = rand_pd(rng, 2 * nhist)
    mu = rng.normal(size=n)
    F = po.Factor(alpha, B, D)
    X, _ = F.rand_and_logpdf(mu, po.randn_fill(2024, n, N))
    Sig = F.dense(); v = np.diag(Sig)
    Rc = Sig / np.sqrt(np.outer(v, v))
    nchecks = 2 * n + n * (n - 1) // 2
    tol = st.norm.ppf(1 - (0.01 / nchecks) / 2) / np.sqrt(N)
    assert np.all(np.abs(X.mean(1) - mu) < tol * np.sqrt(v))
    assert np.all(np.abs(X.var(1) - v) < tol * np.sqrt(2) * v)
    Re = np.corrcoef(X)
    iu = np.triu_indices(n, 1)
    assert np.all(np.abs(np.arctanh(Re[iu]) - np.arctanh(Rc[iu])) < tol)


def test_isonormal_one_iteration_is_exact():
    """test/singlepath.jl:13-41: for logp = -|x|^2/2 one L-BFGS step lands on the mode; then
    s = y = -theta0, gilbert_init gives alpha = 1, B D B' = 0, mu = 0, Sigma = I, size(B) = (d, 2)."""
    rng = np.random.default_rng(0)
    for d in (1, 5, 10, 100):
        th0 = rng.normal(size=d)
        theta = np.stack([th0, np.zeros(d)])
        grad = -theta
        res = po.path_fit_elbo(theta, grad, 6, po.GaussTarget(np.zeros(d), np.ones(d)), 10,
                               np.array([1, 2], dtype=np.uint64))
        assert list(res["j_eff"]) == [0, 1] and res["n_rejected"] == 0
        np.testing.assert_allclose(res["mu"][1], 0, atol=1e-6)
        assert abs(res["logdet"][1]) < 1e-6
        alpha_all, hl, hs, _ = po.lbfgs_history(theta, grad, 6)
        B, D = po.lbfgs_inverse_hessian(alpha_all[1], (theta[1] - theta[0])[:, None], (grad[0] - grad[1])[:, None])
        assert B.shape == (d, 2)
        np.testing.assert_allclose(np.diag(alpha_all[1]) + B @ D @ B.T, np.eye(d), atol=1e-6)
        assert res["best_iter"] == 1 and abs(res["elbo"][1] - (d / 2) * np.log(2 * np.pi)) < 1e-9 * d + 1e-9


# ---- targets --------------------------------------------------------------------------------------
def test_lowrank_gauss_target_against_dense():
    rng = np.random.default_rng(2)
    d, r = 40, 5
    sig2 = np.exp(rng.uniform(-1, 1, d)); W = rng.normal(size=(d, r)); mean = rng.normal(size=d)
    Wd = W / sig2[:, None]
    Cm = np.eye(r) + W.T @ Wd
    G = np.linalg.inv(np.linalg.cholesky(Cm))
    tgt = po.GaussTarget(mean, 1 / sig2, Wd, G)
    X = rng.normal(size=(d, 9))
    Sig = np.diag(sig2) + W @ W.T
    e = X - mean[:, None]
    np.testing.assert_allclose(tgt.logp(X), -0.5 * np.einsum("ij,ij->j", e, np.linalg.solve(Sig, e)), rtol=1e-10)


def test_funnel_formula():
    """docs/src/examples/quickstart.md:229-234"""
    rng = np.random.default_rng(3)
    d = 12
    X = rng.normal(size=(d, 5))
    tau = X[0]
    exp = ((tau / 3) ** 2 + (d - 1) * tau + np.sum((X[1:] * np.exp(-tau / 2)) ** 2, axis=0)) / -2
    np.testing.assert_allclose(po.FunnelTarget(d).logp(X), exp, rtol=1e-13)


# ---- PSIS --------------------------------------------------------------------------------------------
def psis_numpy(logw):
    """Independent NumPy restatement of the published algorithm (Vehtari et al. 2024, appendix;
    Zhang & Stephens 2009) using scipy.stats.genpareto for the quantiles."""
    x = np.array(logw, dtype=float)
    S = len(x)
    M = int(min(-(-S // 5), np.ceil(3 * np.sqrt(S))))
    k = np.nan
    if M >= 5:
        order = np.argsort(x, kind="stable")
        cut = x[order[S - M - 1]]
        tail = order[S - M:]
        if np.all(np.isfinite(x[tail])):
            lmax = x[tail[-1]]
            mu_s = np.exp(cut - lmax)
            w = np.exp(x[tail] - lmax) - mu_s
            if np.any(w != 0):
                n = M
                m = 30 + int(np.floor(np.sqrt(n)))
                b = 1 / w[-1] + (1 - np.sqrt(m / (np.arange(1, m + 1) - 0.5))) / (3 * w[(n + 2) // 4 - 1])
                ks = np.log1p(-b[:, None] * w).mean(axis=1)
                L = n * (np.log(-b / ks) - ks - 1)
                wt = np.exp(L - L.max()); wt /= wt.sum()
                bp = np.sum(b * wt)
                k = np.log1p(-bp * w).mean()
                sigma = -k / bp
                k = (k * n + 5) / (n + 10)
                p = (np.arange(1, n + 1) - 0.5) / n
                q = st.genpareto(c=k, scale=sigma).ppf(p)
                x[tail] = np.minimum(np.log(q + mu_s), 0) + lmax
    x -= np.logaddexp.reduce(x)
    return x, np.exp(x), k, M


@pytest.mark.parametrize("S,df", [(1000, 3.0), (64000, 5.0), (200, 1.5), (30, 2.0)])
def test_psis_matches_published_algorithm(S, df):
    rng = np.random.default_rng(S)
    lr = st.t(df).rvs(S, random_state=rng) * 1.5 - 3.0
    lw, w, k, M = po.psis(lr)
    lw2, w2, k2, M2 = psis_numpy(lr)
    assert M == M2 == po.lib().pfo_psis_tail_length(S)
    assert abs(k - k2) < 1e-9
    np.testing.assert_allclose(lw, lw2, rtol=1e-10, atol=1e-10)
    assert abs(w.sum() - 1) < 1e-12                      # test/resample.jl:108
    # smoothing keeps the tail ordered and capped at the raw maximum
    order = np.argsort(lr, kind="stable")
    assert np.all(np.diff(lw[order[S - M:]]) >= -1e-12)


def test_psis_small_and_degenerate():
    # M < 5: no smoothing, only normalisation (PSIS.jl warns)
    lr = np.array([0.1, -0.3, 0.5, 0.0, 1.0, -2.0])
    lw, w, k, M = po.psis(lr)
    assert M < 5 and np.isnan(k)
    np.testing.assert_allclose(lw, lr - np.logaddexp.reduce(lr), rtol=1e-14)
    # test/resample.jl:36-49: only the first component carries weight
    lw = np.full((10, 4), -1000.0); lw[:, 0] = 0.0
    _, w, k, M = po.psis(lw.T.ravel())   # column-major vec: n fastest, k slowest
    np.testing.assert_allclose(w[:10], 0.1, rtol=1e-12)
    assert np.all(w[10:] == 0)
    idx = po.sample_weighted(w, 20, seed=5)
    assert np.all(idx < 10)              # all from component 1


def test_gpd_fit_recovers_shape():
    rng = np.random.default_rng(8)
    x = np.sort(st.genpareto(c=0.4, scale=2.0).rvs(4000, random_state=rng))
    sigma, k = po.gpd_fit(x)
    assert abs(k - 0.4) < 0.08 and abs(sigma - 2.0) < 0.25


# ---- resampling ------------------------------------------------------------------------------------
def test_sample_weighted_distribution_and_determinism():
    rng = np.random.default_rng(4)
    w = rng.dirichlet(np.ones(50))
    idx = po.sample_weighted(w, 400_000, seed=77)
    assert np.array_equal(idx, po.sample_weighted(w, 400_000, seed=77))
    freq = np.bincount(idx, minlength=50) / len(idx)
    assert np.abs(freq - w).max() < 5 * np.sqrt(w.max() / len(idx))
    # explicit uniforms: inverse CDF on the fixed-point table
    u = np.array([0.0, 0.25, 0.5, 0.999999])
    idx = po.sample_weighted(np.array([0.25, 0.25, 0.5]), 4, uniforms=u)
    assert list(idx) == [0, 1, 2, 2]
    # zero-weight entries are never drawn
    w = np.array([0.0, 0.5, 0.0, 0.5, 0.0])
    assert set(po.sample_weighted(w, 1000, seed=1)) == {1, 3}


def test_sample_uniform_and_norep():
    idx = po.sample_uniform(40, 20, seed=3)
    assert idx.min() >= 0 and idx.max() < 40
    w = np.random.default_rng(0).dirichlet(np.ones(30))
    idx = po.sample_weighted_norep(w, 5, seed=9)
    assert len(set(idx)) == 5                             # test/resample.jl:31-34


def test_log_ratio_ordering_identity():
    """test/resample.jl:62-89: ratios[(k-1)N + n] = logp(x_nk) - logpdf(comp_k, x_nk)."""
    d, N, K = 2, 5, 3
    tgt = po.GaussTarget(np.zeros(d), np.ones(d), offset=-d / 2 * np.log(2 * np.pi))
    ratios = []
    for k in range(1, K + 1):
        F = po.Factor(np.ones(d), np.zeros((d, 0)), np.zeros((0, 0)))
        mu = np.full(d, float(k))
        X, logq = F.rand_and_logpdf(mu, po.randn_fill(k, d, N))
        exp = st.multivariate_normal(np.zeros(d), np.eye(d)).logpdf(X.T) - \
            st.multivariate_normal(mu, np.eye(d)).logpdf(X.T)
        got = tgt.logp(X) - F.logpdf(mu, X)
        np.testing.assert_allclose(got, exp, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(tgt.logp(X) - logq, exp, rtol=1e-10, atol=1e-10)
        ratios.append(got)
    assert np.concatenate(ratios).shape == (N * K,)


def test_oracle_single_path_recovers_the_reference_literal_covariance():
    """pins the whole oracle chain (driver -> history -> compact form -> factor -> ELBO argmax) on the reference's own
    known answer: test/singlepath.jl:67-95 -- N(0, Sigma) with the literal 5 x 5 Sigma, history 6, ndraws_elbo = 500,
    fit_distribution.Sigma ~ Sigma with rtol 0.1 (Frobenius, Julia's isapprox)."""
    Sigma = np.array([[2.71, 0.5, 0.19, 0.07, 1.04], [0.5, 1.11, -0.08, -0.17, -0.08], [0.19, -0.08, 0.26, 0.07, -0.7],
                      [0.07, -0.17, 0.07, 0.11, -0.21], [1.04, -0.08, -0.7, -0.21, 8.65]])
    lam, V = np.linalg.eigh(Sigma)
    s2 = 0.5 * lam.min()
    W = V * np.sqrt(lam - s2)                                          # Sigma = s2 I + W W'
    a = np.full(5, 1.0 / s2)
    Wd = W * a[:, None]
    G = np.linalg.inv(np.linalg.cholesky(np.eye(5) + W.T @ Wd))
    tg = po.GaussTarget(np.zeros(5), a, Wd, G)
    P = np.linalg.inv(Sigma)
    x = np.random.default_rng(1).normal(size=(5, 3))
    np.testing.assert_allclose(tg.logp(x), -0.5 * np.einsum("in,ij,jn->n", x, P, x), rtol=1e-11)
    for seed in (38, 5):
        x0 = np.random.default_rng(seed).normal(size=5)
        pts, lps, grads = po.optimize_trace(tg, x0, 6)
        seeds = np.arange(len(pts), dtype=np.uint64) + np.uint64(1000 * seed)
        ref = po.path_fit_elbo(pts, grads, 6, tg, 500, seeds)
        l = ref["best_iter"]
        assert l >= 1
        alpha_all, hl, hs, _ = po.lbfgs_history(pts, grads, 6)
        j = int(hl[l])
        S = np.stack([pts[s + 1] - pts[s] for s in hs[l, :j]], axis=1)
        Y = np.stack([grads[s] - grads[s + 1] for s in hs[l, :j]], axis=1)
        B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
        Sfit = np.diag(alpha_all[l]) + B @ D @ B.T
        assert np.linalg.norm(Sfit - Sigma) <= 0.1 * max(np.linalg.norm(Sfit), np.linalg.norm(Sigma))
