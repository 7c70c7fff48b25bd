"""Pins the CPU oracle's linear algebra against the reference's own fixtures and identities.

Mirrors /root/reference/test/inverse_hessian.jl:8-44 (literal S0/Y0 fixture + explicit dense Byrd
formula, four history layouts) and /root/reference/test/woodbury.jl:18-404 (every factor operation
against dense algebra, including n < m), plus an independent LAPACK check of the Householder
convention through SciPy (dgeqrf, the convention Julia's `qr` uses).
"""
import json
import os

import numpy as np
import pytest
import scipy.linalg as sla

from oracle import pf_oracle as po


def explicit_byrd(alpha, S, Y):
    """lbfgs_inverse_hessian_explicit, test/inverse_hessian.jl:8-14"""
    H0 = np.diag(alpha)
    B = np.hstack([H0 @ Y, S])
    R = np.triu(S.T @ Y)
    E = np.diag(np.diag(R))
    Rinv = np.linalg.inv(R)
    j = S.shape[1]
    D = np.block([[np.zeros((j, j)), -Rinv], [-Rinv.T, Rinv.T @ (E + Y.T @ H0 @ Y) @ Rinv]])
    return H0 + B @ D @ B.T


@pytest.fixture(scope="module")
def s0y0(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "lbfgs_S0Y0.json")))
    S = np.array(g["S0_columns"]).T.copy()
    Y = np.array(g["Y0_columns"]).T.copy()
    return S, Y


def test_byrd_compact_matches_explicit_on_reference_fixture(s0y0):
    S, Y = s0y0
    n, J = S.shape
    rng = np.random.default_rng(0)
    alpha = rng.uniform(0.1, 1.0, n)
    # empty history (test/inverse_hessian.jl:28)
    B, D = po.lbfgs_inverse_hessian(alpha, S[:, :0], Y[:, :0])
    assert B.shape == (n, 0) and D.shape == (0, 0)
    # partial history (:30-32)
    B, D = po.lbfgs_inverse_hessian(alpha, S[:, :3], Y[:, :3])
    H = np.diag(alpha) + B @ D @ B.T
    np.testing.assert_allclose(H, explicit_byrd(alpha, S[:, :3], Y[:, :3]), rtol=1e-12, atol=1e-13)
    # full history (:41-43)
    B, D = po.lbfgs_inverse_hessian(alpha, S, Y)
    np.testing.assert_allclose(np.diag(alpha) + B @ D @ B.T, explicit_byrd(alpha, S, Y), rtol=1e-11, atol=1e-12)
    assert B.shape == (n, 2 * J)
    np.testing.assert_allclose(B[:, :J], alpha[:, None] * Y)
    np.testing.assert_allclose(B[:, J:], S)
    np.testing.assert_allclose(D[:J, :J], 0)
    np.testing.assert_allclose(D, D.T, atol=1e-9 * np.abs(D).max())


def test_ring_rotated_history_is_reordered_oldest_first(s0y0):
    """test/inverse_hessian.jl:34-39: a rotated ring buffer gives the same H.  We drive the
    trace walk (src/inverse_hessian.jl:43-63) so that the ring wraps and check hist_src."""
    S, Y = s0y0
    n, nh = S.shape
    # build a synthetic trace whose steps are exactly S0 columns / Y0 columns
    theta = np.zeros((nh + 1, n))
    grad = np.zeros((nh + 1, n))
    for l in range(nh):
        theta[l + 1] = theta[l] + S[:, l]
        grad[l + 1] = grad[l] - Y[:, l]
    J = 3
    alpha_all, hist_len, hist_src, rej = po.lbfgs_history(theta, grad, J)
    assert rej == 0
    assert list(hist_len) == [0, 1, 2, 3, 3, 3]
    assert list(hist_src[3, :3]) == [0, 1, 2]
    assert list(hist_src[4, :3]) == [1, 2, 3]      # ring wrapped: oldest first
    assert list(hist_src[5, :3]) == [2, 3, 4]
    # alpha recurrence = repeated gilbert_init (src/inverse_hessian.jl:55)
    a = np.ones(n)
    for l in range(nh):
        a = po.gilbert_init(a, S[:, l], Y[:, l])
        np.testing.assert_allclose(alpha_all[l + 1], a, rtol=1e-14)
    # H at the last point equals the explicit formula on the last J pairs
    B, D = po.lbfgs_inverse_hessian(alpha_all[5], S[:, 2:5], Y[:, 2:5])
    np.testing.assert_allclose(np.diag(alpha_all[5]) + B @ D @ B.T,
                               explicit_byrd(alpha_all[5], S[:, 2:5], Y[:, 2:5]), rtol=1e-10, atol=1e-12)


def test_curvature_rejection_counts_and_keeps_state():
    """src/inverse_hessian.jl:47-58: y.s <= eps |y|^2 -> rejected, history/alpha unchanged."""
    rng = np.random.default_rng(3)
    n = 6
    theta = rng.normal(size=(4, n))
    grad = -theta.copy()            # iso normal: y = s  -> accepted
    grad[2] = grad[1] + (theta[2] - theta[1])   # y = -s on step 2 -> rejected
    alpha_all, hist_len, hist_src, rej = po.lbfgs_history(theta, grad, 5)
    assert rej >= 1
    assert hist_len[2] == hist_len[1]
    np.testing.assert_array_equal(alpha_all[2], alpha_all[1])


def test_gilbert_init_formula():
    rng = np.random.default_rng(1)
    a, s, y = rng.uniform(0.5, 2, 7), rng.normal(size=7), rng.normal(size=7)
    aa = np.sum(y * a * y); b = y @ s; c = np.sum(s * s / a)
    exp = b / (aa / a + y**2 - (aa / c) * (s / a) ** 2)
    np.testing.assert_allclose(po.gilbert_init(a, s, y), exp, rtol=1e-14)


@pytest.mark.parametrize("n,m", [(10, 8), (5, 8), (8, 8), (30, 12), (3, 12)])
def test_householder_matches_lapack_dgeqrf(n, m):
    rng = np.random.default_rng(n * 100 + m)
    A = rng.normal(size=(n, m))
    QR, tau = po.householder_qr(A)
    (qr_raw, tau_l), _ = sla.qr(A, mode="raw")
    np.testing.assert_allclose(QR, qr_raw, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(tau, tau_l, rtol=1e-12, atol=1e-14)


def rand_pd(rng, n):
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    return (Q * rng.uniform(0.05, 1.0, n)) @ Q.T


@pytest.mark.parametrize("n", [5, 10])
def test_woodbury_factor_identities(n):
    """test/woodbury.jl:18-115,217-273,311-402 with m = 8 (n = 5 covers n < m, k = min(n, m))."""
    m = 8
    rng = np.random.default_rng(n)
    alpha = rng.uniform(0.1, 1.0, n)
    B = rng.normal(size=(n, m))
    D = rand_pd(rng, m)
    F = po.Factor(alpha, B, D)
    assert F.status == 0 and F.k == min(n, m)
    W = np.diag(alpha) + B @ D @ B.T
    I = np.eye(n)
    Rm = F.lmul_R(I)                      # Matrix(R) = lmul!(R, I)  (src/woodbury.jl:104-106)
    Lm = F.lmul_L(I)
    np.testing.assert_allclose(Lm, Rm.T, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(Rm.T @ Rm, W, rtol=1e-11, atol=1e-12)          # W = R'R
    np.testing.assert_allclose(F.ldiv_R(Rm), I, atol=1e-11)                   # R \ R
    np.testing.assert_allclose(F.ldiv_L(Lm), I, atol=1e-11)
    X = rng.normal(size=(n, 7))
    np.testing.assert_allclose(F.lmul_R(X), Rm @ X, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(F.lmul_L(X), Lm @ X, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(F.ldiv_R(X), np.linalg.solve(Rm, X), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(F.ldiv_L(X), np.linalg.solve(Lm, X), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(F.mul_W(X), W @ X, rtol=1e-11, atol=1e-12)     # mul!(y, W, x)
    x = rng.normal(size=n)
    np.testing.assert_allclose(F.mul_W(x), W @ x, rtol=1e-11, atol=1e-12)
    sign, ld = np.linalg.slogdet(W)
    assert sign > 0
    assert abs(F.logdet - ld) < 1e-10                                          # logdet :217-222
    # invquad / quad (:367-402)
    np.testing.assert_allclose(np.sum(F.ldiv_L(X) ** 2, axis=0), np.einsum("ij,ij->j", X, np.linalg.solve(W, X)),
                               rtol=1e-9)
    np.testing.assert_allclose(np.sum(F.lmul_R(X) ** 2, axis=0), np.einsum("ij,ij->j", X, W @ X), rtol=1e-10)


def test_factor_matches_scipy_pipeline():
    """pdfactorize (src/woodbury.jl:201-207) step by step through SciPy/LAPACK."""
    rng = np.random.default_rng(11)
    n, m = 12, 6
    alpha = rng.uniform(0.2, 2.0, n)
    B = rng.normal(size=(n, m))
    D = rand_pd(rng, m)
    F = po.Factor(alpha, B, D)
    U = np.sqrt(alpha)
    (qr_raw, tau), R = sla.qr(B / U[:, None], mode="raw")
    np.testing.assert_allclose(F.QR, qr_raw, rtol=1e-12, atol=1e-13)
    Cm = np.eye(m) + np.triu(qr_raw[:m]) @ D @ np.triu(qr_raw[:m]).T
    V = sla.cholesky(Cm, lower=False)
    np.testing.assert_allclose(F.V, V, rtol=1e-11, atol=1e-12)
    # lmul!(Q, x) through dormqr
    x = rng.normal(size=(n, 3))
    Qx, = sla.lapack.dormqr("L", "N", qr_raw, tau, np.asfortranarray(x), 3 * n)[:1]
    z = x.copy(order="F")
    po.lib().pfo_apply_q(n, m, po._p(F.QR), po._p(F.tau), 0, po._p(z), 3)
    np.testing.assert_allclose(z, Qx, rtol=1e-12, atol=1e-13)


def test_not_pd_status():
    n, m = 6, 2
    alpha = np.ones(n); alpha[2] = -1.0
    F = po.Factor(alpha, np.zeros((n, m)), np.zeros((m, m)))
    assert F.status == 1                                  # A not PD  (src/woodbury.jl:202)
    B = np.zeros((n, m)); B[0, 0] = 1.0; B[1, 1] = 1.0
    D = -5.0 * np.eye(m)
    F = po.Factor(np.ones(n), B, D)
    assert F.status == 2                                  # C = I + R D R' not PD (:205)


def test_fit_mean_is_theta_plus_sigma_grad():
    """test/mvnormal.jl:28: mu = theta + Sigma * grad."""
    rng = np.random.default_rng(5)
    n, m = 9, 4
    alpha = rng.uniform(0.2, 2.0, n)
    B = rng.normal(size=(n, m)); D = rand_pd(rng, m)
    F = po.Factor(alpha, B, D)
    th, g = rng.normal(size=n), rng.normal(size=n)
    np.testing.assert_allclose(F.fit_mean(th, g), th + F.dense() @ g, rtol=1e-11, atol=1e-12)


def test_oracle_lbfgs_driver_matches_host_driver_and_converges():
    """pfo_optimize_trace (the checker of the device optimiser) against the independent numpy driver
    pfmi/optimize.py and against scipy's finite-difference gradient check; Optim's own trajectory is third party
    (parity unpinned)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pathfinder.jl_amd"))
    os.environ.setdefault("PFMI_NO_TORCH", "1")
    from pfmi import targets, optimize, hostrng
    from helpers import oracle_target
    from scipy.optimize import check_grad
    for tg, scale in [(targets.t_iso(10), 2), (targets.t_diag(30, 1), 2), (targets.t_lowrank(50, 8, 2), 2), (targets.t_funnel(12), 3)]:
        ot = oracle_target(tg)
        x0 = hostrng.HostRNG(3).rand(tg.d) * 2 * scale - scale
        err = check_grad(lambda x: po.logp_grad(ot, x)[0], lambda x: po.logp_grad(ot, x)[1], x0)
        assert err < 1e-4 * max(1.0, np.abs(po.logp_grad(ot, x0)[1]).max())
        if tg.kind == 1:
            continue                                      # funnel: unbounded, trajectories are chaotic
        P, L, G = po.optimize_trace(ot, x0, 6)
        tr = optimize.optimize_with_trace(tg, x0, history_length=6)
        n = min(len(P), len(tr), 10)
        np.testing.assert_allclose(P[:n], tr.points[:n], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(G[:n], tr.gradients[:n], rtol=1e-8, atol=1e-10)
        assert np.abs(G[-1]).max() <= 1e-8 and np.all(np.diff(L) >= -1e-12)
        for l in (0, len(P) // 2, len(P) - 1):            # the recorded (logp, grad) belong to the recorded point
            lp, g = po.logp_grad(ot, P[l])
            assert abs(lp - L[l]) <= 1e-12 * max(1, abs(lp)) and np.allclose(g, G[l], rtol=1e-12, atol=1e-14)


def test_inverse_hessian_reproduces_the_lbfgs_direction_on_the_banana():
    """reference test/inverse_hessian.jl:46-77: with the optimiser's own initialisation H0 = (y's / y'y) I (`nocedal_wright_scaling`)
    the compact-form inverse Hessian of every trace point, applied to the gradient, is the direction the L-BFGS optimiser actually
    took: dot(p, s) / (|p| |s|) = 1, and no update is rejected.  The trace comes from this repo's host driver (two-loop recursion,
    the same initialisation); the inverse Hessian from the oracle's restatement of src/inverse_hessian.jl:98-133."""
    from pfmi.optimize import optimize_with_trace
    from helpers import Banana
    n, J = 10, 5
    rng = np.random.default_rng(7)
    total = 0
    for _ in range(4):
        tr = optimize_with_trace(Banana(n), 10 * rng.normal(size=n), J, 1000)
        total += len(tr)
        _check_directions(tr, n, J)
    assert total > 40


def test_oracle_hinit_switch_is_the_nocedal_wright_walk():
    """The oracle's `Hinit` switch (pfo_set_hinit(1): alpha = fill(y's / y'y) at every accepted step, test/inverse_hessian.jl:49) against
    a literal NumPy walk of src/inverse_hessian.jl:25-66 with that Hinit, on banana traces; and the reference's property itself through the
    oracle's own walk + compact form + factor: H_l * grad_l is parallel to the step the optimiser took (test/inverse_hessian.jl:62-76)."""
    from helpers import Banana
    from pfmi.optimize import optimize_with_trace
    n, J = 10, 5
    rng = np.random.default_rng(11)
    try:
        po.set_hinit("nocedal_wright")
        for _ in range(3):
            tr = optimize_with_trace(Banana(n), 10 * rng.normal(size=n), J, 1000)
            P, G = tr.points, tr.gradients
            alpha_all, hist_len, hist_src, rej = po.lbfgs_history(P, G, J)
            assert rej == 0
            alpha = np.ones(n)
            for l in range(1, len(tr)):
                s, y = P[l] - P[l - 1], G[l - 1] - G[l]
                if y @ s > 1e-12 * (y @ y):
                    alpha = np.full(n, (y @ s) / (y @ y))
                np.testing.assert_allclose(alpha_all[l], alpha, rtol=1e-15)
            for l in range(1, len(tr) - 1):
                j = hist_len[l]
                src = hist_src[l, :j]
                S = np.stack([P[q + 1] - P[q] for q in src], axis=1)
                Y = np.stack([G[q] - G[q + 1] for q in src], axis=1)
                B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
                F = po.Factor(alpha_all[l], B, D)
                p = F.mul_W(G[l][:, None])[:, 0]
                step = P[l + 1] - P[l]
                assert abs((p @ step) / np.linalg.norm(p) / np.linalg.norm(step) - 1) < 1e-8
    finally:
        po.set_hinit("gilbert")
    a0 = po.lbfgs_history(P, G, J)[0]                      # back to gilbert_init: a different diagonal
    assert not np.allclose(a0[-1], alpha_all[-1])


def _check_directions(tr, n, J):
    P, G = tr.points, tr.gradients
    S, Y, nrej, cosines = [], [], 0, []
    alpha = np.ones(n)                                      # H0 = I before the first update (src/inverse_hessian.jl:38-39)
    for l in range(len(tr) - 1):
        if l > 0:
            s, y = P[l] - P[l - 1], G[l - 1] - G[l]         # :45-46
            if y @ s > 1e-12 * (y @ y):                     # :47
                S.append(s); Y.append(y)
                S, Y = S[-J:], Y[-J:]
                alpha = np.full(n, (y @ s) / (y @ y))       # nocedal_wright_scaling
            else:
                nrej += 1
        H = np.diag(alpha)
        if S:
            B, D = po.lbfgs_inverse_hessian(alpha, np.array(S).T, np.array(Y).T)
            H = H + B @ D @ B.T
        p, step = H @ G[l], P[l + 1] - P[l]
        cosines.append((p @ step) / np.linalg.norm(p) / np.linalg.norm(step))
    assert nrej == 0
    np.testing.assert_allclose(cosines, 1.0, rtol=0, atol=1e-8)


def test_lapack_baseline_leg_equals_oracle_on_same_normals():
    """tests/cpu_lapack_baseline.py (bench.py's `kind: "lapack"` CPU leg: dgeqrf / dormqr / dtrmm on d x N blocks) against the scalar
    oracle on identical standard normals: same Householder convention => same draws, logq, mean, logdet."""
    import cpu_lapack_baseline as cl
    import pfmi
    from helpers import make_traces
    tg = pfmi.t_lowrank(40, r=4, seed=3)
    tr = make_traces(tg, 1, 5)[0]
    J, N = 6, 64
    alpha_all, hl, hs, _ = po.lbfgs_history(tr.points, tr.gradients, J)
    rng = np.random.default_rng(0)
    nchk = 0
    for l in (1, 2, len(tr) // 2, len(tr) - 1):
        j = int(hl[l])
        S = np.stack([tr.points[s + 1] - tr.points[s] for s in hs[l, :j]], axis=1)
        Y = np.stack([tr.gradients[s] - tr.gradients[s + 1] for s in hs[l, :j]], axis=1)
        B, D = po.lbfgs_inverse_hessian(alpha_all[l], S, Y)
        Fo = po.Factor(alpha_all[l], B, D)
        Fl = cl.factor(alpha_all[l], B, D)
        assert abs(Fl["logdet"] - Fo.logdet) <= 1e-10 * (1 + abs(Fo.logdet))
        mu_o = Fo.fit_mean(tr.points[l], tr.gradients[l])
        mu_l = cl.fit_mean(Fl, tr.points[l], tr.gradients[l])
        np.testing.assert_allclose(mu_l, mu_o, rtol=1e-9, atol=1e-10)
        U = np.asfortranarray(rng.standard_normal((tg.d, N)))
        Xo, lqo = Fo.rand_and_logpdf(mu_o, U)
        Xl, lql = cl.rand_and_logpdf(Fl, mu_l, U.copy(order="F"))
        np.testing.assert_allclose(lql, lqo, rtol=1e-10, atol=1e-10)
        W = Fo.dense()
        # the draws agree up to the sign convention of the reflectors only when R's diagonal is well separated from 0; the law always does
        np.testing.assert_allclose(np.cov(Xl), np.cov(Xo), atol=0.5 * np.abs(W).max())
        if np.abs(np.diag(Fo.QR[:Fo.k, :Fo.k])).min() > 1e-6 * np.abs(np.diag(Fo.QR[:Fo.k, :Fo.k])).max():
            np.testing.assert_allclose(Xl, Xo, rtol=1e-7, atol=1e-8)
            nchk += 1
    assert nchk >= 2
    elbo, nd = cl.path_elbo(tr.points, tr.gradients, J, tg, 200, 1, nfits=5)
    assert nd == 5 * 200 and np.all(np.isfinite(elbo[1:6]))
