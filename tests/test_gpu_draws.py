"""GPU parity of the draws: `rand_and_logpdf` / `unwhiten!` (reference src/mvnormal.jl:24-39, src/woodbury.jl:136-143, 401-406) on identical normals
and on the in-kernel generator (device normals bit-identical to the oracle's), the draw kernels against each other (lane-per-draw, two-pass MFMA,
streaming writer), the reference's 300 000-draw consistency check (test/mvnormal.jl:66-109)."""
import json
import os
import warnings

import numpy as np
import pytest

from helpers import demo_device_target, fit_seeds, make_traces, oracle_factor_from_gpu, oracle_target
from oracle import pf_oracle as po
import margins as mg
from gpu_common import CASES, MIN_STRICT, _setup, _well_conditioned, _with_kernel

pytestmark = pytest.mark.gpu


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
