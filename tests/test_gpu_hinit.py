"""GPU: the `Hinit` keyword of lbfgs_inverse_hessians through the boundary (pfmi_fit_batch_ex / pfmi_set_hinit; reference
src/inverse_hessian.jl:25, forwarded by fit_mvnormals src/mvnormal.jl:14-16).  The reference's own test of it
(test/inverse_hessian.jl:46-76) is a PROPERTY: with Hinit = nocedal_wright_scaling -- the initialisation of the optimiser itself --
every fitted inverse Hessian applied to the gradient is the step the optimiser took, and no update is rejected."""
import numpy as np
import pytest

from helpers import Banana, fit_seeds, make_traces, oracle_target
from oracle import pf_oracle as po
import margins as mg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pfmi_mod():
    import pfmi
    return pfmi


def test_nocedal_wright_hinit_reproduces_the_optimisers_steps_on_the_banana(pfmi_mod):
    """test/inverse_hessian.jl:46-76 on the GPU: banana target, n = 10, history_length = 5, traces of this repo's host L-BFGS (two-loop
    recursion, gamma = s'y / y'y: the initialisation Optim.LBFGS uses); fit_mvnormals(...; Hinit = nocedal_wright) on the device, then
    H * grad through the device factor (pfmi_woodbury_apply(MUL) = mul!(y, W, x), src/woodbury.jl:346-349): cos(H grad, step) = 1 for
    every trace point, 0 rejected updates.  With the default gilbert_init the same check fails -- the switch really switches."""
    pfmi = pfmi_mod
    n, J = 10, 5
    rng = np.random.default_rng(3)
    eng = pfmi.Engine(0)
    total = 0
    for rep in range(3):
        tr = pfmi.optimize_with_trace(Banana(n), 10 * rng.normal(size=n), history_length=J, maxiters=1000)
        P, G = tr.points, tr.gradients
        dists, nrej = pfmi.fit_mvnormals(P, G, history_length=J, engine=eng, Hinit="nocedal_wright")
        assert nrej == 0                                       # test/inverse_hessian.jl:75
        cos = []
        for l in range(len(tr) - 1):
            p = dists[l].Sigma.mul(G[l][:, None])[:, 0]
            step = P[l + 1] - P[l]
            cos.append((p @ step) / np.linalg.norm(p) / np.linalg.norm(step))
        mg.check("hinit:banana", "1 - cos(H grad, step)", np.max(np.abs(np.array(cos) - 1.0)), bound=1e-8)
        total += len(cos)
        # against the oracle's walk with the same switch: alpha bit for bit up to the last ulp, j_eff / status array-equal
        try:
            po.set_hinit("nocedal_wright")
            alpha_all, hl, hs, rej = po.lbfgs_history(P, G, J)
        finally:
            po.set_hinit("gilbert")
        status, jeff, logdet, nr = eng.fit_status()
        np.testing.assert_array_equal(jeff, hl)
        assert rej == 0 and np.all(status == 0)
        for l in (0, 1, len(tr) // 2, len(tr) - 1):
            np.testing.assert_allclose(eng.get_fit(l, int(jeff[l]))["alpha"], alpha_all[l], rtol=1e-14)
        # the default Hinit gives another H: the property does NOT hold (so the check above is not vacuous)
        d0, _ = pfmi.fit_mvnormals(P, G, history_length=J, engine=eng)
        l = len(tr) // 2
        p0 = d0[l].Sigma.mul(G[l][:, None])[:, 0]
        step = P[l + 1] - P[l]
        assert abs((p0 @ step) / np.linalg.norm(p0) / np.linalg.norm(step) - 1.0) > 1e-6
    assert total > 30
    eng.close()


@pytest.mark.parametrize("tname,d,J", [("lr", 50, 6), ("diag", 700, 4), ("funnel", 3000, 10), ("diag", 12000, 4)])
def test_hinit_switch_in_every_history_kernel_against_the_oracle(pfmi_mod, tname, d, J, monkeypatch):
    """the three history walks (register sets d <= 2048, lean 2048 < d <= 10 240, register sets beyond; memory-resident via the hook) with
    Hinit = nocedal_wright against the oracle's walk with the same switch: status / j_eff / rejected array-equal, alpha, logdet, mu within
    SURVEY 8(d)'s tolerances; and pfmi_set_hinit (the context default) equals pfmi_fit_batch_ex (the per-call keyword)."""
    pfmi = pfmi_mod
    tg = {"lr": lambda: pfmi.t_lowrank(d, r=8, seed=2), "diag": lambda: pfmi.t_diag(d, seed=1), "funnel": lambda: pfmi.t_funnel(d)}[tname]()
    traces = make_traces(tg, 2, 5, scale=10.0 if tname == "funnel" else 2.0, history_length=J, maxiters=14 if d > 1000 else 60)
    otg = oracle_target(tg)
    eng = pfmi.Engine(0)
    eng.set_target(tg)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    outs = {}
    for mode in ("default", "mem"):
        if mode == "mem":
            monkeypatch.setenv("PFMI_HISTORY_KERNEL", "mem")
        eng.fit_batch(J, hinit="nocedal_wright")
        status, jeff, logdet, nrej = eng.fit_status()
        outs[mode] = (status.copy(), jeff.copy(), logdet.copy(), nrej.copy())
        monkeypatch.delenv("PFMI_HISTORY_KERNEL", raising=False)
        try:
            po.set_hinit("nocedal_wright")
            for k, tr in enumerate(traces):
                p0, P = int(eng.offsets[k]), len(tr)
                ref = po.path_fit_elbo(tr.points, tr.gradients, J, otg, 0, np.zeros(P, dtype=np.uint64))
                alpha_all = po.lbfgs_history(tr.points, tr.gradients, J)[0]
                np.testing.assert_array_equal(status[p0:p0 + P], ref["status"])
                np.testing.assert_array_equal(jeff[p0:p0 + P], ref["j_eff"])
                assert nrej[k] == ref["n_rejected"]
                ok = ref["status"] == 0
                mg.check(f"hinit:{tname}{d}:{mode}", "logdet", mg.rel(logdet[p0:p0 + P][ok], ref["logdet"][ok]))
                for l in (1, P // 2, P - 1):
                    f = eng.get_fit(p0 + l, int(jeff[p0 + l]))
                    np.testing.assert_allclose(f["alpha"], alpha_all[l], rtol=1e-13)
                    assert np.ptp(f["alpha"]) == 0.0                # a SCALAR diagonal: fill(y's / y'y)
                    if ok[l]:
                        mg.check(f"hinit:{tname}{d}:{mode}", "mu", np.max(np.abs(f["mu"] - ref["mu"][l])) / (1 + np.abs(ref["mu"][l]).max()))
        finally:
            po.set_hinit("gilbert")
    for a, b in zip(outs["default"], outs["mem"]):
        np.testing.assert_array_equal(a, b) if a.dtype.kind == "i" else np.testing.assert_allclose(a, b, rtol=1e-12, equal_nan=True)
    # the context default: the same bits as the keyword
    eng.set_hinit("nocedal_wright")
    eng.fit_batch(J)
    s2, j2, l2, n2 = eng.fit_status()
    eng.set_hinit("gilbert")
    np.testing.assert_array_equal(s2, outs["default"][0]); np.testing.assert_array_equal(j2, outs["default"][1])
    np.testing.assert_array_equal(l2, outs["default"][2])
    eng.fit_batch(J)
    l3 = eng.fit_status()[2]
    assert not np.array_equal(l3[np.isfinite(l3)][2:], l2[np.isfinite(l2)][2:])      # gilbert_init again: different fits
    eng.close()


def test_streamed_pipeline_honours_the_context_hinit(pfmi_mod):
    """pfmi_set_hinit + pfmi_stream_enqueue: the streamed dataflow (segmented walk, the state handed from segment to segment) gives the
    bits of the packed route pfmi_optimize_batch ; pfmi_fit_batch_ex(.., nocedal_wright) ; pfmi_elbo_batch on the same starting points."""
    pfmi = pfmi_mod
    d, K, J, N, maxit = 120, 5, 6, 128, 200
    tg = pfmi.t_lowrank(d, r=8, seed=2)
    x0 = pfmi.HostRNG(4).rand(K * d).reshape(K, d) * 4 - 2
    cap = maxit + 1
    tab = pfmi.hostrng.rand_u64(21, np.arange(K * cap, dtype=np.uint64), 9)
    eng = pfmi.Engine(0)
    eng.set_target(tg)
    npts = eng.optimize_batch(x0, J, maxit)
    eng.fit_batch(J, hinit="nocedal_wright")
    sp = np.concatenate([np.concatenate([[np.uint64(0)], tab[k * cap:k * cap + int(npts[k]) - 1]]) for k in range(K)]).astype(np.uint64)
    ep, sep, bp = eng.elbo_batch(N, sp)
    stp = eng.fit_status()
    eng.set_hinit("nocedal_wright")
    eng.stream_enqueue(x0, N, tab, J, maxit)
    ns = eng.stream_wait()
    es, ses, bs = eng.elbo_batch_wait()
    sts = eng.fit_status()
    eng.set_hinit("gilbert")
    assert np.array_equal(ns, npts) and np.array_equal(bs, bp)
    off = np.concatenate([[0], np.cumsum(npts)])
    for k in range(K):
        sl = slice(k * cap, k * cap + int(npts[k]))
        pk = slice(int(off[k]), int(off[k + 1]))
        np.testing.assert_array_equal(sts[0][sl], stp[0][pk])
        np.testing.assert_array_equal(sts[1][sl], stp[1][pk])
        np.testing.assert_array_equal(sts[2][sl], stp[2][pk])
        np.testing.assert_array_equal(es[sl][1:], ep[pk][1:])
    eng.close()
