"""GPU parity tests added in round 4 (VERDICT r3 "next" #1, #2):

* BASELINE config 5 at ONE GPU'S FULL SHARE -- 32 paths, maxiters = 1000, d = 10^4, J = 10, N_e = 2000: 32 000 fits, 6.4 x 10^7 ELBO
  draws, ~70 GB of factors, buffers of 6.4 x 10^9 elements, histories a thousand steps long -- with full-size properties, oracle
  comparisons on fits sampled across the WHOLE trace (first / middle / last 20 of 4 paths) and the per-draw log densities of the
  winners;
* the lifetime / error-path fixes of ADVICE r3: a context destroyed before its communicator, closures that raise, the hook table.
All calls go through the C ABI of libpfmi.so.
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from helpers import fit_seeds, oracle_factor_from_gpu, oracle_target
from oracle import pf_oracle as po
import margins as mg
from test_gpu_parity_r2 import _factor, _wc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pfmi_mod():
    import pfmi
    return pfmi


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


# ---- ADVICE r3 -----------------------------------------------------------------------------------------------------------------
def test_context_destroyed_before_its_communicator(pfmi_mod):
    """A host with unordered finalisers (Julia's GC at exit) may destroy a pfmi_ctx before the pfmi_comm that borrows it:
    pfmi_destroy closes the group first, the later pfmi_comm_destroy only frees the shell, calls in between report PFMI_ERR_STATE."""
    L = pfmi_mod.lib()
    ctx, comm = C.c_void_p(), C.c_void_p()
    assert L.pfmi_create(C.c_int32(0), C.byref(ctx)) == 0
    arr = (C.c_void_p * 1)(ctx)
    assert L.pfmi_comm_init_all(C.c_int32(1), arr, C.byref(comm)) == 0
    world = C.c_int32()
    assert L.pfmi_comm_info(comm, C.byref(world), None, None) == 0 and world.value == 1
    assert L.pfmi_destroy(ctx) == 0                                     # the context goes FIRST
    assert L.pfmi_comm_info(comm, C.byref(world), None, None) == -3     # PFMI_ERR_STATE: the communicator is closed
    assert b"destroyed" in L.pfmi_last_error()
    k, m = C.c_double(), C.c_int64()
    assert L.pfmi_comm_pool_psis(comm, C.byref(k), C.byref(m)) == -3
    assert L.pfmi_comm_destroy(comm) == 0                               # no use-after-free: only the shell is left
    # and the usual order still works
    assert L.pfmi_create(C.c_int32(0), C.byref(ctx)) == 0
    arr = (C.c_void_p * 1)(ctx)
    assert L.pfmi_comm_init_all(C.c_int32(1), arr, C.byref(comm)) == 0
    assert L.pfmi_comm_destroy(comm) == 0 and L.pfmi_destroy(ctx) == 0


def test_debug_hook_table(pfmi_mod):
    """pfmi_debug_set: the explicit form of the PFMI_* test hooks (the environment is honoured only under PFMI_DEBUG_HOOKS=1)."""
    L = pfmi_mod.lib()
    assert L.pfmi_debug_set(b"NOT_A_HOOK", b"1") == -1
    d, J = 200, 6
    tg = pfmi_mod.t_diag(d, seed=1)
    eng = pfmi_mod.Engine(0)
    try:
        eng.set_target(tg)
        x0 = pfmi_mod.HostRNG(3).rand(2 * d).reshape(2, d) * 4 - 2
        eng.optimize_batch(x0, J, 30)
        eng.fit_batch(J)
        seeds = fit_seeds(eng.P, 2)
        e0 = eng.elbo_batch(256, seeds)[0]
        assert L.pfmi_debug_set(b"PFMI_ELBO_KERNEL", b"lane") == 0     # another kernel, the same numbers to roundoff
        eng.profile(2)
        e1 = eng.elbo_batch(256, seeds)[0]
        assert L.pfmi_debug_set(b"PFMI_ELBO_KERNEL", None) == 0
        e2 = eng.elbo_batch(256, seeds)[0]
        eng.profile(0)
        f = np.isfinite(e0)
        np.testing.assert_array_equal(e0[f], e2[f])
        assert np.max(np.abs(e1[f] - e0[f]) / (1 + np.abs(e0[f]))) <= 1e-10 and not np.array_equal(e1[f], e0[f])
    finally:
        eng.close()


def test_python_closures_that_raise_are_reraised(pfmi_mod):
    """An exception inside a Python logp closure (host callback or torch device closure) used to be printed and swallowed by ctypes,
    leaving stale memory to be reduced into ELBOs: it now fills its block with NaN and is re-raised by the Engine call."""
    d, J = 20, 4
    calls = {"n": 0}

    def bad(x):
        calls["n"] += 1
        if calls["n"] > 3:
            raise ZeroDivisionError("closure failed")
        return float(-0.5 * (x @ x))

    tg = pfmi_mod.CallbackTarget(d, bad, grad=lambda x: -x)
    good = pfmi_mod.t_iso(d)
    eng = pfmi_mod.Engine(0)
    try:
        tr = pfmi_mod.optimize_with_trace(good, pfmi_mod.HostRNG(1).rand(d) * 4 - 2, history_length=J)
        eng.set_target(tg)
        eng.set_traces([tr.points], [tr.gradients])
        eng.fit_batch(J)
        with pytest.raises(ZeroDivisionError):
            eng.elbo_batch(16, fit_seeds(eng.P, 1))
        assert tg.pending_error is None                                  # consumed: the next call starts clean
        import torch

        def tbad(X):
            raise RuntimeError("torch closure failed")

        tt = pfmi_mod.TorchDeviceTarget(d, tbad, host=good)
        eng.set_target(tt)
        eng.set_traces([tr.points], [tr.gradients])
        eng.fit_batch(J)
        with pytest.raises(RuntimeError, match="torch closure failed"):
            eng.elbo_batch(16, fit_seeds(eng.P, 1))
        assert torch.cuda.is_available()
    finally:
        eng.close()


# ---- dimensions beyond the register kernels (VERDICT r3 missing #5: d > 16 384 was refused) ------------------------------------------
def test_dimension_beyond_16384_runs_the_whole_hot_path(pfmi_mod):
    """d = 20 000: the memory-resident history walk (pf_history_mem_kernel), the column-by-column fit, the streamed ELBO scan and the
    streaming draw writer -- every stage of the hot path -- against the oracle on traces from the host driver (the device L-BFGS stops
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
        # the public mirror picks the host driver at this size instead of failing in the device optimiser
        from pfmi.api import _use_device_optimizer
        assert not _use_device_optimizer(tg, "auto") and _use_device_optimizer(pfmi_mod.t_diag(16384, 1), "auto")
    finally:
        eng.close()


@pytest.mark.parametrize("tname,d,J", [("diag", 700, 6), ("funnel", 3000, 10), ("diag", 12000, 4)])
def test_memory_resident_history_walk_matches_the_register_kernels(pfmi_mod, tname, d, J, monkeypatch):
    """PFMI_HISTORY_KERNEL=mem at sizes the register kernels own: the same accepted updates, ring sources and rejections; alpha to
    roundoff (the memory-resident walk divides like the reference, the register kernels carry 1 / alpha)."""
    tg = pfmi_mod.t_funnel(d) if tname == "funnel" else pfmi_mod.t_diag(d, 1)
    sc = 10.0 if tname == "funnel" else 2.0
    x0 = pfmi_mod.HostRNG(23).rand(2 * d).reshape(2, d) * 2 * sc - sc
    out = {}
    for mode in ("default", "mem"):
        if mode == "mem":
            monkeypatch.setenv("PFMI_HISTORY_KERNEL", "mem")
        else:
            monkeypatch.delenv("PFMI_HISTORY_KERNEL", raising=False)
        e = pfmi_mod.Engine(0)
        try:
            e.set_target(tg)
            e.optimize_batch(x0, J, 40)
            e.fit_batch(J)
            st, je, ld, nr = e.fit_status()
            pts = sorted({1, e.P // 2, e.P - 1})
            out[mode] = (st, je, nr, ld, [e.get_fit(p, int(je[p]))["alpha"] for p in pts])
        finally:
            e.close()
    a, b = out["default"], out["mem"]
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_array_equal(x, y)
    okm = a[0] == 0
    mg.check(f"history mem d={d}", "logdet vs register walk", np.max(np.abs(a[3][okm] - b[3][okm]) / (1 + np.abs(a[3][okm]))), 1e-10)
    for x, y in zip(a[4], b[4]):
        mg.check(f"history mem d={d}", "alpha vs register walk", np.max(np.abs(x - y) / x), 1e-10)


# ---- PSIS pools beyond the LDS tail capacity (VERDICT r3 missing #5: tails > 4095, i.e. S > 1 863 225, were refused) -------------------
@pytest.mark.parametrize("case", ["big_t3", "forced_t4", "forced_ties", "forced_with_inf"])
def test_psis_large_tail_route(pfmi_mod, case, monkeypatch):
    """M + 1 > 4096: every (key, index) pair is sorted in global memory and the tail is fitted on the sorted run
    (pf_psis_bigtail_kernel).  At S = 2.2 x 10^6 (M = 4450) against the oracle; forced at smaller S (PFMI_PSIS_KERNEL=big) against the
    regular route on the inputs that stress the selection (ties at the cutoff, -Inf log ratios) and against the oracle."""
    rng = np.random.default_rng(5)
    if case == "big_t3":
        S = 2_200_000
        lr = rng.standard_t(3, size=S) * 2.0 - 0.5
    elif case == "forced_t4":
        S = 300_000
        lr = rng.standard_t(4, size=S) * 2.0 - 1.0
    elif case == "forced_ties":
        S = 64_000
        lr = np.round(rng.normal(size=S), 1)
    else:
        S = 100_000
        lr = rng.normal(size=S)
        lr[rng.integers(0, S, 80)] = -np.inf
    eng = pfmi_mod.Engine(0)
    try:
        if case != "big_t3":
            b = eng.psis(lr)                                       # regular route
            monkeypatch.setenv("PFMI_PSIS_KERNEL", "big")
        a = eng.psis(lr)
        monkeypatch.delenv("PFMI_PSIS_KERNEL", raising=False)
    finally:
        eng.close()
    lw, w, k, M = po.psis(lr)
    assert a["tail_length"] == M and (case != "big_t3" or M + 1 > 4096)
    cfg = f"PSIS large tail {case} S={S}"
    fin = np.isfinite(lw)
    np.testing.assert_array_equal(np.isfinite(a["log_weights"]), fin)
    mg.check(cfg, "psis_logw", np.max(np.abs(a["log_weights"][fin] - lw[fin]) / (1 + np.abs(lw[fin]))))
    mg.check(cfg, "psis_w", np.max(np.abs(a["weights"] - w)) / np.max(w))
    mg.check(cfg, "pareto_k", abs(a["pareto_shape"] - k) / (1 + abs(k)))
    assert abs(a["weights"].sum() - 1.0) <= 1e-12
    if case != "big_t3":
        assert a["tail_length"] == b["tail_length"]
        assert abs(a["pareto_shape"] - b["pareto_shape"]) <= 1e-13 * (1 + abs(b["pareto_shape"]))
        assert np.max(np.abs(a["log_weights"][fin] - b["log_weights"][fin])) <= 1e-12 * (1 + np.abs(b["log_weights"][fin]).max())


# ---- large results in page-locked memory (include/pfmi.h: pfmi_host_alloc) ------------------------------------------------------------
def test_large_results_arrive_in_page_locked_memory(pfmi_mod, monkeypatch):
    """The draws / pool arrays that the Python host hands to the library are page-locked above 16 MB (one DMA transfer instead of the
    runtime's staged copy): the same bytes as into ordinary memory, the block is recycled when its arrays die."""
    import gc
    from pfmi import core
    d, K, J, N_r = 600, 3, 5, 2000                                            # pool: 600 x 2000 x 3 doubles = 28.8 MB
    tg = pfmi_mod.t_lowrank(d, r=8, seed=4)
    eng = pfmi_mod.Engine(0)
    try:
        eng.set_target(tg)
        x0 = pfmi_mod.HostRNG(5).rand(K * d).reshape(K, d) * 4 - 2
        eng.optimize_batch(x0, J)
        eng.fit_batch(J)
        eng.elbo_batch(64, fit_seeds(eng.P, 3))
        pts = np.array([int(eng.offsets[k]) + 2 for k in range(K)], dtype=np.int64)
        eng.pool_build(N_r, pts, fit_seeds(K, 9))
        monkeypatch.setattr(core, "_PIN_MIN_BYTES", 1 << 62)
        X_plain, lr_plain = eng.pool_get()
        monkeypatch.setattr(core, "_PIN_MIN_BYTES", 16 << 20)
        core._pin_free.clear()
        X_pin, lr_pin = eng.pool_get()
        assert X_pin.flags["F_CONTIGUOUS"] and X_pin.shape == X_plain.shape
        np.testing.assert_array_equal(X_pin, X_plain)
        np.testing.assert_array_equal(lr_pin, lr_plain)
        # the block returns to the pool with its last view and is handed out again
        view = X_pin[:, :10, 0]
        del X_pin
        gc.collect()
        assert sum(len(v) for v in core._pin_free.values()) == 0
        del view
        gc.collect()
        assert sum(len(v) for v in core._pin_free.values()) == 1
        addr = next(v[0] for v in core._pin_free.values() if v)
        Y = core.result_empty((d, N_r, K))
        assert Y.ctypes.data == addr and sum(len(v) for v in core._pin_free.values()) == 0
    finally:
        eng.close()
