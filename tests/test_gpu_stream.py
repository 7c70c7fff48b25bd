"""GPU tests of the streaming pipeline (pfmi_stream_enqueue / pfmi_stream_wait, round 5):

optimise + fit + ELBO scan as ONE enqueued dataflow -- the fits and scans of the trace points a path has already produced run while
the paths are still being optimised -- must be BIT-IDENTICAL to the packed route
    pfmi_optimize_batch ; pfmi_fit_batch ; pfmi_elbo_batch_enqueue
(reference src/singlepath.jl:285-325 per run, src/multipath.jl:190-208 over runs); only the layout differs: trace point l of path k is
slot k * (maxiters + 1) + l.  All calls go through the C ABI of libpfmi.so.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pfmi_mod():
    import pfmi
    return pfmi


def _packed(pfmi, tg, x0, J, maxiters, N, sd_stride, N_r, ndraws):
    K = x0.shape[0]
    cap = maxiters + 1
    e = pfmi.Engine(0)
    e.set_target(tg)
    npts = e.optimize_batch(x0, J, maxiters)
    e.fit_batch(J)
    # fit l of run k uses value l - 1 of the run's predrawn stream; a failed run the value behind the L it consumed
    seeds = np.concatenate([np.concatenate([[np.uint64(0)], sd_stride[k * cap:k * cap + int(npts[k]) - 1]]) for k in range(K)]).astype(np.uint64)
    e.elbo_batch_enqueue(N, seeds)
    fail = np.array([sd_stride[k * cap + int(npts[k]) - 1] for k in range(K)], dtype=np.uint64)
    e.pool_build_best(N_r, fail)
    comm = pfmi.Comm.init_all([e])
    res, idx, draws = comm.psis_resample(ndraws, seed=9)
    status, jeff, logdet, nrej = e.fit_status()
    elbo, se, best = e.elbo_batch_wait()
    traces = [e.get_trace(k) for k in range(K)]
    fit = e.get_fit(int(e.offsets[0]) + int(best[0]), int(jeff[int(e.offsets[0]) + int(best[0])])) if best[0] > 0 else None
    out = dict(npts=npts, off=e.offsets.copy(), status=status, jeff=jeff, logdet=logdet, nrej=nrej, elbo=elbo, se=se, best=best,
               res=res, idx=idx, draws=draws, traces=traces, fit=fit)
    comm.close(); e.close()
    return out


def _streamed(pfmi, tg, x0, J, maxiters, N, sd_stride, N_r, ndraws, repeat=1):
    K = x0.shape[0]
    e = pfmi.Engine(0)
    e.set_target(tg)
    outs = []
    for _ in range(repeat):
        e.stream_enqueue(x0, N, sd_stride, J, maxiters)
        npts = e.stream_wait()                 # schedules the pipeline until the last segment is out; does not wait for the GPU
        e.pool_build_best(N_r, None)
        comm = pfmi.Comm.init_all([e])
        res, idx, draws = comm.psis_resample(ndraws, seed=9)
        status, jeff, logdet, nrej = e.fit_status()
        elbo, se, best = e.elbo_batch_wait()
        traces = [e.get_trace(k) for k in range(K)]
        fit = e.get_fit(int(e.offsets[0]) + int(best[0]), int(jeff[int(e.offsets[0]) + int(best[0])])) if best[0] > 0 else None
        outs.append(dict(npts=npts, off=e.offsets.copy(), status=status, jeff=jeff, logdet=logdet, nrej=nrej, elbo=elbo, se=se, best=best,
                         res=res, idx=idx, draws=draws, traces=traces, fit=fit))
        comm.close()
    e.close()
    return outs


def _compare(a, s, K, cap):
    """a: packed route, s: streamed (fixed stride)"""
    np.testing.assert_array_equal(a["npts"], s["npts"])
    np.testing.assert_array_equal(a["nrej"], s["nrej"])
    np.testing.assert_array_equal(a["best"], s["best"])
    for k in range(K):
        n = int(a["npts"][k])
        pa, ps = int(a["off"][k]), k * cap
        for name in ("status", "jeff", "logdet", "elbo", "se"):
            np.testing.assert_array_equal(a[name][pa:pa + n], s[name][ps:ps + n], err_msg=f"{name} path {k}")
        assert np.all(s["status"][ps + n:ps + cap] == 4), "slots a path never reached carry PFMI_FIT_ABSENT"
        assert np.all(np.isnan(s["elbo"][ps + n:ps + cap]))
        for i in range(3):
            np.testing.assert_array_equal(a["traces"][k][i], s["traces"][k][i])
    np.testing.assert_array_equal(a["idx"], s["idx"])
    np.testing.assert_array_equal(a["draws"], s["draws"])
    np.testing.assert_equal(a["res"]["pareto_shape"], s["res"]["pareto_shape"])       # (NaN == NaN here: a pool of 64 draws has no tail to fit)
    assert a["res"]["tail_length"] == s["res"]["tail_length"]
    if a["fit"] is not None:
        for name in ("alpha", "B", "D", "qr_factors", "T", "V", "mu"):
            np.testing.assert_array_equal(a["fit"][name], s["fit"][name], err_msg=name)
        assert a["fit"]["logdet"] == s["fit"]["logdet"]


CASES = [
    # name, target, K, J, maxiters, N, init scale
    ("lr64", lambda m: m.t_lowrank(64, r=8, seed=2), 4, 6, 60, 256, 2.0),
    ("diag100", lambda m: m.t_diag(100, seed=1), 8, 6, 1000, 1000, 2.0),
    ("lr1000", lambda m: m.t_lowrank(1000, r=8, seed=2), 8, 6, 1000, 1000, 2.0),
    ("funnel50_j10", lambda m: m.t_funnel(50), 5, 10, 300, 200, 10.0),
    ("iso10_short", lambda m: m.t_iso(10), 3, 6, 20, 64, 2.0),
    ("funnel3000_j10", lambda m: m.t_funnel(3000), 4, 10, 80, 256, 10.0),         # d > 1024: panel fit, streamed factor blocks (KC = 20)
    ("diag16_many", lambda m: m.t_diag(16, seed=1), 300, 6, 40, 64, 2.0),      # more paths than CUs: nothing on the device waits for anything
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_streamed_pipeline_is_bit_identical_to_the_packed_route(pfmi_mod, case):
    name, mk, K, J, maxiters, N, scale = case
    tg = mk(pfmi_mod)
    d = tg.d
    cap = maxiters + 1
    x0 = pfmi_mod.HostRNG(31).rand(K * d).reshape(K, d) * 2 * scale - scale
    if name.startswith("iso"):
        x0[0] = 0.0                             # already at the optimum: one trace point, L = 0 -- a FAILED run (src/singlepath.jl:299)
    sd = pfmi_mod.hostrng.rand_u64(77, np.arange(K * cap, dtype=np.uint64), 9)
    a = _packed(pfmi_mod, tg, x0, J, maxiters, N, sd, N, 200)
    assert int(a["npts"].min()) >= 1
    s1, s2 = _streamed(pfmi_mod, tg, x0, J, maxiters, N, sd, N, 200, repeat=2)
    _compare(a, s1, K, cap)
    _compare(a, s2, K, cap)                     # the same context again: every buffer / event / flag is reusable


def test_stream_refuses_what_it_cannot_run(pfmi_mod):
    e = pfmi_mod.Engine(0)
    tg = pfmi_mod.t_diag(16, seed=1)
    e.set_target(tg)
    with pytest.raises(pfmi_mod._lib.PfmiError):  # history_length beyond the tuned kernels
        e.stream_enqueue(np.zeros((2, 16)), 16, np.zeros(2 * 11, dtype=np.uint64), 20, 10)
    cb = pfmi_mod.CallbackTarget(16, lambda x: float(-0.5 * (x @ x)))
    e.set_target(cb)
    with pytest.raises(pfmi_mod._lib.PfmiError):  # a host closure cannot be optimised on the device
        e.stream_enqueue(np.zeros((2, 16)), 16, np.zeros(2 * 11, dtype=np.uint64), 6, 10)
    e.close()


def test_seeds_handed_over_later_and_deferred_downloads_equal_the_plain_calls(pfmi_mod):
    """pfmi_stream_enqueue(seeds = NULL) + pfmi_stream_seeds: the optimiser starts before the host has drawn the streams;
    pfmi_comm_psis_resample_enqueue / _wait with pfmi_defer_downloads in between: fit statuses, ELBO table and PSIS weights arrive with the ONE
    wait of the pooled stage.  Same bits as the plain sequence."""
    tg = pfmi_mod.t_lowrank(200, r=8, seed=2)
    K, J, maxiters, N = 6, 6, 80, 512
    cap = maxiters + 1
    x0 = pfmi_mod.HostRNG(5).rand(K * 200).reshape(K, 200) * 4 - 2
    sd = pfmi_mod.hostrng.rand_u64(78, np.arange(K * cap, dtype=np.uint64), 9)
    a = _streamed(pfmi_mod, tg, x0, J, maxiters, N, sd, N, 300)[0]          # (np.empty arrays: a cancelled deferral leaves them as they were)
    e = pfmi_mod.Engine(0)
    e.set_target(tg)
    comm = pfmi_mod.Comm.init_all([e])
    for rep in range(2):
        e.stream_enqueue(x0, N, None, J, maxiters)
        assert not e.stream_pump()                  # nothing but the optimiser runs before the seeds are there
        with pytest.raises(pfmi_mod._lib.PfmiError):
            e.stream_wait()
        e.stream_seeds(sd)
        npts = e.stream_wait()
        e.pool_build_best(N, None)
        comm.psis_resample_enqueue(300, seed=9)
        e.defer(1)
        st = e.fit_status()
        el = e.elbo_batch_wait()
        ws = e.psis_weights(K * N)
        e.defer(0)
        res, idx, draws = comm.psis_resample_wait()
        np.testing.assert_array_equal(npts, a["npts"])
        for i in range(4):
            np.testing.assert_array_equal(st[i], [a["status"], a["jeff"], a["logdet"], a["nrej"]][i])
        np.testing.assert_array_equal(el[0], a["elbo"]); np.testing.assert_array_equal(el[1], a["se"]); np.testing.assert_array_equal(el[2], a["best"])
        np.testing.assert_array_equal(idx, a["idx"]); np.testing.assert_array_equal(draws, a["draws"])
        w, lw = e.psis_weights(K * N)
        np.testing.assert_array_equal(ws[0], w); np.testing.assert_array_equal(ws[1], lw)
        assert abs(w.sum() - 1) < 1e-12
    # cancelled deferral: nothing is delivered later
    e.defer(1)
    junk = e.fit_status()
    before = [x.copy() for x in junk]
    e.defer(-1)
    e.sync()
    for x, y in zip(junk, before):
        np.testing.assert_array_equal(x, y)
    comm.close(); e.close()


def test_results_do_not_depend_on_where_the_segments_are_cut(pfmi_mod):
    """PFMI_STREAM_POLICY / _PUB / _MINLEN move the segment boundaries and the publication rate; the results are the same bits."""
    L = pfmi_mod.lib()
    tg = pfmi_mod.t_lowrank(300, r=8, seed=2)
    K, J, maxiters, N = 5, 6, 150, 512
    cap = maxiters + 1
    x0 = pfmi_mod.HostRNG(8).rand(K * 300).reshape(K, 300) * 4 - 2
    sd = pfmi_mod.hostrng.rand_u64(79, np.arange(K * cap, dtype=np.uint64), 9)
    ref = _streamed(pfmi_mod, tg, x0, J, maxiters, N, sd, N, 200)[0]
    hooks = [(b"PFMI_STREAM_POLICY", b"0"), (b"PFMI_STREAM_POLICY", b"1"), (b"PFMI_STREAM_POLICY", b"2"), (b"PFMI_STREAM_PUB", b"4"),
             (b"PFMI_STREAM_PUB", b"32"), (b"PFMI_STREAM_MINLEN", b"64")]
    for key, val in hooks:
        assert L.pfmi_debug_set(key, val) == 0
        try:
            got = _streamed(pfmi_mod, tg, x0, J, maxiters, N, sd, N, 200)[0]
        finally:
            assert L.pfmi_debug_set(key, None) == 0
        _compare_streamed(ref, got, K, cap)


def _compare_streamed(a, s, K, cap):
    for name in ("npts", "nrej", "best", "status", "jeff", "idx", "draws"):
        np.testing.assert_array_equal(a[name], s[name], err_msg=name)
    for k in range(K):
        n = int(a["npts"][k])
        for name in ("logdet", "elbo", "se"):
            np.testing.assert_array_equal(a[name][k * cap:k * cap + n], s[name][k * cap:k * cap + n], err_msg=f"{name} path {k}")
    np.testing.assert_equal(a["res"]["pareto_shape"], s["res"]["pareto_shape"])


def test_an_abandoned_streaming_call_is_drained_before_its_memory_is_reused(pfmi_mod):
    """pfmi_stream_enqueue without its pfmi_stream_wait, then the packed route / pfmi_destroy on the same context: what is in flight finishes
    first (the optimiser still writes the staging trace), nothing is corrupted, nothing hangs."""
    tg = pfmi_mod.t_lowrank(500, r=8, seed=2)
    K, J, maxiters, N = 6, 6, 200, 256
    x0 = pfmi_mod.HostRNG(12).rand(K * 500).reshape(K, 500) * 4 - 2
    sd = pfmi_mod.hostrng.rand_u64(80, np.arange(K * (maxiters + 1), dtype=np.uint64), 9)

    def packed(e):
        npts = e.optimize_batch(x0, J, maxiters)
        e.fit_batch(J)
        seeds = pfmi_mod.hostrng.rand_u64(81, np.arange(e.P, dtype=np.uint64), 9)
        return npts, e.elbo_batch(N, seeds)

    e0 = pfmi_mod.Engine(0)
    e0.set_target(tg)
    ref = packed(e0)
    e0.close()
    e = pfmi_mod.Engine(0)
    e.set_target(tg)
    e.stream_enqueue(x0, N, sd, J, maxiters)                # ... and never waited for
    got = packed(e)
    np.testing.assert_array_equal(got[0], ref[0])
    for a, b in zip(got[1], ref[1]):
        np.testing.assert_array_equal(a, b)
    e.stream_enqueue(x0, N, None, J, maxiters)              # (not even its seeds have arrived)
    e.close()                                               # pfmi_destroy with the optimiser in flight


def test_the_packed_entry_points_work_on_the_streaming_layout(pfmi_mod):
    """After a streamed call the context holds the fixed-stride layout; pfmi_fit_batch / pfmi_elbo_batch (a refit, a rescan with other draws)
    take slots there: the same numbers as the streamed call produced, absent slots left alone."""
    tg = pfmi_mod.t_lowrank(200, r=8, seed=2)
    K, J, maxiters, N = 4, 6, 90, 256
    cap = maxiters + 1
    x0 = pfmi_mod.HostRNG(15).rand(K * 200).reshape(K, 200) * 4 - 2
    tab = pfmi_mod.hostrng.rand_u64(83, np.arange(K * cap, dtype=np.uint64), 9)
    e = pfmi_mod.Engine(0)
    e.set_target(tg)
    e.stream_enqueue(x0, N, tab, J, maxiters)
    npts = e.stream_wait()
    st0 = e.fit_status()
    el0 = e.elbo_batch_wait()
    slot_seeds = np.zeros(K * cap, dtype=np.uint64)
    for k in range(K):
        slot_seeds[k * cap + 1:(k + 1) * cap] = tab[k * cap:(k + 1) * cap - 1]
    e.fit_batch(J)
    st1 = e.fit_status()
    el1 = e.elbo_batch(N, slot_seeds)
    for k in range(K):
        sl = slice(k * cap, k * cap + int(npts[k]))
        for a, b in zip(st0[:3], st1[:3]):
            np.testing.assert_array_equal(a[sl], b[sl])
        np.testing.assert_array_equal(el0[0][sl], el1[0][sl]); np.testing.assert_array_equal(el0[1][sl], el1[1][sl])
        assert np.all(st1[0][k * cap + int(npts[k]):(k + 1) * cap] == 4)
    np.testing.assert_array_equal(st0[3], st1[3]); np.testing.assert_array_equal(el0[2], el1[2])
    X, lp, lq = e.draws(int(el1[2][0]), int(slot_seeds[int(el1[2][0])]), 16)          # slot of run 0's winner (offset 0)
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(lq))
    e.close()


def test_public_call_retries_and_failed_runs_are_the_same_streamed_or_packed(pfmi_mod, monkeypatch):
    """pfmi.multipathfinder on a built-in target takes the streaming pipeline; PFMI_NO_STREAM=1 forces the packed route.  A run that starts AT the
    optimum has no fit (L = 0: a failed try, src/singlepath.jl:299): with ntries = 3 it is retried from a sampled point (the other runs keep the
    streams they had), with ntries = 1 it stays failed and draws from fit_distributions[1] with the seed its rng yields next (:231-233).  Same
    draws, component ids, k-hat, tries and fit iterations on both routes."""
    import warnings
    d = 40
    tg = pfmi_mod.t_diag(d, seed=3)
    rng0 = pfmi_mod.HostRNG(4)
    inits = [tg.mean.copy(), rng0.rand(d) * 4 - 2, rng0.rand(d) * 4 - 2, rng0.rand(d) * 4 - 2]

    def run(ntries):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e = pfmi_mod.Engine(0)
            r = pfmi_mod.multipathfinder(tg, 300, init=[x.copy() for x in inits], ndraws_elbo=128, ntries=ntries, rng=pfmi_mod.HostRNG(6), engine=e,
                                         maxiters=80)
            out = (r.draws.copy(), r.draw_component_ids.copy(), r.psis_result.pareto_shape, [p.num_tries for p in r.pathfinder_results],
                   [p.fit_iteration for p in r.pathfinder_results], [p.success for p in r.pathfinder_results],
                   [len(p.optim_trace) for p in r.pathfinder_results], r.pathfinder_results[1].draws.copy())
            e.close()
            return out

    for ntries in (3, 1):
        monkeypatch.delenv("PFMI_NO_STREAM", raising=False)
        s = run(ntries)
        monkeypatch.setenv("PFMI_NO_STREAM", "1")
        p = run(ntries)
        monkeypatch.delenv("PFMI_NO_STREAM", raising=False)
        np.testing.assert_array_equal(s[0], p[0]); np.testing.assert_array_equal(s[1], p[1]); np.testing.assert_array_equal(s[7], p[7])
        np.testing.assert_equal(s[2], p[2])
        assert s[3:7] == p[3:7], (s[3:7], p[3:7])
        if ntries == 3:
            assert s[3][0] == 2 and all(s[5]), s[3:6]                 # the run that started at the optimum needed its second try
        else:
            assert s[3][0] == 1 and not s[5][0] and s[4][0] == 0 and all(s[5][1:]), s[3:6]


def test_streamed_pipeline_random_shapes(pfmi_mod):
    """a dozen random (target, K, d, J, maxiters, N) shapes -- register / panel fit kernels, resident / streamed factor blocks, every padding of the
    history block up to 32 columns, paths that stop at maxiters -- streamed against packed, bit for bit"""
    rs = np.random.RandomState(20260929)
    for it in range(12):
        d = int(rs.choice([7, 33, 64, 130, 257, 700, 1024, 1500, 2300]))
        J = int(rs.choice([1, 2, 3, 5, 6, 8, 10, 13, 16]))
        K = int(rs.choice([1, 2, 3, 7, 12]))
        maxiters = int(rs.choice([5, 17, 40, 100]))
        N = int(rs.choice([16, 64, 100, 300]))
        kind = int(rs.randint(3))
        tg = pfmi_mod.t_diag(d, seed=1 + it) if kind == 0 else pfmi_mod.t_lowrank(d, r=int(rs.choice([1, 4, 8, 16])), seed=2 + it) if kind == 1 else pfmi_mod.t_funnel(d)
        scale = 10.0 if kind == 2 else 2.0
        cap = maxiters + 1
        x0 = pfmi_mod.HostRNG(100 + it).rand(K * d).reshape(K, d) * 2 * scale - scale
        sd = pfmi_mod.hostrng.rand_u64(200 + it, np.arange(K * cap, dtype=np.uint64), 9)
        a = _packed(pfmi_mod, tg, x0, J, maxiters, N, sd, N, 50)
        s = _streamed(pfmi_mod, tg, x0, J, maxiters, N, sd, N, 50)[0]
        try:
            _compare(a, s, K, cap)
        except AssertionError as ex:
            raise AssertionError(f"shape {it}: kind {kind} d {d} J {J} K {K} maxiters {maxiters} N {N}: {ex}") from ex
