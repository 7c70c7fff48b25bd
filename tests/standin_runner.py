#!/usr/bin/env python
"""Executes libpfmi's G > 1 collective path on ONE GPU (VERDICT r2 "next" #1).

Run in its own process by tests/test_gpu_multirank.py:  python tests/standin_runner.py <scenario>
with PFMI_RCCL_LIB = tests/rccl_standin/librccl_standin.so (the in-process RCCL stand-in, test infrastructure) and
PFMI_COMM_ALLOW_SHARED_GPU = 1, so that csrc/comm_rccl.hip forms a world of G ranks among contexts that all sit on GPU 0 and runs
its real code: the shard-size handshake, the G-way all-gather of the log-ratio shards (padded + compacted when the shards are unequal),
replicated PSIS / index selection, and the OWNER-ONLY assembly of the result (round 6): zero-copy stores of every context's own columns into
the caller's host array (pfmi_comm_init_all), ncclSend / ncclRecv of the owned columns to rank 0 (pfmi_comm_init_rank).

Contract (reference test/multipath.jl:107-140, the `ntasks` invariance, extended to the GPU count): for the same seeds the pooled
stage must give BIT-IDENTICAL k-hat, indices and draws for every G.

Scenarios
  c4        BASELINE config 3 / 4: K = 64 device-made traces, d = 1000, J = 6, N = 1000; G = 1 vs G in {2, 4, 8} contexts
            (8 paths per context at G = 8 = config 4's sharding), with / without replacement, host uniforms, separate and fused calls
  c5        config 5's shape, small: d = 10^4, J = 10, K = 8, funnel; G in {2, 4}
  threads   one host THREAD per rank, pfmi_comm_init_rank with a shipped id (the process-per-GPU mode), G = 4
  uneven    K = 20 runs over G in {3, 8} contexts (the reference accepts any nruns, src/multipath.jl:131-146; its own test uses 20,
            test/multipath.jl:12-85): bit-identical to G = 1; the same in process-per-GPU mode (K = 10 over 4 threads)
  mismatch  process-per-GPU mode with different draws per run / a rank without a pool: every rank returns the error, nobody hangs
  api       pfmi.multipathfinder(engines=[...]) at G = 2, 4 against the single-engine call, bit for bit
"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "pathfinder.jl_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

import pfmi  # noqa: E402
from pfmi.hostrng import rand_u64  # noqa: E402

MASTER = 20260928


def _inputs(tg, K, scale):
    run_seeds = rand_u64(MASTER, np.arange(K, dtype=np.uint64), 9)
    x0 = np.stack([pfmi.HostRNG(int(s)).rand(tg.d) * 2 * scale - scale for s in run_seeds])
    return run_seeds, x0


def _local_stage(eng, tg, x0, run_seeds, J, N, maxiters):
    """optimise -> fit -> ELBO -> winners picked on the device -> pool; everything a rank does before the exchange"""
    eng.set_target(tg)
    npts = eng.optimize_batch(x0, J, maxiters)
    eng.fit_batch(J)
    seeds = np.concatenate([rand_u64(int(s), np.arange(n, dtype=np.uint64), 10) for s, n in zip(run_seeds, npts)])
    eng.elbo_batch_enqueue(N, seeds)
    eng.pool_build_best(N)
    return npts, seeds


def _pooled_variants(comm, d, ndraws, S):
    """every mode of the pooled stage; returns a dict of arrays to compare across G"""
    out = {}
    r = comm.pool_psis()
    out["k"], out["M"] = r["pareto_shape"], r["tail_length"]
    out["idx"], out["draws"] = comm.resample(ndraws, seed=MASTER)
    out["idx_nr"], out["draws_nr"] = comm.resample(ndraws, replace=False, seed=MASTER + 1)
    u = np.random.default_rng(3).random(ndraws)
    out["idx_u"], out["draws_u"] = comm.resample(ndraws, uniforms=u)
    out["idx_uni"], out["draws_uni"] = comm.resample(ndraws, importance=False, seed=MASTER + 2)
    r2, out["idx_f"], out["draws_f"] = comm.psis_resample(ndraws, seed=MASTER)          # fused, one synchronisation
    assert r2["pareto_shape"] == out["k"] and r2["tail_length"] == out["M"]
    np.testing.assert_array_equal(out["idx_f"], out["idx"])
    np.testing.assert_array_equal(out["draws_f"], out["draws"])
    r3, out["idx_f0"], out["draws_f0"] = comm.psis_resample(ndraws, importance=False, seed=MASTER + 2)
    assert np.isnan(r3["pareto_shape"]) and r3["tail_length"] == 0
    np.testing.assert_array_equal(out["idx_f0"], out["idx_uni"])
    assert out["idx"].min() >= 0 and out["idx"].max() < S and len(np.unique(out["idx_nr"])) == ndraws
    return out


def _compare(ref, got, tag):
    for key in ref:
        a, b = ref[key], got[key]
        if isinstance(a, np.ndarray):
            np.testing.assert_array_equal(a, b, err_msg=f"{tag}: {key}")
        else:
            assert a == b or (a != a and b != b), (tag, key, a, b)


def _blocks(K, G):
    """contiguous blocks of paths, the first K % G one longer (pfmi.api._blocks)"""
    base, rem = divmod(K, G)
    b = [0]
    for g in range(G):
        b.append(b[-1] + base + (1 if g < rem else 0))
    return b


def _sharded(tg, K, J, N, ndraws, maxiters, scale, Gs, check_owner=True):
    run_seeds, x0 = _inputs(tg, K, scale)
    # ---- G = 1: one context, no RCCL involved at all
    e1 = pfmi.Engine(0)
    npts1, seeds1 = _local_stage(e1, tg, x0, run_seeds, J, N, maxiters)
    c1 = pfmi.Comm.init_all([e1])
    assert c1.info() == dict(world=1, nlocal=1, rccl_version=0)
    ref = _pooled_variants(c1, tg.d, ndraws, K * N)
    pool1, lr1 = e1.pool_get()
    elbo1, se1, best1 = e1.elbo_batch_wait()
    c1.close()
    print(f"G=1: P={e1.P} khat={ref['k']:.6f} tail={ref['M']}", flush=True)
    for G in Gs:
        bk = _blocks(K, G)
        engs = [pfmi.Engine(0) for _ in range(G)]
        for g, e in enumerate(engs):
            sl = slice(bk[g], bk[g + 1])
            npts, seeds = _local_stage(e, tg, x0[sl], run_seeds[sl], J, N, maxiters)
            np.testing.assert_array_equal(npts, npts1[sl])
        comm = pfmi.Comm.init_all(engs)
        info = comm.info()
        assert info == dict(world=G, nlocal=G, rccl_version=99999), info      # the stand-in, G ranks counted by the library itself
        got = _pooled_variants(comm, tg.d, ndraws, K * N)
        _compare(ref, got, f"G={G}")
        # every rank's shard is the corresponding block of the G = 1 pool, and its ELBO table the corresponding block
        for g, e in enumerate(engs):
            _, lr = e.pool_get(draws=False)
            np.testing.assert_array_equal(lr, lr1[bk[g] * N:bk[g + 1] * N])
            el, _, bs = e.elbo_batch_wait()
            p0 = int(e1.offsets[bk[g]])
            np.testing.assert_array_equal(el, elbo1[p0:p0 + e.P])
            np.testing.assert_array_equal(bs, best1[bk[g]:bk[g + 1]])
        if check_owner:                                   # the selected columns really come from different owners
            owners = np.unique(np.searchsorted(np.array(bk[1:]) * N, ref["idx_uni"], side="right"))
            assert len(owners) > 1 or G == 1                # (the UNIFORM selection: the weighted one may be degenerate -- k-hat 12 at config 5's shape)
            print(f"G={G}: bit-identical; selected columns owned by ranks {owners.tolist()}", flush=True)
        comm.close()
        for e in engs:
            e.close()
    e1.close()
    return ref


def scenario_c4():
    tg = pfmi.t_lowrank(1000, r=8, seed=2)
    ref = _sharded(tg, 64, 6, 1000, 1000, 1000, 2.0, (2, 4, 8))
    print("c4 ok", ref["k"])


def scenario_c5():
    tg = pfmi.t_funnel(10000)
    _sharded(tg, 8, 10, 256, 256, 14, 10.0, (2, 4))
    print("c5 ok")


def scenario_threads(K=16, G=4):
    """process-per-GPU mode (pfmi_comm_init_rank), the ranks being threads of this process: each thread owns one context, forms the
    group with the shipped id and makes the same sequence of pfmi_comm_* calls.  The d x ndraws result lives on rank 0 only (the owners
    SEND their columns there); the other ranks get k-hat, the tail length and the indices (replicated) and NaN where they asked for draws."""
    tg = pfmi.t_lowrank(200, r=8, seed=2)
    J, N, nd = 6, 256, 300
    bk = _blocks(K, G)
    run_seeds, x0 = _inputs(tg, K, 2.0)
    e1 = pfmi.Engine(0)
    _local_stage(e1, tg, x0, run_seeds, J, N, 200)
    c1 = pfmi.Comm.init_all([e1])
    ref = _pooled_variants(c1, tg.d, nd, K * N)
    c1.close()
    uid = pfmi.Comm.unique_id()
    res, errs = [None] * G, []

    def work(g):
        try:
            e = pfmi.Engine(0)
            sl = slice(bk[g], bk[g + 1])
            _local_stage(e, tg, x0[sl], run_seeds[sl], J, N, 200)
            comm = pfmi.Comm.init_rank(e, G, g, uid)
            assert comm.info() == dict(world=G, nlocal=1, rccl_version=99999)
            res[g] = _pooled_variants(comm, tg.d, nd, K * N)
            comm.close()
            e.close()
        except Exception as ex:  # pragma: no cover
            errs.append((g, repr(ex)))

    ths = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for g in range(G):
        if g == 0:
            _compare(ref, res[g], f"thread rank {g}")
        else:                                              # not the root: everything replicated is there, the draws are NaN (never stale)
            for key in ref:
                if key.startswith("draws"):
                    assert np.all(np.isnan(res[g][key])), (g, key)
                else:
                    _compare({key: ref[key]}, {key: res[g][key]}, f"thread rank {g}")
    e1.close()
    print(f"threads ok (K={K}, G={G}, blocks {bk})")


def scenario_uneven():
    """any nruns over any G: K = 20 over 3 and 8 contexts (blocks 7 7 6 / 3 3 3 3 2 2 2 2), pageable and page-locked destinations,
    then the process-per-GPU mode with K = 10 over 4 threads (3 3 2 2)"""
    tg = pfmi.t_lowrank(300, r=8, seed=2)
    _sharded(tg, 20, 6, 256, 400, 300, 2.0, (3, 8))
    tg2 = pfmi.t_diag(9000, seed=1)                       # 9000 x 300 x 8 B = 21.6 MB: the result array is page-locked (direct stores)
    _sharded(tg2, 5, 6, 128, 300, 12, 2.0, (2, 4))
    scenario_threads(K=10, G=4)
    print("uneven ok")


def scenario_mismatch():
    """every rank must come back with an error (no hang) when the draws per run differ or a rank has no pool"""
    tg = pfmi.t_lowrank(64, r=8, seed=2)
    J, N, G = 6, 128, 2
    for case in ("unequal", "nopool"):
        uid = pfmi.Comm.unique_id()
        out = [None] * G

        def work(g):
            e = pfmi.Engine(0)
            try:
                Kl = 3
                run_seeds, x0 = _inputs(tg, Kl, 2.0)
                e.set_target(tg)
                if not (case == "nopool" and g == 1):
                    _local_stage(e, tg, x0, run_seeds, J, N // 2 if (case == "unequal" and g == 1) else N, 50)
                comm = pfmi.Comm.init_rank(e, G, g, uid)
                try:
                    comm.pool_psis()
                    out[g] = "no error"
                except pfmi.PfmiError as ex:
                    out[g] = str(ex)
                finally:
                    comm.close()
            finally:
                e.close()

        ths = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        [t.start() for t in ths]
        [t.join(timeout=120) for t in ths]
        assert not any(t.is_alive() for t in ths), "a rank is blocked in a collective"
        print(case, out, flush=True)
        assert all(o is not None and o != "no error" for o in out), out
        if case == "unequal":
            assert all("differ" in o for o in out), out
        else:
            assert all("cannot take part" in o for o in out), out
    print("mismatch ok")


def scenario_api():
    """pfmi.multipathfinder(engines=[...]): ONE host thread drives G contexts (src/multipath.jl:190-225) -- result identical to the
    single-engine call for every G"""
    tg = pfmi.t_lowrank(120, r=8, seed=2)
    for nruns, Gs in ((8, (2, 4)), (10, (4,))):           # (10 runs over 4 engines: blocks 3 3 2 2)
        _api_case(tg, nruns, Gs)
    print("api ok")


def _api_case(tg, nruns, Gs):
    kw = dict(nruns=nruns, ndraws_elbo=128, history_length=6, maxiters=300)
    r1 = pfmi.multipathfinder(tg, 200, rng=pfmi.HostRNG(5), **kw)
    for G in Gs:
        engs = [pfmi.Engine(0) for _ in range(G)]
        rg = pfmi.multipathfinder(tg, 200, rng=pfmi.HostRNG(5), engines=engs, **kw)
        np.testing.assert_array_equal(rg.draws, r1.draws)
        np.testing.assert_array_equal(rg.draw_component_ids, r1.draw_component_ids)
        assert rg.psis_result.pareto_shape == r1.psis_result.pareto_shape
        np.testing.assert_array_equal(rg.psis_result.weights, r1.psis_result.weights)
        for a, b in zip(rg.pathfinder_results, r1.pathfinder_results):
            assert a.fit_iteration == b.fit_iteration and a.success == b.success and a.draw_seed == b.draw_seed
            np.testing.assert_array_equal(a.draws, b.draws)
            np.testing.assert_array_equal(a.fit_distribution.mu, b.fit_distribution.mu)
        # resample() on the sharded result: stored draws reproduce, fresh candidates and no-replacement equal the single-engine call
        for rkw in (dict(), dict(replace=False), dict(ndraws_per_run=96), dict(importance=False)):
            a = pfmi.resample(rg, 150, rng=pfmi.HostRNG(9), **rkw)
            b = pfmi.resample(r1, 150, rng=pfmi.HostRNG(9), **rkw)
            np.testing.assert_array_equal(a.draws, b.draws)
            np.testing.assert_array_equal(a.draw_component_ids, b.draw_component_ids)
        print(f"api nruns={nruns} G={G} ok", flush=True)
        for e in engs:
            e.close()


if __name__ == "__main__":
    assert os.environ.get("PFMI_RCCL_LIB"), "run through tests/test_gpu_multirank.py (PFMI_RCCL_LIB must point at the stand-in)"
    {"c4": scenario_c4, "c5": scenario_c5, "threads": scenario_threads, "mismatch": scenario_mismatch, "api": scenario_api,
     "uneven": scenario_uneven}[sys.argv[1]]()
