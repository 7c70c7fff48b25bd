"""world_size-2 `gloo` test of the multi-GPU PROTOCOL (pfmi/distributed.py = csrc/comm_rccl.hip restated over torch.distributed) on CPU.

The engine-specific steps (per-path ELBO/draws, PSIS, index draw, column gather) are played by the CPU oracle, so what is under test
is the protocol the C library runs over RCCL: uneven contiguous shards, the all-gather padded to the largest shard + compaction to
the k-major pool order, replicated PSIS / index selection, the owners SENDING their selected columns to rank 0.  The 2-rank result
must be IDENTICAL to the single-process result (the reference's ntasks invariance, test/multipath.jl:107-140, extended to the GPU count).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pipeline(rank, world, dist):
    """One rank's share of a small multipath job; returns (draws (d, ndraws), idx, pareto_k)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "pathfinder.jl_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import pfmi
    from pfmi.distributed import pooled_psis_resample, shard_paths
    from helpers import oracle_target
    from oracle import pf_oracle as po

    K, d, N, J, ndraws = 5, 12, 40, 6, 64               # 5 paths over 2 ranks: blocks 3 + 2 (unequal shards)
    tg = pfmi.t_lowrank(d, r=3, seed=4)
    otg = oracle_target(tg)
    k0, k1 = shard_paths(K, world, rank)
    run_seeds = pfmi.hostrng.rand_u64(77, np.arange(K, dtype=np.uint64), 9)    # keyed by the GLOBAL path index
    pools, lrs = [], []
    for k in range(k0, k1):
        tr = pfmi.optimize_with_trace(tg, pfmi.HostRNG(int(run_seeds[k])).rand(d) * 4 - 2, history_length=J)
        seeds = pfmi.hostrng.rand_u64(int(run_seeds[k]), np.arange(len(tr), dtype=np.uint64), 10)
        r = po.path_fit_elbo(tr.points, tr.gradients, J, otg, N, seeds, want_draws=True)
        pools.append(r["draws"])
        lrs.append(r["logp"] - r["logq"])
    pool = np.concatenate(pools, axis=1)                       # (d, K_local * N), k-major
    lr_local = torch.from_numpy(np.concatenate(lrs))
    shard_sizes = [(shard_paths(K, world, r)[1] - shard_paths(K, world, r)[0]) * N for r in range(world)]
    state = {}

    def psis_fn(t):
        lw, w, k, M = po.psis(t.numpy())
        state["w"] = w
        return dict(pareto_shape=k, tail_length=M)

    def sample_fn(S):
        return po.sample_weighted(state["w"], ndraws, seed=123)

    def columns_fn(cols):
        assert np.all((cols >= k0 * N) & (cols < k1 * N)), "asked for a column this rank does not own"
        return pool[:, cols - k0 * N]

    res, idx, out = pooled_psis_resample(dist if world > 1 else None, lr_local, shard_sizes, d, ndraws,
                                         psis_fn=psis_fn, sample_fn=sample_fn, columns_fn=columns_fn)
    return out, idx, res["pareto_shape"]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        draws, idx, k = _pipeline(rank, world, dist)
        q.put((rank, draws, idx, k))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_single_process():
    import torch.multiprocessing as mp
    ref_draws, ref_idx, ref_k = _pipeline(0, 1, None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owners = np.unique(ref_idx // (3 * 40))                    # (rank 0 owns pool columns 0 .. 119, rank 1 the rest)
    assert len(owners) == 2, "the selection should touch both ranks' columns"
    for rank, draws, idx, k in got:
        np.testing.assert_array_equal(idx, ref_idx)            # replicated, deterministic index selection
        if rank == 0:
            np.testing.assert_array_equal(draws, ref_draws)    # the owners' columns arrive at their selection positions
        else:
            assert draws is None                               # the result lives on rank 0
        assert k == ref_k


def test_shard_paths_contract():
    from pfmi.distributed import shard_paths
    assert [shard_paths(64, 8, r) for r in (0, 7)] == [(0, 8), (56, 64)]
    assert [shard_paths(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]      # any nruns: the first K % G blocks one longer
    assert [shard_paths(20, 8, r)[1] - shard_paths(20, 8, r)[0] for r in range(8)] == [3, 3, 3, 3, 2, 2, 2, 2]
    with pytest.raises(ValueError):
        shard_paths(3, 4, 0)


def _verify_worker(rank, world, port, q, tamper):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "pathfinder.jl_amd"))
    from pfmi.distributed import result_fingerprint, sharded_equals_single
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = np.arange(50, dtype=np.int64) * 3
        draws = np.linspace(-1, 1, 200).reshape(4, 50)
        ref = result_fingerprint(float("nan"), 17, idx, draws) if rank == 0 else None       # NaN k-hat compares by bit pattern
        mine_draws = draws.copy()
        if tamper and rank == 1:
            mine_draws[3, 49] = np.nextafter(mine_draws[3, 49], 2.0)                         # ONE bit on ONE rank
        ok, bad = sharded_equals_single(dist, result_fingerprint(float("nan"), 17, idx, mine_draws), ref)
        q.put((rank, ok, bad))
        ok2, _ = sharded_equals_single(dist, result_fingerprint(1.0, 17, idx, draws), None)  # no reference: None everywhere
        q.put((rank, ok2, ["second"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tamper", [False, True])
def test_sharded_equals_single_verdict_is_collective(tamper):
    """bench.py's self-check of a multi-GPU run: every rank compares its copy of the sharded answer with rank 0's
    single-GPU reference; one differing bit on one rank makes the verdict False on EVERY rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, q, tamper)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    first = [g for g in got if g[2] != ["second"]]
    second = [g for g in got if g[2] == ["second"]]
    assert len(first) == 2 and all(g[1] is (not tamper) for g in first), first
    if tamper:
        assert [g[2] for g in first if g[0] == 1] == [["draws_sha256"]] and [g[2] for g in first if g[0] == 0] == [[]]
    assert all(g[1] is None for g in second)
