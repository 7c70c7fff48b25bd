"""The collective path at world size 1 on the GPU (RCCL through the C ABI: `pfmi_comm_*`; torch's nccl group on engine-owned memory), bench.py's
rank accounting, and a context destroyed before its communicator.  The G > 1 sharding scenarios run in tests/test_gpu_multirank.py."""
from concurrent.futures import ThreadPoolExecutor
import ctypes as C
import json
import os
import warnings

import numpy as np
import pytest

from helpers import fit_seeds, make_traces, oracle_factor_from_gpu, oracle_target
from oracle import pf_oracle as po
import margins as mg
from gpu_common import _setup

pytestmark = pytest.mark.gpu


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


# ---- collectives behind the C ABI -------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["init_all", "init_all_local", "init_rank"])
def test_comm_rccl_world1_equals_local_path(pfmi_mod, eng, mode):
    """pfmi_comm_* (csrc/comm_rccl.hip: ncclAllGather of the log-ratio shards, replicated PSIS / index selection, owner gather,
    ncclAllReduce) in the only RCCL world a 1-GPU box allows.  Both ways of forming the group -- ncclCommInitAll (one process,
    G contexts: a single Julia caller) and ncclCommInitRank with a shipped id (one process per GPU) -- must reproduce the
    single-GPU calls bit for bit (result invariance under the GPU count, test/multipath.jl:107-140 extended to G)."""
    tg = pfmi_mod.t_lowrank(50, r=8, seed=2)
    traces = make_traces(tg, 4, 11)
    eng.set_target(tg)
    eng.set_traces([t.points for t in traces], [t.gradients for t in traces])
    eng.fit_batch(6)
    seeds = fit_seeds(eng.P, 4)
    elbo, se, best = eng.elbo_batch(64, seeds)
    pts = [int(eng.offsets[k]) + int(best[k]) for k in range(4)]
    eng.pool_build(96, pts, seeds[pts])
    pool, lr = eng.pool_get()
    ref = eng.psis(lr)
    ref_idx = eng.resample_indices(len(lr), 40, seed=9)
    ref_norep = eng.resample_indices(len(lr), 40, replace=False, seed=9)
    if mode == "init_all":                                           # the REAL librccl in the only world a 1-GPU box allows
        os.environ["PFMI_COMM_FORCE_RCCL"] = "1"
        try:
            comm = pfmi_mod.Comm.init_all([eng])
        finally:
            os.environ.pop("PFMI_COMM_FORCE_RCCL", None)
    elif mode == "init_all_local":                                   # round 3: a world of one context does not touch RCCL at all
        comm = pfmi_mod.Comm.init_all([eng])
    else:
        comm = pfmi_mod.Comm.init_rank(eng, 1, 0, pfmi_mod.Comm.unique_id())
    try:
        info = comm.info()
        assert info["world"] == 1 and info["nlocal"] == 1
        assert (info["rccl_version"] == 0) if mode == "init_all_local" else (info["rccl_version"] > 20000)
        r_f, idx_f, draws_f = comm.psis_resample(40, seed=9)         # the fused entry (one synchronisation)
        assert r_f["pareto_shape"] == ref["pareto_shape"] and r_f["tail_length"] == ref["tail_length"]
        np.testing.assert_array_equal(idx_f, ref_idx)
        np.testing.assert_array_equal(draws_f, pool.reshape(tg.d, -1, order="F")[:, ref_idx])
        res = comm.pool_psis()
        assert res["pareto_shape"] == ref["pareto_shape"] and res["tail_length"] == ref["tail_length"]
        idx, draws = comm.resample(40, seed=9)
        np.testing.assert_array_equal(idx, ref_idx)
        np.testing.assert_array_equal(draws, pool.reshape(tg.d, -1, order="F")[:, ref_idx])
        idx2, draws2 = comm.resample(40, replace=False, seed=9)
        np.testing.assert_array_equal(idx2, ref_norep)
        np.testing.assert_array_equal(draws2, pool.reshape(tg.d, -1, order="F")[:, ref_norep])
        u = np.random.default_rng(1).random(25)
        idx3, _ = comm.resample(25, uniforms=u, want_draws=False)
        np.testing.assert_array_equal(idx3, po.sample_weighted(ref["weights"], 25, uniforms=u))
    finally:
        comm.close()
    with pytest.raises(pfmi_mod.PfmiError, match="share GPU"):
        e2 = pfmi_mod.Engine(0)
        try:
            pfmi_mod.Comm.init_all([eng, e2])                          # one rank per GPU
        finally:
            e2.close()


# ---- bench.py contract on the GPU box ---------------------------------------------------------------------------
@pytest.mark.parametrize("comm_mode", ["c_abi"])
def test_bench_force_dist_counts_its_ranks(comm_mode):
    """bench.py --gpus 1 --force-dist: the N > 1 code path (pfmi_comm_init_rank: RCCL through the C ABI) in a single-rank world; the JSON
    line reports the ranks counted through the collective.  (Round 6: the torch.distributed fallback data path is gone.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "1",
                        "--npaths", "8", "--dim", "100", "--target", "diag", "--no-cpu-baseline", "--verify-sharding"], capture_output=True,
                       text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["ranks_in_collective"] == 1
    # round 4: --verify-sharding through the torch.distributed world (broadcast of rank 0's single-GPU reference, MIN all-reduce of the
    # per-rank verdicts) -- here the "sharded" run IS a world of one, so it must equal the recomputation bit for bit
    assert line["sharded_equals_single"] is True, line.get("sharded_equals_single_note")
    if comm_mode == "c_abi":
        assert line["rccl_version"] and line["rccl_version"] > 0
    assert line["config"]["collective_backend"].startswith("RCCL" if comm_mode == "c_abi" else "torch.distributed")
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


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
