"""Multi-GPU orchestration of the pooled stage.

Paths are independent until pooling (reference src/multipath.jl:190-208 vs :215-225), so they are sharded
in contiguous blocks over the ranks (one process per GPU) and everything up to the per-draw log importance
ratios is local.  The data path then has exactly ONE collective with a real exchange: an all-gather of the
fp64 log-ratio shards (K/G * N_r doubles per rank -- 64 KB per GPU at config 4) over RCCL/xGMI, after which
PSIS and the index selection are REPLICATED deterministically on every rank (same code, same seed, integer
CDF => identical indices for any G).  The selected columns live on the rank that owns their path; each rank
fills its own columns into a zero (d x ndraws) buffer and a sum all-reduce assembles the result (8 MB at
config 4; the draw pool itself is never exchanged).

The functions take the engine-specific steps as callables so that the same orchestration is exercised on CPU
(`gloo`, world_size 2, tests/test_distributed_cpu.py with the oracle standing in for the engine) and on the
GPUs (`nccl` == RCCL, bench.py).
"""


def shard_paths(K, world, rank):
    """Contiguous block of paths owned by `rank`; pool order stays k-major (src/resample.jl:93)."""
    if K % world != 0:
        raise ValueError(f"npaths={K} must be divisible by the number of ranks {world} "
                         "(equal log-ratio shards keep the result independent of the GPU count)")
    per = K // world
    return rank * per, (rank + 1) * per


def pooled_psis_resample(dist, lr_local, lr_all, out, *, psis_fn, sample_fn, gather_fn, sync_fn=None, min_world=2):
    """Run the pooled stage collectively.

    dist       torch.distributed (initialised) or None for a single process
    lr_local   1-D tensor: this rank's log ratios, k-major / n fastest (K_local * N_r)
    lr_all     1-D tensor receiving the gathered pool (K * N_r); ignored when dist is None
    out        1-D tensor (d * ndraws) receiving the resampled draws on every rank
    psis_fn    (lr_all tensor) -> psis result (weights stay inside the engine)
    sample_fn  (S) -> index array (identical on every rank)
    gather_fn  (idx, out tensor) -> fills `out` with this rank's owned columns, zeros elsewhere
    sync_fn    optional device synchronisation between engine work and collectives
    min_world  collectives are issued when the world has at least this many ranks (1: also in a single-rank world,
               used to exercise the RCCL path on a 1-GPU box)
    returns    (psis result, idx)
    """
    collective = dist is not None and dist.is_initialized() and dist.get_world_size() >= min_world
    if collective:
        dist.all_gather_into_tensor(lr_all, lr_local)      # the single exchange step of the data path
        if sync_fn:
            sync_fn()
        pooled = lr_all
    else:
        pooled = lr_local
    res = psis_fn(pooled)
    idx = sample_fn(int(pooled.numel()))
    gather_fn(idx, out)
    if collective:
        if sync_fn:
            sync_fn()
        dist.all_reduce(out)                               # every column is owned by exactly one rank
        if sync_fn:
            sync_fn()
    return res, idx


# ---- self-verification of a sharded run ---------------------------------------------------------------------
def result_fingerprint(pareto_k, tail_length, idx, draws):
    """What must be IDENTICAL for any GPU count (test/multipath.jl:107-140 extended to G): k-hat (bit pattern, NaN included), the
    PSIS tail length, the resample indices and the (d x ndraws) result, the arrays as SHA-256 of their bytes."""
    import hashlib
    import numpy as np
    return {"pareto_k_bits": np.float64(pareto_k).view(np.uint64).item(), "tail_length": int(tail_length),
            "idx_sha256": hashlib.sha256(np.ascontiguousarray(idx, dtype=np.int64).tobytes()).hexdigest(),
            "draws_sha256": hashlib.sha256(np.ascontiguousarray(draws, dtype=np.float64).tobytes()).hexdigest()}


def sharded_equals_single(dist, mine, reference, device=None):
    """Every rank compares the fingerprint of ITS copy of the sharded result with rank 0's single-GPU reference.

    dist       torch.distributed (initialised) or None
    mine       result_fingerprint(...) of this rank's sharded answer
    reference  result_fingerprint(...) of all K paths on one context -- needed on rank 0 only (None elsewhere); None on rank 0 too
               means "could not be computed" and the answer is None on every rank
    device     torch device for the flag all-reduce (None: CPU, for gloo)
    returns    (True | False | None, per-field mismatches seen by THIS rank)
    """
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if reference is None:
            return None, []
        bad = [k for k in reference if mine.get(k) != reference[k]]
        return not bad, bad
    import torch
    box = [reference if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ref = box[0]
    if ref is None:
        return None, []
    bad = [k for k in ref if mine.get(k) != ref[k]]
    flag = torch.tensor([0.0 if bad else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # one rank that disagrees makes the verdict False everywhere
    return bool(flag.item() == 1.0), bad
