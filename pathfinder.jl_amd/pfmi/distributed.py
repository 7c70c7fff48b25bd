"""Multi-GPU orchestration of the pooled stage -- the PROTOCOL of csrc/comm_rccl.hip restated over torch.distributed.

The product path is `pfmi_comm_*` (RCCL called directly from libpfmi.so).  This module keeps the same protocol in a form that runs on
CPU (`gloo`, world_size 2, tests/test_distributed_cpu.py, the oracle standing in for the engine), plus the fingerprint helpers bench.py
uses to verify a sharded run against a single-GPU recomputation.  Until round 5 it was also a fallback data path of bench.py; that
fallback is gone (one orchestration to keep in sync).

Paths are independent until pooling (reference src/multipath.jl:190-208 vs :215-225): contiguous blocks of paths per rank, ANY nruns
over any world size (the first K % G blocks one path longer; src/multipath.jl:131-146).  ONE collective with real content: an
all-gather of the fp64 log-ratio shards, padded to the largest shard and compacted back to the k-major pool order of
src/resample.jl:93.  PSIS and the index selection then run REPLICATED and deterministically on every rank.  The selected columns are
sent by their OWNERS to rank 0, which holds the d x ndraws result; nothing else is exchanged, the draw pool never moves.
"""


def shard_paths(K, world, rank):
    """Contiguous block of paths owned by `rank`; pool order stays k-major (src/resample.jl:93).  Any K >= world."""
    if K < world:
        raise ValueError(f"npaths={K} is smaller than the number of ranks {world}: every rank needs at least one path")
    base, rem = divmod(K, world)
    k0 = rank * base + min(rank, rem)
    return k0, k0 + base + (1 if rank < rem else 0)


def pooled_psis_resample(dist, lr_local, shard_sizes, d, ndraws, *, psis_fn, sample_fn, columns_fn, min_world=2):
    """Run the pooled stage collectively (CPU tensors: the gloo model of csrc/comm_rccl.hip).

    dist         torch.distributed (initialised) or None for a single process
    lr_local     1-D float64 tensor: this rank's log ratios, k-major / n fastest (K_local * N_r)
    shard_sizes  list of every rank's K_r * N_r (the handshake of comm_rccl.hip; known to the caller here)
    d, ndraws    shape of the result
    psis_fn      (pooled 1-D tensor) -> psis result (weights stay inside the engine)
    sample_fn    (S) -> int64 index array (identical on every rank)
    columns_fn   (global pool columns owned by this rank, int64 array) -> (d, n) array of those columns
    returns      (psis result, idx, draws or None): draws (d, ndraws) on rank 0 (and in a single process), None elsewhere
    """
    import numpy as np
    import torch
    collective = dist is not None and dist.is_initialized() and dist.get_world_size() >= min_world
    world = dist.get_world_size() if collective else 1
    rank = dist.get_rank() if collective else 0
    offs = np.concatenate([[0], np.cumsum(shard_sizes)]).astype(np.int64) if collective else np.array([0, lr_local.numel()])
    if collective:
        smax = int(max(shard_sizes))
        pad = torch.zeros(smax, dtype=torch.float64)
        pad[:lr_local.numel()] = lr_local
        allp = torch.empty(world * smax, dtype=torch.float64)
        dist.all_gather_into_tensor(allp, pad)               # the single exchange step of the data path
        pooled = torch.cat([allp[r * smax:r * smax + int(shard_sizes[r])] for r in range(world)])     # compaction: k-major pool order
    else:
        pooled = lr_local
    res = psis_fn(pooled)
    idx = np.asarray(sample_fn(int(pooled.numel())), dtype=np.int64)
    owner = np.searchsorted(offs[1:], idx, side="right")
    mine = np.flatnonzero(owner == rank)
    cols = np.asarray(columns_fn(idx[mine]), dtype=np.float64).reshape(d, len(mine))
    if not collective:
        out = np.empty((d, ndraws), order="F")
        out[:, mine] = cols
        return res, idx, out
    if rank != 0:
        if len(mine):
            dist.send(torch.from_numpy(np.ascontiguousarray(cols.T)), dst=0)      # exactly the owned columns, in selection order
        return res, idx, None
    out = np.empty((d, ndraws), order="F")
    out[:, mine] = cols
    for r in range(1, world):
        pos = np.flatnonzero(owner == r)
        if len(pos):
            buf = torch.empty((len(pos), d), dtype=torch.float64)
            dist.recv(buf, src=r)
            out[:, pos] = buf.numpy().T
    return res, idx, out


# ---- self-verification of a sharded run ---------------------------------------------------------------------
def result_fingerprint(pareto_k, tail_length, idx, draws):
    """What must be IDENTICAL for any GPU count (test/multipath.jl:107-140 extended to G): k-hat (bit pattern, NaN included), the
    PSIS tail length, the resample indices and the (d x ndraws) result, the arrays as SHA-256 of their bytes.  draws = None (a rank of
    a process-per-GPU group other than rank 0: the result lives on rank 0) leaves the draws out of the comparison."""
    import hashlib
    import numpy as np
    fp = {"pareto_k_bits": np.float64(pareto_k).view(np.uint64).item(), "tail_length": int(tail_length),
          "idx_sha256": hashlib.sha256(np.ascontiguousarray(idx, dtype=np.int64).tobytes()).hexdigest()}
    if draws is not None:
        fp["draws_sha256"] = hashlib.sha256(np.ascontiguousarray(draws, dtype=np.float64).tobytes()).hexdigest()
    return fp


def sharded_equals_single(dist, mine, reference, device=None):
    """Every rank compares the fingerprint of ITS copy of the sharded result with rank 0's single-GPU reference (the fields it holds).

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
        bad = [k for k in reference if k in mine and mine[k] != reference[k]]
        return not bad, bad
    import torch
    box = [reference if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ref = box[0]
    if ref is None:
        return None, []
    bad = [k for k in ref if k in mine and mine[k] != ref[k]]
    if dist.get_rank() == 0 and "draws_sha256" not in mine:
        bad.append("draws_sha256")                              # rank 0 must hold the result
    flag = torch.tensor([0.0 if bad else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # one rank that disagrees makes the verdict False everywhere
    return bool(flag.item() == 1.0), bad
