"""Shared helpers for the parity tests: synthetic traces + oracle-side target objects."""
import numpy as np

from oracle import pf_oracle as po


def oracle_target(t):
    """pfmi target -> oracle target (same parameters)."""
    if t.kind == 1:
        return po.FunnelTarget(t.d)
    return po.GaussTarget(t.mean, t.a, t.Wd if t.r else None, t.G if t.r else None, t.offset)


def make_traces(target, K, seed, scale=2.0, history_length=6, maxiters=1000):
    import pfmi
    rng = pfmi.HostRNG(seed)
    traces = []
    for k in range(K):
        x0 = rng.rand(target.d) * 2 * scale - scale
        traces.append(pfmi.optimize_with_trace(target, x0, history_length=history_length, maxiters=maxiters))
    return traces


def fit_seeds(P, seed):
    import pfmi
    return pfmi.hostrng.rand_u64(seed, np.arange(P, dtype=np.uint64), 9)
