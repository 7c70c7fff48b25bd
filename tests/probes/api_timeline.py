"""usage: python tests/probes/api_timeline.py [K] -- wall-clock timeline of ONE pfmi.multipathfinder call (d = 1000, N = 1000): when each engine /
communicator method and host helper starts and how long it takes (median over 10 calls per entry, in call order)"""
import functools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pathfinder.jl_amd")]
import numpy as np  # noqa: E402
import pfmi  # noqa: E402
import pfmi.api as api  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tg = pfmi.t_lowrank(1000, r=8, seed=2)
eng = pfmi.Engine(0)
kw = dict(nruns=K, ndraws_elbo=1000, history_length=6, engine=eng)
for _ in range(3):
    pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
LOG = []
T0 = [0.0]


def wrap(obj, name):
    f = getattr(obj, name)

    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        LOG.append((name, t0 - T0[0], time.perf_counter() - t0))
        return r
    setattr(obj, name, g)


for n in ("stream_enqueue", "stream_wait", "optimize_batch_enqueue", "optimize_batch_wait", "fit_batch", "elbo_batch_enqueue", "elbo_batch_wait",
          "pool_build_best", "fit_status", "set_target", "psis_weights"):
    wrap(pfmi.Engine, n)
wrap(pfmi.core.Comm, "psis_resample")
for n in ("rand_u64_multi", "_run_paths", "_assemble_path", "_comm_for"):
    wrap(api, n)
runs = []
for _ in range(10):
    LOG.clear()
    T0[0] = time.perf_counter()
    pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
    tot = time.perf_counter() - T0[0]
    agg = {}
    order = []
    for name, st, du in LOG:
        if name not in agg:
            agg[name] = [st, 0.0, 0]
            order.append(name)
        agg[name][1] += du
        agg[name][2] += 1
    runs.append((tot, order, agg))
runs.sort(key=lambda r: r[0])
tot, order, agg = runs[len(runs) // 2]
res = pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
nfit = sum(len(r.optim_trace) - 1 for r in res.pathfinder_results)
print(f"K={K}: wall {tot * 1e3:.3f} ms (median of 10); this call's instance: {nfit} fits, longest path {max(len(r.optim_trace) for r in res.pathfinder_results)} points")
for name in order:
    st, du, n = agg[name]
    print(f"  {st * 1e3:8.3f} ms  {name:24s} {du * 1e3:8.3f} ms  x{n}")
