"""usage: python tests/probes/stream_probe.py [K] [d] [reps] -- end-to-end step (x0 on the host -> resampled draws on the host) through the packed route
(optimize_batch ; fit_batch ; elbo_batch_enqueue ; pool ; PSIS ; resample) and through the streaming pipeline (pfmi_stream_enqueue), same
seeds; prints both wall-clocks and whether the results are bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pathfinder.jl_amd")]
import pfmi  # noqa: E402
from pfmi.hostrng import rand_u64  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
J, maxiters, N = 6, 1000, 1000
cap = maxiters + 1
master = 20260928
tg = pfmi.t_lowrank(d, r=8, seed=2)
run_seeds = rand_u64(master, np.arange(K, dtype=np.uint64), 9)
x0 = np.stack([pfmi.HostRNG(int(s)).rand(d) * 4 - 2 for s in run_seeds])
tab = np.concatenate([rand_u64(int(s), np.arange(1, cap + 1, dtype=np.uint64), 10) for s in run_seeds])
eng = pfmi.Engine(0)
eng.set_target(tg)
comm = pfmi.Comm.init_rank(eng, 1, 0, pfmi.Comm.unique_id()) if os.environ.get("STREAM_RCCL") else pfmi.Comm.init_all([eng])    # STREAM_RCCL=1: a world of one rank THROUGH RCCL
npts = eng.optimize_batch(x0, J, maxiters)
seeds = np.concatenate([rand_u64(int(s), np.arange(n, dtype=np.uint64), 10) for s, n in zip(run_seeds, npts)])
out = {}


def packed():
    eng.optimize_batch(x0, J, maxiters)
    eng.fit_batch(J)
    eng.elbo_batch_enqueue(N, seeds)
    eng.pool_build_best(N)
    res, idx, dr = comm.psis_resample(1000, seed=master)
    elbo, se, best = eng.elbo_batch_wait()
    out["p"] = (res["pareto_shape"], idx, dr, best)


def streamed():
    eng.stream_enqueue(x0, N, tab, J, maxiters)
    eng.stream_wait()
    eng.pool_build_best(N)
    res, idx, dr = comm.psis_resample(1000, seed=master)
    elbo, se, best = eng.elbo_batch_wait()
    out["s"] = (res["pareto_shape"], idx, dr, best)


runs = (("packed", packed), ("streamed", streamed), ("packed", packed), ("streamed", streamed))
if os.environ.get("STREAM_ONLY"):
    packed()
    runs = (("streamed", streamed),)
for name, fn in runs:
    for _ in range(3):
        fn()
    eng.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"{name:9s} K={K} d={d}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}   (fits {int(npts.sum()) - K}, longest path {int(npts.max())}, shortest {int(npts.min())})")
p, s = out["p"], out["s"]
same = p[0] == s[0] and np.array_equal(p[1], s[1]) and np.array_equal(p[2], s[2]) and np.array_equal(p[3], s[3])
print("bit-identical:", same)
if os.environ.get("STREAM_STAGES"):
    eng.profile(2)
    for _ in range(5):
        streamed()
    eng.sync()
    for nm in ("optimize", "history", "fit", "elbo_draws", "elbo_reduce", "psis", "resample"):
        ms, n = eng.kernel_time(nm)
        print(f"  {nm:12s} {ms / max(n, 1):.4f} ms x {n / 5:.0f} per step")
