import os, sys, time
sys.path.insert(0, "pathfinder.jl_amd")
import numpy as np, pfmi
import pfmi.api as api
tg = pfmi.t_lowrank(1000, r=8, seed=2)
eng = pfmi.Engine(0)
kw = dict(nruns=64, ndraws_elbo=1000, history_length=6, engine=eng)
for _ in range(3): pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
# instrument selected functions
import functools
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for n in ("optimize_batch_enqueue", "optimize_batch_wait", "fit_batch", "elbo_batch_enqueue", "elbo_batch_wait", "pool_build_best", "fit_status", "set_target"):
    wrap(pfmi.Engine, n)
wrap(pfmi.core.Comm, "psis_resample")
wrap(api, "rand_u64_multi"); wrap(api, "_run_paths"); wrap(api, "_assemble_path")
N = 10
t0 = time.perf_counter()
for _ in range(N): pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
tot = (time.perf_counter() - t0) / N
print(f"wall {tot*1e3:.2f} ms per call")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"  {k:26s} {v/N*1e3:7.3f} ms")
