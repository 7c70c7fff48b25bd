"""usage: python tests/probes/stream_steps.py -- 120 streamed end-to-end steps at 8 paths (d = 1000): mean / median / max and the outliers (a host pause -- e.g. a full
garbage collection -- shows as ONE step of tens of milliseconds: the calling thread schedules the pipeline)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pathfinder.jl_amd")]
import numpy as np, pfmi
from pfmi.hostrng import rand_u64
K, d, J, maxiters, N = 8, 1000, 6, 1000, 1000
cap = maxiters + 1
tg = pfmi.t_lowrank(d, r=8, seed=2)
run_seeds = rand_u64(20260928, np.arange(K, dtype=np.uint64), 9)
x0 = np.stack([pfmi.HostRNG(int(s)).rand(d) * 4 - 2 for s in run_seeds])
tab = np.concatenate([rand_u64(int(s), np.arange(1, cap + 1, dtype=np.uint64), 10) for s in run_seeds])
eng = pfmi.Engine(0); eng.set_target(tg); comm = pfmi.Comm.init_all([eng])
def streamed():
    eng.stream_enqueue(x0, N, tab, J, maxiters)
    eng.stream_wait()
    eng.pool_build_best(N)
    comm.psis_resample_enqueue(1000, seed=1)
    eng.defer(1); el = eng.elbo_batch_wait(); eng.defer(0)
    return comm.psis_resample_wait()
for _ in range(3): streamed()
ts = []
for i in range(120):
    t0 = time.perf_counter(); streamed(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("mean %.3f median %.3f max %.3f" % (ts.mean(), np.median(ts), ts.max()))
print("steps slower than 1.3 x median:", [(int(i), round(float(t), 2)) for i, t in enumerate(ts) if t > 1.3 * np.median(ts)])
