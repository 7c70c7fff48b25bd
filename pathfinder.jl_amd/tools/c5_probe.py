import sys, time, numpy as np
sys.path.insert(0,'pathfinder.jl_amd'); sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import pfmi
from helpers import fit_seeds
d, J, N, K = 10000, 10, 2000, 4
tg = pfmi.t_funnel(d)
rng = pfmi.HostRNG(1)
t0=time.time()
traces = [pfmi.optimize_with_trace(tg, rng.rand(d)*20-10, history_length=J, maxiters=40) for _ in range(K)]
print("traces", [len(t)-1 for t in traces], time.time()-t0)
e = pfmi.Engine(0); e.set_target(tg); e.set_traces([t.points for t in traces],[t.gradients for t in traces])
e.profile(True)
t0=time.time(); e.fit_batch(J); e.sync(); print("fit s", time.time()-t0)
st, je, ld, nr = e.fit_status(); print("status counts", np.bincount(st), "jeff max", je.max(), "rej", nr)
seeds = fit_seeds(e.P, 3)
t0=time.time(); elbo, se, best = e.elbo_batch(N, seeds); e.sync(); dt=time.time()-t0
nd=(e.P-K)*N
print("elbo s", dt, "draws/s", nd/dt, "best", best, "finite", np.isfinite(elbo).sum())
pts=[int(e.offsets[k])+int(best[k]) for k in range(K)]
t0=time.time(); e.pool_build(N, pts, seeds[pts]); e.sync(); print("pool s", time.time()-t0)
ptr,cnt=e.pool_log_ratios_dev(); r=e.psis_dev(ptr,cnt); print("psis k", r["pareto_shape"], "sumw", r["weights"].sum())
idx=e.resample_indices(cnt, 2000, seed=1); X=e.pool_gather(idx); print("draws", X.shape, np.isfinite(X).all())
for n in ("history","fit","elbo_draws","elbo_draws_x","psis","resample"): print(n, e.kernel_time(n))
