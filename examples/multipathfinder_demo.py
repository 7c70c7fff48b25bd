"""README usage example: multi-path Pathfinder on a built-in target, then single-path on a Python closure (needs an MI355X)."""
import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pathfinder.jl_amd"))
import numpy as np, pfmi
target = pfmi.t_lowrank(200, r=8, seed=2)
res = pfmi.multipathfinder(target, 1000, nruns=8, ndraws_elbo=200, rng=pfmi.HostRNG(1))
print(res)
run = res.pathfinder_results[0]
print(run.fit_iteration, run.elbo_estimates[run.fit_iteration - 1].value, type(run.fit_distribution.Sigma))
f = pfmi.CallbackTarget(5, logp=lambda x: -0.5 * float(x @ x), grad=lambda x: -x, logp_batch=lambda X: -0.5 * (X * X).sum(0))
single = pfmi.pathfinder(f, init=np.ones(5), ndraws=100)
print(single)
