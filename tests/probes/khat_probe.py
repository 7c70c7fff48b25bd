"""Pareto k-hat of multipathfinder at config 3's shape for target variants (which ones can Pathfinder fit at d = 1000, J = 6?)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "pathfinder.jl_amd"))
import numpy as np
import pfmi
d = 1000
rng = pfmi.HostRNG(2)
logsig = -0.5 + 1.0 * rng.rand(d); W = rng.randn(d * 8).reshape(d, 8); m = rng.randn(d)
variants = {
    "headline": pfmi.GaussTarget(m, np.exp(2 * logsig), W),
    "unit diag, r=4, W~N(0,1)": pfmi.GaussTarget(m, np.ones(d), W[:, :4]),
    "unit diag, r=4, W*0.1": pfmi.GaussTarget(m, np.ones(d), 0.1 * W[:, :4]),
    "diag e^{+-0.05}, r=4, W*0.1": pfmi.GaussTarget(m, np.exp(0.2 * logsig), 0.1 * W[:, :4]),
    "diag e^{+-0.5}, no W": pfmi.GaussTarget(m, np.exp(2 * logsig)),
    "unit diag, r=8, W~N(0,1)": pfmi.GaussTarget(m, np.ones(d), W),
}
eng = pfmi.Engine(0)
for name, tg in variants.items():
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = pfmi.multipathfinder(tg, 1000, nruns=64, ndraws_elbo=1000, rng=pfmi.HostRNG(20260928), engine=eng)
    L = [len(p.optim_trace) - 1 for p in r.pathfinder_results]
    print(f"{name:32s} khat {r.psis_result.pareto_shape:7.3f}  fits {sum(L):6d}  best iter median {int(np.median([p.fit_iteration for p in r.pathfinder_results]))}", flush=True)
