"""exploration script (not a test): qf kernel vs lane kernel agreement + timing"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, pfmi
from helpers import fit_seeds
eng = pfmi.Engine(0)
def run(tg, K, J, N, scale=2, maxit=1000, modes=("lane", "qf")):
    eng.set_target(tg)
    x0 = pfmi.HostRNG(3).rand(K * tg.d).reshape(K, tg.d) * 2 * scale - scale
    npts = eng.optimize_batch(x0, J, maxit)
    eng.fit_batch(J)
    seeds = fit_seeds(eng.P, 1)
    out = {}
    for m in modes:
        os.environ["PFMI_ELBO_KERNEL"] = m
        eng.elbo_batch(N, seeds)
        eng.sync(); t0 = time.perf_counter()
        elbo, se, best = eng.elbo_batch(N, seeds)
        eng.sync(); dt = time.perf_counter() - t0
        pts = sorted({1, min(3, eng.P - 1), eng.P // 2, eng.P - 1})
        logs = [eng.elbo_logs(p, N) for p in pts]
        out[m] = (elbo, se, best, logs, dt)
    ref = out[modes[0]]
    for m in modes[1:]:
        o = out[m]
        ok = np.isfinite(ref[0])
        e_err = np.max(np.abs(o[0][ok] - ref[0][ok]) / (1 + np.abs(ref[0][ok])))
        lp_err = max(np.max(np.abs(a[0] - b[0]) / (1 + np.abs(b[0]))) for a, b in zip(o[3], ref[3]))
        lq_err = max(np.max(np.abs(a[1] - b[1]) / (1 + np.abs(b[1]))) for a, b in zip(o[3], ref[3]))
        nfit = eng.P - K
        print(f"d={tg.d} J={J} N={N} P={eng.P}: {m} vs {modes[0]}: elbo {e_err:.2e} lp {lp_err:.2e} lq {lq_err:.2e} best_equal {np.array_equal(o[2], ref[2])} nan_equal {np.array_equal(np.isnan(o[0]), np.isnan(ref[0]))}"
              f" | {modes[0]} {ref[4]*1e3:.2f} ms, {m} {o[4]*1e3:.2f} ms ({nfit*N/o[4]:.3e} draws/s)")
if len(sys.argv) > 1 and sys.argv[1] == "c3":
    run(pfmi.t_lowrank(1000, 8, 2), 64, 6, 1000, modes=("mfma", "qf"))
elif len(sys.argv) > 1 and sys.argv[1] == "c5":
    run(pfmi.t_funnel(10000), 4, 10, 2000, scale=10, maxit=60, modes=("qf", "qf"))
elif len(sys.argv) > 1 and sys.argv[1] == "big":
    run(pfmi.t_lowrank(1000, 8, 2), 64, 6, 1000, modes=("mfma", "qf"))
    run(pfmi.t_funnel(10000), 4, 10, 2000, scale=10, maxit=60, modes=("lane", "qf"))
else:
    run(pfmi.t_iso(10), 2, 6, 100)
    run(pfmi.t_diag(30, 1), 2, 6, 200)
    run(pfmi.t_lowrank(50, 8, 2), 2, 6, 200)
    run(pfmi.t_lowrank(300, 8, 2), 2, 6, 500)
    run(pfmi.t_funnel(12), 2, 6, 100, scale=10, maxit=40)
    run(pfmi.t_diag(30, 1), 2, 10, 200)
    run(pfmi.t_lowrank(50, 8, 2), 2, 16, 200)
    run(pfmi.t_diag(3000, 1), 2, 6, 200, maxit=30)
    run(pfmi.t_funnel(2500), 2, 10, 300, scale=10, maxit=30)
    run(pfmi.t_lowrank(1000, 8, 2), 8, 6, 1000, modes=("mfma", "qf"))
