"""exploration script (not a test): cluster fit kernel vs memory-resident kernel (PFMI_FIT_KERNEL=mem)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, pfmi
eng = pfmi.Engine(0)
def run(tg, K, J, scale, maxit):
    eng.set_target(tg)
    x0 = pfmi.HostRNG(3).rand(K * tg.d).reshape(K, tg.d) * 2 * scale - scale
    eng.optimize_batch(x0, J, maxit)
    out = {}
    for mode in ("mem", "cluster"):
        os.environ["PFMI_FIT_KERNEL"] = mode
        eng.fit_batch(J); eng.sync()
        t0 = time.perf_counter(); eng.fit_batch(J); eng.sync(); dt = time.perf_counter() - t0
        st, je, ld, nr = eng.fit_status()
        fits = [eng.get_fit(p, int(je[p])) for p in sorted({1, eng.P // 2, eng.P - 1})]
        out[mode] = (st, ld, fits, dt)
    a, b = out["mem"], out["cluster"]
    ok = a[0] == 0
    print(f"d={tg.d} J={J} P={eng.P}: status_equal {np.array_equal(a[0], b[0])} logdet err {np.max(np.abs(a[1][ok]-b[1][ok])/(1+np.abs(a[1][ok]))):.2e} "
          + " ".join(f"{k} {max((np.max(np.abs(fa[k]-fb[k]))/(1e-300+np.max(np.abs(fa[k]))) if fa[k].size else 0.0) for fa, fb in zip(a[2], b[2])):.1e}" for k in ("mu", "qr_factors", "T", "V"))
          + f" | mem {a[3]*1e3:.2f} ms cluster {b[3]*1e3:.2f} ms")
run(pfmi.t_diag(3000, 1), 2, 6, 2, 30)
run(pfmi.t_lowrank(2000, 8, 2), 4, 6, 2, 60)
run(pfmi.t_funnel(2500), 2, 10, 10, 30)
run(pfmi.t_funnel(10000), 8, 10, 10, 200)
