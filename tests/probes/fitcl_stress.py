"""stress (not a test): cluster fit kernel vs memory-resident kernel over many fits / repetitions -- every logdet and mu must agree"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd"))
import numpy as np, pfmi
eng = pfmi.Engine(0)
bad = 0; total = 0
for (tgf, d, K, J, scale, maxit, reps) in [(pfmi.t_funnel, 10000, 8, 10, 10, 200, 6), (lambda d: pfmi.t_diag(d, 1), 3000, 16, 6, 2, 100, 10),
                                           (lambda d: pfmi.t_lowrank(d, 8, 2), 2000, 16, 6, 2, 100, 10)]:
    tg = tgf(d); eng.set_target(tg)
    x0 = pfmi.HostRNG(3).rand(K * d).reshape(K, d) * 2 * scale - scale
    eng.optimize_batch(x0, J, maxit)
    os.environ["PFMI_FIT_KERNEL"] = "mem"; eng.fit_batch(J); st0, je0, ld0, _ = eng.fit_status()
    os.environ["PFMI_FIT_KERNEL"] = "cluster"
    for r in range(reps):
        eng.fit_batch(J); st, je, ld, _ = eng.fit_status()
        ok = st0 == 0
        err = np.abs(ld[ok] - ld0[ok]) / (1 + np.abs(ld0[ok]))
        nb = int((err > 1e-9).sum()) + int((st != st0).sum())
        bad += nb; total += int(ok.sum())
    print(f"d={d} J={J} P={eng.P}: {reps} repetitions, mismatches so far {bad} of {total} fits", flush=True)
print("TOTAL mismatches", bad, "of", total)
