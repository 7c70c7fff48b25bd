"""time of the device L-BFGS at config 3 (64 paths, d = 1000, J = 6) and at d = 100 / 10^4"""
import sys, time
sys.path.insert(0, "pathfinder.jl_amd")
import numpy as np
import pfmi
eng = pfmi.Engine(0)
for name, tg, K, J, sc, mi in (("C3", pfmi.targets.t_lowrank(1000), 64, 6, 2.0, 1000), ("d100", pfmi.targets.t_lowrank(100), 64, 6, 2.0, 1000),
                               ("funnel1e4", pfmi.targets.t_funnel(10000), 8, 10, 2.0, 200),
                               ("diag12000", pfmi.targets.t_diag(12000), 4, 6, 2.0, 200), ("lr16_600", pfmi.targets.t_lowrank(600, 16, 3), 16, 6, 2.0, 1000)):
    d = tg.d
    eng.set_target(tg)
    x0 = pfmi.HostRNG(2024).rand(K * d).reshape(K, d) * 2 * sc - sc
    for _ in range(2):
        npts = eng.optimize_batch(x0, J, mi)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        npts = eng.optimize_batch(x0, J, mi)
    eng.sync()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name}: {dt:.3f} ms  points/path mean {npts.mean():.1f} max {npts.max()}  -> {dt * 1e3 / npts.max():.2f} us per iteration of the longest path")
