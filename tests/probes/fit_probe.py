"""exploration script (not a test): fit kernel time vs history length / dimension"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd"))
import numpy as np, pfmi
eng = pfmi.Engine(0)
for d in (250, 1000):
    tg = pfmi.t_lowrank(d, 8, 2)
    eng.set_target(tg)
    x0 = pfmi.HostRNG(5).rand(64 * d).reshape(64, d) * 4 - 2
    eng.optimize_batch(x0, 6)
    for J in (1, 2, 4, 6, 8):
        eng.fit_batch(J); eng.sync()
        eng.profile(True); eng.fit_batch(J); eng.sync()
        print(f"d={d} P={eng.P} J={J}: history {eng.kernel_time('history')[0]:.3f} ms  fit {eng.kernel_time('fit')[0]:.3f} ms")
        eng.profile(False)
