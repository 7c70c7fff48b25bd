"""exploration script (not a test): wall-clock of the public multipathfinder() at config 3"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd"))
import numpy as np, pfmi, cProfile, pstats
tg = pfmi.t_lowrank(1000, 8, 2)
eng = pfmi.Engine(0)
for rep in range(3):
    t0 = time.perf_counter()
    res = pfmi.multipathfinder(tg, 1000, nruns=64, ndraws_elbo=1000, rng=pfmi.HostRNG(1), engine=eng)
    print("multipathfinder wall %.1f ms  pareto_k %.2f" % ((time.perf_counter() - t0) * 1e3, res.psis_result.pareto_shape))
pr = cProfile.Profile(); pr.enable()
res = pfmi.multipathfinder(tg, 1000, nruns=64, ndraws_elbo=1000, rng=pfmi.HostRNG(1), engine=eng)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
