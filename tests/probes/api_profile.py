import cProfile, pstats, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pathfinder.jl_amd"))
import pfmi
tg = pfmi.t_lowrank(1000, r=8, seed=2)
eng = pfmi.Engine(0)
kw = dict(nruns=64, ndraws_elbo=1000, history_length=6, engine=eng)
for _ in range(2):
    pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw); ts.append(time.perf_counter() - t0)
print("wall ms", [round(t * 1e3, 2) for t in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
