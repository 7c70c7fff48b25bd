import sys, cProfile, pstats, io
sys.path.insert(0, "pathfinder.jl_amd")
import pfmi
tg = pfmi.t_lowrank(1000, r=8, seed=2)
eng = pfmi.Engine(0)
kw = dict(nruns=64, ndraws_elbo=1000, history_length=6, engine=eng)
for _ in range(3): pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
