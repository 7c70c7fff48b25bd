"""Where does the public API's wall time go?  (config 3 through pfmi.multipathfinder; cProfile of one warm call)"""
import cProfile, pstats, sys, time, io
sys.path.insert(0, "pathfinder.jl_amd")
import numpy as np
import pfmi

d, K, J, N_e, ndraws = 1000, 64, 6, 1000, 1000
tg = pfmi.targets.t_lowrank(d)
eng = pfmi.Engine(0)
master = 2024
kw = dict(nruns=K, ndraws_elbo=N_e, history_length=J, engine=eng, init_scale=2.0, maxiters=1000)
for _ in range(2):
    pfmi.multipathfinder(tg, ndraws, rng=pfmi.HostRNG(master), **kw)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); pfmi.multipathfinder(tg, ndraws, rng=pfmi.HostRNG(master), **kw); ts.append((time.perf_counter() - t0) * 1e3)
print("wall ms:", [round(t, 2) for t in ts])
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    pfmi.multipathfinder(tg, ndraws, rng=pfmi.HostRNG(master), **kw)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
