"""usage: python tests/probes/api_cprofile.py [K] -- where the public call's host time goes (cProfile of pfmi.multipathfinder, d = 1000, N = 1000)"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pathfinder.jl_amd")]
import pfmi  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tg = pfmi.t_lowrank(1000, r=8, seed=2)
eng = pfmi.Engine(0)
kw = dict(nruns=K, ndraws_elbo=1000, history_length=6, engine=eng)
for _ in range(3):
    pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
ts = []
for _ in range(15):
    t0 = time.perf_counter()
    pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"K={K}: api wall median {sorted(ts)[len(ts) // 2]:.3f} ms  min {min(ts):.3f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    pfmi.multipathfinder(tg, 1000, rng=pfmi.HostRNG(20260928), **kw)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
