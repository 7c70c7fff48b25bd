"""device-closure scan at config 3's shape on a sample of paths: writer / reader kernel times (run on the GPU box).
usage: [PFMI_LIB_PATH=...] python tests/probes/devcb_probe.py [npaths] [d] [J] [N] [target]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "pathfinder.jl_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import pfmi  # noqa: E402
from bench import _demo_device_target  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
J = int(sys.argv[3]) if len(sys.argv) > 3 else 6
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
tname = sys.argv[5] if len(sys.argv) > 5 else "lowrank"
maxit = int(sys.argv[6]) if len(sys.argv) > 6 else 1000
tg = {"lowrank": lambda: pfmi.t_lowrank(d, r=8, seed=2), "funnel": lambda: pfmi.t_funnel(d), "diag": lambda: pfmi.t_diag(d, 1)}[tname]()
sc = 10.0 if tname == "funnel" else 2.0
x0 = pfmi.HostRNG(1).rand(K * d).reshape(K, d) * 2 * sc - sc
e = pfmi.Engine(0)
e.set_target(tg)
e.optimize_batch(x0, J, maxit)
trs = [e.get_trace(k, logp=False) for k in range(K)]
e.set_target(_demo_device_target(pfmi, tg))
e.set_traces([t[0] for t in trs], [t[2] for t in trs])
e.fit_batch(J)
seeds = pfmi.hostrng.rand_u64(5, np.arange(e.P, dtype=np.uint64), 9)
e.elbo_batch(N, seeds)
e.sync()
t0 = time.perf_counter()
for _ in range(3):
    e.elbo_batch_enqueue(N, seeds)
el = e.elbo_batch_wait()
dt = (time.perf_counter() - t0) / 3
import hashlib  # noqa: E402
print("elbo table sha1", hashlib.sha1(np.ascontiguousarray(el[0]).tobytes()).hexdigest()[:16], "best", el[2].tolist())
e.profile(True)
e.elbo_batch(N, seeds)
tw, nw = e.kernel_time("elbo_draws_x")
tr, nr = e.kernel_time("device_callback")
nd = (e.P - K) * N
print(f"lib={os.environ.get('PFMI_LIB_PATH', 'default')} fits={e.P - K} draws={nd} wall={dt * 1e3:.3f} ms writer={tw:.3f} ms ({8.0 * d * nd / tw / 1e6:.0f} GB/s) "
      f"reader={tr:.3f} ms ({8.0 * d * nd / tr / 1e6:.0f} GB/s) total {16.0 * d * nd / dt / 1e9:.0f} GB/s")
