"""Can the draw writer and the user's reader kernel overlap?  Two engines (two streams) on one GPU, each scanning half of the
sample through the device closure, against one engine scanning all of it (run on the GPU box; XW_WAVES=8 variant via PFMI_LIB_PATH)."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "pathfinder.jl_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import pfmi  # noqa: E402
from bench import _demo_device_target  # noqa: E402

K, d, J, N = 16, 1000, 6, 1000
tg = pfmi.t_lowrank(d, r=8, seed=2)
x0 = pfmi.HostRNG(1).rand(K * d).reshape(K, d) * 4 - 2
e = pfmi.Engine(0)
e.set_target(tg)
e.optimize_batch(x0, J, 1000)
trs = [e.get_trace(k, logp=False) for k in range(K)]
dt_ = _demo_device_target(pfmi, tg)


def make(paths):
    g = pfmi.Engine(0)
    g.set_target(dt_)
    g.set_traces([trs[k][0] for k in paths], [trs[k][2] for k in paths])
    g.fit_batch(J)
    sd = pfmi.hostrng.rand_u64(5, np.arange(g.P, dtype=np.uint64), 9)
    g.elbo_batch(N, sd)
    return g, sd


for chunk in (os.environ.get("PFMI_DEVCB_CHUNK_MB", "2048"),):
    eall, sall = make(range(K))
    ea, sa = make(range(0, K // 2))
    eb, sb = make(range(K // 2, K))
    nd = (eall.P - K) * N
    for rep in range(2):
        eall.sync(); t0 = time.perf_counter()
        eall.elbo_batch_enqueue(N, sall); eall.elbo_batch_wait()
        t_one = time.perf_counter() - t0
        ea.sync(); eb.sync(); t0 = time.perf_counter()
        ea.elbo_batch_enqueue(N, sa); eb.elbo_batch_enqueue(N, sb)
        ea.elbo_batch_wait(); eb.elbo_batch_wait()
        t_two = time.perf_counter() - t0
        print(f"lib={os.environ.get('PFMI_LIB_PATH', 'default')} chunkMB={chunk} fits={eall.P - K}: one stream {t_one * 1e3:.2f} ms "
              f"({16.0 * d * nd / t_one / 1e9:.0f} GB/s), two streams {t_two * 1e3:.2f} ms ({16.0 * d * nd / t_two / 1e9:.0f} GB/s)")
