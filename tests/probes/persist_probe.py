"""usage: python tests/probes/persist_probe.py [K] -- the ELBO scan as one work-queue launch (PFMI_QF_PERSIST = number of workgroups) against the
regular launch (a workgroup per fit + tail pieces): time per scan and bit-identity of the ELBO table (d = 1000, J = 6, N = 1000)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pathfinder.jl_amd")]
import numpy as np  # noqa: E402
import pfmi  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tg = pfmi.t_lowrank(1000, r=8, seed=2)
L = pfmi.lib()
eng = pfmi.Engine(0)
eng.set_target(tg)
x0 = pfmi.HostRNG(3).rand(K * 1000).reshape(K, 1000) * 4 - 2
eng.optimize_batch(x0, 6, 1000)
eng.fit_batch(6)
seeds = pfmi.hostrng.rand_u64(5, np.arange(eng.P, dtype=np.uint64), 9)
ref = None
for mode in (None, b"256", b"248", b"224", b"512", None):
    assert L.pfmi_debug_set(b"PFMI_QF_PERSIST", mode) == 0
    eng.elbo_batch(1000, seeds)
    ts = []
    for _ in range(5):
        eng.sync()
        t0 = time.perf_counter()
        el = eng.elbo_batch(1000, seeds)
        ts.append((time.perf_counter() - t0) * 1e3)
    if ref is None:
        ref = el
    same = np.array_equal(el[0], ref[0], equal_nan=True) and np.array_equal(el[2], ref[2])
    print(f"K={K} fits={eng.P - K}: {'regular launch' if mode is None else 'work queue of ' + mode.decode() + ' workgroups':32s} {min(ts):8.3f} ms (min of 5)  bit-identical {same}")
