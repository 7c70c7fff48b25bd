"""exploration script (not a test): sha256 of the draw writer's output (x, logq) over the shapes of
test_draw_writer_matches_lane_kernel_and_oracle_normals -- run it under two builds (PFMI_LIB_PATH) and diff the lines:
a restructured writer must reproduce the previous one bit for bit.  usage: PFMI_DEBUG_HOOKS=1 python tests/probes/xw_bits.py"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd"))
os.environ["PFMI_DEBUG_HOOKS"] = "1"
os.environ["PFMI_ELBO_KERNEL"] = "xw"
import numpy as np
import pfmi

SHAPES = [("lr", 1000, 6, 1000, 2.0, 25), ("lr", 130, 6, 200, 2.0, 25), ("diag", 10, 6, 64, 2.0, 25), ("diag", 33, 3, 17, 2.0, 12),
          ("funnel", 500, 10, 300, 3.0, 30), ("lr", 3000, 10, 272, 2.0, 20), ("diag", 2500, 16, 100, 2.0, 24), ("funnel", 10000, 10, 160, 10.0, 12),
          ("lr", 1000, 6, 2500, 2.0, 12), ("diag", 64, 2, 1000, 2.0, 12), ("lr", 1008, 8, 512, 2.0, 12)]
eng = pfmi.Engine(0)
for tname, d, J, N, scale, maxit in SHAPES:
    tg = {"diag": lambda d: pfmi.t_diag(d, 1), "lr": lambda d: pfmi.t_lowrank(d, 8, 2), "funnel": pfmi.t_funnel}[tname](d)
    eng.set_target(tg)
    x0 = pfmi.HostRNG(3).rand(2 * d).reshape(2, d) * 2 * scale - scale
    eng.optimize_batch(x0, J, maxit)
    eng.fit_batch(J)
    status = eng.fit_status()[0]
    h = hashlib.sha256()
    for p in sorted({1, eng.P - 1, int(eng.offsets[1]) + 1, eng.P // 2}):
        if status[p] != 0:
            continue
        X, lp, lq = eng.draws(p, 1000 + p, N)
        h.update(np.ascontiguousarray(X).tobytes()); h.update(lq.tobytes())
        X2, _, lq2 = eng.draws(p, 1000 + p, 40, n0=53)
        h.update(np.ascontiguousarray(X2).tobytes()); h.update(lq2.tobytes())
    pp = [int(eng.offsets[k]) + 1 for k in range(2)]
    eng.pool_build(N, pp, np.array([77, 78], dtype=np.uint64))
    pool, lr = eng.pool_get()
    h.update(np.ascontiguousarray(pool).tobytes())
    print(f"{tname} d={d} J={J} N={N}: {h.hexdigest()[:32]} finite={bool(np.isfinite(pool).all())}")
