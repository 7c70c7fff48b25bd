import os, sys, time
os.environ["PFMI_DEBUG_HOOKS"] = "1"          # the library reads PFMI_HISTORY_KERNEL only then
sys.path.insert(0, "pathfinder.jl_amd")
import numpy as np, pfmi
out = {}
for d, J, K, tname in ((10000, 10, 8, "funnel"), (6000, 6, 4, "diag"), (7500, 8, 3, "diag"), (3000, 6, 8, "diag"), (4096, 6, 8, "diag")):
    tg = pfmi.t_funnel(d) if tname == "funnel" else pfmi.t_diag(d, 1)
    sc = 10.0 if tname == "funnel" else 2.0
    x0 = pfmi.HostRNG(5).rand(K * d).reshape(K, d) * 2 * sc - sc
    res = {}
    for mode in ("lean", "prefetch"):
        if mode == "prefetch": os.environ["PFMI_HISTORY_KERNEL"] = "prefetch"
        else: os.environ.pop("PFMI_HISTORY_KERNEL", None)
        e = pfmi.Engine(0)
        e.set_target(tg)
        e.optimize_batch(x0, J, 200)
        e.profile(2)
        for _ in range(3): e.fit_batch(J)
        st, je, ld, nr = e.fit_status()
        ms, n = e.kernel_time("history")
        f = e.get_fit(e.P - 1, int(je[-1]))
        res[mode] = (st.copy(), je.copy(), nr.copy(), ld.copy(), f["alpha"].copy(), f["mu"].copy(), ms / n)
        e.close()
    a, b = res["lean"], res["prefetch"]
    same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a[:6], b[:6]))
    print(f"d={d} J={J} K={K} {tname}: lean {a[6]:.3f} ms, prefetch {b[6]:.3f} ms, bit-identical {same}")
