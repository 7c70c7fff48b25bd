"""exploration script (not a test): panel-fit time at config 5's shape against the number of resident workgroups
(PFMI_FIT_PANEL_GRID: does a per-workgroup scratch that stays inside the 256 MB MALL beat more workgroups on HBM?)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd"))
import numpy as np, pfmi
K, d, J, maxit = 8, 10000, 10, 200
eng = pfmi.Engine(0)
eng.set_target(pfmi.t_funnel(d))
x0 = pfmi.HostRNG(5).rand(K * d).reshape(K, d) * 20 - 10
eng.optimize_batch(x0, J, maxit)
L = pfmi.lib()
ref = None
for g in sys.argv[1:] or ["0"]:
    L.pfmi_debug_set(b"PFMI_FIT_PANEL_GRID", g.encode() if g != "0" else None)
    eng.fit_batch(J); eng.sync()
    eng.profile(True)
    for _ in range(3):
        eng.fit_batch(J)
    eng.sync()
    t, n = eng.kernel_time("fit")
    st = eng.fit_status()
    ld = st[2].copy()
    if ref is None:
        ref = ld
    same = np.array_equal(np.nan_to_num(ld), np.nan_to_num(ref))
    print(f"grid={g}: P={eng.P} fit {t / n:.3f} ms per launch ({n} launches), logdet identical to the first setting: {same}")
    eng.profile(False)
