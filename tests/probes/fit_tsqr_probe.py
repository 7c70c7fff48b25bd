"""exploration script (not a test): the TSQR + Householder-reconstruction fit (default for 1024 < d <= 16384) against the left-looking panel
kernel (PFMI_FIT_KERNEL=panel) and the column-by-column kernel (=mem): outputs and time at several shapes.
usage: python tests/probes/fit_tsqr_probe.py [c5] [small]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd"))
import numpy as np, pfmi
L = pfmi.lib()
shapes = {"small": [("diag", 1500, 4, 2, 12), ("lr", 3000, 6, 2, 14), ("funnel", 6000, 10, 2, 16), ("diag", 12000, 10, 2, 14), ("diag", 2000, 16, 2, 22)],
          "c5": [("funnel", 10000, 10, 8, 200)],
          "big": [("diag", 20000, 5, 4, 30), ("diag", 32768, 10, 4, 40)],
          "j16": [("diag", 2000, 16, 2, 22), ("diag", 8000, 16, 8, 120), ("funnel", 10000, 16, 8, 100)]}
which = [a for a in sys.argv[1:] if a in shapes] or ["small"]
for w in which:
    for tname, d, J, K, maxit in shapes[w]:
        tg = {"diag": lambda d: pfmi.t_diag(d, 1), "lr": lambda d: pfmi.t_lowrank(d, 8, 2), "funnel": pfmi.t_funnel}[tname](d)
        eng = pfmi.Engine(0)
        eng.set_target(tg)
        x0 = pfmi.HostRNG(4).rand(K * d).reshape(K, d) * (20 if tname == "funnel" else 4) - (10 if tname == "funnel" else 2)
        if d <= 16384:
            eng.optimize_batch(x0, J, maxit)
        else:                                                  # the device optimiser stops at 16 384 coordinates: host driver, uploaded traces
            from pfmi.optimize import optimize_with_trace
            trs = [optimize_with_trace(tg, x0[k], history_length=J, maxiters=maxit) for k in range(K)]
            eng.set_traces([t.points for t in trs], [t.gradients for t in trs])
        res = {}
        for kern in ("tsqr", "panel", "mem"):
            L.pfmi_debug_set(b"PFMI_FIT_KERNEL", kern.encode())               # ("tsqr" also forces the TSQR kernel where the panel kernel is the default: KPAD = 32)
            eng.fit_batch(J); eng.sync()
            eng.profile(True)
            for _ in range(3):
                eng.fit_batch(J)
            eng.sync()
            t, n = eng.kernel_time("fit")
            eng.profile(False)
            st, je, ld, nr = eng.fit_status()
            pts = sorted(set([1, 2, 3, 5, eng.P // 2, eng.P - 1]) & set(range(eng.P)))
            res[kern] = (t / n, st.copy(), ld.copy(), {p: eng.get_fit(p, int(je[p])) for p in pts})
        L.pfmi_debug_set(b"PFMI_FIT_KERNEL", None)
        t0, st0, ld0, f0 = res["tsqr"]
        line = f"{tname} d={d} J={J} P={eng.P}: " + "  ".join(f"{k} {res[k][0]:.3f} ms" for k in res)
        for other in ("panel", "mem"):
            t1, st1, ld1, f1 = res[other]
            ok = st0 == st1
            e_ld = np.nanmax(np.abs(ld0 - ld1) / (1 + np.abs(ld1)))
            e_mu = max(np.abs(f0[p]["mu"] - f1[p]["mu"]).max() / (1 + np.abs(f1[p]["mu"]).max()) for p in f0)
            e_qr = max((np.abs(f0[p]["qr_factors"] - f1[p]["qr_factors"]).max() / max(np.abs(f1[p]["qr_factors"]).max(), 1e-300)) if f0[p]["j"] else 0.0 for p in f0)
            e_T = max(np.abs(f0[p]["T"] - f1[p]["T"]).max() if f0[p]["j"] else 0.0 for p in f0)
            e_V = max(np.abs(f0[p]["V"] - f1[p]["V"]).max() / max(np.abs(f1[p]["V"]).max(), 1e-300) if f0[p]["j"] else 0.0 for p in f0)
            e_D = max(np.abs(f0[p]["D"] - f1[p]["D"]).max() / max(np.abs(f1[p]["D"]).max(), 1e-300) if f0[p]["j"] else 0.0 for p in f0)
            line += f"\n    vs {other}: status equal {bool(np.all(ok))}  logdet {e_ld:.2e}  mu {e_mu:.2e}  QR {e_qr:.2e}  T {e_T:.2e}  V {e_V:.2e}  D {e_D:.2e}"
        print(line, flush=True)
        eng.close()
