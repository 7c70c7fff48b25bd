"""fuzz (not a test): random shapes -- single-pass scan vs lane kernel (per-draw logp / logq), register / panel vs memory-resident fit
kernel (logdet, mu), history walk (effective history, sources, rejections, alpha) vs the oracle, device L-BFGS sanity.  Prints the worst discrepancies; any NaN pattern mismatch is reported."""
import os, sys
os.environ["PFMI_DEBUG_HOOKS"] = "1"          # the library reads the PFMI_*_KERNEL selectors only then
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "pathfinder.jl_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, pfmi
from helpers import fit_seeds
from oracle import pf_oracle as po
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
eng = pfmi.Engine(0)
worst = dict(lp=0.0, lq=0.0, elbo=0.0, ld=0.0, mu=0.0); bad = 0; ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for case in range(ncase):
    d = int(rng.choice([3, 7, 16, 17, 33, 64, 100, 255, 256, 257, 600, 1024, 1025, 1500, 2048, 2561, 5121, 7000]))
    J = int(rng.integers(1, 11)); K = int(rng.integers(1, 4)); N = int(rng.choice([64, 65, 100, 333, 800, 1000]))
    if d > 1500: N = min(N, 100)
    kind = rng.choice(["iso", "diag", "lr", "funnel"])
    tg = {"iso": lambda: pfmi.t_iso(d), "diag": lambda: pfmi.t_diag(d, int(rng.integers(1, 99))),
          "lr": lambda: pfmi.t_lowrank(d, int(rng.choice([3, 8, 11, 16])), int(rng.integers(1, 99))), "funnel": lambda: pfmi.t_funnel(d)}[kind]()
    eng.set_target(tg)
    sc = 10.0 if kind == "funnel" else 2.0
    x0 = rng.uniform(-sc, sc, size=(K, d))
    eng.optimize_batch(x0, J, 25 if kind == "funnel" else 60)
    res = {}
    for fk in ("", "mem"):
        if fk: os.environ["PFMI_FIT_KERNEL"] = fk
        else: os.environ.pop("PFMI_FIT_KERNEL", None)
        eng.fit_batch(J)
        st, je, ld, _ = eng.fit_status()
        res[fk] = (st.copy(), ld.copy(), eng.get_fit(eng.P - 1, int(je[-1]))["mu"])
    os.environ.pop("PFMI_FIT_KERNEL", None)
    eng.fit_batch(J)
    st, je, ld, nr = eng.fit_status()
    for k in range(K):                                   # history walk vs the oracle
        th, _, gr = eng.get_trace(k, logp=False)
        alpha_all, hl, hs, nrej = po.lbfgs_history(th, gr, J)
        p0 = int(eng.offsets[k])
        if not np.array_equal(je[p0:p0 + len(th)], hl) or int(nr[k]) != int(nrej): bad += 1; print("HISTORY MISMATCH", d, J, kind, k)
        pl = p0 + len(th) - 1
        f = eng.get_fit(pl, int(je[pl]))
        worst["alpha"] = max(worst.get("alpha", 0.0), float(np.max(np.abs(f["alpha"] - alpha_all[-1]) / np.abs(alpha_all[-1]))))
    if not np.array_equal(res[""][0], res["mem"][0]): bad += 1; print("STATUS MISMATCH", d, J, kind)
    ok = res[""][0] == 0
    if ok.any():
        worst["ld"] = max(worst["ld"], float(np.max(np.abs(res[""][1][ok] - res["mem"][1][ok]) / (1 + np.abs(res["mem"][1][ok])))))
        if ok[-1]: worst["mu"] = max(worst["mu"], float(np.max(np.abs(res[""][2] - res["mem"][2]) / (1 + np.abs(res["mem"][2])))))
    seeds = fit_seeds(eng.P, int(rng.integers(1, 1000)))
    out = {}
    for mode in ("lane", "qf"):
        os.environ["PFMI_ELBO_KERNEL"] = mode
        elbo, se, best = eng.elbo_batch(N, seeds)
        pts = sorted({1 if eng.P > 1 else 0, eng.P // 2, eng.P - 1})
        out[mode] = (elbo.copy(), [eng.elbo_logs(p, N) for p in pts])
    os.environ.pop("PFMI_ELBO_KERNEL", None)
    a, b = out["qf"], out["lane"]
    if not np.array_equal(np.isnan(a[0]), np.isnan(b[0])): bad += 1; print("NAN PATTERN MISMATCH", d, J, N, kind)
    fin = np.isfinite(b[0]) & np.isfinite(a[0])
    if fin.any(): worst["elbo"] = max(worst["elbo"], float(np.max(np.abs(a[0][fin] - b[0][fin]) / (1 + np.abs(b[0][fin])))))
    for (lpa, lqa), (lpb, lqb) in zip(a[1], b[1]):
        f2 = np.isfinite(lpb) & np.isfinite(lpa)
        if not np.array_equal(np.isfinite(lpa), np.isfinite(lpb)): bad += 1; print("LOGP FINITENESS MISMATCH", d, J, N, kind)
        if f2.any(): worst["lp"] = max(worst["lp"], float(np.max(np.abs(lpa[f2] - lpb[f2]) / (1 + np.abs(lpb[f2])))))
        worst["lq"] = max(worst["lq"], float(np.nanmax(np.abs(lqa - lqb) / (1 + np.abs(lqb)))))
print("cases", ncase, "pattern mismatches", bad, "worst relative discrepancies", worst)
