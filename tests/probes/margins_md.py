"""gpurun_out/parity_margins.json (written by the GPU test session, tests/margins.py) -> profiles/rNN_parity_margins.md"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_margins.json"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04_parity_margins.md"
rows = json.load(open(src))["rows"]
out = ["# Parity margins: largest deviation GPU vs oracle observed by the `-m gpu` test session", "",
       "Written by `tests/margins.py` (every oracle comparison goes through `margins.check`, which asserts `bound` and records the",
       "maximum).  `contract` = SURVEY.md 8(d).  Deviations are normalised as the contract states them (`|a-b| / (1+|b|)` for",
       "log densities, ELBO, SE, logdet; `max|dW| / max|W|`; `max|d mu| / (1+max|mu|)`; per-column relative for draws).", "",
       "| config | quantity | comparisons | max deviation | asserted bound | contract | margin (bound / max) | note |", "|---|---|---|---|---|---|---|---|"]
for r in rows:
    m = r["max"]
    margin = "inf" if m == 0 else f"{r['bound'] / m:.1f}x" if r["bound"] != float("inf") and r["bound"] < 1e300 else "-"
    b = "recorded only" if r["bound"] > 1e300 else f"{r['bound']:.0e}"
    c = "-" if r.get("contract") is None else f"{r['contract']:.0e}"
    out.append(f"| {r['config']} | {r['quantity']} | {r['n']} | {m:.2e} | {b} | {c} | {margin} | {r.get('why') or ''} |")
open(dst, "w").write("\n".join(out) + "\n")
print(dst, len(rows), "rows")
