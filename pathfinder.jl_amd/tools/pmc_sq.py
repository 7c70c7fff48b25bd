#!/usr/bin/env python
"""Per-kernel sums of the counters in rocprofv3 --pmc result databases, as a markdown table.
usage: pmc_sq.py "<title>" pass1.db [pass2.db ...] > profiles/<name>.md
Counters are summed over every launch of a kernel family (name up to the template arguments) in the run."""
import glob
import sqlite3
import sys

title = sys.argv[1]
print(f"# {title}\n\n| kernel | counter | sum over launches | launches | per launch | avg launch (ms) |\n|---|---|---|---|---|---|")
for pat in sys.argv[2:]:
    for path in sorted(glob.glob(pat)):
        con = sqlite3.connect(path)
        rows = con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) "
                           "from counters_collection group by kernel_name, counter_name order by sum(duration) desc").fetchall()
        for name, ctr, val, n, dur in rows:
            short = name.split("(")[0][:70]
            if not any(k in short for k in ("pf_elbo_qf", "pf_elbo_xw", "pfx_", "pf_fit", "pf_history", "pf_psis", "pf_elbo_mfma")):
                continue
            print(f"| `{short}` | {ctr} | {val:.4g} | {n} | {val / n:.4g} | {dur / 1e6:.3f} |")
