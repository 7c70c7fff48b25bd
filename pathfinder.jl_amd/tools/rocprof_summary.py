#!/usr/bin/env python
"""Dump the per-kernel statistics of a rocprofv3 (--kernel-trace --stats) results database as markdown.
usage: rocprof_summary.py results.db "title / command line" > profiles/<name>.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print(f"# {title}\n")
print("| kernel | calls | total (ms) | avg (ms) | % of GPU time |\n|---|---|---|---|---|")
for name, calls, tot, avg, pct in rows:
    print(f"| `{name[:110]}` | {calls} | {tot / 1e3:.3f} | {avg / 1e3:.4f} | {pct:.2f} |")

# kernels that one call site launches more than once with different grids (the ELBO scan: a main launch + a tail launch cut into
# one-batch pieces, csrc/elbo_qf_kernel.hip) -- per grid, so that the per-launch averages can be compared with bench.py's figures
try:
    per_grid = list(db.execute(
        "select name, grid_x, grid_y, workgroup_x, count(*), sum(duration), avg(duration) from kernels "
        "where name like '%pf_elbo_qf_kernel%' group by name, grid_x, grid_y order by sum(duration) desc"))
    if len(per_grid) > 1:
        print("\nELBO scan by launch grid (threads; one scan per step = ONE launch since the shared-constants cut: grid y = whole fits + pieces of the last round; the larger grids are the public-API runs of bench.py):\n")
        print("| kernel | grid x, y (threads) | calls | total (ms) | avg (ms) |\n|---|---|---|---|---|")
        tot = 0.0
        gx_main = min(r[1] for r in per_grid)        # the main launch has one workgroup column (grid x = workgroup size)
        nscan = sum(r[4] for r in per_grid if r[1] == gx_main)
        for name, gx, gy, wx, calls, sm, avg in per_grid:
            print(f"| `{name[:60]}` | {gx}, {gy} | {calls} | {sm / 1e6:.3f} | {avg / 1e6:.4f} |")
            tot += sm
        print(f"\nscans: {nscan}; time per scan: {tot / 1e6 / max(nscan, 1):.4f} ms")
except sqlite3.Error as ex:                      # older rocprofv3 schema without the per-dispatch view
    print(f"\n(per-grid breakdown unavailable: {ex})")
