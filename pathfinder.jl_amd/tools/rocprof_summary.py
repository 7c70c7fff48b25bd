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
