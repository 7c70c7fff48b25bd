#!/usr/bin/env python
"""Summarise HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: pmc_traffic.py fetch.db write.db "<command>" > profiles/pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so it is doubled."""
import json
import sqlite3
import sys


def per_launch(db, counter, pattern):
    con = sqlite3.connect(db)
    rows = list(con.execute(
        "select grid_size, sum(value), count(distinct dispatch_id), avg(duration) from counters_collection "
        "where kernel_name like ? and counter_name = ? group by grid_size order by grid_size desc", (pattern, counter)))
    return rows


pat = "%pf_elbo_qf_kernel%"
f = per_launch(sys.argv[1], "FETCH_SIZE", pat)
w = per_launch(sys.argv[2], "WRITE_SIZE", pat)
out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "kernel": "pf_elbo_qf_kernel (single-pass ELBO scan, largest grid)",
       "fetch_correction": 2.0}
g, v, n, dur = f[0]
out["fetch_bytes_per_launch"] = 2.0 * v / n * 1024
out["launches_profiled"] = n
out["avg_duration_ms_under_pmc"] = dur / 1e6
g, v, n, dur = w[0]
out["write_bytes_per_launch"] = v / n * 1024
out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
print(json.dumps(out, indent=1))
