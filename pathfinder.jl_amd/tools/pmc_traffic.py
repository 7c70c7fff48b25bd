#!/usr/bin/env python
"""Summarise HBM traffic of the hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: pmc_traffic.py fetch.db write.db "<command>" > profiles/<tag>_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so it is doubled.  Per kernel family the launch shape that moves the most
bytes is reported (the main launch of the scan; one block of fits for the device-closure writer / reader)."""
import json
import sqlite3
import sys


def per_launch(db, counter, pattern):
    con = sqlite3.connect(db)
    return list(con.execute(
        "select grid_size, sum(value), count(distinct dispatch_id), avg(duration) from counters_collection "
        "where kernel_name like ? and counter_name = ? group by grid_size order by sum(value) desc", (pattern, counter)))


out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "fetch_correction": 2.0, "kernels": {}}
for label, pat in (("pf_elbo_qf_kernel (single-pass ELBO scan)", "%pf_elbo_qf_kernel%"),
                   ("pf_elbo_xw_kernel (draw writer: device-closure scans, pools)", "%pf_elbo_xw_kernel%"),
                   ("pfx_gauss_kernel (examples/device_logp: the user's closure reading the draws)", "%pfx_%"),
                   ("pf_fit_reg_kernel", "%pf_fit_reg_kernel%")):
    f, w = per_launch(sys.argv[1], "FETCH_SIZE", pat), per_launch(sys.argv[2], "WRITE_SIZE", pat)
    if not f or not w:
        continue
    g, v, n, dur = f[0]
    k = {"fetch_bytes_per_launch": 2.0 * v / n * 1024, "launches_profiled": n, "avg_duration_ms_under_pmc": dur / 1e6}
    g, v, n, dur = w[0]
    k["write_bytes_per_launch"] = v / n * 1024
    k["traffic_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
    out["kernels"][label] = k
scan = out["kernels"].get("pf_elbo_qf_kernel (single-pass ELBO scan)")
if scan:                                          # top-level keys of round 2's file, kept for readers of profiles/pmc_traffic.json
    out.update(kernel="pf_elbo_qf_kernel (single-pass ELBO scan, largest grid)", **scan)
print(json.dumps(out, indent=1))
