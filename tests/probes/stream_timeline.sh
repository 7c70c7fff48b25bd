# usage (GPU box): bash tests/probes/stream_timeline.sh [K] [d]   -- every dispatch and copy of the LAST streamed step of tests/probes/stream_probe.py
# (rocprofv3 --kernel-trace --memory-copy-trace), with queue ids: which segments overlap, where the CUs idle
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stl
STREAM_ONLY=1 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/stl -o q -- python $R/tests/probes/stream_probe.py ${1:-8} ${2:-1000} 2 > /tmp/stl.log 2>&1
tail -3 /tmp/stl.log | cut -c1-300
python - <<'PY'
import sqlite3, glob
con = sqlite3.connect(glob.glob('/tmp/stl/*.db')[0])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print(cols)
qc = 'queue_id' if 'queue_id' in cols else ('queue' if 'queue' in cols else None)
sc = 'stream_id' if 'stream_id' in cols else ('stream' if 'stream' in cols else None)
sel = "select start, end, name" + (", " + qc if qc else ", 0") + (", " + sc if sc else ", 0") + (", grid_x, grid_y, workgroup_x" if 'grid_x' in cols else ", 0, 0, 1") + " from kernels"
ev = [(s, e, n[:44], q, st, gx * max(gy, 1) // max(wx, 1)) for s, e, n, q, st, gx, gy, wx in con.execute(sel)]
try:
    ev += [(s, e, 'COPY %s %d B' % (n, b), -1, -1, 0) for s, e, n, b in con.execute("select start, end, name, size from memory_copies")]
except Exception as x:
    print('no memory_copies view:', x)
ev.sort()
ls = [i for i, v in enumerate(ev) if 'pf_lbfgs' in v[2]]
i0 = ls[-1]
t0 = ev[i0][0]
for s, e, n, q, st, wg in ev[i0:]:
    print('%9.1f us .. %9.1f us (%8.1f us)  q%s s%s  %6d wg  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, wg, n))
PY
