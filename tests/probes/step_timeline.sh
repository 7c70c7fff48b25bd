# usage (GPU box): bash tests/probes/step_timeline.sh [bench.py flags ...]   -- every dispatch and copy of the LAST timed step, with the idle
# time before it (rocprofv3 --kernel-trace --memory-copy-trace; bench.py --minimal)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stl
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/stl -o q -- python $R/bench.py --steps 3 --warmup 1 --minimal --no-cpu-baseline --no-pmc "$@" > /tmp/stl.log 2>&1
tail -1 /tmp/stl.log | cut -c1-300
python - <<'PY'
import sqlite3, glob
con = sqlite3.connect(glob.glob('/tmp/stl/*.db')[0])
ev = [(s, e, n[:60]) for s, e, n in con.execute("select start, end, name from kernels")]
try:
    ev += [(s, e, 'COPY %s %d B' % (n, b)) for s, e, n, b in con.execute("select start, end, name, size from memory_copies")]
except Exception as x:
    print('no memory_copies view:', x, [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")][:40])
ev.sort()
# the last step starts at the last history-walk dispatch
hs = [i for i, v in enumerate(ev) if 'pf_history' in v[2]]
i0 = hs[-1]
t0, prev = ev[i0][0], ev[i0][0]
busy = 0
for s, e, n in ev[i0:]:
    print('%9.1f us  +%7.1f idle  %8.1f us  %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    busy += e - s
    prev = max(prev, e)
print('span %.1f us, busy %.1f us, idle %.1f us, %d events' % ((prev - t0) / 1e3, busy / 1e3, (prev - t0 - busy) / 1e3, len(ev) - i0))
PY
