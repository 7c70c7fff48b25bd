# usage (GPU box): bash tests/probes/qf_durations.sh <lib tag|default> ...   -- per-dispatch durations of the scan kernel under rocprofv3
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  if [ "$t" = default ]; then L=""; else L="PFMI_LIB_PATH=$R/pathfinder.jl_amd/build/variants/libpfmi_$t.so"; fi
  rm -rf /tmp/qfd_$t
  env $L rocprofv3 --kernel-trace -d /tmp/qfd_$t -o q -- python $R/bench.py --steps 4 --warmup 1 --minimal --no-cpu-baseline --no-pmc > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
con=sqlite3.connect(glob.glob('/tmp/qfd_$t/*.db')[0])
rows=con.execute("select grid_x/workgroup_x, grid_y, duration/1e6 from kernels where name like '%pf_elbo_qf%' order by start").fetchall()
print('$t', [(r[0], r[1], round(r[2],3)) for r in rows[2:10]])
rows=con.execute("select name, avg(duration)/1e6, count(*) from kernels group by name order by sum(duration) desc").fetchall()
print('   ', [(r[0][5:22], round(r[1],3), r[2]) for r in rows[:6]])
PY
done
