set -x
mkdir -p gpurun_out/r06
V=$PWD/pathfinder.jl_amd/build/variants
PFMI_LIB_PATH=$V/libpfmi_tsprof.so timeout 300 python tests/probes/fit_tsqr_probe.py c5 > gpurun_out/r06/tsqr_prof1.txt 2>&1; grep "TS_PROF\|funnel" gpurun_out/r06/tsqr_prof1.txt | head -8
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
B="python $R/tests/probes/fit_tsqr_probe.py c5"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/ts_f -o f -- $B > $R/gpurun_out/r06/ts_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/ts_w -o w -- $B > $R/gpurun_out/r06/ts_w.log 2>&1
cd $R
python - <<'P'
import sqlite3, glob
for tag in ("f", "w"):
    db = glob.glob(f"gpurun_out/ts_{tag}/*.db")[0]
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")]
    for row in con.execute(f"select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) from {v[0]} "
                           "where kernel_name like '%fit%' group by kernel_name, counter_name"):
        print(tag, row[0][:70], row[1], "%.4g per launch" % (row[2] / row[3]), "launches", row[3], "avg ms %.3f" % (row[4] / 1e6))
P
