# usage (on the GPU box): bash tests/probes/fit_c5.sh   -- HBM traffic + SQ counters of the fit kernel at the config-5 shape (d = 10^4, J = 10)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --npaths 8 --dim 10000 --target funnel --history 10 --ndraws-elbo 2000 --ndraws 2000 --init-scale 10 --maxiters 200 --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/fc5_f -o f -- $B > $R/gpurun_out/fc5_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/fc5_w -o w -- $B > $R/gpurun_out/fc5_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $R/gpurun_out/fc5_a -o a -- $B > $R/gpurun_out/fc5_a.log 2>&1
cd $R
python - <<'P'
import sqlite3, glob
for tag in ("f", "w", "a"):
    db = glob.glob(f"gpurun_out/fc5_{tag}/*.db")[0]
    con = sqlite3.connect(db)
    for row in con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) from counters_collection "
                           "where kernel_name like '%fit%' or kernel_name like '%history%' group by kernel_name, counter_name"):
        print(tag, row[0][:60], row[1], "%.4g per launch" % (row[2] / row[3]), "launches", row[3], "avg ms %.3f" % (row[4] / 1e6))
P
