# usage (GPU box): bash tests/probes/pmc_icache.sh   -- instruction-cache counters of the kernels of one bench step
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ic_a /tmp/ic_b
B="python $R/bench.py --steps 1 --warmup 0 --minimal --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d /tmp/ic_a -o a -- $B > /tmp/ic_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/ic_b -o b -- $B > /tmp/ic_b.log 2>&1
tail -3 /tmp/ic_a.log
python - <<'PY'
import sqlite3, glob
for pat in ('/tmp/ic_a/*.db', '/tmp/ic_b/*.db'):
    for path in glob.glob(pat):
        con = sqlite3.connect(path)
        rows = con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name order by sum(duration) desc").fetchall()
        for name, ctr, val, n, dur in rows:
            short = name.split('(')[0][:44]
            if any(k in short for k in ('pf_elbo_qf', 'pf_fit_reg', 'pf_lbfgs', 'pf_history', 'pf_psis_kernel', 'pf_elbo_mfma')):
                print(f"{short:46s} {ctr:28s} {val / n:14.4g} per launch  ({n} launches, {dur / 1e6:.3f} ms)")
PY
