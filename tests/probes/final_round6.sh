# usage (GPU box): bash tests/probes/final_round6.sh <tag>   -- round 6: final_round.sh plus this round's extra lines (config 2 with its CPU baseline, the
# single-thread scheduler through the RCCL stand-in, the large-d fit kernels side by side with their HBM traffic, writer variants)
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tests/probes/final_round.sh $TAG
python bench.py --npaths 8 --dim 100 --target diag --no-pmc --cpu-seconds 8 > $O/${TAG}_c2_bench_line.json 2>> $O/${TAG}_bench.err
env PFMI_RCCL_LIB=$R/tests/rccl_standin/librccl_standin.so PFMI_COMM_ALLOW_SHARED_GPU=1 PFMI_DEBUG_HOOKS=1 timeout 600 python bench.py --gpus 8 --single-process --steps 10 --warmup 2 --no-pmc --no-cpu-baseline > $O/${TAG}_single_process_8_standin_bench_line.json 2>> $O/${TAG}_bench.err
env PFMI_DEBUG_HOOKS=1 timeout 600 python bench.py --gpus 1 --single-process --npaths 8 --steps 20 --warmup 2 --no-pmc --no-cpu-baseline > $O/${TAG}_single_process_1_bench_line.json 2>> $O/${TAG}_bench.err
timeout 600 python tests/probes/fit_tsqr_probe.py small c5 j16 > $O/${TAG}_fit_kernels.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/ts_$c; rocprofv3 --kernel-trace --pmc $c -d $O/ts_$c -o p -- python $R/tests/probes/fit_tsqr_probe.py c5 > /dev/null 2>&1; done )
python - <<PY > $O/${TAG}_fit_traffic.txt
import sqlite3, glob
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python tests/probes/fit_tsqr_probe.py c5   (1 608 fits, d = 10^4, J = 10; KiB per launch; FETCH x 2 on gfx950)")
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob("$O/ts_%s/*.db" % c)[0]
    con = sqlite3.connect(db)
    v = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith("counters_collection")][0]
    for row in con.execute(f"select kernel_name, sum(value), count(distinct dispatch_id), avg(duration) from {v} where kernel_name like '%pf_fit%' group by kernel_name"):
        per = row[1] / row[2] * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
        tot.setdefault(row[0], {})[c] = per
        print(c, row[0][:60], "%.4g KiB per launch" % (row[1] / row[2]), "launches", row[2], "avg ms %.3f" % (row[3] / 1e6))
for k, v in tot.items():
    print(k[:60], "-> %.2f MB per fit (FETCH x 2 + WRITE) / 1608 fits" % ((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) / 1608 / 1e6))
PY
cat $O/${TAG}_fit_traffic.txt
for i in 1 2; do XW_AB_C5=1 timeout 900 bash tests/probes/xw_ab.sh default xw_nofair xw_w12g1 xw_w16g1 xw_w12g2; done > $O/${TAG}_writer_variants.txt 2>&1
python tests/probes/margins_md.py $O/parity_margins.json $O/${TAG}_parity_margins.md
for f in c2_bench_line single_process_8_standin_bench_line single_process_1_bench_line; do python - <<PY
import json
try:
    l=json.loads(open("$O/${TAG}_$f.json").read().strip().split("\n")[-1])
    print("$f", l["ms_per_step"], "ms;", "e2e", l.get("multipathfinder_wall_ms_incl_device_lbfgs"), "api", l.get("multipathfinder_api_wall_ms"), {k: l[k] for k in l if "schedule" in k or "single" in k})
except Exception as e: print("$f", "ERR", e)
PY
done
