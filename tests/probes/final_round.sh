# usage (GPU box): bash tests/probes/final_round.sh <tag>   -- the evidence of a round's final build in one session: the GPU test suite, smoke, the
# default bench line (with counters and CPU baseline), the kernel-trace / PMC summaries and the other configs' bench lines -> gpurun_out/<tag>_*
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
(timeout 2400 python -m pytest tests -q -x -m gpu > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log)
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.log 2>&1; echo "rc=$?" >> $O/${TAG}_smoke.log)
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
bash tests/probes/profile_round.sh $TAG > $O/${TAG}_profile_round.log 2>&1
python bench.py --npaths 8 --no-cpu-baseline --no-pmc > $O/${TAG}_c4_share_bench_line.json 2>> $O/${TAG}_bench.err
python bench.py --npaths 8 --dim 100 --target diag --no-cpu-baseline --no-pmc > $O/${TAG}_c2_bench_line.json 2>> $O/${TAG}_bench.err
python bench.py --npaths 8 --dim 10000 --target funnel --history 10 --ndraws-elbo 2000 --ndraws 2000 --init-scale 10 --maxiters 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/${TAG}_c5_shape_bench_line.json 2>> $O/${TAG}_bench.err
TAG=$TAG bash tests/probes/c5_share_profile.sh > $O/${TAG}_c5_share.log 2>&1
grep -n "passed\|failed" $O/${TAG}_pytest.log | tail -2; tail -2 $O/${TAG}_smoke.log
for f in bench_line c4_share_bench_line c2_bench_line c5_shape_bench_line c5_share_bench_line; do python - <<PY
import json
try:
    l=json.loads(open("$O/${TAG}_$f.json").read().strip().split("\n")[-1])
    print("$f", l["ms_per_step"], "ms;", {k: v["ms"] for k, v in l["stages_ms"].items()}, "e2e", l.get("multipathfinder_wall_ms_incl_device_lbfgs"), "api", l.get("multipathfinder_api_wall_ms"), "roofline", (l.get("roofline") or {}).get("frac"))
except Exception as e: print("$f", "ERR", e)
PY
done
