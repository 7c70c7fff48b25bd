set -x
mkdir -p gpurun_out/r06
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/r06/t10.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t10.log
tail -12 gpurun_out/r06/t10.log
TAG=r06a bash tests/probes/c5_share_profile.sh > gpurun_out/r06/c5_share_a.log 2>&1; tail -14 gpurun_out/r06/c5_share_a.log
python - <<'PY'
import json
for f in ("gpurun_out/r06a_c5_share_bench_line.json",):
    l=json.loads(open(f).read().strip().split("\n")[-1])
    print(f, l["ms_per_step"], {k: v["ms"] for k, v in l["stages_ms"].items()})
PY
bash tests/probes/fit_c5.sh > gpurun_out/r06/fit_c5_pmc.txt 2>&1; grep "fit" gpurun_out/r06/fit_c5_pmc.txt | head -20
