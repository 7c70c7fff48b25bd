set -x
mkdir -p gpurun_out/r06
( timeout 900 python -m pytest tests/test_gpu_elbo.py tests/test_gpu_configs.py -q -m gpu -x -k "callback or closure or config3_exact" ) > gpurun_out/r06/t16.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t16.log; tail -5 gpurun_out/r06/t16.log
for i in 1 2; do XW_AB_C5=1 timeout 600 bash tests/probes/xw_ab.sh default; PFMI_DEBUG_HOOKS=1 PFMI_DEVCB_OVERLAP=0 XW_AB_C5=1 timeout 600 bash tests/probes/xw_ab.sh default; done
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > gpurun_out/r06/bench16.json 2> gpurun_out/r06/bench16.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r06/bench16.json") if x.startswith("{")][-1])
d=l["device_callback_target"]; print({k:d[k] for k in d if k not in ("note","sample","measured_hbm_bytes_per_launch","writer_issue_floor")}); print(l["self_checks"], l["ms_per_step"])
PY
