# round 6, session 2: full GPU suite on the rebuilt library (comm rewrite, hinit, writer v2)
set -x
mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_hinit.py tests/test_gpu_stream.py tests/test_gpu_callback.py -x -q -m gpu ) > gpurun_out/r06/t2a.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t2a.log
tail -30 gpurun_out/r06/t2a.log
( time timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multirank.py --deselect tests/test_gpu_hinit.py --deselect tests/test_gpu_stream.py --deselect tests/test_gpu_callback.py ) > gpurun_out/r06/t2b.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t2b.log
tail -40 gpurun_out/r06/t2b.log
XW_AB_C5=1 timeout 300 bash tests/probes/xw_ab.sh default > gpurun_out/r06/xw_ab2.txt 2>&1; cat gpurun_out/r06/xw_ab2.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-pmc --cpu-seconds 6 > gpurun_out/r06/bench2.json 2> gpurun_out/r06/bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r06/bench2.json") if x.startswith("{")][-1])
print(json.dumps(l["callback_target"], indent=1)); print(l["self_checks"], l["value"], l["ms_per_step"])
PY
