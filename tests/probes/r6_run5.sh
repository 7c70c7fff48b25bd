# round 6, session after the container was re-created: full GPU suite, smoke, default bench line, writer probe
set -x
mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests -q -m gpu ) > gpurun_out/r06/t5.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t5.log
tail -30 gpurun_out/r06/t5.log
(python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke5.log 2>&1; echo "rc=$?" >> gpurun_out/r06/smoke5.log); tail -3 gpurun_out/r06/smoke5.log
timeout 900 python bench.py > gpurun_out/r06/bench5.json 2> gpurun_out/r06/bench5.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/r06/bench5.json
XW_AB_C5=1 timeout 300 bash tests/probes/xw_ab.sh default > gpurun_out/r06/xw_ab5.txt 2>&1; cat gpurun_out/r06/xw_ab5.txt
