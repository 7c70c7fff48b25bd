set -x
mkdir -p gpurun_out/r06
( time timeout 1800 python -m pytest tests -q -m gpu -x ) > gpurun_out/r06/t11.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t11.log
tail -8 gpurun_out/r06/t11.log
for i in 1 2; do XW_AB_C5=1 timeout 900 bash tests/probes/xw_ab.sh default xw_nofair xw_w12g1 xw_w16g1 xw_w16g1nofair xw_w12g2; done > gpurun_out/r06/xw_ab11.txt 2>&1; cat gpurun_out/r06/xw_ab11.txt
