set -x
mkdir -p gpurun_out/r06
( time timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_hinit.py tests/test_gpu_stream.py tests/test_gpu_callback.py -q -m gpu ) > gpurun_out/r06/t3.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t3.log
tail -60 gpurun_out/r06/t3.log
cp pathfinder.jl_amd/lib/libpfmi.so pathfinder.jl_amd/build/variants/libpfmi_cur.so
for i in 1 2; do XW_AB_C5=1 timeout 600 bash tests/probes/xw_ab.sh base v2 cur; done > gpurun_out/r06/xw_ab3.txt 2>&1; cat gpurun_out/r06/xw_ab3.txt
