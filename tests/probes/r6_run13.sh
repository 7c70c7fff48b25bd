set -x
mkdir -p gpurun_out/r06
V=$PWD/pathfinder.jl_amd/build/variants
timeout 600 python tests/probes/fit_tsqr_probe.py small c5 j16 > gpurun_out/r06/tsqr_probe4.txt 2>&1; echo rc=$?; grep -v "vs mem" gpurun_out/r06/tsqr_probe4.txt
PFMI_LIB_PATH=$V/libpfmi_tsprof.so timeout 300 python tests/probes/fit_tsqr_probe.py c5 > gpurun_out/r06/tsqr_prof4.txt 2>&1; grep "TS_PROF" gpurun_out/r06/tsqr_prof4.txt | head -3
( timeout 900 python -m pytest tests/test_gpu_fit.py -q -m gpu -k "panel_fit or memory_resident_fit or large_d" -x ) > gpurun_out/r06/t13.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t13.log; tail -3 gpurun_out/r06/t13.log
