# round 6: first run of the TSQR fit kernel
set -x
mkdir -p gpurun_out/r06
timeout 600 python tests/probes/fit_tsqr_probe.py small c5 > gpurun_out/r06/tsqr_probe1.txt 2>&1; echo rc=$?; cat gpurun_out/r06/tsqr_probe1.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "panel_fit or memory_resident_fit or large_d" -x ) > gpurun_out/r06/t6.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t6.log; tail -15 gpurun_out/r06/t6.log
