set -x
mkdir -p gpurun_out/r06
timeout 600 python tests/probes/fit_tsqr_probe.py j16 > gpurun_out/r06/tsqr_probe_j16.txt 2>&1; echo rc=$?; cat gpurun_out/r06/tsqr_probe_j16.txt
