set -x
mkdir -p gpurun_out/r06
V=$PWD/pathfinder.jl_amd/build/variants
for i in 1 2; do
timeout 600 python tests/probes/fit_tsqr_probe.py c5 j16 2>&1 | grep "funnel\|diag"
PFMI_LIB_PATH=$V/libpfmi_tw1.so timeout 600 python tests/probes/fit_tsqr_probe.py c5 j16 2>&1 | grep "funnel\|diag"
done
PFMI_LIB_PATH=$V/libpfmi_tsprof.so timeout 300 python tests/probes/fit_tsqr_probe.py c5 2>&1 | grep "TS_PROF" | head -2
PFMI_LIB_PATH=$V/libpfmi_tsprof1.so timeout 300 python tests/probes/fit_tsqr_probe.py c5 2>&1 | grep "TS_PROF" | head -2
timeout 600 python tests/probes/fit_tsqr_probe.py small 2>&1 | grep -v "vs mem"
