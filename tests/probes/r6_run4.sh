set -x
mkdir -p gpurun_out/r06
cp pathfinder.jl_amd/lib/libpfmi.so pathfinder.jl_amd/build/variants/libpfmi_cur.so
for i in 1 2 3; do XW_AB_C5=1 timeout 600 bash tests/probes/xw_ab.sh cur fair1 fair2; done > gpurun_out/r06/xw_ab4.txt 2>&1; cat gpurun_out/r06/xw_ab4.txt
env PFMI_RCCL_LIB=$PWD/tests/rccl_standin/librccl_standin.so PFMI_COMM_ALLOW_SHARED_GPU=1 PFMI_DEBUG_HOOKS=1 timeout 600 python bench.py --gpus 8 --single-process --steps 10 --warmup 2 > gpurun_out/r06/bench_sp8.json 2> gpurun_out/r06/bench_sp8.err; echo rc=$?; tail -c 1800 gpurun_out/r06/bench_sp8.json
env PFMI_DEBUG_HOOKS=1 timeout 600 python bench.py --gpus 1 --single-process --npaths 8 --steps 20 --warmup 2 > gpurun_out/r06/bench_sp1.json 2> gpurun_out/r06/bench_sp1.err; echo rc=$?; tail -c 1500 gpurun_out/r06/bench_sp1.json
