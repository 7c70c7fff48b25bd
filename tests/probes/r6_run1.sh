# round 6, first session: new tests, writer variants (bit identity + A/B), bench self-check
set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_hinit.py tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r06/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r06/t1.log
tail -15 gpurun_out/r06/t1.log
V=$PWD/pathfinder.jl_amd/build/variants
for t in base v2 v2w16g1 v2w12g1; do PFMI_LIB_PATH=$V/libpfmi_$t.so timeout 300 python tests/probes/xw_bits.py > gpurun_out/r06/xw_bits_$t.txt 2>&1; done
md5sum gpurun_out/r06/xw_bits_*.txt
XW_AB_C5=1 timeout 900 bash tests/probes/xw_ab.sh base v2 v2w16g1 v2w12g1 v2w12g2 v2nofair v2w16g1nofair > gpurun_out/r06/xw_ab.txt 2>&1
cat gpurun_out/r06/xw_ab.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/r06/bench1.json 2> gpurun_out/r06/bench1.err; echo "bench rc=$?"
tail -c 2500 gpurun_out/r06/bench1.json
