# usage (on the GPU box): bash tests/probes/profile_round.sh <tag>
# kernel-trace stats, HBM traffic (two separate --pmc passes) and SQ counters of `bench.py`; results under gpurun_out/<tag>_*
TAG=$1; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc"
B1="python $R/bench.py --steps 1 --warmup 0 --minimal --with-devcb --no-cpu-baseline --no-pmc"
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o s -- $B > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o f -- $B1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o w -- $B1 > /dev/null 2>&1
cd $R
python pathfinder.jl_amd/tools/rocprof_summary.py gpurun_out/prof_stats/s_results.db "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc (1x MI355X)" > gpurun_out/${TAG}_bench_kernel_stats.md
python pathfinder.jl_amd/tools/pmc_traffic.py gpurun_out/prof_fetch/f_results.db gpurun_out/prof_write/w_results.db "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 1 --warmup 0 --minimal --with-devcb --no-cpu-baseline --no-pmc (two separate passes, 1x MI355X, $TAG)" > gpurun_out/${TAG}_pmc_traffic.json
bash tests/probes/pmc_cmd.sh ${TAG}_sq 'qf\|xw\|pfx\|fit_reg' $B1 > /dev/null 2>&1
head -16 gpurun_out/${TAG}_bench_kernel_stats.md; cat gpurun_out/${TAG}_pmc_traffic.json
