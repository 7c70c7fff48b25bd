R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --npaths 32 --dim 10000 --target funnel --history 10 --ndraws-elbo 2000 --ndraws 2000 --init-scale 10 --maxiters 1000 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline"
$B > $R/gpurun_out/${TAG:-r05}_c5_share_bench_line.json 2> $R/gpurun_out/${TAG:-r05}_c5.err
rm -rf $R/gpurun_out/prof_c5
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c5 -o s -- $B > $R/gpurun_out/${TAG:-r05}_c5_share_bench_line_under_rocprof.json 2> $R/gpurun_out/${TAG:-r05}_c5_rocprof.err
cd $R
python pathfinder.jl_amd/tools/rocprof_summary.py gpurun_out/prof_c5/s_results.db "${TAG:-r05}: rocprofv3 --kernel-trace --stats -- python bench.py --npaths 32 --dim 10000 --target funnel --history 10 --ndraws-elbo 2000 --ndraws 2000 --init-scale 10 --maxiters 1000 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline (one GPU share of config 5, 1x MI355X, final build)" > gpurun_out/${TAG:-r05}_c5_share_kernel_stats.md
rm -rf gpurun_out/prof_c5
head -12 gpurun_out/${TAG:-r05}_c5_share_kernel_stats.md
