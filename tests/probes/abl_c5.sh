A="--npaths 8 --dim 10000 --target funnel --history 10 --ndraws-elbo 2000 --ndraws 2000 --init-scale 10 --maxiters 200 --steps 3 --warmup 1 --no-cpu-baseline"
for L in "" $(ls pathfinder.jl_amd/build/abl/libpfmi_abl?.so); do
  if [ -n "$L" ]; then export PFMI_LIB_PATH=$PWD/$L; fi
  python bench.py $A 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$L', l['ms_per_step'], l['stages_ms']['elbo_draws'])"
  python bench.py --history 8 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$L J=8', l['ms_per_step'], l['stages_ms']['elbo_draws'], l['config']['fits_total'])"
done
