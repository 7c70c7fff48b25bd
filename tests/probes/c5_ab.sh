# usage: bash tests/probes/c5_ab.sh <tag> ...   -- config-5 shape (d = 10^4, J = 10, N = 2000, funnel, 8 paths x 200 iterations): scan time
R=$GRAFT_REPO_ROOT
for t in "$@"; do
  if [ "$t" = default ]; then L=""; else L="PFMI_LIB_PATH=$R/pathfinder.jl_amd/build/variants/libpfmi_$t.so"; fi
  env $L python $R/bench.py --npaths 8 --dim 10000 --target funnel --history 10 --ndraws-elbo 2000 --ndraws 2000 --init-scale 10 --maxiters 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); st=l['stages_ms']; print('$t', 'step', l['ms_per_step'], 'draws/s %.3e' % l['value'], 'scan', st['elbo_draws'], 'fit', st['fit'], 'pool', st['elbo_draws_x'])"
done
