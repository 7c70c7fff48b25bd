# usage: bash tests/probes/scan_ab.sh <tag> [<tag> ...]   -- config-3 step / scan time of experiment builds (build/variants/libpfmi_<tag>.so;
# "default" = the committed library), two runs each (run on the GPU box)
R=$GRAFT_REPO_ROOT
for t in "$@"; do
  for rep in 1 2; do
    if [ "$t" = default ]; then L=""; else L="PFMI_LIB_PATH=$R/pathfinder.jl_amd/build/variants/libpfmi_$t.so"; fi
    env $L python $R/bench.py --no-cpu-baseline --no-pmc --steps 10 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$t', 'step', l['ms_per_step'], 'scan', l['stages_ms']['elbo_draws']['ms'], 'fit', l['stages_ms']['fit']['ms'], 'api', l['multipathfinder_api_wall_ms'], 'e2e', l['multipathfinder_wall_ms_incl_device_lbfgs'])"
  done
done
