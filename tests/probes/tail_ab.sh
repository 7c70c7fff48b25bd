# usage (GPU box): bash tests/probes/tail_ab.sh [bench.py flags]   -- scan time at 8 paths per GPU under the three cuts of the last fits:
# one launch with shared constants (default), the round-3 two launches, one workgroup per fit
R=$GRAFT_REPO_ROOT
for mode in default PFMI_QF_TWO_LAUNCHES PFMI_QF_NO_TAIL; do
  for rep in 1 2; do
    if [ "$mode" = default ]; then E=""; else E="PFMI_DEBUG_HOOKS=1 $mode=1"; fi
    env $E python $R/bench.py --npaths 8 --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$mode', 'fits', l['config']['fits_total'], 'step', l['ms_per_step'], 'scan', l['stages_ms']['elbo_draws']['ms'])"
  done
done
