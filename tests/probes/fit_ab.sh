# usage (on the GPU box): bash tests/probes/fit_ab.sh <tag> ...  -- fit / history time at config 3 (64 paths, d = 1000, J = 6) of experiment builds
V=$PWD/pathfinder.jl_amd/build/variants
for lib in "$@"; do
  if [ $lib = default ]; then unset PFMI_LIB_PATH; else export PFMI_LIB_PATH=$V/libpfmi_$lib.so; fi
  python - <<'P'
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "pathfinder.jl_amd"))
import numpy as np, pfmi
eng = pfmi.Engine(0)
d = 1000
eng.set_target(pfmi.t_lowrank(d, 8, 2))
x0 = pfmi.HostRNG(5).rand(64 * d).reshape(64, d) * 4 - 2
eng.optimize_batch(x0, 6)
for K8 in (False, True):
    if K8:
        eng.optimize_batch(x0[:8], 6)
    eng.fit_batch(6); eng.sync()
    eng.profile(True)
    for _ in range(5): eng.fit_batch(6)
    eng.sync()
    tf, nf = eng.kernel_time("fit"); th, nh = eng.kernel_time("history")
    print(f"lib={os.environ.get('PFMI_LIB_PATH', 'default').split('/')[-1]} P={eng.P}: fit {tf / nf:.4f} ms  history {th / nh:.4f} ms")
    eng.profile(False)
P
done
