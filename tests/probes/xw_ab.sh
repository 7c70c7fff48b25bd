# usage (on the GPU box): bash tests/probes/xw_ab.sh <tag> [<tag> ...]   -- writer time of experiment builds (variant.sh) at config 3's and
# config 5's shape; "default" = the shipped library
V=$PWD/pathfinder.jl_amd/build/variants
for lib in "$@"; do
  if [ $lib = default ]; then unset PFMI_LIB_PATH; else export PFMI_LIB_PATH=$V/libpfmi_$lib.so; fi
  python tests/probes/devcb_probe.py 8 1000 6 1000 lowrank 2>&1 | grep lib= | sed "s|$V/||"
  [ -n "$XW_AB_C5" ] && python tests/probes/devcb_probe.py 4 10000 10 2000 funnel 100 2>&1 | grep lib= | sed "s|$V/||"
done
