# usage: bash tests/probes/variant.sh <tag> <source stem> [-DNAME=VALUE ...]
# builds pathfinder.jl_amd/build/variants/libpfmi_<tag>.so = the committed objects with csrc/<stem>.hip recompiled under the given
# defines (experiment builds, selected at run time with PFMI_LIB_PATH; never the shipped library)
set -e
TAG=$1; STEM=$2; shift 2
R=$(cd "$(dirname "$0")/../.." && pwd)/pathfinder.jl_amd
mkdir -p $R/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result "$@" -c $R/csrc/$STEM.hip -o $R/build/variants/${STEM}_$TAG.o
OBJS=$(ls $R/build/*.o | grep -v "/$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libpfmi_$TAG.so $OBJS $R/build/variants/${STEM}_$TAG.o -ldl
echo built $R/build/variants/libpfmi_$TAG.so
