#!/bin/bash
# A/B kernel builds: recompiles ONE source of madm_amd/csrc with extra flags and links it with the shipped objects into
# build/libmadm_hip_<name>.so (git-ignored, travels with gpurun; select with MADM_HIP_LIB=build/libmadm_hip_<name>.so).
# usage: tools/build_variant.sh <name> <file.hip> [extra hipcc flags...]     e.g.  tools/build_variant.sh h16stamps conv3x3_h16.hip -DH16_STAMPS
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/madm_amd/csrc
mkdir -p $R/build
make -s -C $C
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
[ "$src" = conv3x3_h16.hip ] && NOPK=""
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$C -Wall -Wno-unused-function $NOPK "$@" -c $C/$src -o $R/build/${src%.hip}_$name.o 2>&1 | grep -v "is not a recognized feature" || true
objs=$(for f in $C/*.o; do [ "$(basename $f)" = "${src%.hip}.o" ] && echo $R/build/${src%.hip}_$name.o || echo $f; done)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o $R/build/libmadm_hip_$name.so
ls -la $R/build/libmadm_hip_$name.so
