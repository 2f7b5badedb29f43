#!/bin/bash
# Builds an experimental variant of the library next to the product:
#   tools/variant.sh <name> <rt_hip source> [extra hipcc flags...]  ->  raytracing_amd/librt_amd_<name>.so
set -e
NAME=$1; SRC=$2; shift 2
D=$(dirname "$0")/../raytracing_amd
mkdir -p $D/csrc/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -I$D/../include -I$D/csrc --offload-arch=gfx950 "$@" -x hip -c "$SRC" -o $D/csrc/build/rt_hip_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/librt_amd_$NAME.so $D/csrc/build/rt_hip_$NAME.o \
  $D/csrc/build/kd_build_gpu.o $D/csrc/build/gather.o $D/csrc/build/host_math.o $D/csrc/build/scene_ctors.o $D/csrc/build/kd_build.o $D/csrc/build/render_pipe.o $D/csrc/build/mesh.o -lpthread -ldl
echo built $D/librt_amd_$NAME.so
