#!/bin/bash
# usage: tools/build_photo_variant.sh NAME [flags for photometric.hip only] -> variants/NAME.so
# (the other objects come from build/obj of the last regular build)
set -e
cd "$(dirname "$0")/.."
mkdir -p variants build/vobj
N=$1; shift
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden -DDW_BUILD \
  -mllvm -amdgpu-sched-strategy=max-ilp "$@" -Iinclude -Idepth_estimation_amd/csrc \
  -c depth_estimation_amd/csrc/photometric.hip -o build/vobj/$N.o -save-temps=obj 2>/dev/null || \
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden -DDW_BUILD \
  -mllvm -amdgpu-sched-strategy=max-ilp "$@" -Iinclude -Idepth_estimation_amd/csrc \
  -c depth_estimation_amd/csrc/photometric.hip -o build/vobj/$N.o
others=$(ls build/obj/*.o | grep -v photometric.o)
hipcc --offload-arch=gfx950 -shared -fPIC build/vobj/$N.o $others -o variants/$N.so
