#!/bin/bash
# usage: tools/kmeta.sh [flags] -- register / spill counts of the tile kernels for a photometric.hip build with the given flags
cd "$(dirname "$0")/.."
mkdir -p build/meta
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden -DDW_BUILD \
  -mllvm -amdgpu-sched-strategy=max-ilp "$@" -Iinclude -Idepth_estimation_amd/csrc --cuda-device-only -S \
  depth_estimation_amd/csrc/photometric.hip -o build/meta/photometric.s
grep -E "^\s+\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|name):" build/meta/photometric.s | awk '{printf "%s %s  ", $1, $2} /vgpr_spill/ {print ""}' | grep tile
grep -c v_readlane build/meta/photometric.s; grep -c v_writelane build/meta/photometric.s
