#!/bin/bash
# usage: tools/build_variant.sh NAME [extra hipcc flags / -D...]  -> variants/NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
N=$1; shift
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fvisibility=hidden -DDW_BUILD "$@" \
  -Iinclude -Idepth_estimation_amd/csrc depth_estimation_amd/csrc/*.hip -o variants/$N.so
