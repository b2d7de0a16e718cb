#!/bin/bash
# usage: tools/build_variant.sh SOURCE(.hip basename without extension) NAME [flags] -> variants/NAME.so
# One source file rebuilt with extra flags, linked with the other objects of the last regular build.
set -e
cd "$(dirname "$0")/.."
mkdir -p variants build/vobj
SRC=$1; N=$2; shift 2
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden -DDW_BUILD \
  "$@" -Iinclude -Idepth_estimation_amd/csrc -c depth_estimation_amd/csrc/$SRC.hip -o build/vobj/$N.o
hipcc --offload-arch=gfx950 -shared -fPIC build/vobj/$N.o $(ls build/obj/*.o | grep -v "/$SRC.o") -o variants/$N.so
