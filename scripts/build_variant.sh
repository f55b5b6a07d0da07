#!/bin/bash
# build_variant.sh <name> [-DFLAG ...]  ->  build/variants/libort_<name>.so  (A/B builds of the engine, same flags as build.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
n=$1; shift
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -fno-slp-vectorize "$@" \
  -o build/variants/libort_$n.so opticalraytracing.jl_amd/csrc/ort_hip.hip
