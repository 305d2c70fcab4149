#!/bin/bash
# builds lib/libagcn_hip_<tag>.so with extra -D flags: tools/scratch/build_variant.sh nt1 -DAGCN_NT=1
set -e
cd "$(dirname "$0")/../.."
PKG=adaptive-graph-convolutional-network_amd
tag=$1; shift
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -Wno-unused-function -Wno-pass-failed "$@" \
  -o $PKG/lib/libagcn_hip_$tag.so $PKG/csrc/*.hip $PKG/csrc/*.cpp
