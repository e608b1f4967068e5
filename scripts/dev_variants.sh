#!/bin/bash
# Developer helper: builds variants of libwr_hip.so with extra compiler flags into build_dev/.
# Usage: bash scripts/dev_variants.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/../wgpu-renderer_amd/csrc" || exit 1
mkdir -p ../../build_dev
while [ $# -ge 2 ]; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function $2 -shared -o ../../build_dev/libwr_$1.so wr_api.hip wr_gltf.cpp wr_jpeg.cpp -lz -ldl 2>&1 | grep -E "error" ; echo built $1 ) &
  shift; shift
done
wait
