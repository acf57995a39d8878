#!/bin/bash
# Build an A/B variant of the kernel library: scripts/build_variant.sh NAME [extra hipcc flags for gemm.hip ...]
# -> crog_amd/libcrog_hip_NAME.so (select with CROG_LIB=crog_amd/libcrog_hip_NAME.so).  Other objects come from the main build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p crog_amd/csrc/build_$name
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc "$@" -c crog_amd/csrc/gemm.hip -o crog_amd/csrc/build_$name/gemm.o
objs=""
for f in api norm eltwise head conv_aux attn ssg preprocess; do objs="$objs crog_amd/csrc/build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o crog_amd/libcrog_hip_$name.so crog_amd/csrc/build_$name/gemm.o $objs
echo built crog_amd/libcrog_hip_$name.so
