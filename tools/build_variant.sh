#!/bin/bash
# Developer tool: builds a variant of the library with extra compiler flags (kernel experiments).
# -DNLPS_DEV=1: the only build that reads NLPS_* environment switches and accepts the NLPS_ABL_* / NLPS_PHASE_TIMING macros.
#   tools/build_variant.sh NAME [-DNLPS_...=v ...]   ->  build/exp/lib_NAME.so   (run with NLPS_GPU_LIB=... tools/kbench.py)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/exp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -munsafe-fp-atomics \
  -fvisibility=hidden -fvisibility-inlines-hidden -DNLPS_DEV=1 "$@" -o build/exp/lib_$name.so \
  nl-partsol_amd/csrc/nlps_gpu.hip nl-partsol_amd/csrc/nlps_io.cpp
echo build/exp/lib_$name.so
