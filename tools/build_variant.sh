#!/bin/bash
# Build a variant of libsfm_hip.so with extra compiler flags (e.g. -DSFM_ABLATE_FOLD) into
# tools/ab/<name>.so, for A/B timing with SFM_HIP_LIBRARY=<path> python bench.py ...
# usage: tools/build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/ab
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=${PRELOAD:-2} "$@" \
  -shared -o tools/ab/$name.so sfm_amd/csrc/sfm_hip_capi.hip sfm_amd/csrc/sfm_hip_cascade.hip sfm_amd/csrc/sfm_hip_geometry.hip \
  sfm_amd/csrc/sfm_hip_replicate.hip sfm_amd/csrc/sfm_hip_partition.cc -ldl
echo tools/ab/$name.so
