#!/bin/bash
# Build an experimental variant of the library into scratch/exp/ (timing experiments; loaded with RT_AMD_LIB=...).
#   scripts/exp_build.sh NAME [-DFLAG=... ...]      -> scratch/exp/librt_amd_NAME.so
# -DRT_EXP_ONLY_TREE compiles only the plain hierarchy kernels (C4, cover_image): 20 s instead of minutes.
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/exp
name=$1
shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -ffp-contract=off "$@" \
    raytracer_amd/csrc/rt_amd.hip -o "scratch/exp/librt_amd_$name.so"
