#!/bin/bash
# Build knock-out variants of the render kernel (RT_EXP bitmask, rt_amd.hip) into scratch/exp/ for
# timing experiments: which phase of the area-light shading pass costs what.  The images of these
# variants are WRONG by construction; only their kernel time is read.
#   scripts/exp_variants.sh 1 2 4 7      -> scratch/exp/librt_amd_exp<N>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/exp
for v in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -ffp-contract=off -DRT_EXP="$v" \
        raytracer_amd/csrc/rt_amd.hip -o "scratch/exp/librt_amd_exp$v.so" &
done
wait
ls -la scratch/exp
