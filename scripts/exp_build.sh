#!/bin/bash
# Build an experimental variant of the library into scratch/exp/ (timing experiments; loaded with RT_AMD_LIB=...).
#   scripts/exp_build.sh NAME [-DFLAG=... ...]      -> scratch/exp/librt_amd_NAME.so
# -DRT_EXP_ONLY_TREE compiles only the plain hierarchy kernels (C4, cover_image): seconds instead of minutes.
set -e
cd "$(dirname "$0")/.."
name=$1
shift
make -s -j8 kernels LIBDIR="scratch/exp/$name" OBJDIR="scratch/exp/$name/obj" EXTRA_HIPFLAGS="$*"
cp "scratch/exp/$name/librt_amd.so" "scratch/exp/librt_amd_$name.so"
rm -rf "scratch/exp/$name"
