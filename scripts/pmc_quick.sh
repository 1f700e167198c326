#!/bin/bash
# Two SQ counter passes (instruction mix, wave-cycle breakdown) over bench.py; honours RT_AMD_LIB.
# Usage: scripts/pmc_quick.sh <tag> [bench args...]  ->  gpurun_out/prof_<tag>/
set -u
tag=${1:?tag}
shift
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=(bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@")
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d "$out/pmcA" -o pmcA --output-format csv -- python3 "${args[@]}" > "$out/pmcA.log" 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -d "$out/pmcB" -o pmcB --output-format csv -- python3 "${args[@]}" > "$out/pmcB.log" 2>&1 || exit 1
python3 scripts/pmc_summary.py "$out" | grep -A40 "render_kernel<false>" | grep -v "^#.*true" | head -34
