#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats, then PMC counters, one pass
# per run (never combined with a trace domain).  Usage: scripts/pmc.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/{stats,pmcA,pmcB,pmcC,pmcD}/ ; copy the summaries into profiles/.
set -u
tag=${1:?tag}
shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
args=(bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@")

run() {  # name, rocprof options...
    local name=$1
    shift
    timeout -s KILL 150 rocprofv3 "$@" -d "$out/$name" -o "$name" --output-format csv -- python3 "${args[@]}" \
        > "$out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}

run stats --kernel-trace --stats &&
run pmcA --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
run pmcB --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run pmcC --pmc FETCH_SIZE GRBM_GUI_ACTIVE &&
run pmcD --pmc WRITE_SIZE GRBM_GUI_ACTIVE
rc=$?
find "$out" -name '*.csv' | sort
exit $rc
