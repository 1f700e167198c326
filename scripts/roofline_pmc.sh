#!/bin/bash
# Hardware counters of the render kernel for the roofline of bench.py: rocprofv3 --pmc passes over
# scripts/render_loop.py (one workload, the f64-chain kernel and the FP32-only kernel in one run), each
# pass a run of its own and never combined with a trace domain (MI355X_MICROARCH.md).
#   scripts/roofline_pmc.sh <tag> <workload> [render_loop args...]
# Output: gpurun_out/pmc_<tag>/<workload>/pass*/ + gpurun_out/pmc_<tag>/<workload>.json (per-launch means over the
# launches after each kernel's first — cold caches, and for scenes with glass the raster tile order — per
# kernel); scripts/roofline_pmc_merge.py folds the json files into profiles/pmc.json, which bench.py reads.
set -u
tag=${1:?tag}
wl=${2:?workload}
shift 2
cd "$(dirname "$0")/.." || exit 1
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag/$wl
mkdir -p "$out"
pass() {  # name counters...
    local name=$1
    shift
    (cd "$out" && timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$name" -o "$name" --output-format csv -- python3 "$root/scripts/render_loop.py" "$wl" --reps 4 "${EXTRA[@]}" > "$name.log" 2>&1)
    local rc=$?
    echo "$wl $name rc=$rc"
    return $rc
}
EXTRA=("$@")
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU &&
pass p2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS &&
pass p3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR &&
pass p4 FETCH_SIZE GRBM_GUI_ACTIVE &&
pass p5 WRITE_SIZE GRBM_GUI_ACTIVE
rc=$?
python3 - "$out" "$wl" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, wl = sys.argv[1], sys.argv[2]
rows = defaultdict(lambda: defaultdict(list))
for p in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    first = {}  # per kernel: its first dispatch of this pass (cold code and caches, raster tile order) is left out
    recs = [r for r in csv.DictReader(open(p)) if "render_kernel" in r["Kernel_Name"]]
    for r in recs:
        k = r["Kernel_Name"]
        first[k] = min(first.get(k, 1 << 62), int(r["Dispatch_Id"]))
    for r in recs:
        if int(r["Dispatch_Id"]) != first[r["Kernel_Name"]]:
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())} for k, d in rows.items()}
json.dump({wl: res}, open(f"{out}.json", "w"), indent=1)
for k, d in res.items():
    print(k)
    for c in sorted(d):
        print(f"   {c:28s} {d[c]:18.1f}")
PY
exit $rc
