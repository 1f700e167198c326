#!/bin/bash
# HBM bytes per launch of the render kernel from the L2 memory-side counters, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc runs (they do
# not fit one pass), no trace domain in the same run, the program directly after `--`.
# Writes gpurun_out/traffic_<tag>.json; copy it to profiles/traffic.json, which bench.py reports as
# roofline.traffic.   Usage: scripts/traffic.sh <tag> [bench args...]
set -u
tag=${1:?tag}
shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/traffic_$tag
mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $c -d "$out/$c" -o $c --output-format csv -- python3 bench.py --steps 5 --warmup 2 \
        --no-cpu-baseline "$@" > "$out/$c.log" 2>&1 || exit 1
done
python3 - "$out" "$tag" <<'EOF'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = {}
    for p in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "render_kernel<false" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    (k, v), = vals.items()
    res["kernel"] = k
    res[c + "_KiB_per_launch"] = sum(v) / len(v)
    res["launches_" + c] = len(v)
# FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM section): doubled.
res["hbm_bytes_per_launch"] = int(1024 * (2 * res["FETCH_SIZE_KiB_per_launch"] + res["WRITE_SIZE_KiB_per_launch"]))
res["note"] = "FETCH_SIZE doubled per the gfx950 correction (calibrated for streaming reads only; an upper estimate here)"
res["tag"] = tag
json.dump(res, open(f"gpurun_out/traffic_{tag}.json", "w"), indent=1)
print(json.dumps(res))
EOF
