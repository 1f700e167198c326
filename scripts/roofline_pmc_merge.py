#!/usr/bin/env python3
"""Folds the per-workload counter files of scripts/roofline_pmc.sh into profiles/pmc.json (read by bench.py).

    python scripts/roofline_pmc_merge.py gpurun_out/pmc_<tag> [--note TEXT]

profiles/pmc.json: {workload: {kernel name: {counter: mean per launch, "launches": n}}, "_meta": {...}}.
The counters belong to ONE build of the kernels: re-collect after a change that moves the instruction mix.
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    note = sys.argv[sys.argv.index("--note") + 1] if "--note" in sys.argv else ""
    dst = os.path.join(ROOT, "profiles", "pmc.json")
    table = {}
    if os.path.exists(dst):
        with open(dst) as f:
            table = json.load(f)
    for p in sorted(glob.glob(os.path.join(src, "*.json"))):
        with open(p) as f:
            for wl, kernels in json.load(f).items():
                table[wl] = {k.replace("void ", "").replace("(rtk::KParams)", ""): v for k, v in kernels.items()}
    rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    table["_meta"] = {"source": os.path.relpath(src, ROOT), "collected_at_commit_or_later": rev, "note": note,
                      "units": "SQ_* = events per launch (cycle counters in quad-cycles summed over waves); FETCH_SIZE/WRITE_SIZE = KiB; "
                               "GRBM_GUI_ACTIVE = busy cycles summed over the 8 XCDs",
                      "workload_sizes": "each workload at its BASELINE.json resolution, counter jitter seed 0 where it has area lights"}
    with open(dst, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print("wrote", dst, "workloads:", [k for k in table if k != "_meta"])


if __name__ == "__main__":
    main()
