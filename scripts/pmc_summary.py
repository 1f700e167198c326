#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of scripts/pmc.sh: per kernel, mean counter value per dispatch.

    python scripts/pmc_summary.py gpurun_out/prof_<tag> [> profiles/<round>_<tag>_pmc.txt]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    rows = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
    for path in sorted(glob.glob(os.path.join(root, "pmc*", "*_counter_collection.csv"))):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for path in glob.glob(os.path.join(root, "stats", "*_kernel_stats.csv")):
        print("# kernel stats:", path)
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                print("  %-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (r["Name"][:60], r["Calls"], r["AverageNs"],
                                                                         r["TotalDurationNs"], r["Percentage"]))
    for k in sorted(rows):
        print(f"# {k}")
        for c in sorted(rows[k]):
            v = rows[k][c]
            print(f"  {c:28s} dispatches={len(v):3d} mean={sum(v) / len(v):16.1f}")
        d = {c: sum(v) / len(v) for c, v in rows[k].items()}
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d and d["SQ_WAVES"]:
            print(f"  -> VALU insts per wave      {d['SQ_INSTS_VALU'] / d['SQ_WAVES']:12.1f}")
            for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if c in d:
                    print(f"  -> {c[9:]:5s} insts per wave     {d[c] / d['SQ_WAVES']:12.1f}")
        if "SQ_WAVE_CYCLES" in d:
            wc = d["SQ_WAVE_CYCLES"]
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY",
                      "SQ_WAIT_INST_ANY"):
                if c in d:
                    print(f"  -> {c:22s} / SQ_WAVE_CYCLES = {d[c] / wc:6.3f}")
            if "SQ_BUSY_CYCLES" in d:
                print(f"  -> mean waves resident (WAVE_CYCLES / BUSY_CYCLES) = {wc / d['SQ_BUSY_CYCLES']:6.2f} (per SQ, quad-cycles)")


if __name__ == "__main__":
    main()
