#!/usr/bin/env python3
"""Registers, spills and scratch of every render_kernel instantiation (hipcc's kernel-resource-usage
remarks), one line per kernel.  Usage: scripts/kernel_resources.py [-DRT_EXP=..]"""
import re
import subprocess
import sys

from concurrent.futures import ThreadPoolExecutor


def one(group):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-mllvm", "-memssa-check-limit=2000", f"-DRT_KGROUP={group}",
           "-Rpass-analysis=kernel-resource-usage", "-c", "raytracer_amd/csrc/rt_kernels.hip", "-o", f"/tmp/rt_res_{group}.o"]
    return subprocess.run(cmd + sys.argv[1:], capture_output=True, text=True).stderr


with ThreadPoolExecutor(8) as ex:  # one translation unit per kernel group (rt_kernels.hip)
    err = "\n".join(ex.map(one, [0, 1, 3, 4, 5, 7]))
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|Name): (\S+)", line)
    if m:
        cur = m.group(2)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-66s %5s %5s %6s %6s %8s %4s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "occ"))
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = name.replace("void rtk::", "").replace("(rtk::KParams)", "")
    print("%-66s %5d %5d %6d %6d %8d %4d" % (name[:66], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", -1),
                                             r.get("SGPRs Spill", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1)))
