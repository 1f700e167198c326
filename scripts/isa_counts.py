#!/usr/bin/env python3
"""Instruction mix of every render_kernel instantiation, from the device assembly: scalar loads, vector (global / flat /
buffer) loads, scratch traffic, LDS instructions, total.  A kernel whose record loads stopped being scalar (the MemorySSA
budget) shows here as s_load falling and global_load rising.  Usage: scripts/isa_counts.py [-D...] [filter]"""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

flags = [a for a in sys.argv[1:] if a.startswith("-")]
filt = [a for a in sys.argv[1:] if not a.startswith("-")]


def one(group):
    out = f"/tmp/rt_isa_{group}.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-mllvm", "-memssa-check-limit=2000",
           f"-DRT_KGROUP={group}", "--cuda-device-only", "-S", "raytracer_amd/csrc/rt_kernels.hip", "-o", out]
    subprocess.run(cmd + flags, check=True, capture_output=True)
    return open(out).read()


with ThreadPoolExecutor(8) as ex:
    texts = list(ex.map(one, [0, 1, 3, 4, 5, 7]))
print("%-66s %6s %6s %6s %6s %6s %6s %7s" % ("kernel", "s_load", "vload", "vstore", "scr_ld", "scr_st", "ds", "total"))
for text in texts:
    cur, cnt = None, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN3rtk13render_kernel\S+):", line)
        if m:
            cur, cnt = m.group(1), dict(s_load=0, vload=0, vstore=0, scr_ld=0, scr_st=0, ds=0, total=0)
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip().replace("void rtk::", "").replace("(rtk::KParams)", "")
            if not filt or any(f in name for f in filt):
                print("%-66s %6d %6d %6d %6d %6d %6d %7d" % (name[:66], cnt["s_load"], cnt["vload"], cnt["vstore"], cnt["scr_ld"], cnt["scr_st"], cnt["ds"], cnt["total"]))
            cur = None
            continue
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        cnt["total"] += 1
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            cnt["s_load"] += 1
        elif op.startswith(("global_load", "flat_load", "buffer_load")):
            cnt["vload"] += 1
        elif op.startswith(("global_store", "flat_store", "buffer_store", "global_atomic", "flat_atomic")):
            cnt["vstore"] += 1
        elif op.startswith("scratch_load"):
            cnt["scr_ld"] += 1
        elif op.startswith("scratch_store"):
            cnt["scr_st"] += 1
        elif op.startswith("ds_"):
            cnt["ds"] += 1
