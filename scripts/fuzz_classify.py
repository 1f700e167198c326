#!/usr/bin/env python3
"""Names the class of every outlier a GPU fuzz sweep listed (CPU only): for each `seed N: (x,y) err` line of the
fuzz_<family>.txt files scripts/fuzz_worlds.py --gpu wrote into a directory, the f64 oracle and the scalar spec
(tests/cpp/hostsim.cpp) trace the pixel, and the first path node where they part is printed — or "gpu-only" when the
spec agrees with the oracle and only the kernel's 1-ulp rcp / rsq flips the pixel.

    python scripts/fuzz_classify.py gpurun_out/r3/fuzz_new
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import census_classify as cc
import helpers as H
names = ["node","t","px","py","pz","nx","ny","nz","inside","n1","n2","pr","pg","pb","L0","L1","L2","L3","refl","refr","kind","refined","e22","e23"]
for fam in ["random_world", "random_field", "random_stress", "random_mesh"]:
    path = os.path.join(sys.argv[1], f"fuzz_{fam}.txt")
    if not os.path.exists(path):
        continue
    for line in open(path):
        m = re.match(r"seed (\d+): \((\d+),(\d+)\) (\S+)", line)
        if not m: continue
        seed, x, y, err = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))
        name = f"{fam}:{seed}"
        xy = np.array([[x, y]], dtype=np.int32)
        oe, oc, se, scn, rgb = cc.traces(name, 160, 96, 0, 0, xy)
        want = H.oracle_probe(name, 160, 96, xy)
        d_sim = np.abs(rgb[0].astype(np.float64) - want[0]).max()
        cls = cc.classify(oe[0], oc[0], se[0], scn[0]) if d_sim > 1e-4 else ("gpu-only", -1)
        o = {int(e[0]): e for e in oe[0][:oc[0]]}; s = {int(e[0]): e for e in se[0][:scn[0]]}
        detail = ""
        for node in sorted(set(o) | set(s)):
            a, b = o.get(node), s.get(node)
            if a is None or b is None: detail = f"node {node} only on one side"; break
            diff = [nm for nm, va, vb in zip(names, a, b) if abs(va - vb) > 1e-6 and nm not in ("kind", "refined")]
            if diff: detail = f"node {node}: {diff} okind {int(a[20])} skind {int(b[20])} refined {int(b[21])}"; break
        print(f"{name:22s} ({x},{y}) gpu err {err:.2e} spec err {d_sim:.2e} {cls} {detail}")
