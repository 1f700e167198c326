#!/usr/bin/env python3
"""Path traces of one pixel by the f64 oracle and by the scalar FP32 spec, side by side (census debugging).

    python scripts/trace_pixel.py SCENE WIDTH HEIGHT X Y [jitter seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import census_classify as CC  # noqa: E402

name, w, h, x, y = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
jitter = int(sys.argv[6]) if len(sys.argv) > 6 else 0
seed = int(sys.argv[7]) if len(sys.argv) > 7 else 0
oe, oc, se, sn, rgb = CC.traces(name, w, h, jitter, seed, np.array([[x, y]]))
np.set_printoptions(precision=9, linewidth=200)
for tag, ev, n in (("oracle", oe[0], oc[0]), ("spec  ", se[0], sn[0])):
    for e in sorted(ev[:n], key=lambda e: e[0]):
        print(f"{tag} node {int(e[0]):2d} t={e[1]:.9g} p=({e[2]:.8g} {e[3]:.8g} {e[4]:.8g}) n=({e[5]:.6f} {e[6]:.6f} {e[7]:.6f}) in={int(e[8])} "
              f"n1={e[9]:.4g} n2={e[10]:.4g} pat=({e[11]:.3f} {e[12]:.3f} {e[13]:.3f}) I={e[14:18]} refl={int(e[18])} refr={int(e[19])} kind={int(e[20])}")
print("spec rgb", rgb[0], "class", CC.classify(oe[0], oc[0], se[0], sn[0]))
