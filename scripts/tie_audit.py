#!/usr/bin/env python3
"""Counts, at each config's full resolution, how many nearest-hit decisions the widened tie window of
rt_math.h (1e-5 max(1, t/4)) takes differently from the reference's fixed 1e-5 window
(intersection.rs:106-113).  CPU only: the scalar spec of the kernel arithmetic evaluates both rules
on every path node (tests/cpp/hostsim.cpp, hostsim_tie_audit).

    python scripts/tie_audit.py [names...]  >> profiles/r02_parity_census.txt
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from raytracer_amd import ExampleScene, RtOptions  # noqa: E402

FULL = {
    "checkered_walls_metallic_sphere": (1280, 720), "striped_sphere_multiple_lights": (1920, 1080),
    "glass_sphere_checkered_floor_aerial_view": (1920, 1080), "multiple_glass_and_metallic_spheres": (3840, 2160),
    "cover_image": (3840, 2160), "shapes_zoo": (1920, 1080), "obj_mesh": (1920, 1080),
}
S = H.hostsim()
print("# Tie-window audit (scripts/tie_audit.py): nearest-hit decisions per frame, how many the widened window")
print("# 1e-5 max(1, t/4) takes differently from the fixed 1e-5 window, and how many hits lie at t > 4 (window wider than 1e-5)")
for mode, flags in (("FP32 only", 2), ("with the f64 chain", 0)):
    for name in (sys.argv[1:] or list(FULL)):
        w, h = FULL[name]
        ex = ExampleScene(name, w, h)
        out = np.zeros((h, w, 3), dtype=np.float32)
        st = (C.c_uint64 * 3)()
        err = C.create_string_buffer(256)
        opt = RtOptions()
        opt.flags = flags
        rc = S.hostsim_tie_audit(C.cast(ex.scene_desc, C.c_void_p), C.cast(ex.camera_desc, C.c_void_p), C.cast(C.byref(opt), C.c_void_p),
                                 out.ctypes.data_as(C.c_void_p), st, err, 256)
        assert rc == 0, err.value
        print(f"  {name:42s} {w}x{h} {mode:18s}: {st[0]:10d} hits, {st[1]:6d} decided differently ({st[1] / max(1, st[0]):.1e}), "
              f"{st[2]:10d} at t > 4", flush=True)
