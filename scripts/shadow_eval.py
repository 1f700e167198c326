#!/usr/bin/env python3
"""CPU check of the exact chain's shadow-ray decisions on the pixels the last GPU census listed as outliers.

    python scripts/shadow_eval.py gpurun_out/census3 [gpurun_out/census_amb64 ...]

For every <scene>_<jitter>.npz of the census directories the scalar spec (tests/cpp/hostsim.cpp, the kernel's
arithmetic) renders the listed pixels again and they are compared with the oracle's colours stored beside them:
how many of the census outliers are left, and what the shadow walks did (rays, rays left to f64, answers changed).
"""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from raytracer_amd import ExampleScene, RtOptions  # noqa: E402

S = H.hostsim()
S.hostsim_trace_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
S.hostsim_shadow_stats.argtypes = [C.c_void_p, C.c_int]
seen = set()
tot_before = tot_after = 0
for d in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(d, "*.npz"))):
        key = os.path.basename(path)
        if key in seen:
            continue
        seen.add(key)
        g = np.load(path)
        n = len(g["x"])
        if n == 0:
            continue
        w, h, jitter, seed = int(g["width"]), int(g["height"]), int(g["jitter"]), int(g["seed"])
        name = key.rsplit("_", 1)[0]
        ex = ExampleScene(name, w, h)
        opt = RtOptions()
        opt.jitter_mode, opt.seed = jitter, seed
        xy = np.ascontiguousarray(np.stack([g["x"], g["y"]], axis=1), dtype=np.int32)
        ev = np.zeros((n, 64, 24))
        cnt = np.zeros(n, dtype=np.int32)
        rgb = np.zeros((n, 3), dtype=np.float32)
        st = (C.c_uint64 * 7)()
        S.hostsim_shadow_stats(st, 1)
        rc = S.hostsim_trace_pixels(C.cast(ex.scene_desc, C.c_void_p), C.cast(ex.camera_desc, C.c_void_p), C.cast(C.byref(opt), C.c_void_p),
                                    xy.ctypes.data, n, ev.ctypes.data, 64, cnt.ctypes.data, rgb.ctypes.data)
        assert rc == 0
        S.hostsim_shadow_stats(st, 1)
        err = np.abs(rgb.astype(np.float64) - g["ref"]).max(axis=1)
        left = int((err > 1e-4).sum())
        tot_before += n
        tot_after += left
        print(f"{key:58s} census outliers {n:4d} -> spec {left:4d}   shadow rays {st[0]:7d}, unsure {st[1]:5d}, f64 tests {st[2]:4d}, pixels again {st[3]:4d}"
              + ("   left: " + " ".join(f"({x},{y}):{e:.1e}" for (x, y), e in zip(xy[err > 1e-4][:6], err[err > 1e-4][:6])) if left else ""))
print(f"total {tot_before} -> {tot_after}")
