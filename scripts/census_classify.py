#!/usr/bin/env python3
"""Names the class of every parity outlier the census found (CPU only; scripts/parity_census.py wrote
the outlier lists on the GPU box).

    python scripts/census_classify.py gpurun_out/census [--max 4000] > profiles/r02_parity_census.txt

For each outlier pixel the f64 oracle and the scalar FP32 spec of the kernel arithmetic (tests/cpp/
hostsim.cpp) both trace the pixel's path tree and record, per shaded path node, the decisions taken
there: surface hit (point, normal), pattern colour, light intensity per light, n1/n2, which children
were traced.  The FIRST node (in path-tree order) where the two disagree on a discrete decision names
the class:
  tie          another surface at (almost) the same t — the corner line of two walls, coincident faces
  silhouette   another surface (or none) at a different t — the ray grazes an edge of a shape
  pattern      same surface, other pattern colour — the point lies on a stripe/checker/ring boundary
  shadow       same surface and colour, other light intensity — a shadow ray grazes an occluder
  n1n2 / tir   other refractive indices / other total-internal-reflection decision
  continuous   no discrete decision differs: accumulated rounding error above 1e-4 (deep glass paths)
  gpu-only     the scalar spec agrees with the oracle; only the kernel's 1-ulp rcp/rsq flips the pixel
`depth` is the path-tree level of that node (0 = the primary hit).
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from raytracer_amd import ExampleScene, RtOptions  # noqa: E402

EV, CAP = 24, 64
# oracle Shape::Kind {CUBE, CYLINDER, GROUP, PLANE, SMOOTH_TRIANGLE, SPHERE, TRIANGLE} -> RT_PRIM_* of the product
ORACLE_KIND = {0: 2, 1: 3, 3: 1, 4: 5, 5: 0, 6: 4}


def traces(name, w, h, jitter, seed, xy):
    n = len(xy)
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    L = H.oracle()
    L.oracle_trace_pixels.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    oe = np.zeros((n, CAP, EV))
    oc = np.zeros(n, dtype=np.int32)
    assert L.oracle_trace_pixels(name.encode(), w, h, jitter, seed, xy.ctypes.data, n, oe.ctypes.data, CAP, oc.ctypes.data) == 0
    S = H.hostsim()
    S.hostsim_trace_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    ex = ExampleScene(name, w, h)
    opt = RtOptions()
    opt.jitter_mode, opt.seed = jitter, seed
    se = np.zeros((n, CAP, EV))
    scn = np.zeros(n, dtype=np.int32)
    rgb = np.zeros((n, 3), dtype=np.float32)
    assert S.hostsim_trace_pixels(C.cast(ex.scene_desc, C.c_void_p), C.cast(ex.camera_desc, C.c_void_p), C.cast(C.byref(opt), C.c_void_p),
                                  xy.ctypes.data, n, se.ctypes.data, CAP, scn.ctypes.data, rgb.ctypes.data) == 0
    return oe, oc, se, scn, rgb


def level(node):
    return int(np.floor(np.log2(node + 1)))


def classify(oe, on, se, sn):
    o = {int(e[0]): e for e in oe[:on]}
    s = {int(e[0]): e for e in se[:sn]}
    for node in sorted(set(o) | set(s)):
        a, b = o.get(node), s.get(node)
        if a is None or b is None:
            # a node only one side shaded: its parent agreed on tracing the child, so one ray missed everything
            return "silhouette", level(node)
        scale = max(1.0, float(np.abs(a[2:5]).max()))
        same_surface = np.abs(a[2:5] - b[2:5]).max() <= 1e-3 * scale and np.dot(a[5:8], b[5:8]) > 0.99 and ORACLE_KIND.get(int(a[20]), -1) == int(b[20])
        if not same_surface:
            close = abs(a[1] - b[1]) <= 1e-4 * max(1.0, abs(a[1]))
            return ("tie" if close else "silhouette"), level(node)
        if np.abs(a[11:14] - b[11:14]).max() > 1e-6:
            grad = 0 < np.abs(a[11:14] - b[11:14]).max() < 1e-3
            return ("pattern(gradient)" if grad else "pattern"), level(node)
        if np.abs(a[14:18] - b[14:18]).max() > 1e-9:
            return "shadow", level(node)
        # (the scalar spec only works n1/n2 out for transparent materials; the oracle always does)
        if (a[19] or b[19]) and (abs(a[9] - b[9]) > 1e-6 or abs(a[10] - b[10]) > 1e-6):
            return "n1n2", level(node)
        if a[18] != b[18] or a[19] != b[19]:
            return "tir", level(node)
    return "continuous", max(level(n) for n in o) if o else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--max", type=int, default=4000)
    a = ap.parse_args()
    summary = {}
    cj = os.path.join(a.dir, "census.json")
    if os.path.exists(cj):
        for r in json.load(open(cj)):
            summary[(r["scene"], r["jitter"])] = r
    print("# Parity census — HIP path vs f64 oracle, every config at its BASELINE resolution (scripts/parity_census.py on MI355X,")
    print("# classes by scripts/census_classify.py).  bad = pixels with a channel off by more than 1e-4.\n")
    for path in sorted(glob.glob(os.path.join(a.dir, "*.npz"))):
        g = np.load(path)
        w, h, jitter, seed = int(g["width"]), int(g["height"]), int(g["jitter"]), int(g["seed"])
        name = os.path.basename(path).rsplit("_", 1)[0]
        key = (name, "counter" if jitter else "centre")
        r = summary.get(key, {})
        xs, ys = g["x"], g["y"]
        n = len(xs)
        print(f"## {name} {w}x{h} jitter={key[1]}: bad {n} of {r.get('pixels_compared', '?')} = {r.get('bad_frac', float('nan')):.2e}"
              f"{' (SAMPLED bands)' if r.get('sampled_bands') else ''}, max {r.get('max_abs', float('nan')):.2e}, median {r.get('median_abs', float('nan')):.1e}; "
              f"PNG bytes off by >1: {r.get('png_bytes_off_by_more_than_1', '?')} in {r.get('png_pixels_off_by_more_than_1', '?')} pixels, by 1: {r.get('png_bytes_off_by_1', '?')}")
        if n == 0:
            print()
            continue
        idx = np.arange(n)
        if n > a.max:
            idx = np.sort(np.random.default_rng(0).choice(n, a.max, replace=False))
        xy = np.stack([xs[idx], ys[idx]], axis=1)
        oe, oc, se, scn, rgb = traces(name, w, h, jitter, seed, xy)
        classes = Counter()
        worst = {}
        for k in range(len(idx)):
            d_sim = np.abs(rgb[k].astype(np.float64) - g["ref"][idx[k]]).max()
            if d_sim <= 1e-4:
                cls, lev = "gpu-only", -1  # the scalar spec of the same arithmetic agrees with the oracle here
            else:
                cls, lev = classify(oe[k], oc[k], se[k], scn[k])
            err = float(np.abs(g["gpu"][idx[k]].astype(np.float64) - g["ref"][idx[k]]).max())
            classes[(cls, lev)] += 1
            if err > worst.get((cls, lev), (0, None))[0]:
                worst[(cls, lev)] = (err, (int(xy[k, 0]), int(xy[k, 1])))
        tot = sum(classes.values())
        note = f" (classified a random {tot} of {n})" if tot < n else ""
        print(f"   class (depth){note}:")
        for (cls, lev), c in sorted(classes.items(), key=lambda kv: -kv[1]):
            e, at = worst[(cls, lev)]
            print(f"     {cls:18s} depth {lev:2d}: {c:6d}  ({100.0 * c / tot:5.1f} %)  worst {e:.2e} at pixel {at}")
        print()


if __name__ == "__main__":
    main()
