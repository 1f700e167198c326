#!/usr/bin/env python3
"""Fuzz parity sweep: random_world:<seed> scenes (scenes/examples.inc) against the f64 oracle.

    python scripts/fuzz_worlds.py [--first 0 --count 100 --width 96 --height 64] [--gpu] [--classify]

Without --gpu the scalar spec of the kernel arithmetic (tests/cpp/hostsim.cpp) is compared, with --gpu the
kernels through the C ABI.  Prints one line per scene that has a pixel above 1e-4 and a total; --classify names
the class of each such pixel (scripts/census_classify.py: oracle and scalar-spec path traces).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import helpers as H  # noqa: E402
from raytracer_amd import ExampleScene, RtOptions, Scene  # noqa: E402

TOL = 1e-4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--width", type=int, default=96)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--jitter-seed", type=int, default=5)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--family", default="random_world", choices=["random_world", "random_field", "random_stress", "random_mesh"])
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--classify", action="store_true")
    a = ap.parse_args()
    w, h = a.width, a.height
    bad = pixels = 0
    classes = {}
    for seed in range(a.first, a.first + a.count):
        name = f"{a.family}:{seed}"
        ex = ExampleScene(name, w, h)
        o = RtOptions()
        o.jitter_mode, o.seed = 1, a.jitter_seed
        if a.gpu:
            sc = Scene(ex.scene_desc, 0)
            img, _ = sc.render(ex.camera_desc, o)
            sc.close()
        else:
            img = H.hostsim_render(ex, o)
        ref, _, _ = H.oracle_render(name, w, h, jitter=1, seed=a.jitter_seed, threads=a.threads)
        d = np.abs(img - ref).max(axis=-1)
        ys, xs = np.nonzero(d > TOL)
        pixels += w * h
        bad += len(xs)
        if len(xs) == 0:
            continue
        line = f"seed {seed}: " + " ".join(f"({x},{y}) {d[y, x]:.2e}" for x, y in zip(xs, ys))
        if a.classify:
            import census_classify as CC
            xy = np.stack([xs, ys], axis=1)
            oe, oc, se, scn, rgb = CC.traces(name, w, h, 1, a.jitter_seed, xy)
            for k in range(len(xs)):
                if np.abs(rgb[k].astype(np.float64) - ref[ys[k], xs[k]]).max() <= TOL:
                    cls = ("gpu-only", -1)
                else:
                    cls = CC.classify(oe[k], oc[k], se[k], scn[k])
                classes[cls] = classes.get(cls, 0) + 1
                line += f" {cls[0]}@{cls[1]}"
        print(line, flush=True)
    print(f"{a.family} {'gpu' if a.gpu else 'scalar spec'}: seeds {a.first}..{a.first + a.count - 1} at {w}x{h}: {bad} of {pixels} pixels above {TOL} = {bad / max(1, pixels):.2e}")
    for k, v in sorted(classes.items(), key=lambda kv: -kv[1]):
        print(f"   {k[0]} depth {k[1]}: {v}")


if __name__ == "__main__":
    main()
