#!/usr/bin/env python3
"""Renders one workload a few times with and without the f64 chain (for rocprofv3 --pmc / --kernel-trace runs:
the two modes are different kernel instantiations, so one run reports both).

    python3 scripts/render_loop.py WORKLOAD [WIDTH HEIGHT] [--reps N] [--jitter counter|center]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import ExampleScene, RtOptions, Scene  # noqa: E402

FULL = {
    "checkered_walls_metallic_sphere": (1280, 720), "striped_sphere_multiple_lights": (1920, 1080),
    "glass_sphere_checkered_floor_aerial_view": (1920, 1080), "multiple_glass_and_metallic_spheres": (3840, 2160),
    "cover_image": (3840, 2160), "cover_image_area": (3840, 2160), "shapes_zoo": (1920, 1080), "obj_mesh": (1920, 1080),
}
ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("size", nargs="*", type=int)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--jitter", default="counter")
ap.add_argument("--modes", default="f64,f32")
a = ap.parse_args()
w, h = a.size if len(a.size) == 2 else FULL[a.workload]
ex = ExampleScene(a.workload, w, h)
sc = Scene(ex.scene_desc, 0)
for mode in a.modes.split(","):
    opt = RtOptions()
    opt.jitter_mode = 1 if a.jitter == "counter" else 0
    opt.flags = 2 if mode == "f32" else 0
    ms = []
    for _ in range(a.reps):
        _, st = sc.render(ex.camera_desc, opt)
        ms.append(st.kernel_ms)
    print(f"{a.workload} {w}x{h} {mode}: kernel ms {min(ms):.3f} (best of {a.reps})", flush=True)
sc.close()
