#!/usr/bin/env python3
"""Triangle-soup scaling (the mesh path): a floor plus N small random triangles, one point light, 1080p.
    python3 scripts/soup_perf.py [N ...]      (RT_AMD_LANE=0/1 forces the walk mode of the hierarchy kernels)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from raytracer_amd import ExampleScene, RtOptions, Scene  # noqa: E402

cam = ExampleScene("striped_sphere_multiple_lights", 1920, 1080)
for n in [int(a) for a in sys.argv[1:]] or [10_000, 100_000, 1_000_000]:
    keep = []
    d = H.triangle_soup_desc(n, keep)
    t0 = time.time()
    sc = Scene(C.pointer(d), 0)
    t_create = time.time() - t0
    ms = []
    for k in range(4):
        _, st = sc.render(cam.camera_desc, RtOptions())
        if k:
            ms.append(st.kernel_ms)
    _, st = sc.render(cam.camera_desc, RtOptions(), stats=True)
    rays = st.primary_rays + st.reflection_rays + st.refraction_rays + st.shadow_rays
    print(f"{n} triangles: scene_create {t_create:.2f} s, 1080p kernel {min(ms):.3f} ms, {rays / 1e6:.1f} M rays, {rays / min(ms) / 1e3:.0f} Mrays/s "
          f"[{sc.kernel_name(cam.camera_desc)}]", flush=True)
    sc.close()
