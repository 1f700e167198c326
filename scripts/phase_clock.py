#!/usr/bin/env python3
"""With a library built with -DRT_PHASE_CLOCK (scripts/exp_build.sh): where the waves of ONE band of 8 rows spend
their time (the library prints the phase ticks of a stats render to stderr).
    RT_AMD_LIB=scratch/exp/librt_amd_clock.so python3 scripts/phase_clock.py WORKLOAD W H BAND [--f32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import ExampleScene, RtOptions, Scene  # noqa: E402

name, w, h, band = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ex = ExampleScene(name, w, h)
sc = Scene(ex.scene_desc, 0)
o = RtOptions()
o.jitter_mode = 1
o.flags = 2 if "--f32" in sys.argv else 0
if band >= 0:
    o.tile_rows, o.tile_first, o.tile_stride = 8, band, 1 << 20
for _ in range(2):
    _, st = sc.render(ex.camera_desc, o, stats=True)
    print(f"band {band}: stats kernel {st.kernel_ms:.3f} ms, rays cast {st.rays_cast()}", flush=True)
