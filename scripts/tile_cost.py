#!/usr/bin/env python3
"""Where a frame's critical path is: kernel time of every 8-row band rendered ALONE (all its tiles run at once, so
the time is that of its slowest tile), next to the rays its tiles cast (rt_render_stats_noalloc's per-tile map).

    python3 scripts/tile_cost.py WORKLOAD WIDTH HEIGHT [--f32]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import ExampleScene, RtOptions, Scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("width", type=int)
ap.add_argument("height", type=int)
ap.add_argument("--f32", action="store_true")
a = ap.parse_args()
ex = ExampleScene(a.workload, a.width, a.height)
sc = Scene(ex.scene_desc, 0)
out = torch.zeros((a.height + 8, a.width, 3), dtype=torch.float32, device="cuda:0")
base = RtOptions()
base.jitter_mode = 1
base.flags = 2 if a.f32 else 0
st, tiles = sc.render_stats_noalloc(ex.camera_desc, out.data_ptr(), base, 0, want_tiles=True)
tx = (a.width + 7) // 8
tiles = np.asarray(tiles).reshape(-1, tx)
print(f"{a.workload} {a.width}x{a.height}: {tiles.size} tiles, rays cast per tile: mean {tiles.mean():.0f} median {np.median(tiles):.0f} "
      f"p99 {np.quantile(tiles, 0.99):.0f} max {tiles.max()}")
_, stw = sc.render(ex.camera_desc, base)
print(f"whole frame kernel {stw.kernel_ms:.3f} ms")
bands = (a.height + 7) // 8
rows = []
for b in range(bands):
    o = RtOptions()
    o.jitter_mode, o.flags = 1, base.flags
    o.tile_rows, o.tile_first, o.tile_stride = 8, b, 1 << 20
    best = 1e9
    for _ in range(2):
        _, s = sc.render(ex.camera_desc, o)
        best = min(best, s.kernel_ms)
    rows.append((best, b, int(tiles[b].max()), int(tiles[b].sum())))
rows.sort(reverse=True)
for ms, b, mx, sm in rows[:8]:
    print(f"band {b:4d}: alone {ms:.3f} ms, heaviest tile {mx} rays, band {sm} rays")
print("median band alone:", f"{np.median([r[0] for r in rows]):.3f} ms")
