#!/usr/bin/env python3
"""Full-frame parity census: every config at its BASELINE resolution, HIP path vs the f64 oracle.

    python scripts/parity_census.py [--out gpurun_out/census] [--threads 16] [--rows-4k N] [names...]

Per scene (deterministic scenes with centre jitter; area-light scenes ALSO with the shared counter RNG,
seed 0 — the bench's own mode) the whole frame is rendered on the GPU through the C ABI and by the oracle
on the host threads, and compared per channel:
  pixels above 1e-4 (count, fraction, max error), PNG bytes differing by more than 1, by exactly 1.
The outlier coordinates with both colours go to <out>/<scene>_<jitter>.npz; scripts/census_classify.py
(CPU only: oracle + scalar spec path traces) names the class of every one of them.
`--rows-4k N` limits the 3840x2160 scenes to N evenly spread 8-row bands if a whole frame is too slow
for the oracle (the count is then scaled and marked as sampled).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from raytracer_amd import ExampleScene, RtOptions, Scene, quantize_u8  # noqa: E402

FULL = {
    "checkered_walls_metallic_sphere": (1280, 720), "striped_sphere_multiple_lights": (1920, 1080),
    "glass_sphere_checkered_floor_aerial_view": (1920, 1080), "multiple_glass_and_metallic_spheres": (3840, 2160),
    "cover_image": (3840, 2160), "cover_image_area": (3840, 2160), "shapes_zoo": (1920, 1080), "obj_mesh": (1920, 1080),
}
AREA = {"checkered_walls_metallic_sphere", "striped_sphere_multiple_lights", "glass_sphere_checkered_floor_aerial_view",
        "cover_image_area", "shapes_zoo"}
TOL = 1e-4


def census(name, w, h, jitter, seed, threads, bands, out_dir):
    ex = ExampleScene(name, w, h)
    sc = Scene(ex.scene_desc, 0)
    opt = RtOptions()
    opt.jitter_mode, opt.seed = jitter, seed
    gpu, st = sc.render(ex.camera_desc, opt)
    sc.close()
    if bands:  # evenly spread 8-row bands
        n_bands = (h + 7) // 8
        pick = np.unique(np.linspace(0, n_bands - 1, bands).astype(int))
        rows = np.concatenate([np.arange(b * 8, min(h, b * 8 + 8)) for b in pick])
    else:
        rows = np.arange(h)
    ref = np.zeros((h, w, 3), dtype=np.float64)
    t0 = time.time()
    if bands:
        for b in pick:
            r, _, _ = H.oracle_render(name, w, h, jitter=jitter, seed=seed, threads=threads, rows=(int(b * 8), int(min(h, b * 8 + 8))))
            ref[b * 8: b * 8 + 8] = r[b * 8: b * 8 + 8]
    else:
        ref, _, _ = H.oracle_render(name, w, h, jitter=jitter, seed=seed, threads=threads)
    sec = time.time() - t0
    g, r = gpu[rows].astype(np.float64), ref[rows]
    d = np.abs(g - r)
    bad = (d > TOL).any(axis=-1)
    q = np.abs(quantize_u8(gpu[rows]).astype(int) - H.oracle_quantize(r).astype(int))
    ys, xs = np.nonzero(bad)
    res = {
        "scene": name, "size": [w, h], "jitter": "counter" if jitter else "centre", "seed": seed,
        "pixels_compared": int(bad.size), "sampled_bands": int(bands) if bands else 0,
        "bad_pixels": int(bad.sum()), "bad_frac": float(bad.mean()), "max_abs": float(d.max()),
        "median_abs": float(np.median(d)), "p999_abs": float(np.quantile(d, 0.999)),
        "png_bytes_off_by_more_than_1": int((q > 1).sum()), "png_bytes_off_by_1": int((q == 1).sum()),
        "png_pixels_off_by_more_than_1": int((q > 1).any(axis=-1).sum()),
        "gpu_kernel_ms": float(st.kernel_ms), "oracle_seconds": sec,
    }
    np.savez_compressed(os.path.join(out_dir, f"{name}_{res['jitter']}.npz"), x=xs.astype(np.int32), y=rows[ys].astype(np.int32),
                        gpu=gpu[rows][ys, xs], ref=r[ys, xs], width=w, height=h, jitter=jitter, seed=seed)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "census"))
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--rows-4k", type=int, default=0)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink every frame (rehearsals)")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    results = []
    cj = os.path.join(a.out, "census.json")
    if os.path.exists(cj):  # earlier calls into the same directory: keep their rows
        with open(cj) as f:
            results = json.load(f)
    for name in (a.names or list(FULL)):
        w, h = FULL[name]
        w, h = max(8, int(w * a.scale)), max(8, int(h * a.scale))
        modes = [(0, 0)] + ([(1, 0)] if name in AREA else [])
        for jitter, seed in modes:
            bands = a.rows_4k if (h >= 2160 and a.rows_4k) else 0
            r = census(name, w, h, jitter, seed, a.threads, bands, a.out)
            results = [x for x in results if (x["scene"], x["jitter"]) != (r["scene"], r["jitter"])] + [r]
            print(f"{name:42s} {w}x{h} {r['jitter']:7s} bad {r['bad_pixels']:6d} / {r['pixels_compared']} = {r['bad_frac']:.2e} "
                  f"max {r['max_abs']:.2e} p99.9 {r['p999_abs']:.1e} | PNG bytes >1: {r['png_bytes_off_by_more_than_1']} "
                  f"(=1: {r['png_bytes_off_by_1']}) | gpu {r['gpu_kernel_ms']:.2f} ms oracle {r['oracle_seconds']:.1f} s", flush=True)
            with open(os.path.join(a.out, "census.json"), "w") as f:
                json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
