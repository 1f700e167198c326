#!/usr/bin/env python3
"""Kernel time of every workload at its config resolution + a parity spot check, in one GPU run.

    python scripts/perf_all.py [--reps 3] [--small] [names...]   > gpurun_out/perf_<tag>.txt

For each scene: render_kernel time (rt_stats.kernel_ms, events on the launch stream; best and mean of
`reps` after one warm-up), unique rays (instrumented variant), Mrays/s, algorithmic TFLOP/s, and the
number of the 256 committed full-resolution f64 probes that differ by more than 1e-4.
`--small` additionally compares a 320x180 render with the oracle (needs a few CPU seconds).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracer_amd import ExampleScene, RtOptions, Scene, flopmodel  # noqa: E402

FULL = {
    "checkered_walls_metallic_sphere": (1280, 720), "striped_sphere_multiple_lights": (1920, 1080),
    "glass_sphere_checkered_floor_aerial_view": (1920, 1080), "multiple_glass_and_metallic_spheres": (3840, 2160),
    "cover_image": (3840, 2160), "cover_image_area": (3840, 2160), "shapes_zoo": (1920, 1080), "obj_mesh": (1920, 1080),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    for name in (a.names or list(FULL)):
        w, h = FULL[name]
        ex = ExampleScene(name, w, h)
        sc = Scene(ex.scene_desc, 0)
        ms = []
        img = None
        for k in range(a.reps + 1):
            opt = RtOptions()  # centre jitter: comparable with the committed probes
            img, st = sc.render(ex.camera_desc, opt)
            if k:
                ms.append(st.kernel_ms)
        g = np.load(os.path.join(ROOT, "tests", "golden", f"probes_{name}.npz"))
        xy, rgb = g["xy"], g["rgb"]
        got = img[xy[:, 1], xy[:, 0]]
        bad = int((np.abs(got - rgb) > 1e-4).any(axis=1).sum())
        opt = RtOptions()
        opt.jitter_mode, opt.seed = 1, 0
        msj = []
        for k in range(a.reps + 1):
            _, st = sc.render(ex.camera_desc, opt)
            if k:
                msj.append(st.kernel_ms)
        opt32 = RtOptions()
        opt32.jitter_mode, opt32.seed, opt32.flags = 1, 0, 2  # RT_FLAG_F32_ONLY: the same frame without the f64 chain
        ms32 = []
        for k in range(a.reps + 1):
            _, st = sc.render(ex.camera_desc, opt32)
            if k:
                ms32.append(st.kernel_ms)
        _, st = sc.render(ex.camera_desc, opt, stats=True)
        d = st.as_dict()
        rays = st.total_rays()
        n_lights = int(ex.scene_desc.contents.n_lights)
        has_nodes = int(ex.scene_desc.contents.n_nodes) > 0
        flops = None if has_nodes else flopmodel.frame_flops(d, n_lights, flopmodel.flat_ray_cost(ex.scene_desc))
        line = (f"{name:42s} {w}x{h} centre: best {min(ms):8.3f} ms mean {np.mean(ms):8.3f} | counter: best {min(msj):8.3f} ms (FP32 only {min(ms32):7.3f}, "
                f"f64 nodes {d.get('exact_nodes', 0) / 1e6:.2f} M) "
                f"| rays {rays / 1e6:9.2f} M -> {rays / min(msj) / 1e3:9.1f} Mrays/s")
        if flops:
            line += f" | {flops / min(msj) / 1e9:6.2f} TFLOP/s ({100 * flops / min(msj) / 1e9 / flopmodel.PEAK_FP32_TFLOPS:4.1f} %)"
        line += f" | probes>1e-4: {bad}/256 max {np.abs(got - rgb).max():.1e}"
        print(line, flush=True)
        if a.small:
            import helpers as H

            exs = ExampleScene(name, 320, 180)
            scs = Scene(exs.scene_desc, 0)
            for jit in (0, 1):
                o = RtOptions()
                o.jitter_mode, o.seed = jit, 7
                im, _ = scs.render(exs.camera_desc, o)
                ref, _, _ = H.oracle_render(name, 320, 180, jitter=jit, seed=7, threads=16)
                sim = H.hostsim_render(exs, o)
                c = H.compare(im, ref)
                print(f"    320x180 jitter={jit}: vs oracle max {c['max_abs']:.2e} bad {c['bad_pixels']} | vs hostsim max "
                      f"{np.abs(im - sim).max():.2e}", flush=True)
            scs.close()
        sc.close()


if __name__ == "__main__":
    main()
