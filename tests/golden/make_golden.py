"""Generates the committed golden fixtures from the f64 oracle (oracle/rt_oracle.hpp).

The reference is Rust and cannot run here (no cargo/rustc), and it ships no golden images, so these
vectors come from the build's own oracle, which is pinned by the reference's known-answer tests
(tests/cpp/oracle_kat.cpp).  They are DATA: inputs (scene name, size, jitter mode, seed) and expected
outputs (f64 colours); nothing of the reference's source is stored.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz, *.json

Fixtures
  images.npz          per scene, 96x54 (test_world: 11x11), jitter = centre: f64 rgb
  images_counter.npz  the area-light scenes again with the counter RNG, seed 3
  probes_<scene>.npz  256 pixel coordinates at the config's FULL resolution + their f64 colours
  c4_spheres.json     the frozen 576-sphere table of config 4 (SplitMix64 seed 20230201): per sphere
                      the first three rows of transform_inverse, rounded to f32, + material index
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402

SMALL = (96, 54)
FULL = {  # BASELINE.json configs
    "checkered_walls_metallic_sphere": (1280, 720),
    "striped_sphere_multiple_lights": (1920, 1080),
    "glass_sphere_checkered_floor_aerial_view": (1920, 1080),
    "multiple_glass_and_metallic_spheres": (3840, 2160),
    "cover_image": (3840, 2160),
    "cover_image_area": (3840, 2160),
    "shapes_zoo": (1920, 1080),
    "obj_mesh": (1920, 1080),
}
AREA_SCENES = ["checkered_walls_metallic_sphere", "striped_sphere_multiple_lights",
               "glass_sphere_checkered_floor_aerial_view", "cover_image_area", "shapes_zoo"]


def main():
    images = {}
    for name in H.SCENES:
        w, h = (11, 11) if name == "test_world" else SMALL
        images[name], _, _ = H.oracle_render(name, w, h, jitter=H.JITTER_CENTER, threads=8)
    np.savez_compressed(os.path.join(HERE, "images.npz"), **images)

    counter = {}
    for name in AREA_SCENES:
        counter[name], _, _ = H.oracle_render(name, *SMALL, jitter=H.JITTER_COUNTER, seed=3, threads=8)
    np.savez_compressed(os.path.join(HERE, "images_counter.npz"), **counter)

    for name, (w, h) in FULL.items():
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        xy = np.stack([rng.integers(0, w, 256), rng.integers(0, h, 256)], axis=1).astype(np.int32)
        rgb = H.oracle_probe(name, w, h, xy, jitter=H.JITTER_CENTER)
        np.savez_compressed(os.path.join(HERE, f"probes_{name}.npz"), width=w, height=h, xy=xy, rgb=rgb)

    # frozen C4 world, as flattened by the product host API (values only)
    from raytracer_amd import ExampleScene
    import ctypes as C

    ex = ExampleScene("multiple_glass_and_metallic_spheres", 64, 36)
    d = ex.scene_desc.contents
    inv = np.ctypeslib.as_array(C.cast(d.prim_inv, C.POINTER(C.c_float)), shape=(d.n_prims, 12)).copy()
    mat = np.ctypeslib.as_array(C.cast(d.prim_material, C.POINTER(C.c_uint32)), shape=(d.n_prims,)).copy()
    typ = np.ctypeslib.as_array(C.cast(d.prim_type, C.POINTER(C.c_uint8)), shape=(d.n_prims,)).copy()
    nxt = np.ctypeslib.as_array(C.cast(d.item_next, C.POINTER(C.c_uint32)), shape=(d.n_items,)).copy()
    ref = np.ctypeslib.as_array(C.cast(d.item_ref, C.POINTER(C.c_uint32)), shape=(d.n_items,)).copy()
    with open(os.path.join(HERE, "c4_spheres.json"), "w") as f:
        json.dump({"seed": 20230201, "n_prims": int(d.n_prims), "n_nodes": int(d.n_nodes), "n_items": int(d.n_items),
                   "prim_type": typ.tolist(), "prim_material": mat.tolist(),
                   "prim_inv": [[float(np.float32(v)) for v in row] for row in inv],
                   "item_ref": ref.tolist(), "item_next": nxt.tolist()}, f)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
