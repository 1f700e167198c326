"""ctypes access to the TEST infrastructure: the f64 oracle (oracle/_build/liboracle.so) and the
scalar FP32 host simulator of the kernel arithmetic (tests/cpp/_build/libhostsim.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use these.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
ORACLE_KAT = os.path.join(ROOT, "oracle", "_build", "oracle_kat")
HOSTSIM_SO = os.path.join(ROOT, "tests", "cpp", "_build", "libhostsim.so")

JITTER_CENTER, JITTER_COUNTER, JITTER_THREAD_RNG = 0, 1, 2

# (scene, default parity resolution) — 16:9 like the examples
SCENES = [
    "test_world", "checkered_walls_metallic_sphere", "striped_sphere_multiple_lights",
    "glass_sphere_checkered_floor_aerial_view", "multiple_glass_and_metallic_spheres",
    "cover_image", "cover_image_area", "shapes_zoo", "obj_mesh",
]

_oracle = None
_hostsim = None


def oracle():
    global _oracle
    if _oracle is None:
        L = C.CDLL(ORACLE_SO)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_render_example.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_render_example_depth.argtypes = L.oracle_render_example.argtypes + [C.c_int]
        L.oracle_probe_pixels.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_quantize_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_quantize_u8.restype = None
        L.oracle_counter_jitter.argtypes = [C.c_uint32] * 6
        L.oracle_counter_jitter.restype = C.c_double
        L.oracle_scene_name.restype = C.c_char_p
        _oracle = L
    return _oracle


ORACLE_STAT_NAMES = ["primary", "reflect", "refract", "shadow", "shaded_hits", "light_cells", "pattern_lookups",
                     "tests_cube", "tests_cylinder", "tests_group_box", "tests_plane", "tests_smooth_triangle",
                     "tests_sphere", "tests_triangle"]


def oracle_render(name: str, width: int, height: int, jitter: int = JITTER_CENTER, seed: int = 0, threads: int = 0,
                  rows: Optional[Tuple[int, int]] = None, want_stats: bool = False, depth: int = -1):
    """f64 oracle render.  Returns (rgb[h, w, 3] float64, stats dict or None, seconds).
    depth: what World::color_at starts with (world.rs:24); -1 = the reference's RECURSION_DEPTH = 5."""
    L = oracle()
    rgb = np.zeros((height, width, 3), dtype=np.float64)
    stats = np.zeros(14, dtype=np.uint64)
    sec = C.c_double()
    r0, r1 = rows if rows else (0, 0)
    rc = L.oracle_render_example_depth(name.encode(), width, height, jitter, seed, threads, r0, r1, rgb.ctypes.data,
                                       stats.ctypes.data if want_stats else None, C.byref(sec), depth)
    if rc != 0:
        raise RuntimeError(L.oracle_last_error().decode())
    sd = dict(zip(ORACLE_STAT_NAMES, (int(v) for v in stats))) if want_stats else None
    return rgb, sd, sec.value


def oracle_probe(name: str, width: int, height: int, xy: np.ndarray, jitter: int = JITTER_CENTER, seed: int = 0) -> np.ndarray:
    L = oracle()
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    out = np.zeros((xy.shape[0], 3), dtype=np.float64)
    rc = L.oracle_probe_pixels(name.encode(), width, height, jitter, seed, xy.ctypes.data, xy.shape[0], out.ctypes.data)
    if rc != 0:
        raise RuntimeError(L.oracle_last_error().decode())
    return out


def oracle_quantize(rgb: np.ndarray) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.empty(rgb.shape, dtype=np.uint8)
    oracle().oracle_quantize_u8(rgb.ctypes.data, rgb.size, out.ctypes.data)
    return out


def hostsim():
    global _hostsim
    if _hostsim is None:
        L = C.CDLL(HOSTSIM_SO)
        L.hostsim_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        _hostsim = L
    return _hostsim


def hostsim_render(example, options=None) -> np.ndarray:
    """Scalar FP32 execution of the kernel arithmetic on an ExampleScene (tests only)."""
    from raytracer_amd import band_rows
    cam = example.camera_desc.contents
    nrows = cam.vsize if options is None else len(band_rows(cam.vsize, options.tile_rows, options.tile_first,
                                                            max(1, options.tile_stride)))
    out = np.zeros((nrows, cam.hsize, 3), dtype=np.float32)
    err = C.create_string_buffer(256)
    rc = hostsim().hostsim_render(C.cast(example.scene_desc, C.c_void_p), C.cast(example.camera_desc, C.c_void_p),
                                  C.cast(C.byref(options), C.c_void_p) if options is not None else None,
                                  out.ctypes.data, err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return out


def hostsim_streams(scene_desc, staged_top: bool = True) -> np.ndarray:
    """The walk streams of a scene as rt_scene_create keeps them on the device, laid out on the host by rt_pack.h
    ([records, 8] uint32 words): eight direction orders + canonical order + eight top regions of 16 records for a scene
    with a hierarchy, the canonical order alone otherwise.  staged_top=False: no top regions picked (the product's
    default: the regions are all zero and the streams have no portals)."""
    L = hostsim()
    L.hostsim_pack_streams.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t()
    rc = L.hostsim_pack_streams(C.cast(scene_desc, C.c_void_p), int(staged_top), None, 0, C.byref(n))
    assert rc == 0, rc
    out = np.zeros((n.value // 32, 8), dtype=np.uint32)
    rc = L.hostsim_pack_streams(C.cast(scene_desc, C.c_void_p), int(staged_top), out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(n))
    assert rc == 0, rc
    return out


def compare(test: np.ndarray, ref: np.ndarray, tol: float = 1e-4) -> dict:
    """Per-channel absolute error summary of an f32 image against the f64 oracle image."""
    d = np.abs(test.astype(np.float64) - ref)
    bad = (d > tol).any(axis=-1)
    return {
        "max_abs": float(d.max()) if d.size else 0.0,
        "bad_pixels": int(bad.sum()),
        "pixels": int(bad.size),
        "bad_frac": float(bad.mean()) if bad.size else 0.0,
        "rmse": float(np.sqrt((d ** 2).mean())) if d.size else 0.0,
    }


def triangle_soup_desc(n: int, keep: list):
    """rt_scene_desc of a floor plus n small random triangles (three materials, one point light).
    `keep` receives the numpy arrays the description points into."""
    rng = np.random.default_rng(42)
    c = rng.uniform(-6, 6, (n, 3)) * np.array([1.0, 0.35, 1.0]) + np.array([0.0, 2.2, 0.0])
    e0 = rng.normal(size=(n, 3)) * 0.12
    e1 = rng.normal(size=(n, 3)) * 0.12
    nrm = np.cross(e0, e1)  # triangle.rs:82-83: normal = e1 x e0 normalised; the sign only flips shading
    nrm = -nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    params = np.zeros((n + 1, 24), dtype=np.float32)
    params[1:, 0:3], params[1:, 3:6], params[1:, 6:9], params[1:, 9:12] = c, e0, e1, nrm
    types = np.full(n + 1, 4, dtype=np.uint8)
    types[0] = 1  # the floor
    inv = np.tile(np.eye(4)[:3, :].reshape(12), (n + 1, 1)).astype(np.float32)
    mats = (np.arange(n + 1) % 3).astype(np.uint32)
    mat_param = np.array([[0.1, 0.9, 0.3, 50, 1.0, 0.0, 0.0, 0], [0.1, 0.7, 0.2, 10, 1.0, 0.2, 0.0, 0],
                          [0.2, 0.8, 0.9, 300, 1.0, 0.0, 0.0, 0]], dtype=np.float32)
    mat_pat = np.arange(3, dtype=np.uint32)
    pat_type = np.zeros(3, dtype=np.uint8)
    pat_a = np.array([[0.9, 0.9, 0.9], [0.8, 0.3, 0.2], [0.2, 0.5, 0.9]], dtype=np.float32)
    pat_inv = np.tile(np.eye(4)[:3, :].reshape(12), (3, 1)).astype(np.float32)
    l_type = np.zeros(1, dtype=np.uint8)
    l_pos = np.array([[-8.0, 12.0, 6.0]], dtype=np.float32)
    l_vec = np.zeros((1, 3), dtype=np.float32)
    l_steps = np.ones(1, dtype=np.uint32)
    l_col = np.ones((1, 3), dtype=np.float32)
    ref, nxt = np.arange(n + 1, dtype=np.uint32), np.arange(1, n + 2, dtype=np.uint32)
    from raytracer_amd import RtSceneDesc
    d = RtSceneDesc()
    d.abi_version = 3
    d.n_prims, d.prim_type, d.prim_inv = n + 1, types.ctypes.data, inv.ctypes.data
    d.prim_material, d.prim_param = mats.ctypes.data, params.ctypes.data
    d.n_items, d.item_ref, d.item_next = n + 1, ref.ctypes.data, nxt.ctypes.data
    d.n_materials, d.mat_param, d.mat_pattern = 3, mat_param.ctypes.data, mat_pat.ctypes.data
    d.n_patterns, d.pat_type, d.pat_color_a = 3, pat_type.ctypes.data, pat_a.ctypes.data
    d.pat_color_b, d.pat_inv = pat_a.ctypes.data, pat_inv.ctypes.data
    d.n_lights, d.light_type, d.light_pos = 1, l_type.ctypes.data, l_pos.ctypes.data
    d.light_uvec, d.light_vvec = l_vec.ctypes.data, l_vec.ctypes.data
    d.light_usteps, d.light_vsteps, d.light_color = l_steps.ctypes.data, l_steps.ctypes.data, l_col.ctypes.data
    keep += [params, types, inv, mats, mat_param, mat_pat, pat_type, pat_a, pat_inv, l_type, l_pos, l_vec, l_steps, l_col, ref, nxt]
    return d
