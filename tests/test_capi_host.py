"""C ABI and host-side logic that need no GPU: the library loads and exports every symbol
include/rt_amd.h declares, argument validation and error behaviour, rt_output_rows band arithmetic,
and the C++ host API's flatten() (SoA layout, group tree, material/pattern tables)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import helpers as H
from raytracer_amd import ExampleScene, RtCameraDesc, RtOptions, RtSceneDesc, band_rows, capi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT_ERR_BAD_ARGUMENT, RT_ERR_UNSUPPORTED, RT_ERR_CAPACITY = -1, -2, -3


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.RT_SYMBOLS), declared ^ set(capi.RT_SYMBOLS)
    L = lib()
    for sym in declared:
        assert getattr(L, sym) is not None
    assert L.rt_abi_version() == capi.RT_ABI_VERSION == 3


def test_struct_layouts_match_the_header():
    # field order/size as in include/rt_amd.h (natural C alignment on x86-64)
    assert C.sizeof(RtCameraDesc) == 4 * 5 + 4 * 12 + 4 + 8 * 15  # f32 fields, has_f64, 15 doubles
    assert C.sizeof(RtOptions) == 7 * 4
    assert C.sizeof(RtSceneDesc) == 8 + 5 * 8 + 27 * 8  # (abi_version, n_prims) + 5 padded counts + 27 pointers


def _arr(ptr, ctype, shape):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=shape)


def test_output_rows_band_arithmetic():
    L = lib()
    cam = RtCameraDesc()
    cam.hsize, cam.vsize = 64, 2160
    rows = C.c_uint32()
    assert L.rt_output_rows(C.byref(cam), None, C.byref(rows)) == 0 and rows.value == 2160
    total = 0
    for world in (1, 2, 3, 4, 8):
        total = 0
        for rank in range(world):
            o = RtOptions()
            o.tile_rows, o.tile_first, o.tile_stride = 8, rank, world
            assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == 0
            assert rows.value == len(band_rows(2160, 8, rank, world))
            total += rows.value
        assert total == 2160  # the bands partition the image
    # ragged last band
    cam.vsize = 21
    o = RtOptions()
    o.tile_rows, o.tile_first, o.tile_stride = 8, 0, 2
    assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == 0 and rows.value == 8 + 5
    assert band_rows(21, 8, 0, 2).tolist() == list(range(0, 8)) + list(range(16, 21))


def test_option_and_camera_validation():
    L = lib()
    rows = C.c_uint32()
    cam = RtCameraDesc()
    cam.hsize, cam.vsize = 0, 10  # camera.rs:110-112 NullDimension
    assert L.rt_output_rows(C.byref(cam), None, C.byref(rows)) == RT_ERR_BAD_ARGUMENT
    assert b"null dimensions" in L.rt_last_error()
    cam.hsize = 10
    o = RtOptions()
    o.max_depth = 9
    assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == RT_ERR_CAPACITY
    o = RtOptions()
    o.jitter_mode = 7
    assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == RT_ERR_UNSUPPORTED
    o = RtOptions()
    o.tile_rows, o.tile_first, o.tile_stride = 8, 2, 2
    assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == RT_ERR_BAD_ARGUMENT
    o = RtOptions()
    o.flags = 8  # an unknown bit (1 = RT_FLAG_GLOBAL_ROWS, 2 = RT_FLAG_F32_ONLY, 4 = RT_FLAG_DEPTH_ZERO)
    assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == RT_ERR_BAD_ARGUMENT
    o = RtOptions()
    o.flags, o.max_depth = 4, 3  # depth 0 is asked for by the flag alone
    assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == RT_ERR_BAD_ARGUMENT
    o.max_depth = 0
    assert L.rt_output_rows(C.byref(cam), C.byref(o), C.byref(rows)) == 0
    assert L.rt_output_rows(None, None, C.byref(rows)) == RT_ERR_BAD_ARGUMENT
    assert L.rt_render(None, C.byref(cam), None, None, None) == RT_ERR_BAD_ARGUMENT


def _copy_desc(ex):
    d = RtSceneDesc()
    C.memmove(C.byref(d), ex.scene_desc, C.sizeof(RtSceneDesc))
    return d


def test_scene_validation_rejects_malformed_descriptions():
    """Validation runs before any device is touched, so it is testable without a GPU."""
    L = lib()
    ex = ExampleScene("multiple_glass_and_metallic_spheres", 16, 9)
    out = C.c_void_p()

    assert L.rt_scene_create(None, 0, C.byref(out)) == RT_ERR_BAD_ARGUMENT
    d = _copy_desc(ex)
    d.abi_version = 99
    assert L.rt_scene_create(C.byref(d), 0, C.byref(out)) == RT_ERR_BAD_ARGUMENT

    d = _copy_desc(ex)
    types = _arr(d.prim_type, C.c_uint8, (d.n_prims,)).copy()
    types[3] = 17
    d.prim_type = types.ctypes.data
    assert L.rt_scene_create(C.byref(d), 0, C.byref(out)) == RT_ERR_UNSUPPORTED

    d = _copy_desc(ex)
    mats = _arr(d.prim_material, C.c_uint32, (d.n_prims,)).copy()
    mats[0] = 1000
    d.prim_material = mats.ctypes.data
    assert L.rt_scene_create(C.byref(d), 0, C.byref(out)) == RT_ERR_BAD_ARGUMENT

    # a skip link that points backwards would make traversal loop forever: must be rejected
    d = _copy_desc(ex)
    nxt = _arr(d.item_next, C.c_uint32, (d.n_items,)).copy()
    nxt[1] = 0
    d.item_next = nxt.ctypes.data
    assert L.rt_scene_create(C.byref(d), 0, C.byref(out)) == RT_ERR_BAD_ARGUMENT
    assert b"item" in L.rt_last_error()

    # a subtree that crosses its parent's end
    d = _copy_desc(ex)
    nxt = _arr(d.item_next, C.c_uint32, (d.n_items,)).copy()
    ref = _arr(d.item_ref, C.c_uint32, (d.n_items,))
    inner = [i for i in range(2, d.n_items) if ref[i] & 0x80000000]
    nxt[inner[0]] = d.n_items if nxt[inner[0]] != d.n_items else nxt[inner[0]]
    parent_end = nxt[1]
    if nxt[inner[0]] > parent_end or True:
        nxt[1] = inner[0] + 1  # parent now ends inside the child's subtree
    d.item_next = nxt.ctypes.data
    assert L.rt_scene_create(C.byref(d), 0, C.byref(out)) == RT_ERR_BAD_ARGUMENT

    d = _copy_desc(ex)
    us = np.array([0], dtype=np.uint32)
    lt = np.array([1], dtype=np.uint8)
    d.light_type, d.light_usteps = lt.ctypes.data, us.ctypes.data
    assert L.rt_scene_create(C.byref(d), 0, C.byref(out)) == RT_ERR_BAD_ARGUMENT


def test_flatten_of_the_example_scenes():
    counts = {  # prims, nodes, lights, non-solid patterns
        "test_world": (2, 0, 1), "checkered_walls_metallic_sphere": (6, 0, 1),
        "striped_sphere_multiple_lights": (2, 0, 2), "glass_sphere_checkered_floor_aerial_view": (7, 0, 1),
        "cover_image": (19, 0, 2), "cover_image_area": (19, 0, 2),
    }
    for name, (np_, nn, nl) in counts.items():
        ex = ExampleScene(name, 16, 9)
        d = ex.scene_desc.contents
        assert (d.n_prims, d.n_nodes, d.n_lights) == (np_, nn, nl), name
        assert d.n_items == d.n_prims + d.n_nodes
        ref = _arr(d.item_ref, C.c_uint32, (d.n_items,))
        assert ref.tolist() == list(range(d.n_prims))  # world.objects order


def test_flatten_deduplicates_materials_and_orders_lights():
    ex = ExampleScene("cover_image", 16, 9)
    d = ex.scene_desc.contents
    assert d.n_materials == 6  # backdrop, glass sphere, WHITE/BLUE/RED/PURPLE (examples/cover-image.rs:15-89)
    mp = _arr(d.mat_param, C.c_float, (d.n_materials, 8))
    assert np.allclose(mp[1], [0.0, 0.2, 1.0, 200.0, 1.5, 0.7, 0.7, 0.0])
    lp = _arr(d.light_pos, C.c_float, (d.n_lights, 3))
    assert lp[0].tolist() == [50.0, 100.0, -50.0] and lp[1].tolist() == [-400.0, 50.0, -10.0]
    ex = ExampleScene("striped_sphere_multiple_lights", 16, 9)
    d = ex.scene_desc.contents
    assert _arr(d.light_usteps, C.c_uint32, (2,)).tolist() == [8, 4]  # left_light listed first (:67)
    uv = _arr(d.light_uvec, C.c_float, (2, 3))
    assert np.allclose(uv[0], [0.5, 0, 0]) and np.allclose(uv[1], [1.0, 0, 0])  # horizontal_dir / cells
    pt = _arr(d.pat_type, C.c_uint8, (d.n_patterns,))
    assert sorted(pt.tolist()) == [0, 1]  # Solid floor + Stripe sphere


def test_flatten_of_the_divided_group_matches_the_frozen_fixture():
    with open(os.path.join(ROOT, "tests", "golden", "c4_spheres.json")) as f:
        g = json.load(f)
    ex = ExampleScene("multiple_glass_and_metallic_spheres", 16, 9)
    d = ex.scene_desc.contents
    assert (d.n_prims, d.n_nodes, d.n_items) == (g["n_prims"], g["n_nodes"], g["n_items"])
    inv = _arr(d.prim_inv, C.c_float, (d.n_prims, 12))
    assert np.array_equal(inv, np.asarray(g["prim_inv"], dtype=np.float32))
    assert _arr(d.item_next, C.c_uint32, (d.n_items,)).tolist() == g["item_next"]
    ref = _arr(d.item_ref, C.c_uint32, (d.n_items,))
    assert ref.tolist() == g["item_ref"]
    # structure implied by group.rs:168-231: every leaf run under one node has < 256 children
    # unless the node could not be split further; the root group is item 1 and spans to the end
    assert ref[0] == 0 and (ref[1] & 0x80000000) and g["item_next"][1] == d.n_items
    # boxes bound their children: sphere centres lie inside their node's box
    nmin = _arr(d.node_min, C.c_float, (d.n_nodes, 3))
    nmax = _arr(d.node_max, C.c_float, (d.n_nodes, 3))
    assert np.all(nmin[0] <= nmin.min(axis=0) + 1e-6) and np.all(nmax[0] >= nmax.max(axis=0) - 1e-6)


def test_camera_desc_matches_the_reference_camera_tests():
    ex = ExampleScene("test_world", 11, 11)
    cam = ex.camera_desc.contents
    # camera.rs:95-136 with fov = pi/2, 11x11: half_width = half_height = 1, pixel_size = 2/11
    assert abs(cam.half_width - 1.0) < 1e-6 and abs(cam.half_height - 1.0) < 1e-6
    assert abs(cam.pixel_size - 2.0 / 11.0) < 1e-7
    inv = np.array(list(cam.inv)).reshape(3, 4)
    assert np.allclose(inv[:, 3], [0, 0, -5])  # view(from=(0,0,-5)) inverse puts the eye at the origin


def test_unknown_scene_is_an_error():
    with pytest.raises(capi.RtError):
        ExampleScene("no_such_scene", 8, 8)


def test_shared_frame_entry_points_validate_their_arguments():
    L = lib()
    frame = C.c_void_p()
    handle = (C.c_ubyte * 64)()
    assert L.rt_frame_create(0, 0, C.byref(frame), C.cast(handle, C.c_void_p)) == RT_ERR_BAD_ARGUMENT  # zero size
    assert L.rt_frame_create(0, 1024, None, C.cast(handle, C.c_void_p)) == RT_ERR_BAD_ARGUMENT
    assert L.rt_frame_open(0, 0, None, C.byref(frame)) == RT_ERR_BAD_ARGUMENT
    assert L.rt_frame_download(0, None, None, 16) == RT_ERR_BAD_ARGUMENT
    assert L.rt_frame_close(0, None) == 0 and L.rt_frame_destroy(0, None) == 0  # nothing to do
    # the GLOBAL_ROWS flag belongs to rt_render_device; the host-output entry points pack their rows
    from raytracer_amd import RT_FLAG_GLOBAL_ROWS
    assert RT_FLAG_GLOBAL_ROWS == 1
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    assert "#define RT_FLAG_GLOBAL_ROWS 1u" in header
