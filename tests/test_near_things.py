"""Near things outside the sphere/cube/plane family (round 3): pixels of the FULL-SIZE shapes_zoo frame that were more
than 1e-4 from the f64 oracle until the kinds of near thing they hang on were handed to the f64 chain
(profiles/r03_parity_census.txt):
  a ray grazing a cylinder's wall or crossing it at a rim             rt_math.h isect_cylinder (kUnsure / soft)
  a cylinder WALL point that got an end's normal (cylinder.rs:172-184) rt_exact.h xprepare: qedge
  a Group's box clipped within FP32 reach of an edge (group.rs:107)    rt_math.h gates_pass_near, rt_exact.h xgates_pass,
                                                                       rt_scene_desc.node_box64
  a shadow ray past a triangle's edge                                  rt_math.h occludes_prim_near
The scalar spec (tests/cpp/hostsim.cpp) is checked here on the CPU; the kernels on the GPU against the same colours."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from raytracer_amd import ExampleScene, RtOptions

W, Hh = 1920, 1080
# (x, y, jitter mode): the outliers of the round-2 and first round-3 censuses of shapes_zoo, centre and counter jitter
PIXELS = [(562, 206, 0), (967, 558, 0), (1424, 727, 0), (856, 777, 0), (1489, 801, 0), (1249, 952, 0), (1472, 956, 0),
          (1260, 962, 0), (562, 206, 1), (1267, 523, 1), (1424, 727, 1), (1489, 801, 1), (1472, 956, 1), (1264, 984, 1),
          (1269, 999, 1)]


def _oracle(jitter):
    xy = np.array([(x, y) for x, y, j in PIXELS if j == jitter], dtype=np.int32)
    return xy, H.oracle_probe("shapes_zoo", W, Hh, xy, jitter=jitter, seed=0)


@pytest.mark.parametrize("jitter", [0, 1])
def test_scalar_spec_follows_the_reference_at_the_former_outliers(jitter):
    xy, want = _oracle(jitter)
    S = H.hostsim()
    S.hostsim_trace_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    ex = ExampleScene("shapes_zoo", W, Hh)
    opt = RtOptions()
    opt.jitter_mode, opt.seed = jitter, 0
    n, cap = len(xy), 64
    ev = np.zeros((n, cap, 24))
    cnt = np.zeros(n, dtype=np.int32)
    rgb = np.zeros((n, 3), dtype=np.float32)
    assert S.hostsim_trace_pixels(C.cast(ex.scene_desc, C.c_void_p), C.cast(ex.camera_desc, C.c_void_p), C.cast(C.byref(opt), C.c_void_p),
                                  xy.ctypes.data, n, ev.ctypes.data, cap, cnt.ctypes.data, rgb.ctypes.data) == 0
    err = np.abs(rgb.astype(np.float64) - want).max(axis=1)
    assert (err <= 1e-4).all(), list(zip(xy.tolist(), err.tolist()))


def test_without_the_f64_boxes_the_f32_ones_are_widened():
    """node_box64 is optional: a description without it still renders (the f32 boxes stand in), with it the clipped-box
    pixels follow the reference."""
    ex = ExampleScene("shapes_zoo", 96, 54)
    full = H.hostsim_render(ex, RtOptions())
    d = ex.scene_desc.contents if hasattr(ex.scene_desc, "contents") else ex.scene_desc
    kept = d.node_box64
    assert kept  # the C++ mirror fills it
    d.node_box64 = None
    try:
        plain = H.hostsim_render(ex, RtOptions())
    finally:
        d.node_box64 = kept
    assert (np.abs(plain - full).max(axis=-1) > 1e-4).mean() <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("jitter", [0, 1])
def test_kernels_follow_the_reference_at_the_former_outliers(jitter):
    from raytracer_amd import Scene
    xy, want = _oracle(jitter)
    ex = ExampleScene("shapes_zoo", W, Hh)
    opt = RtOptions()
    opt.jitter_mode, opt.seed = jitter, 0
    sc = Scene(ex.scene_desc, 0)
    try:
        img, st = sc.render(ex.camera_desc, opt, stats=True)
    finally:
        sc.close()
    err = np.abs(img[xy[:, 1], xy[:, 0]].astype(np.float64) - want).max(axis=1)
    assert (err <= 1e-4).all() and st.exact_nodes > 0, list(zip(xy.tolist(), err.tolist()))


# cover_image_area's last four census pixels (profiles/r03_parity_census.txt): points on the corner line of the room's cube,
# 1e-15 inside the face next to theirs — dark in the reference, whose shadow rays leave through that face at t ~ 1e-15.
ROOM_CORNER = np.array([(1497, 416), (2627, 1126), (2694, 1411), (2558, 1862)], dtype=np.int32)


def test_scalar_spec_follows_the_reference_on_the_rooms_corner_line():
    want = H.oracle_probe("cover_image_area", 3840, 2160, ROOM_CORNER)
    S = H.hostsim()
    S.hostsim_trace_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    ex = ExampleScene("cover_image_area", 3840, 2160)
    opt = RtOptions()
    n, cap = len(ROOM_CORNER), 64
    ev = np.zeros((n, cap, 24))
    cnt = np.zeros(n, dtype=np.int32)
    rgb = np.zeros((n, 3), dtype=np.float32)
    assert S.hostsim_trace_pixels(C.cast(ex.scene_desc, C.c_void_p), C.cast(ex.camera_desc, C.c_void_p), C.cast(C.byref(opt), C.c_void_p),
                                  ROOM_CORNER.ctypes.data, n, ev.ctypes.data, cap, cnt.ctypes.data, rgb.ctypes.data) == 0
    assert np.abs(rgb.astype(np.float64) - want).max() <= 1e-4


@pytest.mark.gpu
def test_kernels_follow_the_reference_on_the_rooms_corner_line():
    from raytracer_amd import Scene
    want = H.oracle_probe("cover_image_area", 3840, 2160, ROOM_CORNER)
    ex = ExampleScene("cover_image_area", 3840, 2160)
    sc = Scene(ex.scene_desc, 0)
    try:
        img, _ = sc.render(ex.camera_desc, RtOptions())
    finally:
        sc.close()
    err = np.abs(img[ROOM_CORNER[:, 1], ROOM_CORNER[:, 0]].astype(np.float64) - want).max(axis=1)
    assert (err <= 1e-4).all(), err


# Pixels of the seeded fuzz families (160x96, centre jitter) that were above 1e-4 at some point of round 3 and what closed
# them (profiles/r03_group_boxes.txt): the whole f64 over point in the records of the second count (field:98, stress:169),
# triangle edges for the shadow rays (mesh:102, mesh:130), path rays that leave a noise point (field:113, field:134,
# stress:39, stress:162, stress:180, stress:189), the container window at grazing hits (stress:41), the triangle a ray starts
# on as its container (mesh:107), a near miss that hid another one (mesh:2).
FUZZ = [("random_field:98", 107, 51), ("random_stress:169", 40, 59), ("random_mesh:102", 88, 54), ("random_mesh:130", 79, 78),
        ("random_field:113", 142, 44), ("random_field:134", 156, 53), ("random_stress:39", 62, 64), ("random_stress:162", 152, 73),
        ("random_stress:180", 51, 71), ("random_stress:189", 67, 25), ("random_stress:41", 18, 64), ("random_mesh:107", 33, 69),
        ("random_mesh:2", 31, 53), ("random_field:218", 42, 15)]  # (field:218: a cylinder's wall point on a rim line, qedge)


@pytest.mark.parametrize("name,x,y", FUZZ)
def test_scalar_spec_follows_the_reference_at_former_fuzz_outliers(name, x, y):
    xy = np.array([(x, y)], dtype=np.int32)
    want = H.oracle_probe(name, 160, 96, xy)
    S = H.hostsim()
    S.hostsim_trace_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    ex = ExampleScene(name, 160, 96)
    opt = RtOptions()
    ev = np.zeros((1, 64, 24))
    cnt = np.zeros(1, dtype=np.int32)
    rgb = np.zeros((1, 3), dtype=np.float32)
    assert S.hostsim_trace_pixels(C.cast(ex.scene_desc, C.c_void_p), C.cast(ex.camera_desc, C.c_void_p), C.cast(C.byref(opt), C.c_void_p),
                                  xy.ctypes.data, 1, ev.ctypes.data, 64, cnt.ctypes.data, rgb.ctypes.data) == 0
    assert np.abs(rgb[0].astype(np.float64) - want[0]).max() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name,x,y", FUZZ)
def test_kernels_follow_the_reference_at_former_fuzz_outliers(name, x, y):
    from raytracer_amd import Scene
    want = H.oracle_probe(name, 160, 96, np.array([(x, y)], dtype=np.int32))
    ex = ExampleScene(name, 160, 96)
    sc = Scene(ex.scene_desc, 0)
    try:
        img, _ = sc.render(ex.camera_desc, RtOptions())
    finally:
        sc.close()
    assert np.abs(img[y, x].astype(np.float64) - want[0]).max() <= 1e-4


@pytest.mark.parametrize("name", H.SCENES)
def test_no_decision_of_the_spec_hangs_on_the_last_ulp_of_rcp_sqrt_rsq(name):
    """The hardware's v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 may differ from the host's exact results by an ulp.  With every such
    result of the scalar spec moved by -1, 0 or +1 ulp (hostsim_set_ulp_noise: a hash of the argument and a seed chooses) the
    golden images still come out within 1e-4 of the oracle: what FP32 cannot decide is a near thing and goes to f64."""
    import os
    S = H.hostsim()
    S.hostsim_set_ulp_noise.argtypes = [C.c_uint]
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "images.npz"))[name]
    h, w, _ = ref.shape
    try:
        for seed in (1, 2):
            S.hostsim_set_ulp_noise(seed)
            img = H.hostsim_render(ExampleScene(name, w, h), RtOptions())
            c = H.compare(img, ref, 1e-4)
            assert c["bad_pixels"] == 0, (seed, c)  # measured 0 on all nine scenes for seeds 1..3
    finally:
        S.hostsim_set_ulp_noise(0)
