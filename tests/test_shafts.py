"""The kernel's shaft tests (rt_math.h "Shafts") must never rule out a primitive that occludes.

rt_kernel.h answers the shadow rays of a (shading point, area light) pair without casting them when
the shaft from the point to the light provably misses every primitive, and skips single records the
same way on flat scenes.  That is only allowed to change the speed, never the image (light.rs:196-208
casts every ray).  The test-only host simulator evaluates the SAME rt_math.h tests at every shading
point of a render and then tests every ruled-out primitive exactly against rays to the jittered point
and to the four corners of every light cell: `violations` must be zero.  The same goes for the
opposite shortcut — a point declared to lie wholly in one primitive's umbra (all cells occluded
without casting): every one of those rays must be occluded in the exact test.  The `-m gpu` parity suite then
shows the images themselves unchanged against the f64 oracle.
"""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from raytracer_amd import ExampleScene, RtCameraDesc, RtOptions, RtSceneDesc

W, HGT = 48, 27


def audit(scene_desc_ptr, camera_desc_ptr, seed=0, depth=3):
    L = H.hostsim()
    L.hostsim_shaft_audit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    cam = C.cast(camera_desc_ptr, C.POINTER(RtCameraDesc)).contents
    opt = RtOptions()
    opt.jitter_mode, opt.seed, opt.max_depth = 1, seed, depth
    img = np.zeros((cam.vsize, cam.hsize, 3), dtype=np.float32)
    out = np.zeros(10, dtype=np.uint64)
    err = C.create_string_buffer(256)
    rc = L.hostsim_shaft_audit(C.cast(scene_desc_ptr, C.c_void_p), C.cast(camera_desc_ptr, C.c_void_p),
                               C.cast(C.byref(opt), C.c_void_p), img.ctypes.data, out.ctypes.data, err, 256)
    assert rc == 0, err.value.decode()
    return dict(pairs=int(out[0]), ruled_out=int(out[1]), open=int(out[2]), violations=int(out[3]), umbra=int(out[4]),
                umbra_violations=int(out[5]), front=int(out[6]), dull=int(out[7]), phong_violations=int(out[8]),
                behind=int(out[9]))


@pytest.mark.parametrize("name", ["checkered_walls_metallic_sphere", "striped_sphere_multiple_lights",
                                  "glass_sphere_checkered_floor_aerial_view", "cover_image_area", "shapes_zoo"])
def test_no_occluder_is_ruled_out_in_the_example_scenes(name):
    ex = ExampleScene(name, W, HGT)
    r = audit(ex.scene_desc, ex.camera_desc, seed=3)
    assert r["pairs"] > 0 and r["violations"] == 0, r
    # the test has teeth only if the shafts do rule things out
    assert r["ruled_out"] > 0 and r["open"] > 0, r
    # the bounds that drop the per-cell gates of the Phong sums: used, and never contradicted by a cell
    assert r["front"] > 0 and r["dull"] > 0 and r["behind"] > 0 and r["phong_violations"] == 0, r


@pytest.mark.parametrize("name", ["striped_sphere_multiple_lights", "glass_sphere_checkered_floor_aerial_view",
                                  "cover_image_area"])
def test_points_declared_fully_occluded_have_no_unoccluded_ray(name):
    """The opposite shortcut (shaft_blocked_by_ball / _plane): a point declared to lie in a primitive's
    umbra gets lit = 0 without casting, so EVERY ray to every cell (jittered point and corners) must be
    occluded in the exact any-hit test."""
    ex = ExampleScene(name, 96, 54)
    r = audit(ex.scene_desc, ex.camera_desc, seed=2, depth=4)
    assert r["umbra"] > 0 and r["umbra_violations"] == 0 and r["violations"] == 0, r


def _affine(rng, big=False):
    """Random invertible object->world transform (rotation, non-uniform scale, sometimes shear)."""
    a = rng.normal(size=(3, 3))
    q, _ = np.linalg.qr(a)
    s = np.diag(rng.uniform(0.2, 2.5 if big else 1.2, 3))
    m = q @ s
    if rng.random() < 0.3:
        sh = np.eye(3)
        sh[0, 1] = rng.uniform(-0.7, 0.7)
        m = m @ sh
    t = np.array([rng.uniform(-4, 4), rng.uniform(0.0, 3.0), rng.uniform(-4, 4)])
    f = np.eye(4)
    f[:3, :3], f[:3, 3] = m, t
    return f


def random_scene(rng, n_objects, keep):
    types, invs, params, mats = [], [], [], []
    # a floor, sometimes a back wall
    types.append(1), invs.append(np.eye(4)), params.append(np.zeros(24)), mats.append(0)
    if rng.random() < 0.5:
        wall = np.eye(4)
        c, s_ = np.cos(np.pi / 2), np.sin(np.pi / 2)
        wall[:3, :3] = np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]])
        wall[2, 3] = -6.0
        types.append(1), invs.append(np.linalg.inv(wall)), params.append(np.zeros(24)), mats.append(1)
    if rng.random() < 0.4:  # a tilted, scaled plane somewhere behind the objects (row norm != 1)
        f = _affine(rng, big=True)
        f[:3, 3] = np.array([rng.uniform(-2, 2), rng.uniform(-1.0, 0.5), rng.uniform(-9, -5)])
        types.append(1), invs.append(np.linalg.inv(f)), params.append(np.zeros(24)), mats.append(3)
    for _ in range(n_objects):
        t = int(rng.choice([0, 0, 2, 2, 3]))
        f = _affine(rng, big=rng.random() < 0.2)
        p = np.zeros(24)
        if t == 3:
            lo = rng.uniform(-1.5, 0.0)
            p[0], p[1], p[2] = lo, lo + rng.uniform(0.3, 2.5), float(rng.random() < 0.6)
        types.append(t), invs.append(np.linalg.inv(f)), params.append(p), mats.append(int(rng.integers(0, 4)))
    n = len(types)
    prim_type = np.array(types, dtype=np.uint8)
    prim_inv = np.stack([m[:3, :].reshape(12) for m in invs]).astype(np.float32)
    prim_param = np.stack(params).astype(np.float32)
    prim_mat = np.array(mats, dtype=np.uint32)
    # ambient diffuse specular shininess ior reflectivity transparency 0
    mat_param = np.array([[0.1, 0.9, 0.3, 50, 1.0, 0.0, 0.0, 0], [0.1, 0.7, 0.0, 10, 1.0, 0.3, 0.0, 0],
                          [0.0, 0.2, 1.0, 200, 1.5, 0.6, 0.8, 0], [0.2, 0.8, 0.9, 300, 1.0, 0.0, 0.0, 0]], dtype=np.float32)
    mat_pat = np.arange(4, dtype=np.uint32)
    pat_type = np.zeros(4, dtype=np.uint8)
    pat_a = rng.uniform(0.2, 1.0, (4, 3)).astype(np.float32)
    pat_b = pat_a.copy()
    pat_inv = np.tile(np.eye(4)[:3, :].reshape(12), (4, 1)).astype(np.float32)
    n_l = int(rng.integers(1, 3))
    l_type = np.ones(n_l, dtype=np.uint8)
    l_pos, l_u, l_v, l_us, l_vs = [], [], [], [], []
    for _ in range(n_l):
        us, vs = int(rng.integers(2, 9)), int(rng.integers(2, 9))
        near = rng.random() < 0.35  # a light that hangs low between the objects
        pos = np.array([rng.uniform(-6, 6), rng.uniform(0.5, 3.0) if near else rng.uniform(4, 12), rng.uniform(-6, 6)])
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        size = rng.uniform(0.5, 6.0, 2)
        l_pos.append(pos), l_u.append(q[:, 0] * size[0] / us), l_v.append(q[:, 1] * size[1] / vs), l_us.append(us), l_vs.append(vs)
    l_pos, l_u, l_v = (np.array(x, dtype=np.float32) for x in (l_pos, l_u, l_v))
    l_us, l_vs = np.array(l_us, dtype=np.uint32), np.array(l_vs, dtype=np.uint32)
    l_col = rng.uniform(0.3, 1.0, (n_l, 3)).astype(np.float32)
    item_ref = np.arange(n, dtype=np.uint32)
    item_next = np.arange(1, n + 1, dtype=np.uint32)
    arrays = [prim_type, prim_inv, prim_param, prim_mat, mat_param, mat_pat, pat_type, pat_a, pat_b, pat_inv, l_type, l_pos,
              l_u, l_v, l_us, l_vs, l_col, item_ref, item_next]
    keep += arrays
    d = RtSceneDesc()
    d.abi_version = 3
    d.n_prims, d.prim_type, d.prim_inv = n, prim_type.ctypes.data, prim_inv.ctypes.data
    d.prim_material, d.prim_param = prim_mat.ctypes.data, prim_param.ctypes.data
    d.n_nodes, d.node_min, d.node_max = 0, None, None
    d.n_items, d.item_ref, d.item_next = n, item_ref.ctypes.data, item_next.ctypes.data
    d.n_materials, d.mat_param, d.mat_pattern = 4, mat_param.ctypes.data, mat_pat.ctypes.data
    d.n_patterns, d.pat_type, d.pat_color_a = 4, pat_type.ctypes.data, pat_a.ctypes.data
    d.pat_color_b, d.pat_inv = pat_b.ctypes.data, pat_inv.ctypes.data
    d.n_lights, d.light_type, d.light_pos = n_l, l_type.ctypes.data, l_pos.ctypes.data
    d.light_uvec, d.light_vvec = l_u.ctypes.data, l_v.ctypes.data
    d.light_usteps, d.light_vsteps, d.light_color = l_us.ctypes.data, l_vs.ctypes.data, l_col.ctypes.data
    return d


@pytest.mark.parametrize("n_objects", [1, 3, 6, 12, 30])  # <= 8 primitives: flat stream; more: hierarchy
def test_no_occluder_is_ruled_out_in_random_scenes(n_objects):
    cam = ExampleScene("striped_sphere_multiple_lights", 40, 24)  # eye (0, 3, 5) looking at (0, 1, 0)
    tot = dict(pairs=0, ruled_out=0, open=0, violations=0, umbra=0, umbra_violations=0, front=0, dull=0, phong_violations=0, behind=0)
    for seed in range(16):
        rng = np.random.default_rng(1000 * n_objects + seed)
        keep = []
        d = random_scene(rng, n_objects, keep)
        r = audit(C.pointer(d), cam.camera_desc, seed=seed)
        for k in tot:
            tot[k] += r[k]
    assert tot["pairs"] > 1000 and tot["violations"] == 0, tot
    assert tot["ruled_out"] > 0 and tot["umbra"] > 0 and tot["umbra_violations"] == 0, tot
    assert tot["front"] > 0 and tot["dull"] > 0 and tot["behind"] > 0 and tot["phong_violations"] == 0, tot


def many_planes_scene(rng, keep, n_planes=70):
    """Flat stream longer than the 64-bit record mask: a floor, n_planes - 1 more planes underneath it
    (never visible, but in the stream), and two spheres (<= 2 bounded primitives: no hierarchy)."""
    d = random_scene(rng, 2, keep)
    n0 = d.n_prims
    types = np.ctypeslib.as_array(C.cast(d.prim_type, C.POINTER(C.c_uint8)), (n0,)).copy()
    invs = np.ctypeslib.as_array(C.cast(d.prim_inv, C.POINTER(C.c_float)), (n0, 12)).copy()
    mats = np.ctypeslib.as_array(C.cast(d.prim_material, C.POINTER(C.c_uint32)), (n0,)).copy()
    keepi = [i for i in range(n0) if types[i] != 1 or i == 0]  # drop a possible wall, keep floor + spheres
    types, invs, mats = types[keepi], invs[keepi], mats[keepi]
    extra = []
    for k in range(1, n_planes):
        m = np.eye(4)
        m[1, 3] = float(k)  # inverse of translation(0, -k, 0)
        extra.append(m[:3, :].reshape(12))
    order = [0] + list(range(len(types), len(types) + len(extra))) + list(range(1, len(types)))  # spheres LAST (records >= 63)
    types = np.concatenate([types, np.ones(len(extra), dtype=np.uint8)])[order]
    invs = np.concatenate([invs, np.array(extra, dtype=np.float32)])[order]
    mats = np.concatenate([mats, np.zeros(len(extra), dtype=np.uint32)])[order]
    n = len(types)
    params = np.zeros((n, 24), dtype=np.float32)
    ref, nxt = np.arange(n, dtype=np.uint32), np.arange(1, n + 1, dtype=np.uint32)
    types, invs, mats = np.ascontiguousarray(types), np.ascontiguousarray(invs), np.ascontiguousarray(mats)
    keep += [types, invs, mats, params, ref, nxt]
    d.n_prims, d.prim_type, d.prim_inv = n, types.ctypes.data, invs.ctypes.data
    d.prim_material, d.prim_param = mats.ctypes.data, params.ctypes.data
    d.n_items, d.item_ref, d.item_next = n, ref.ctypes.data, nxt.ctypes.data
    return d


def test_no_occluder_is_ruled_out_on_a_flat_stream_longer_than_the_mask():
    cam = ExampleScene("striped_sphere_multiple_lights", 40, 24)
    keep = []
    d = many_planes_scene(np.random.default_rng(7), keep)
    r = audit(C.pointer(d), cam.camera_desc, seed=1)
    assert r["pairs"] > 100 and r["violations"] == 0 and r["ruled_out"] > 0, r
