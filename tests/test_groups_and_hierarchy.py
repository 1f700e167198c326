"""Groups as gates + the device's own hierarchy (raytracer_amd/csrc/rt_pack.h), on the CPU through the
test-only host simulator (the `-m gpu` suite repeats the gate cases on the real kernel).

Reference semantics under test (group.rs:106-119, bounding_box.rs:81-84): a Group's children are
tested iff the ray's LINE hits the Group's box, `tmin < tmax` STRICT — so a zero-thickness box hides
its children, and a box elsewhere hides them too, whatever the children's own
geometry says.  The device walks its own hierarchy and must still honour exactly that.
"""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from raytracer_amd import ExampleScene, RtOptions, RtSceneDesc

NODE = 0x80000000


def grouped_desc(ex, first, box_min, box_max, keep):
    """Copy of ex's flat description with primitives [first, n) wrapped into ONE Group with the
    given box.  `keep` receives the numpy arrays the description points into."""
    d = RtSceneDesc()
    C.memmove(C.byref(d), ex.scene_desc, C.sizeof(RtSceneDesc))
    n = d.n_prims
    assert d.n_nodes == 0 and d.n_items == n
    ref = np.array(list(range(first)) + [NODE | 0] + list(range(first, n)), dtype=np.uint32)
    nxt = np.arange(1, n + 2, dtype=np.uint32)
    nxt[first] = n + 1  # the Group's subtree runs to the end
    mn = np.asarray(box_min, dtype=np.float32)
    mx = np.asarray(box_max, dtype=np.float32)
    keep += [ref, nxt, mn, mx]
    d.n_items, d.item_ref, d.item_next = n + 1, ref.ctypes.data, nxt.ctypes.data
    d.n_nodes, d.node_min, d.node_max = 1, mn.ctypes.data, mx.ctypes.data
    return d


class Wrapped:  # what hostsim_render expects: .scene_desc / .camera_desc pointers
    def __init__(self, ex, desc):
        self._ex, self._d = ex, desc
        self.scene_desc = C.pointer(desc)
        self.camera_desc = ex.camera_desc


def truncated(ex, n_items):
    d = RtSceneDesc()
    C.memmove(C.byref(d), ex.scene_desc, C.sizeof(RtSceneDesc))
    d.n_items = n_items  # the other primitives are simply not referenced
    return d


W, HGT = 64, 36


@pytest.fixture(scope="module")
def cover():
    return ExampleScene("cover_image", W, HGT)


def test_group_with_an_enclosing_box_changes_nothing(cover):
    keep = []
    d = grouped_desc(cover, 2, [-2000, -2000, -2000], [2000, 2000, 2000], keep)
    plain = H.hostsim_render(cover, RtOptions())
    gated = H.hostsim_render(Wrapped(cover, d), RtOptions())
    assert np.array_equal(plain, gated)


@pytest.mark.parametrize("box", [([-2000, 0.5, -2000], [2000, 0.5, 2000]),      # zero thickness: never entered
                                 ([5000, 5000, 5000], [5001, 5001, 5001])])  # elsewhere
def test_group_box_hides_its_children(cover, box):
    keep = []
    d = grouped_desc(cover, 2, box[0], box[1], keep)
    gated = H.hostsim_render(Wrapped(cover, d), RtOptions())
    only_two = H.hostsim_render(Wrapped(cover, truncated(cover, 2)), RtOptions())
    plain = H.hostsim_render(cover, RtOptions())
    assert np.array_equal(gated, only_two)
    assert np.abs(gated - plain).max() > 0.1  # the cubes and the sphere really are gone


def test_hierarchy_is_invisible_in_the_image():
    """19 primitives > the flat limit: the device stream of cover_image is a hierarchy.  Its image
    equals the f64 oracle's (which tests every object, world.rs:32-41) at the deterministic bar."""
    ex = ExampleScene("cover_image", 96, 54)
    sim = H.hostsim_render(ex, RtOptions())
    ref, _, _ = H.oracle_render("cover_image", 96, 54, threads=8)
    assert H.compare(sim, ref, 1e-4)["bad_pixels"] <= 2


def test_tie_order_survives_the_hierarchy():
    """Two coincident unit spheres with different materials among > 8 primitives: every hit is an
    exact tie, and the one listed FIRST in world.objects must win (Intersection::sort is stable,
    intersection.rs:105-120) although the hierarchy does not visit primitives in that order."""
    ex = ExampleScene("cover_image", 48, 27)
    d = RtSceneDesc()
    C.memmove(C.byref(d), ex.scene_desc, C.sizeof(RtSceneDesc))
    n = d.n_prims
    inv = np.ctypeslib.as_array(C.cast(d.prim_inv, C.POINTER(C.c_float)), shape=(n, 12)).copy()
    mat = np.ctypeslib.as_array(C.cast(d.prim_material, C.POINTER(C.c_uint32)), shape=(n,)).copy()
    typ = np.ctypeslib.as_array(C.cast(d.prim_type, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    cubes = [i for i in range(n) if typ[i] == 2]
    a, b = cubes[0], cubes[-1]
    inv[b] = inv[a]  # cube b now coincides with cube a
    imgs = {}
    for first_mat, tag in ((mat[a], "a_first"), (mat[b], "b_first")):
        m2 = mat.copy()
        m2[a] = first_mat
        m2[b] = mat[b] if tag == "a_first" else mat[a]
        d.prim_inv, d.prim_material = inv.ctypes.data, m2.ctypes.data
        imgs[tag] = H.hostsim_render(Wrapped(ex, d), RtOptions())
    if mat[a] != mat[b]:
        # swapping which material sits on the EARLIER primitive changes the picture: the earlier one wins
        assert np.abs(imgs["a_first"] - imgs["b_first"]).max() > 1e-3
    # and the earlier one winning means: identical to the scene where the later duplicate is absent
    d.prim_material = mat.ctypes.data
    with_dup = H.hostsim_render(Wrapped(ex, d), RtOptions())
    ref = np.array([i for i in range(n) if i != b], dtype=np.uint32)
    nxt = np.arange(1, n, dtype=np.uint32)
    d.n_items, d.item_ref, d.item_next = n - 1, ref.ctypes.data, nxt.ctypes.data
    without = H.hostsim_render(Wrapped(ex, d), RtOptions())
    # shadow rays and n1/n2 see the duplicate too, but an opaque duplicate of an opaque cube adds nothing
    assert np.abs(with_dup - without).max() <= 1e-6
