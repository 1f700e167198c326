"""rt_render_stats_noalloc — the one entry point that allocates nothing (include/rt_amd.h): image into
the caller's device buffer, the instrumented kernel's scratch (counters + per-tile ray map) in a static
device array of the library.  Checked through the C ABI against rt_render_stats, the entry point it is a
variant of: same image bit for bit, same counters, on two calls in a row, the second with a smaller
input (the scratch keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import ExampleScene, RtOptions, RtStats, Scene, capi, lib

pytestmark = pytest.mark.gpu

FIELDS = ["primary_rays", "reflection_rays", "refraction_rays", "shadow_rays", "shaded_hits", "light_cells",
          "pattern_lookups", "container_passes", "shadow_rays_cast", "pixels"]


def _both(name, w, h, opt):
    import torch

    ex = ExampleScene(name, w, h)
    sc = Scene(ex.scene_desc, 0)
    ref_img, ref_st = sc.render(ex.camera_desc, opt, stats=True)  # rt_render_stats
    out = torch.full((h, w, 3), -7.0, dtype=torch.float32, device="cuda:0")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())  # the fill runs on torch's stream: order it before the render on `s`
    st, tiles = sc.render_stats_noalloc(ex.camera_desc, out.data_ptr(), opt, s.cuda_stream, want_tiles=True)
    got = out.cpu().numpy()
    sc.close()
    return ref_img, ref_st, got, st, tiles


def test_no_alloc_entry_point_matches_rt_render_stats_twice_in_a_row():
    opt = RtOptions()
    opt.jitter_mode, opt.seed = 1, 4
    # first call: a deep glass scene (all counters non-zero); second: a SMALLER frame of another scene
    for name, w, h in (("glass_sphere_checkered_floor_aerial_view", 333, 187), ("cover_image", 96, 54)):
        ref_img, ref_st, got, st, tiles = _both(name, w, h, opt)
        assert np.array_equal(got, ref_img), name                      # bit-exact image
        for f in FIELDS:
            assert getattr(st, f) == getattr(ref_st, f), (name, f)     # same counters
        assert st.kernel_ms > 0
        assert tiles.shape == ((h + 7) // 8, (w + 7) // 8)
        assert int(tiles.sum()) == st.rays_cast()                      # the map adds up to the rays traced
        assert tiles.min() >= 1                                        # every tile was written (primary rays)


def test_no_alloc_entry_point_rejects_what_its_scratch_cannot_hold():
    ex = ExampleScene("test_world", 4096, 4096)  # 512 x 512 = 262144 tiles > RT_NOALLOC_MAX_TILES
    sc = Scene(ex.scene_desc, 0)
    st = RtStats()
    rc = lib().rt_render_stats_noalloc(sc._h, ex.camera_desc, None, C.c_void_p(8), None, C.byref(st), None)
    assert rc == -3 and b"RT_NOALLOC_MAX_TILES" in lib().rt_last_error()  # RT_ERR_CAPACITY, nothing launched
    # ... while row bands of the same frame fit
    import torch
    o = RtOptions()
    o.tile_rows, o.tile_first, o.tile_stride = 8, 1, 4
    rows = sc.output_rows(ex.camera_desc, o)
    out = torch.zeros((rows, 4096, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()  # (the render below goes to the null stream of HIP, not to torch's current stream)
    st2, _ = sc.render_stats_noalloc(ex.camera_desc, out.data_ptr(), o, 0)
    ref, ref_st = sc.render(ex.camera_desc, o, stats=True)
    assert np.array_equal(out.cpu().numpy(), ref) and st2.primary_rays == ref_st.primary_rays == rows * 4096
    sc.close()
