"""Algorithmic FLOP model of the render path — implementation-independent.

FMA = 2, div/sqrt/rcp/pow = 1 each, minimal affine arithmetic, and the REFERENCE algorithm: every
ray is tested against every top-level primitive (world.rs:32-41); a Group costs one box test plus,
when entered, its children (group.rs:106-119).  `roofline.achieved` in bench.py is this number
divided by the measured kernel time, so a kernel that culls or exits early is credited with the
work it avoided, exactly like a CPU implementation would be.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

# per-test costs (flop), SURVEY.md §8d
COST_PRIM = {0: 60, 1: 13, 2: 55, 3: 79, 4: 80, 5: 80}  # sphere, plane, cube, cylinder(+caps), triangle, smooth
COST_NODE = 22
COST_SHADOW_SETUP = 13
COST_PREPARE = 80
COST_PATTERN = 42
COST_LIGHT_CELL = 50
COST_LIGHT = 12
COST_SCHLICK_REFRACT_CHILD = 20 + 25 + 6

# FP32 vector peak of MI355X (MI355X_MICROARCH.md: 157.3 TFLOP/s = 256 CU x 4 SIMD-32 x 2 flop x
# 2.4 GHz; a wave64 v_fma_f32 issues in 2 cycles, so the peak needs >= 2 waves per SIMD, no packing)
PEAK_FP32_TFLOPS = 157.3
# wave-instructions per second the chip can issue on its vector pipes (one wave64 VALU op / 2 cycles)
PEAK_VALU_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2


def flat_ray_cost(scene_desc) -> float:
    """flop per ray for a scene without groups: the sum over all primitives."""
    d = scene_desc.contents if hasattr(scene_desc, "contents") else scene_desc
    types = np.ctypeslib.as_array(C.cast(d.prim_type, C.POINTER(C.c_uint8)), shape=(d.n_prims,))
    return float(sum(COST_PRIM[int(t)] for t in types))


def ray_cost(scene_desc, oracle_stats: dict | None = None) -> float:
    """flop per ray in reference semantics.  With groups the number of tests per ray depends on
    the rays, so it is measured once by the oracle walking the reference's group tree
    (oracle_stats = counters of a low-resolution oracle render of the same scene)."""
    d = scene_desc.contents if hasattr(scene_desc, "contents") else scene_desc
    if d.n_nodes == 0 or oracle_stats is None:
        return flat_ray_cost(scene_desc)
    rays = oracle_stats["primary"] + oracle_stats["reflect"] + oracle_stats["refract"] + oracle_stats["shadow"]
    tests = (oracle_stats["tests_sphere"] * 60 + oracle_stats["tests_plane"] * 13 + oracle_stats["tests_cube"] * 55 +
             oracle_stats["tests_cylinder"] * 79 + (oracle_stats["tests_triangle"] + oracle_stats["tests_smooth_triangle"]) * 80 +
             oracle_stats["tests_group_box"] * COST_NODE)
    return tests / max(rays, 1)


def frame_flops(stats: dict, n_lights: int, cost_per_ray: float) -> float:
    """Algorithmic flop of one frame from the ray/work counters of rt_render_stats."""
    rays = stats["primary_rays"] + stats["reflection_rays"] + stats["refraction_rays"] + stats["shadow_rays"]
    return (rays * cost_per_ray + stats["shadow_rays"] * COST_SHADOW_SETUP + stats["shaded_hits"] * COST_PREPARE +
            stats["pattern_lookups"] * COST_PATTERN + stats["light_cells"] * COST_LIGHT_CELL +
            stats["shaded_hits"] * n_lights * COST_LIGHT + stats["container_passes"] * COST_SCHLICK_REFRACT_CHILD)
