"""raytracer_amd — MI355X (gfx950) render path for regexPattern/raytracer.

The product is the C-ABI shared library ``_lib/librt_amd.so`` (hand-written HIP kernels, see
``csrc/rt_amd.hip`` and ``include/rt_amd.h``) plus the C++ host API that mirrors the reference's
scene builders (``include/raytracer_amd/raytracer.hpp``).  This Python package is only the thin
ctypes binding used by ``bench.py``, ``__graft_entry__.py`` and the tests; it contains no renderer
and no CPU fallback: if the HIP library is missing, importing :mod:`raytracer_amd.capi` raises.
"""
from .capi import (  # noqa: F401
    RT_FLAG_GLOBAL_ROWS,
    ExampleScene,
    RtCameraDesc,
    RtError,
    RtOptions,
    RtSceneDesc,
    RtStats,
    Scene,
    SharedFrame,
    band_rows,
    example_names,
    lib,
    lib_path,
    quantize_u8,
    scenes_lib,
)

__all__ = [
    "RT_FLAG_GLOBAL_ROWS", "ExampleScene", "RtCameraDesc", "RtError", "RtOptions", "RtSceneDesc", "RtStats", "Scene",
    "SharedFrame", "band_rows", "example_names", "lib", "lib_path", "quantize_u8", "scenes_lib",
]
