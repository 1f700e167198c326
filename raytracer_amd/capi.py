"""ctypes binding of include/rt_amd.h (librt_amd.so) and of the example-scene library.

No compute happens in Python and nothing here falls back to a CPU renderer: every render goes
through ``rt_render`` / ``rt_render_device`` (HIP kernels).  A missing library is a hard error.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "_lib")


class RtError(RuntimeError):
    """A C-ABI call returned a negative rt_status."""


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_prims", C.c_uint32), ("prim_type", C.c_void_p), ("prim_inv", C.c_void_p),
        ("prim_material", C.c_void_p), ("prim_param", C.c_void_p),
        ("n_nodes", C.c_uint32), ("node_min", C.c_void_p), ("node_max", C.c_void_p),
        ("n_items", C.c_uint32), ("item_ref", C.c_void_p), ("item_next", C.c_void_p),
        ("n_materials", C.c_uint32), ("mat_param", C.c_void_p), ("mat_pattern", C.c_void_p),
        ("n_patterns", C.c_uint32), ("pat_type", C.c_void_p), ("pat_color_a", C.c_void_p),
        ("pat_color_b", C.c_void_p), ("pat_inv", C.c_void_p),
        ("n_lights", C.c_uint32), ("light_type", C.c_void_p), ("light_pos", C.c_void_p),
        ("light_uvec", C.c_void_p), ("light_vvec", C.c_void_p), ("light_usteps", C.c_void_p),
        ("light_vsteps", C.c_void_p), ("light_color", C.c_void_p),
        ("prim_inv64", C.c_void_p), ("prim_param64", C.c_void_p), ("mat_ior64", C.c_void_p), ("pat_inv64", C.c_void_p),
        ("light_geom64", C.c_void_p), ("node_box64", C.c_void_p),
    ]


class RtCameraDesc(C.Structure):
    _fields_ = [
        ("hsize", C.c_uint32), ("vsize", C.c_uint32),
        ("pixel_size", C.c_float), ("half_width", C.c_float), ("half_height", C.c_float),
        ("inv", C.c_float * 12),
        ("has_f64", C.c_uint32),
        ("pixel_size64", C.c_double), ("half_width64", C.c_double), ("half_height64", C.c_double),
        ("inv64", C.c_double * 12),
    ]


class RtOptions(C.Structure):
    _fields_ = [
        ("max_depth", C.c_uint32), ("jitter_mode", C.c_uint32), ("seed", C.c_uint32),
        ("tile_rows", C.c_uint32), ("tile_first", C.c_uint32), ("tile_stride", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("reflection_rays", C.c_uint64), ("refraction_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64), ("shaded_hits", C.c_uint64), ("light_cells", C.c_uint64),
        ("pattern_lookups", C.c_uint64), ("container_passes", C.c_uint64), ("shadow_rays_cast", C.c_uint64),
        ("exact_nodes", C.c_uint64), ("pixels", C.c_uint64),
        ("kernel_ms", C.c_float),
    ]

    def total_rays(self) -> int:
        """Rays of the reference's accounting (shadow rays answered without casting included)."""
        return int(self.primary_rays + self.reflection_rays + self.refraction_rays + self.shadow_rays)

    def rays_cast(self) -> int:
        """Rays the kernel really traced against the scene."""
        return int(self.primary_rays + self.reflection_rays + self.refraction_rays + self.shadow_rays_cast)

    def as_dict(self) -> dict:
        return {name: (float(getattr(self, name)) if name == "kernel_ms" else int(getattr(self, name)))
                for name, _ in self._fields_}


# Every symbol include/rt_amd.h declares (tests check that the library exports all of them).
RT_SYMBOLS = [
    "rt_abi_version", "rt_last_error", "rt_device_count", "rt_scene_create", "rt_scene_destroy", "rt_scene_variant", "rt_scene_read_streams",
    "rt_output_rows", "rt_render_device", "rt_render", "rt_render_u8", "rt_scene_reserve", "rt_render_stats", "rt_render_stats_noalloc",
    "rt_quantize_u8_device",
    "rt_quantize_u8", "rt_frame_create", "rt_frame_open", "rt_frame_close", "rt_frame_destroy", "rt_frame_download",
]


def lib_path() -> str:
    # RT_AMD_LIB: another BUILD of the same HIP library (kernel timing experiments); never a fallback
    return os.environ.get("RT_AMD_LIB") or os.path.join(_LIBDIR, "librt_amd.so")


_lib: Optional[C.CDLL] = None
_scenes: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """The HIP product library.  Raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RtError(f"{path} is missing: build it with `make product` (hipcc --offload-arch=gfx950); "
                          "raytracer_amd has no CPU path")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.rt_abi_version.restype = C.c_uint32
        L.rt_last_error.restype = C.c_char_p
        L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.rt_scene_create.argtypes = [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_scene_destroy.restype = None
        L.rt_scene_variant.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.rt_scene_read_streams.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.rt_output_rows.argtypes = [C.POINTER(RtCameraDesc), C.POINTER(RtOptions), C.POINTER(C.c_uint32)]
        L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtOptions), C.c_void_p, C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtOptions), C.c_void_p, C.POINTER(RtStats)]
        L.rt_render_u8.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtOptions), C.c_void_p, C.POINTER(RtStats)]
        L.rt_scene_reserve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.rt_render_stats.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtOptions), C.c_void_p, C.POINTER(RtStats)]
        L.rt_render_stats_noalloc.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtOptions), C.c_void_p, C.c_void_p,
                                              C.POINTER(RtStats), C.c_void_p]
        L.rt_quantize_u8_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
        L.rt_quantize_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.rt_frame_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p]
        L.rt_frame_open.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_frame_close.argtypes = [C.c_int, C.c_void_p]
        L.rt_frame_destroy.argtypes = [C.c_int, C.c_void_p]
        L.rt_frame_download.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def scenes_lib() -> C.CDLL:
    """Example scenes built through the C++ host API (include/raytracer_amd/raytracer.hpp)."""
    global _scenes
    if _scenes is None:
        lib()  # dependency, and the loud failure if the product is missing
        path = os.path.join(_LIBDIR, "librt_scenes.so")
        if not os.path.exists(path):
            raise RtError(f"{path} is missing: build it with `make product`")
        S = C.CDLL(path)
        S.rtx_last_error.restype = C.c_char_p
        S.rtx_scene_build.argtypes = [C.c_char_p, C.c_int, C.c_int]
        S.rtx_scene_build.restype = C.c_void_p
        S.rtx_scene_desc.argtypes = [C.c_void_p]
        S.rtx_scene_desc.restype = C.POINTER(RtSceneDesc)
        S.rtx_camera_desc.argtypes = [C.c_void_p]
        S.rtx_camera_desc.restype = C.POINTER(RtCameraDesc)
        S.rtx_scene_free.argtypes = [C.c_void_p]
        S.rtx_scene_free.restype = None
        S.rtx_render_example.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(RtOptions), C.c_int, C.c_void_p,
                                         C.POINTER(RtStats)]
        S.rtx_render_example_u8.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(RtOptions), C.c_int, C.c_void_p,
                                            C.POINTER(RtStats)]
        S.rtx_scene_name.restype = C.c_char_p
        S.rtx_render_example_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(RtOptions), C.c_int, C.c_char_p]
        S.rtx_png_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _scenes = S
    return _scenes


def _check(rc: int) -> None:
    if rc != 0:
        raise RtError(f"rt_status {rc}: {lib().rt_last_error().decode(errors='replace')}")


def example_names() -> List[str]:
    S = scenes_lib()
    return [S.rtx_scene_name(i).decode() for i in range(S.rtx_scene_count())]


class ExampleScene:
    """A flattened example scene (rt_scene_desc + rt_camera_desc) built by the C++ host API."""

    def __init__(self, name: str, width: int, height: int):
        S = scenes_lib()
        self.name, self.width, self.height = name, width, height
        self._h = S.rtx_scene_build(name.encode(), width, height)
        if not self._h:
            raise RtError(S.rtx_last_error().decode(errors="replace"))
        self.scene_desc = S.rtx_scene_desc(self._h)
        self.camera_desc = S.rtx_camera_desc(self._h)

    def close(self) -> None:
        if self._h:
            scenes_lib().rtx_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """Device-resident scene (rt_scene*) on one GPU."""

    def __init__(self, scene_desc, device: int = 0):
        self._h = C.c_void_p()
        self.device = device
        _check(lib().rt_scene_create(scene_desc, device, C.byref(self._h)))

    def kernel_name(self, camera_desc, options: Optional[RtOptions] = None, stats: bool = False) -> str:
        """Name of the render kernel instantiation a call with this camera and these options launches."""
        v = C.c_uint32()
        _check(lib().rt_scene_variant(self._h, C.byref(v)))
        cam = camera_desc.contents if hasattr(camera_desc, "contents") else camera_desc
        f64 = bool(v.value & 8) and bool(cam.has_f64) and not (options is not None and options.flags & RT_FLAG_F32_ONLY)
        b = lambda x: "true" if x else "false"  # noqa: E731
        return f"render_kernel<{b(stats)}, {b(v.value & 4)}, {b(v.value & 2)}, {b(v.value & 1)}, {b(f64)}, {b(v.value & 16)}, {b(v.value & 32)}>"

    def read_streams(self) -> np.ndarray:
        """rt_scene_read_streams: the walk streams as the device holds them, [streams * records, 8] uint32 words."""
        n = C.c_size_t()
        _check(lib().rt_scene_read_streams(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value // 32, 8), dtype=np.uint32)
        _check(lib().rt_scene_read_streams(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(n)))
        return out

    def output_rows(self, camera_desc, options: Optional[RtOptions] = None) -> int:
        rows = C.c_uint32()
        _check(lib().rt_output_rows(camera_desc, C.byref(options) if options is not None else None, C.byref(rows)))
        return rows.value

    def render(self, camera_desc, options: Optional[RtOptions] = None, stats: bool = False):
        """Camera::render into host memory.  Returns (image[rows, W, 3] float32, RtStats)."""
        cam = camera_desc.contents if hasattr(camera_desc, "contents") else camera_desc
        rows = self.output_rows(camera_desc, options)
        img = np.empty((rows, cam.hsize, 3), dtype=np.float32)
        st = RtStats()
        fn = lib().rt_render_stats if stats else lib().rt_render
        _check(fn(self._h, camera_desc, C.byref(options) if options is not None else None,
                  img.ctypes.data_as(C.c_void_p), C.byref(st)))
        return img, st

    def render_u8(self, camera_desc, options: Optional[RtOptions] = None):
        """Camera::render + Canvas::to_image in one call (rt_render_u8).  Returns (image[rows, W, 3] uint8, RtStats)."""
        cam = camera_desc.contents if hasattr(camera_desc, "contents") else camera_desc
        rows = self.output_rows(camera_desc, options)
        out = np.zeros((rows, cam.hsize, 3), dtype=np.uint8)
        st = RtStats()
        _check(lib().rt_render_u8(self._h, camera_desc, C.byref(options) if options is not None else None,
                                  out.ctypes.data_as(C.c_void_p), C.byref(st)))
        return out, st

    def reserve(self, hsize: int, rows: int) -> None:
        """rt_scene_reserve: set the scene up for frames of up to hsize x rows (no allocation in later calls)."""
        _check(lib().rt_scene_reserve(self._h, hsize, rows))

    def render_device(self, camera_desc, out_ptr: int, options: Optional[RtOptions] = None, stream: int = 0) -> None:
        """Enqueue Camera::render into DEVICE memory at out_ptr on `stream` (no allocation, no wait)."""
        _check(lib().rt_render_device(self._h, camera_desc, C.byref(options) if options is not None else None,
                                      C.c_void_p(out_ptr), C.c_void_p(stream)))

    def render_stats_noalloc(self, camera_desc, out_ptr: int, options: Optional[RtOptions] = None, stream: int = 0,
                             want_tiles: bool = False):
        """rt_render_stats_noalloc: image into DEVICE memory at out_ptr, counters (and the per-tile ray
        map) through the library's static scratch; allocates nothing on the device.  Returns (RtStats,
        tile map [tile rows, tiles per row] uint32 or None)."""
        cam = camera_desc.contents if hasattr(camera_desc, "contents") else camera_desc
        st = RtStats()
        tiles = None
        if want_tiles:
            rows = self.output_rows(camera_desc, options)
            tiles = np.zeros(((rows + 7) // 8, (cam.hsize + 7) // 8), dtype=np.uint32)
        _check(lib().rt_render_stats_noalloc(self._h, camera_desc, C.byref(options) if options is not None else None,
                                             C.c_void_p(out_ptr), C.c_void_p(stream), C.byref(st),
                                             tiles.ctypes.data_as(C.c_void_p) if tiles is not None else None))
        return st, tiles

    def close(self) -> None:
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def band_rows(vsize: int, tile_rows: int, tile_first: int = 0, tile_stride: int = 1) -> np.ndarray:
    """Global image row of every packed output row of a call with these rt_options tile fields
    (index arithmetic only — the rule documented on rt_options in include/rt_amd.h)."""
    if tile_rows == 0:
        return np.arange(vsize, dtype=np.int64)
    rows = []
    b = tile_first
    while b * tile_rows < vsize:
        rows.extend(range(b * tile_rows, min((b + 1) * tile_rows, vsize)))
        b += tile_stride
    return np.asarray(rows, dtype=np.int64)


RT_FLAG_GLOBAL_ROWS = 1
RT_FLAG_F32_ONLY = 2
RT_FLAG_DEPTH_ZERO = 4
RT_NOALLOC_MAX_TILES = 131072
RT_ABI_VERSION = 3


class SharedFrame:
    """One full frame (height x width x 3 f32) in the HBM of ONE GPU that the other ranks of a node
    write their row bands into directly (rt_frame_* in include/rt_amd.h; rt_render_device with
    RT_FLAG_GLOBAL_ROWS).  The owner creates it and hands `handle` (64 bytes) to the other processes,
    which open it; nothing here is a collective."""

    def __init__(self, height: int, width: int, device: int, owner_device: Optional[int] = None,
                 handle: Optional[bytes] = None):
        self.height, self.width, self.device = height, width, device
        self.nbytes = height * width * 3 * 4
        self.ptr = C.c_void_p()
        self.owner = handle is None
        if self.owner:
            h = (C.c_ubyte * 64)()
            _check(lib().rt_frame_create(device, self.nbytes, C.byref(self.ptr), C.cast(h, C.c_void_p)))
            self.handle = bytes(h)
        else:
            assert owner_device is not None and len(handle) == 64
            self.handle = handle
            h = (C.c_ubyte * 64).from_buffer_copy(handle)
            _check(lib().rt_frame_open(device, owner_device, C.cast(h, C.c_void_p), C.byref(self.ptr)))

    def download(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        _check(lib().rt_frame_download(self.device, self.ptr, out.ctypes.data_as(C.c_void_p), self.nbytes))
        return out

    def close(self) -> None:
        if self.ptr:
            L = lib()
            _check(L.rt_frame_destroy(self.device, self.ptr) if self.owner else L.rt_frame_close(self.device, self.ptr))
            self.ptr = C.c_void_p()


def quantize_u8(rgb: np.ndarray) -> np.ndarray:
    """Canvas::to_image on the GPU (rt_quantize_u8)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty(rgb.shape, dtype=np.uint8)
    _check(lib().rt_quantize_u8(rgb.ctypes.data_as(C.c_void_p), rgb.size, out.ctypes.data_as(C.c_void_p)))
    return out
