//! Raw declarations of `include/rt_amd.h` (RT_ABI_VERSION 3).  What `bindgen` would emit, written
//! by hand because the header is small and stable.  UNVERIFIED: no Rust toolchain exists in the
//! environment this was written in; the C++ mirror of the same calls is what is compiled and tested.
//! `tests/test_rust_binding.py` parses this file and the header and compares every function (argument
//! names, order, types), struct field, constant and enum value, so the two cannot drift apart unnoticed.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: u32 = 3;
pub const RT_ITEM_NODE: u32 = 0x8000_0000;
pub const RT_OK: c_int = 0;
pub const RT_ERR_BAD_ARGUMENT: c_int = -1;
pub const RT_ERR_UNSUPPORTED: c_int = -2;
pub const RT_ERR_CAPACITY: c_int = -3;
pub const RT_ERR_HIP: c_int = -4;
pub const RT_ERR_NO_DEVICE: c_int = -5;

pub const RT_PRIM_SPHERE: u8 = 0;
pub const RT_PRIM_PLANE: u8 = 1;
pub const RT_PRIM_CUBE: u8 = 2;
pub const RT_PRIM_CYLINDER: u8 = 3;
pub const RT_PRIM_TRIANGLE: u8 = 4;
pub const RT_PRIM_SMOOTH_TRIANGLE: u8 = 5;
pub const RT_PAT_SOLID: u8 = 0;
pub const RT_PAT_STRIPE: u8 = 1;
pub const RT_PAT_GRADIENT: u8 = 2;
pub const RT_PAT_RING: u8 = 3;
pub const RT_PAT_CHECKER: u8 = 4;
pub const RT_LIGHT_POINT: u8 = 0;
pub const RT_LIGHT_AREA: u8 = 1;
pub const RT_JITTER_CENTER: u32 = 0;
pub const RT_JITTER_COUNTER: u32 = 1;

#[repr(C)]
pub struct rt_scene_desc {
    pub abi_version: u32,
    pub n_prims: u32,
    pub prim_type: *const u8,
    pub prim_inv: *const f32,
    pub prim_material: *const u32,
    pub prim_param: *const f32,
    pub n_nodes: u32,
    pub node_min: *const f32,
    pub node_max: *const f32,
    pub n_items: u32,
    pub item_ref: *const u32,
    pub item_next: *const u32,
    pub n_materials: u32,
    pub mat_param: *const f32,
    pub mat_pattern: *const u32,
    pub n_patterns: u32,
    pub pat_type: *const u8,
    pub pat_color_a: *const f32,
    pub pat_color_b: *const f32,
    pub pat_inv: *const f32,
    pub n_lights: u32,
    pub light_type: *const u8,
    pub light_pos: *const f32,
    pub light_uvec: *const f32,
    pub light_vvec: *const f32,
    pub light_usteps: *const u32,
    pub light_vsteps: *const u32,
    pub light_color: *const f32,
    // optional f64 tables: all four or none (null)
    pub prim_inv64: *const f64,
    pub prim_param64: *const f64,
    pub mat_ior64: *const f64,
    pub pat_inv64: *const f64,
    // optional: position (area: corner), uvec, vvec of every light in f64, nine a light (null: the f32 values)
    pub light_geom64: *const f64,
    // optional: the Groups' boxes in f64, min then max, six a node (null: node_min / node_max)
    pub node_box64: *const f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera_desc {
    pub hsize: u32,
    pub vsize: u32,
    pub pixel_size: f32,
    pub half_width: f32,
    pub half_height: f32,
    pub inv: [f32; 12],
    pub has_f64: u32,
    pub pixel_size64: f64,
    pub half_width64: f64,
    pub half_height64: f64,
    pub inv64: [f64; 12],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_options {
    pub max_depth: u32,
    pub jitter_mode: u32,
    pub seed: u32,
    pub tile_rows: u32,
    pub tile_first: u32,
    pub tile_stride: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub primary_rays: u64,
    pub reflection_rays: u64,
    pub refraction_rays: u64,
    pub shadow_rays: u64,
    pub shaded_hits: u64,
    pub light_cells: u64,
    pub pattern_lookups: u64,
    pub container_passes: u64,
    pub shadow_rays_cast: u64,
    pub exact_nodes: u64,
    pub pixels: u64,
    pub kernel_ms: f32,
}

pub enum rt_scene {}

extern "C" {
    pub fn rt_abi_version() -> u32;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count(count_out: *mut c_int) -> c_int;
    pub fn rt_scene_create(desc: *const rt_scene_desc, device: c_int, scene_out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_variant(scene: *const rt_scene, flags_out: *mut u32) -> c_int;
    pub fn rt_scene_read_streams(scene: *const rt_scene, out: *mut c_void, cap: usize, bytes_out: *mut usize) -> c_int;
    pub fn rt_output_rows(camera: *const rt_camera_desc, options: *const rt_options, rows_out: *mut u32) -> c_int;
    pub fn rt_render_device(
        scene: *mut rt_scene,
        camera: *const rt_camera_desc,
        options: *const rt_options,
        rgb_out_device: *mut f32,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_render(
        scene: *mut rt_scene,
        camera: *const rt_camera_desc,
        options: *const rt_options,
        rgb_out_host: *mut f32,
        stats_out: *mut rt_stats,
    ) -> c_int;
    pub fn rt_render_u8(
        scene: *mut rt_scene,
        camera: *const rt_camera_desc,
        options: *const rt_options,
        rgb8_out_host: *mut u8,
        stats_out: *mut rt_stats,
    ) -> c_int;
    pub fn rt_scene_reserve(scene: *mut rt_scene, hsize: u32, rows: u32) -> c_int;
    pub fn rt_render_stats(
        scene: *mut rt_scene,
        camera: *const rt_camera_desc,
        options: *const rt_options,
        rgb_out_host: *mut f32,
        stats_out: *mut rt_stats,
    ) -> c_int;
    // for callers that cannot allocate: scratch in a static device array of the library
    pub fn rt_render_stats_noalloc(
        scene: *mut rt_scene,
        camera: *const rt_camera_desc,
        options: *const rt_options,
        rgb_out_device: *mut f32,
        stream: *mut c_void,
        stats_out: *mut rt_stats,
        tile_rays_out_host: *mut u32,
    ) -> c_int;
    pub fn rt_quantize_u8_device(rgb_device: *const f32, n: usize, out_device: *mut u8, device: c_int, stream: *mut c_void) -> c_int;
    pub fn rt_quantize_u8(rgb_host: *const f32, n: usize, out_host: *mut u8) -> c_int;
    // one frame shared by the GPUs of a node (rt_options.flags = RT_FLAG_GLOBAL_ROWS)
    pub fn rt_frame_create(device: c_int, bytes: usize, frame_out: *mut *mut c_void, handle_out: *mut rt_ipc_handle) -> c_int;
    pub fn rt_frame_open(device: c_int, owner_device: c_int, handle: *const rt_ipc_handle, frame_out: *mut *mut c_void) -> c_int;
    pub fn rt_frame_close(device: c_int, frame: *mut c_void) -> c_int;
    pub fn rt_frame_destroy(device: c_int, frame: *mut c_void) -> c_int;
    pub fn rt_frame_download(device: c_int, frame: *const c_void, host_out: *mut c_void, bytes: usize) -> c_int;
}

pub const RT_FLAG_GLOBAL_ROWS: u32 = 1;
pub const RT_FLAG_F32_ONLY: u32 = 2;
pub const RT_FLAG_DEPTH_ZERO: u32 = 4;
pub const RT_VARIANT_AREA: u32 = 1;
pub const RT_VARIANT_TREE: u32 = 2;
pub const RT_VARIANT_EXOTIC: u32 = 4;
pub const RT_VARIANT_F64: u32 = 8;
pub const RT_VARIANT_LANE: u32 = 16;
pub const RT_VARIANT_SPAWN: u32 = 32;
pub const RT_MAX_DEPTH: u32 = 8;
pub const RT_MAX_LIGHTS: u32 = 256;
pub const RT_MAX_LIGHT_CELLS: u32 = 32768;
pub const RT_NOALLOC_MAX_TILES: u32 = 131072;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_ipc_handle {
    pub bytes: [u8; 64],
}
