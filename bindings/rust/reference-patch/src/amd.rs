//! `src/amd.rs` for regexPattern/raytracer — the reference-side half of the MI355X render path.
//!
//! UNVERIFIED SOURCE: there is no Rust toolchain in the environment this was written in.  The same
//! flattening logic is compiled and tested in C++ (`include/raytracer_amd/raytracer.hpp: flatten`),
//! against the same C ABI (`include/rt_amd.h`).
//!
//! How to apply (see ../README.md):
//!  * `Cargo.toml`: `[features] amd = ["dep:raytracer-amd-sys"]`,
//!    `[dependencies] raytracer-amd-sys = { path = "…/bindings/rust/raytracer-amd-sys", optional = true }`
//!  * `src/lib.rs`: `#[cfg(feature = "amd")] mod amd;`
//!  * `src/camera.rs`: at the top of `Camera::render`, `#[cfg(feature = "amd")] return crate::amd::render(self, world);`
//!  * visibility: this module reads fields that are private today.  They become `pub(crate)`:
//!    `Camera::{hsize, vsize, pixel_size, half_width, half_height, transform_inverse}` (camera.rs:64-73),
//!    `Pattern3DSpec::{color_a, color_b, transform_inverse}` (pattern.rs:39-44),
//!    `AreaLight::{corner, uvec, usteps, vvec, vsteps, intensity}` (light.rs:81-89),
//!    `Triangle::{e0, e1, normal}` (shape/triangle.rs:48-56), `Transform.0` (transform.rs:63),
//!    `Group::bounding_box()` (shape/group.rs:233).
//!
//! Nothing else in the crate changes: examples build `World`, `Camera`, `Shape`, `Group`, `Material`,
//! `Pattern3D` and `Light` exactly as before and call `camera.render(&world)`.
use std::ffi::CStr;

use raytracer_amd_sys as sys;

use crate::{
    camera::Camera,
    canvas::Canvas,
    color::Color,
    light::Light,
    material::Material,
    pattern::Pattern3D,
    shape::Shape,
    transform::Transform,
    world::World,
};

/// The flat structure-of-arrays tables of `rt_scene_desc`, owned.
#[derive(Default)]
pub(crate) struct FlatScene {
    prim_type: Vec<u8>,
    prim_inv: Vec<f32>,
    prim_material: Vec<u32>,
    prim_param: Vec<f32>,
    node_min: Vec<f32>,
    node_max: Vec<f32>,
    item_ref: Vec<u32>,
    item_next: Vec<u32>,
    mat_param: Vec<f32>,
    mat_pattern: Vec<u32>,
    materials: Vec<Material>, // for de-duplication (Material: PartialEq)
    pat_type: Vec<u8>,
    pat_color_a: Vec<f32>,
    pat_color_b: Vec<f32>,
    pat_inv: Vec<f32>,
    patterns: Vec<Pattern3D>,
    light_type: Vec<u8>,
    light_pos: Vec<f32>,
    light_uvec: Vec<f32>,
    light_vvec: Vec<f32>,
    light_usteps: Vec<u32>,
    light_vsteps: Vec<u32>,
    light_color: Vec<f32>,
    // the same tables in the crate's own f64 (rt_scene_desc.*64): the kernels work the hits that decide
    // something out again with these, so pattern boundaries and silhouettes follow this crate's arithmetic
    prim_inv64: Vec<f64>,
    prim_param64: Vec<f64>,
    mat_ior64: Vec<f64>,
    pat_inv64: Vec<f64>,
    light_geom64: Vec<f64>, // position (area: corner), uvec, vvec: nine a light
    node_box64: Vec<f64>,   // a Group's box, min then max: six a node
}

/// First three rows of an affine 4x4, row-major, f64 -> f32 (rounded once).
pub(crate) fn rows3x4(t: &Transform) -> [f32; 12] {
    let m = &t.0 .0;
    let mut out = [0f32; 12];
    for r in 0..3 {
        for c in 0..4 {
            out[4 * r + c] = m[r][c] as f32;
        }
    }
    out
}

/// The same rows, unrounded.
pub(crate) fn rows3x4_f64(t: &Transform) -> [f64; 12] {
    let m = &t.0 .0;
    let mut out = [0f64; 12];
    for r in 0..3 {
        for c in 0..4 {
            out[4 * r + c] = m[r][c];
        }
    }
    out
}

fn push_color(v: &mut Vec<f32>, c: Color) {
    v.extend_from_slice(&[c.red as f32, c.green as f32, c.blue as f32]);
}

impl FlatScene {
    pub(crate) fn from_world(world: &World) -> Self {
        let mut flat = Self::default();
        for object in &world.objects {
            flat.push_shape(object);
        }
        for light in &world.lights {
            flat.push_light(light);
        }
        flat
    }

    fn pattern_index(&mut self, pattern: &Pattern3D) -> u32 {
        if let Some(i) = self.patterns.iter().position(|p| p == pattern) {
            return i as u32;
        }
        let identity = Transform::default();
        let (tag, a, b, inv) = match pattern {
            Pattern3D::Solid(c) => (sys::RT_PAT_SOLID, *c, *c, identity),
            Pattern3D::Stripe(s) => (sys::RT_PAT_STRIPE, s.color_a, s.color_b, s.transform_inverse),
            Pattern3D::Gradient(s) => (sys::RT_PAT_GRADIENT, s.color_a, s.color_b, s.transform_inverse),
            Pattern3D::Ring(s) => (sys::RT_PAT_RING, s.color_a, s.color_b, s.transform_inverse),
            Pattern3D::Checker(s) => (sys::RT_PAT_CHECKER, s.color_a, s.color_b, s.transform_inverse),
        };
        self.pat_type.push(tag);
        push_color(&mut self.pat_color_a, a);
        push_color(&mut self.pat_color_b, b);
        self.pat_inv.extend_from_slice(&rows3x4(&inv));
        self.pat_inv64.extend_from_slice(&rows3x4_f64(&inv));
        self.patterns.push(pattern.clone());
        (self.patterns.len() - 1) as u32
    }

    fn material_index(&mut self, material: &Material) -> u32 {
        if let Some(i) = self.materials.iter().position(|m| m == material) {
            return i as u32;
        }
        let pattern = self.pattern_index(&material.pattern);
        self.mat_param.extend_from_slice(&[
            material.ambient as f32,
            material.diffuse as f32,
            material.specular as f32,
            material.shininess as f32,
            material.index_of_refraction as f32,
            material.reflectivity as f32,
            material.transparency as f32,
            0.0,
        ]);
        self.mat_pattern.push(pattern);
        self.mat_ior64.push(material.index_of_refraction);
        self.materials.push(material.clone());
        (self.materials.len() - 1) as u32
    }

    /// One item per shape, depth-first pre-order; a Group's subtree is closed by its `item_next`.
    fn push_shape(&mut self, shape: &Shape) {
        if let Shape::Group(group) = shape {
            let bb = group.bounding_box(); // merge of the children's parent_space_bounding_box (group.rs:233-242)
            let node = (self.node_min.len() / 3) as u32;
            self.node_min.extend_from_slice(&[bb.min.0.x as f32, bb.min.0.y as f32, bb.min.0.z as f32]);
            self.node_max.extend_from_slice(&[bb.max.0.x as f32, bb.max.0.y as f32, bb.max.0.z as f32]);
            self.node_box64.extend_from_slice(&[bb.min.0.x, bb.min.0.y, bb.min.0.z, bb.max.0.x, bb.max.0.y, bb.max.0.z]);
            let at = self.item_ref.len();
            self.item_ref.push(sys::RT_ITEM_NODE | node);
            self.item_next.push(0);
            for child in &group.children {
                self.push_shape(child);
            }
            self.item_next[at] = self.item_ref.len() as u32;
            return;
        }
        let cache: &crate::shape::ObjectCache = shape.as_ref();
        let mut param = [0f64; 24]; // filled in f64, rounded once for the f32 table
        let tag = match shape {
            Shape::Sphere(_) => sys::RT_PRIM_SPHERE,
            Shape::Plane(_) => sys::RT_PRIM_PLANE,
            Shape::Cube(_) => sys::RT_PRIM_CUBE,
            Shape::Cylinder(c) => {
                param[0] = c.min;
                param[1] = c.max;
                param[2] = if c.closed { 1.0 } else { 0.0 };
                sys::RT_PRIM_CYLINDER
            }
            Shape::Triangle(t) => {
                fill_triangle(&mut param, t);
                sys::RT_PRIM_TRIANGLE
            }
            Shape::SmoothTriangle(s) => {
                fill_triangle(&mut param, &s.triangle);
                for (k, n) in [s.n0, s.n1, s.n2].iter().enumerate() {
                    param[12 + 3 * k] = n.0.x;
                    param[13 + 3 * k] = n.0.y;
                    param[14 + 3 * k] = n.0.z;
                }
                sys::RT_PRIM_SMOOTH_TRIANGLE
            }
            Shape::Group(_) => unreachable!(),
        };
        let material = self.material_index(&cache.material);
        let prim = self.prim_type.len() as u32;
        self.prim_type.push(tag);
        self.prim_inv.extend_from_slice(&rows3x4(&cache.transform_inverse));
        self.prim_material.push(material);
        self.prim_inv64.extend_from_slice(&rows3x4_f64(&cache.transform_inverse));
        self.prim_param.extend(param.iter().map(|v| *v as f32));
        self.prim_param64.extend_from_slice(&param);
        self.item_ref.push(prim);
        self.item_next.push(self.item_ref.len() as u32);
    }

    fn push_light(&mut self, light: &Light) {
        match light {
            Light::Point(p) => {
                self.light_type.push(sys::RT_LIGHT_POINT);
                self.light_pos.extend_from_slice(&[p.position.0.x as f32, p.position.0.y as f32, p.position.0.z as f32]);
                self.light_uvec.extend_from_slice(&[0.0; 3]);
                self.light_vvec.extend_from_slice(&[0.0; 3]);
                self.light_geom64.extend_from_slice(&[p.position.0.x, p.position.0.y, p.position.0.z, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]);
                self.light_usteps.push(1);
                self.light_vsteps.push(1);
                push_color(&mut self.light_color, p.intensity);
            }
            Light::Area(a) => {
                self.light_type.push(sys::RT_LIGHT_AREA);
                self.light_pos.extend_from_slice(&[a.corner.0.x as f32, a.corner.0.y as f32, a.corner.0.z as f32]);
                self.light_uvec.extend_from_slice(&[a.uvec.0.x as f32, a.uvec.0.y as f32, a.uvec.0.z as f32]);
                self.light_vvec.extend_from_slice(&[a.vvec.0.x as f32, a.vvec.0.y as f32, a.vvec.0.z as f32]);
                self.light_geom64.extend_from_slice(&[
                    a.corner.0.x, a.corner.0.y, a.corner.0.z, a.uvec.0.x, a.uvec.0.y, a.uvec.0.z, a.vvec.0.x, a.vvec.0.y, a.vvec.0.z,
                ]);
                self.light_usteps.push(a.usteps as u32);
                self.light_vsteps.push(a.vsteps as u32);
                push_color(&mut self.light_color, a.intensity);
            }
        }
    }

    /// The C view of the tables; borrows `self`.
    pub(crate) fn desc(&self) -> sys::rt_scene_desc {
        sys::rt_scene_desc {
            abi_version: sys::RT_ABI_VERSION,
            n_prims: self.prim_type.len() as u32,
            prim_type: self.prim_type.as_ptr(),
            prim_inv: self.prim_inv.as_ptr(),
            prim_material: self.prim_material.as_ptr(),
            prim_param: self.prim_param.as_ptr(),
            n_nodes: (self.node_min.len() / 3) as u32,
            node_min: self.node_min.as_ptr(),
            node_max: self.node_max.as_ptr(),
            n_items: self.item_ref.len() as u32,
            item_ref: self.item_ref.as_ptr(),
            item_next: self.item_next.as_ptr(),
            n_materials: self.mat_pattern.len() as u32,
            mat_param: self.mat_param.as_ptr(),
            mat_pattern: self.mat_pattern.as_ptr(),
            n_patterns: self.pat_type.len() as u32,
            pat_type: self.pat_type.as_ptr(),
            pat_color_a: self.pat_color_a.as_ptr(),
            pat_color_b: self.pat_color_b.as_ptr(),
            pat_inv: self.pat_inv.as_ptr(),
            n_lights: self.light_type.len() as u32,
            light_type: self.light_type.as_ptr(),
            light_pos: self.light_pos.as_ptr(),
            light_uvec: self.light_uvec.as_ptr(),
            light_vvec: self.light_vvec.as_ptr(),
            light_usteps: self.light_usteps.as_ptr(),
            light_vsteps: self.light_vsteps.as_ptr(),
            light_color: self.light_color.as_ptr(),
            prim_inv64: self.prim_inv64.as_ptr(),
            prim_param64: self.prim_param64.as_ptr(),
            mat_ior64: self.mat_ior64.as_ptr(),
            pat_inv64: self.pat_inv64.as_ptr(),
            light_geom64: self.light_geom64.as_ptr(),
            node_box64: self.node_box64.as_ptr(),
        }
    }
}

fn fill_triangle(param: &mut [f64; 24], t: &crate::shape::Triangle) {
    let fields = [t.v0.0, t.e0.0, t.e1.0, t.normal.0];
    for (k, v) in fields.iter().enumerate() {
        param[3 * k] = v.x;
        param[3 * k + 1] = v.y;
        param[3 * k + 2] = v.z;
    }
}

fn check(rc: i32) {
    if rc != sys::RT_OK {
        // the reference panics on pool-build failure / poisoned mutex (camera.rs:158-161,175,199)
        let msg = unsafe { CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned();
        panic!("raytracer-amd: rt_status {rc}: {msg}");
    }
}

/// `RT_AMD_SEED` selects the jitter stream of area lights (the reference draws from thread_rng,
/// light.rs:149-152: every run differs; here every run with the same seed is identical).
fn seed_from_env() -> u32 {
    std::env::var("RT_AMD_SEED").ok().and_then(|s| s.parse().ok()).unwrap_or(0)
}

/// `Camera::render` on the GPU (camera.rs:163-208).
pub(crate) fn render(camera: &Camera, world: &World) -> Canvas {
    let flat = FlatScene::from_world(world);
    let desc = flat.desc();
    let cam = sys::rt_camera_desc {
        hsize: camera.hsize as u32,
        vsize: camera.vsize as u32,
        pixel_size: camera.pixel_size as f32,
        half_width: camera.half_width as f32,
        half_height: camera.half_height as f32,
        inv: rows3x4(&camera.transform_inverse),
        has_f64: 1,
        pixel_size64: camera.pixel_size,
        half_width64: camera.half_width,
        half_height64: camera.half_height,
        inv64: rows3x4_f64(&camera.transform_inverse),
    };
    let options = sys::rt_options {
        jitter_mode: sys::RT_JITTER_COUNTER,
        seed: seed_from_env(),
        ..Default::default() // max_depth 0 = RECURSION_DEPTH (world.rs:11)
    };
    let mut rgb = vec![0f32; camera.hsize * camera.vsize * 3];
    unsafe {
        let mut scene = std::ptr::null_mut();
        check(sys::rt_scene_create(&desc, 0, &mut scene));
        let rc = sys::rt_render(scene, &cam, &options, rgb.as_mut_ptr(), std::ptr::null_mut());
        sys::rt_scene_destroy(scene);
        check(rc);
    }
    let mut canvas = Canvas::new(camera.hsize, camera.vsize);
    for y in 0..camera.vsize {
        for x in 0..camera.hsize {
            let p = &rgb[(y * camera.hsize + x) * 3..][..3];
            canvas.write_pixel(
                x,
                y,
                Color {
                    red: p[0] as f64,
                    green: p[1] as f64,
                    blue: p[2] as f64,
                },
            );
        }
    }
    canvas
}
