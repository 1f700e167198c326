/* rt_amd.h — C ABI of librt_amd.so: the MI355X (gfx950) replacement for the per-pixel render
 * path of regexPattern/raytracer.
 *
 * What it replaces (reference file:line, relative to the reference repo root):
 *   Camera::render            src/camera.rs:163-208   -> rt_render / rt_render_device
 *   Canvas::to_image          src/canvas.rs:31-45     -> rt_quantize_u8 / rt_quantize_u8_device
 * and everything those call per pixel: Camera::ray_for_pixel (camera.rs:210-228),
 * World::{color_at,intersect,shade_hit,is_shadowed,reflected_color,refracted_color}
 * (world.rs:24-133), Shape::{intersect,normal_at} + every file under shape/, Intersection::{prepare_computation,
 * find_n1_and_n2,sort,hit} and Computation::schlick (intersection.rs:40-141), Light::intensity_at /
 * AreaLight::intensity_at (light.rs:147-216), Material::lighting (material.rs:118-162),
 * Pattern3D::color_at_object (pattern.rs:58-96).
 *
 * The caller (the reference's Rust side, or the C++ mirror in include/raytracer_amd/) keeps
 * building World / Camera exactly as in the reference's examples and compiles them ONCE into the flat
 * structure-of-arrays description below (values computed in f64, rounded once to f32).  Every
 * pointer is plain host memory owned by the caller and only read during the call that takes it.
 * No C++ or torch types cross this boundary; no function throws or aborts; every function returns
 * RT_OK (0) or a negative rt_status and leaves a message for rt_last_error().
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3u

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_BAD_ARGUMENT = -1,   /* null pointer, zero size, index out of range, malformed tree */
    RT_ERR_UNSUPPORTED = -2,    /* unknown primitive / pattern / light tag                     */
    RT_ERR_CAPACITY = -3,       /* a stated limit exceeded (RT_MAX_*)                          */
    RT_ERR_HIP = -4,            /* a HIP runtime call failed (message holds hipGetErrorString) */
    RT_ERR_NO_DEVICE = -5       /* no usable gfx950 device                                     */
} rt_status;

/* Shape enum, shape.rs:33-41 (Group is not a primitive: it becomes rt node records). */
enum { RT_PRIM_SPHERE = 0, RT_PRIM_PLANE = 1, RT_PRIM_CUBE = 2, RT_PRIM_CYLINDER = 3,
       RT_PRIM_TRIANGLE = 4, RT_PRIM_SMOOTH_TRIANGLE = 5 };
/* Pattern3D enum, pattern.rs:16-31 */
enum { RT_PAT_SOLID = 0, RT_PAT_STRIPE = 1, RT_PAT_GRADIENT = 2, RT_PAT_RING = 3, RT_PAT_CHECKER = 4 };
/* Light enum, light.rs:14-20 */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1 };
/* light.rs:149-152 draws thread_rng u8 / 255; here: 0 = every draw is 0.5 (deterministic parity
 * mode), 1 = counter-based hash of (seed, pixel, path node, light, cell, axis) -> top 8 bits / 255 */
enum { RT_JITTER_CENTER = 0, RT_JITTER_COUNTER = 1 };

#define RT_MAX_DEPTH 8u            /* options.max_depth limit (reference: RECURSION_DEPTH = 5)  */
#define RT_MAX_LIGHTS 256u
#define RT_MAX_LIGHT_CELLS 32768u  /* usteps * vsteps per area light                            */
#define RT_ITEM_NODE 0x80000000u   /* item_ref bit: the item is a node, not a primitive         */

/* Flat scene.  Transforms are affine (bottom row 0,0,0,1 — every public constructor of
 * transform.rs:141-275 and their products/inverses), stored as the first three rows, row-major,
 * 12 floats.  "inv" always means the object's (or pattern's) transform_inverse.
 *
 * Traversal items: the reference tests world.objects in order and, for a Group, its children in
 * order after a box test (world.rs:32-41, group.rs:106-119).  The same order is given here as ONE
 * array of items in depth-first pre-order.  Item i is either primitive item_ref[i], or node
 * (item_ref[i] & ~RT_ITEM_NODE) whose subtree is items (i, item_next[i]); item_next[i] is the
 * item visited when the subtree is skipped (i + 1 for a primitive).  A node's box is the merge of
 * its children's parent_space_bounding_box — what Group::bounding_box() (group.rs:233-242)
 * recomputes for every ray. */
typedef struct rt_scene_desc {
    uint32_t abi_version;          /* RT_ABI_VERSION */

    uint32_t n_prims;
    const uint8_t* prim_type;      /* [n_prims]      RT_PRIM_*                                      */
    const float* prim_inv;         /* [n_prims][12]  transform_inverse, object.rs:9                  */
    const uint32_t* prim_material; /* [n_prims]      index into mat_*                                */
    const float* prim_param;       /* [n_prims][24]  cylinder: min, max, closed(0/1)  (cylinder.rs:42-47)
                                                     triangle: v0[3], e0[3], e1[3], normal[3]       (triangle.rs:48-56)
                                                     smooth:   + n0[3], n1[3], n2[3]                (smooth_triangle.rs:14-19)
                                                     others: ignored.  May be NULL if no primitive needs it. */

    uint32_t n_nodes;
    const float* node_min;         /* [n_nodes][3]   may hold +-inf                                   */
    const float* node_max;         /* [n_nodes][3]                                                   */

    uint32_t n_items;
    const uint32_t* item_ref;      /* [n_items]                                                      */
    const uint32_t* item_next;     /* [n_items]      i < item_next[i] <= n_items                     */

    uint32_t n_materials;
    const float* mat_param;        /* [n_materials][8] ambient, diffuse, specular, shininess,
                                                       index_of_refraction, reflectivity, transparency, 0
                                                       (material.rs:37-87)                           */
    const uint32_t* mat_pattern;   /* [n_materials]  index into pat_*                                */

    uint32_t n_patterns;
    const uint8_t* pat_type;       /* [n_patterns]   RT_PAT_*                                        */
    const float* pat_color_a;      /* [n_patterns][3] (Solid: the colour)                            */
    const float* pat_color_b;      /* [n_patterns][3]                                                */
    const float* pat_inv;          /* [n_patterns][12] pattern transform_inverse (identity for Solid) */

    uint32_t n_lights;             /* world.lights order (world.rs:44)                               */
    const uint8_t* light_type;     /* [n_lights]     RT_LIGHT_*                                      */
    const float* light_pos;        /* [n_lights][3]  point: position; area: corner  (light.rs:42-48,81-89) */
    const float* light_uvec;       /* [n_lights][3]  area: horizontal_dir / usteps                   */
    const float* light_vvec;       /* [n_lights][3]  area: vertical_dir / vsteps                     */
    const uint32_t* light_usteps;  /* [n_lights]     1 for point lights                              */
    const uint32_t* light_vsteps;  /* [n_lights]                                                     */
    const float* light_color;      /* [n_lights][3]  intensity                                       */

    /* Optional (ABI 2): the caller's own f64 values of the tables the hit geometry hangs on — all four
     * or none (NULL).  With them (and rt_camera_desc.has_f64) the kernels carry every path ray a second
     * time in f64 and work the hits that decide something out again with the reference's formulas
     * (raytracer_amd/csrc/rt_exact.h): pattern boundaries, silhouettes, exact ties and deep glass paths
     * then follow the f64 reference instead of FP32 rounding.  Without them the path is FP32 only. */
    const double* prim_inv64;      /* [n_prims][12]  transform_inverse, object.rs:9                  */
    const double* prim_param64;    /* [n_prims][24]  same layout as prim_param                       */
    const double* mat_ior64;       /* [n_materials]  index_of_refraction                             */
    const double* pat_inv64;       /* [n_patterns][12]                                               */
    /* Optional (ABI 3), used with the four tables above: the lights' geometry in the caller's f64 — position
     * (area: corner), uvec, vvec, nine doubles a light (light.rs:42-48,81-89).  With it the shadow rays FP32
     * cannot settle (a ray grazing an occluder, a point within 1e-4 of another surface) are decided from the
     * f64 over point towards the f64 light position (world.rs:73-92).  NULL: the f32 values are widened. */
    const double* light_geom64;    /* [n_lights][9]                                                   */
    /* Optional (ABI 3), used with the four tables above: the Groups' boxes in the caller's f64, min then max.  A
     * ray that clips a Group's box within FP32 reach of an edge is then let in or not by the reference's own test
     * (group.rs:107, bounding_box.rs:81-84: tmin < tmax in f64).  NULL: node_min / node_max are widened. */
    const double* node_box64;      /* [n_nodes][6]                                                    */
} rt_scene_desc;

/* Camera, camera.rs:64-73 after Camera::try_from (camera.rs:95-136). */
typedef struct rt_camera_desc {
    uint32_t hsize, vsize;
    float pixel_size, half_width, half_height;
    float inv[12];                 /* transform_inverse, first three rows                            */
    uint32_t has_f64;              /* nonzero: the four fields below hold the caller's f64 values    */
    double pixel_size64, half_width64, half_height64;
    double inv64[12];
} rt_camera_desc;

/* Row tiles for multi-GPU partitioning: the image is cut into bands of `tile_rows` pixel rows;
 * this call renders bands tile_first, tile_first + tile_stride, ... and packs them one after
 * another into the output (band k of this call starts at output row k * tile_rows).  The defaults
 * (tile_rows = 0) render the whole image.  The jitter hash is keyed on the GLOBAL pixel index, so
 * the image does not depend on the partition: BIT FOR BIT when tile_rows is a multiple of 8 (the
 * kernels work on 8x8 pixel tiles of the packed rows, so every tile then holds the same pixels
 * as in the whole frame; raytracer_amd/partition.py uses 8).  With other band heights the tiles
 * hold other pixels, and two sums follow the tile's composition — a pixel's path nodes added up
 * in the order its wave finished them (ray stealing, lane-mode kernels) and the Phong sums shared
 * out over the idle lanes of a sparse wave — so a pixel can differ in its last bits (<= 1e-6
 * relative; tests/test_gpu_parity.py::test_unaligned_bands_differ_in_the_last_bits_only). */
typedef struct rt_options {
    uint32_t max_depth;            /* world.rs:11 RECURSION_DEPTH; 0 in a zeroed struct means 5
                                      (RT_FLAG_DEPTH_ZERO asks for depth 0 itself)                   */
    uint32_t jitter_mode;          /* RT_JITTER_*                                                    */
    uint32_t seed;
    uint32_t tile_rows, tile_first, tile_stride;
    uint32_t flags;                /* RT_FLAG_* bits; unknown bits are rejected                      */
} rt_options;

/* rt_render_device only: rgb_out_device is a FULL frame (vsize * hsize * 3 floats) and every row is
 * stored at its GLOBAL position instead of packed band after band.  With a frame that lives on
 * another GPU of the node (rt_frame_open) the ranks of a row-band partition write their bands
 * straight into the root GPU's frame over xGMI: the "final gather" of the partition
 * (camera.rs:183-205 collects rows through a Mutex<Canvas>) without a collective and without a
 * staging copy. */
#define RT_FLAG_GLOBAL_ROWS 1u
/* Render with the FP32 path only even though the scene and the camera carry f64 tables (timing the
 * f64 chain's share, parity experiments). */
#define RT_FLAG_F32_ONLY 2u
/* ABI 3.  World::color_at(ray, 0) (world.rs:24): the surface term of the first hit only — reflected_color and
 * refracted_color return BLACK at depth 0 (world.rs:97,119).  A flag because max_depth = 0 in a zeroed struct
 * keeps meaning RECURSION_DEPTH = 5; with this flag max_depth must be 0. */
#define RT_FLAG_DEPTH_ZERO 4u

/* Ray and work counters of one render.  primary/reflection/refraction rays are traced one by one and
 * counted once each (the reference re-traces the recursion once per light, world.rs:59-60; that
 * reference-equivalent count is derived by the caller).  shadow_rays counts every shadow ray of the
 * reference's accounting (light.rs:182-209: one per light cell of every shaded hit), whether the
 * kernel cast it or ANSWERED it without casting (shaft tests, lights behind the surface, materials
 * without diffuse and specular terms); shadow_rays_cast counts only those that were really traced
 * against the scene.  shadow_rays_cast <= shadow_rays. */
typedef struct rt_stats {
    uint64_t primary_rays, reflection_rays, refraction_rays, shadow_rays;
    uint64_t shaded_hits, light_cells, pattern_lookups, container_passes;
    uint64_t shadow_rays_cast;     /* ABI 2 */
    uint64_t exact_nodes;          /* ABI 2: path nodes worked out again in f64 (rt_scene_desc.*64)  */
    uint64_t pixels;
    float kernel_ms;               /* hipEvent time of the render kernel(s) on the scene's stream    */
} rt_stats;

/* Opaque: device-resident copy of a scene on ONE GPU.  Calls that take the same rt_scene must be
 * serialised by the caller (they share the scene's event pair and counter block); distinct scenes are
 * independent, and the tables of a scene are never written after rt_scene_create. */
typedef struct rt_scene rt_scene;

/* Library / device. */
uint32_t rt_abi_version(void);
const char* rt_last_error(void);                      /* thread-local message of the last failure   */
int rt_device_count(int* count_out);

/* Copies the description to `device` (HIP ordinal) and prepares it for rendering.  Validates every
 * index and the item tree (so that traversal always terminates).  Replaces building/borrowing
 * `&World` in Camera::render(&self, world: &World), camera.rs:163. */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** scene_out);
void rt_scene_destroy(rt_scene* scene);

/* ABI 3.  Sets the scene up for frames of up to `hsize` x `rows` pixels ahead of time: the tile-cost table of a
 * scene with transparent materials (rt_render_device below) and the device image the host-output entry points
 * keep.  After it, rt_render_device never allocates or waits for frames up to that size, and rt_render /
 * rt_render_u8 / rt_render_stats call no hipMalloc either.  Optional: without it the first call of a size does
 * the same work itself.  No counterpart in the reference (Camera::render allocates its Canvas per call,
 * camera.rs:164). */
int rt_scene_reserve(rt_scene* scene, uint32_t hsize, uint32_t rows);

/* Which instantiation of the render kernel the scene runs — render_kernel<counters, EXOTIC, TREE, AREA,
 * F64, LANE, SPAWN> in rocprofv3 output: RT_VARIANT_* bits (profiling and bench reports; no counterpart in the
 * reference).  F64 says the scene carries the f64 tables; the kernel with the f64 chain runs when the
 * camera carries its f64 fields too and RT_FLAG_F32_ONLY is not set. */
#define RT_VARIANT_AREA 1u     /* some light is an area light                      */
#define RT_VARIANT_TREE 2u     /* the stream has hierarchy nodes or Group gates    */
#define RT_VARIANT_EXOTIC 4u   /* cylinders or triangles                           */
#define RT_VARIANT_F64 8u      /* rt_scene_desc.*64 given                          */
#define RT_VARIANT_LANE 16u    /* a long hierarchy: every lane walks it on its own */
#define RT_VARIANT_SPAWN 32u   /* some material is reflective or transparent       */
int rt_scene_variant(const rt_scene* scene, uint32_t* flags_out);

/* The scene's walk streams as the device holds them (tests and tools; no counterpart in the reference): 32-byte
 * records {float box[6]; uint32 next; uint32 ref}.  A scene with a hierarchy: the eight direction orders of the stream,
 * which the GPU lays out itself at rt_scene_create, then the canonical order, then the eight staged tops (16 records
 * each) that the lane-mode kernels copy into LDS; any other scene: the canonical order alone.  out may be NULL to ask
 * for the size; *bytes_out receives it.  RT_ERR_CAPACITY if cap is too small.  Synchronous. */
int rt_scene_read_streams(const rt_scene* scene, void* out, size_t cap, size_t* bytes_out);

/* Number of output rows a call with these options produces (<= camera vsize). */
int rt_output_rows(const rt_camera_desc* camera, const rt_options* options, uint32_t* rows_out);

/* Camera::render, camera.rs:163-208.  rgb_out: rows * hsize * 3 floats, row-major, row 0 = top,
 * red/green/blue per pixel — the Canvas contents in canvas.rs:8-12 order.
 *   rt_render_device: rgb_out is DEVICE memory on the scene's GPU; the work is enqueued on
 *                     `stream` (a hipStream_t passed as void*, NULL = the default stream) and the
 *                     call returns without waiting.  Allocates nothing, with one exception: a scene
 *                     with transparent materials keeps a tile-cost table of 20 bytes per 8x8 tile
 *                     (every wave leaves a cost figure of its tile: ten per round of its path loop
 *                     plus one per visibility pass; the next frame with the same tiling starts its
 *                     expensive tiles first — same image, shorter tail).  The first call
 *                     with a frame of 8192 tiles or more that is larger than any before waits for the
 *                     stream, allocates that table and keeps it until rt_scene_destroy
 *                     (RT_AMD_NO_TILE_ORDER=1 in the environment: never).  Calls on one scene must be
 *                     ordered on the device as well (one stream, or events between streams).
 *                     rt_render_stats_noalloc never allocates and does not reorder.
 *   rt_render:        rgb_out is HOST memory; the call renders into a device image the scene keeps (allocated
 *                     on the first call of a size, or by rt_scene_reserve), copies back and waits.
 *                     stats_out may be NULL.
 *   rt_render_u8:     (ABI 3) Camera::render followed by Canvas::to_image (canvas.rs:31-45) in one call, for
 *                     callers that only want the 8-bit image — what every example of the reference does
 *                     (the examples: `camera.render(&world).to_image()`): the frame is quantised on the device
 *                     and ONE byte per channel crosses PCIe instead of four.  rgb8_out_host: rows * hsize * 3
 *                     bytes.  Bit-identical to rt_render + rt_quantize_u8. */
int rt_render_device(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options,
                     float* rgb_out_device, void* stream);
int rt_render(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, float* rgb_out_host,
              rt_stats* stats_out);
int rt_render_u8(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, uint8_t* rgb8_out_host,
                 rt_stats* stats_out);
/* Same image, plus the counters (uses the instrumented kernel variant; slower). rgb may be NULL. */
int rt_render_stats(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options,
                    float* rgb_out_host, rt_stats* stats_out);

/* rt_render_stats for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that may not call hipMalloc — tight serving loops, harnesses that forbid allocation, code
 * that runs beside a graph capture on another stream.  The call itself is SYNCHRONOUS on `stream` (it
 * records events, copies the counters to the host and waits), so it cannot run ON a capturing stream;
 * host outputs land in pageable memory unless the caller passes pinned buffers.  Same kernel, image and
 * counters as rt_render_stats, but
 *   - the image goes to DEVICE memory the caller passes (rows * hsize * 3 floats, as rt_render_device);
 *   - the kernel's scratch (counter block + per-tile ray map) lives in a static device array of the
 *     library: RT_NOALLOC_MAX_TILES * 4 + 256 bytes held on every GPU the library runs on until the
 *     process ends; a call whose rows make more than RT_NOALLOC_MAX_TILES tiles of 8x8 pixels
 *     (3840x2160 = 129600 fits) is rejected with RT_ERR_CAPACITY before anything is launched;
 *   - tile_rays_out_host (may be NULL) receives, per 8x8 tile in raster order of the call's rows
 *     (ceil(hsize/8) per tile row), the rays that tile really traced: the cost map of the frame;
 *   - the call enqueues on `stream`, waits for it and fills stats_out; it calls no hipMalloc/hipFree
 *     and no allocating library routine.
 * Every call on a GPU shares that GPU's scratch, which keeps the last call's contents (the counters
 * are cleared on the stream at the start of each call, every tile entry is overwritten): calls that
 * overlap on one GPU — from any scene — race and must be serialised by the caller. */
#define RT_NOALLOC_MAX_TILES 131072u
int rt_render_stats_noalloc(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options,
                            float* rgb_out_device, void* stream, rt_stats* stats_out,
                            uint32_t* tile_rays_out_host);

/* Canvas::to_image, canvas.rs:31-45: out[i] = (rgb[i] * 255.0) as u8 — truncate toward zero,
 * saturate to 0..255, NaN -> 0.  n = number of floats. */
int rt_quantize_u8_device(const float* rgb_device, size_t n, uint8_t* out_device, int device, void* stream);
int rt_quantize_u8(const float* rgb_host, size_t n, uint8_t* out_host);

/* One frame shared by the GPUs of a node (no counterpart in the reference, whose threads share one
 * Canvas behind a Mutex, camera.rs:176,199-201: this is that shared Canvas for one process per GPU).
 *   rt_frame_create : allocates `bytes` of device memory on `device` and exports an inter-process
 *                     handle for it (hipIpcGetMemHandle).
 *   rt_frame_open   : in ANOTHER process, maps that memory for kernels running on `device`
 *                     (`owner_device` = the HIP ordinal the frame lives on; peer access over xGMI
 *                     is enabled when the two differ).  The pointer is valid for rt_render_device
 *                     with RT_FLAG_GLOBAL_ROWS and for rt_frame_download.
 *   rt_frame_close  : unmaps a pointer obtained from rt_frame_open.
 *   rt_frame_destroy: frees a frame obtained from rt_frame_create (after every opener has closed).
 *   rt_frame_download: copies `bytes` from a frame (created or opened) to host memory, after
 *                     waiting for the device.
 * The caller orders the writers and the reader (e.g. a barrier between the last render and the
 * download); rows written by different ranks never overlap. */
typedef struct rt_ipc_handle {
    unsigned char bytes[64];
} rt_ipc_handle;
int rt_frame_create(int device, size_t bytes, void** frame_out, rt_ipc_handle* handle_out);
int rt_frame_open(int device, int owner_device, const rt_ipc_handle* handle, void** frame_out);
int rt_frame_close(int device, void* frame);
int rt_frame_destroy(int device, void* frame);
int rt_frame_download(int device, const void* frame, void* host_out, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
