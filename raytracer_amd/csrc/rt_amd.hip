// rt_amd.hip — gfx950 render kernels + the C ABI of include/rt_amd.h.
//
// The device code (render_kernel, quantize_kernel) lives in rt_kernel.h; this file is the host
// side of the ABI: validation, upload, launch, copies, error strings.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_aux_kernels.h"
#include "rt_kernel.h"
#include "rt_pack.h"

using namespace rtk;

namespace rtk {  // rt_kernels.hip, one translation unit per group
using KernelFn = void (*)(const KParams);
KernelFn kernel_group_0(int variant);
KernelFn kernel_group_1(int variant);
KernelFn kernel_group_3(int variant);
KernelFn kernel_group_4(int variant);
KernelFn kernel_group_5(int variant);
KernelFn kernel_group_7(int variant);
}  // namespace rtk

namespace {

// ------------------------------------------------------------------------------------------ host
thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

// `copies`: room for that many copies of the table (the first one is filled)
template <class V, class T = typename V::value_type>
int upload(const V& v, const T** out, std::vector<void*>& owned, size_t copies = 1) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T) * copies));
    owned.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RT_OK;
}

struct RenderPlan {
    uint32_t max_depth, tile_rows, tile_first, tile_stride, out_rows;
    bool global_rows, f32_only;
};

int plan_render(const rt_camera_desc* cam, const rt_options* opt, RenderPlan& rp) {
    if (!cam) return fail(RT_ERR_BAD_ARGUMENT, "camera is null");
    if (cam->hsize == 0 || cam->vsize == 0) return fail(RT_ERR_BAD_ARGUMENT, "camera cannot have null dimensions");
    if (static_cast<uint64_t>(cam->hsize) * cam->vsize > 0x7FFFFFFFull) return fail(RT_ERR_CAPACITY, "image too large");
    rt_options o{};
    if (opt) o = *opt;
    if (o.flags & ~(RT_FLAG_GLOBAL_ROWS | RT_FLAG_F32_ONLY | RT_FLAG_DEPTH_ZERO)) return fail(RT_ERR_BAD_ARGUMENT, "options.flags holds an unknown bit");
    rp.global_rows = (o.flags & RT_FLAG_GLOBAL_ROWS) != 0;
    rp.f32_only = (o.flags & RT_FLAG_F32_ONLY) != 0;
    if (o.jitter_mode > RT_JITTER_COUNTER) return fail(RT_ERR_UNSUPPORTED, "unknown jitter mode");
    if (!options_max_depth(&o, rp.max_depth)) return fail(RT_ERR_BAD_ARGUMENT, "RT_FLAG_DEPTH_ZERO goes with max_depth = 0");
    if (rp.max_depth > RT_MAX_DEPTH) return fail(RT_ERR_CAPACITY, "max_depth exceeds RT_MAX_DEPTH");
    if (o.tile_rows == 0) {
        rp.tile_rows = cam->vsize, rp.tile_first = 0, rp.tile_stride = 1, rp.out_rows = cam->vsize;
        return RT_OK;
    }
    if (o.tile_stride == 0) return fail(RT_ERR_BAD_ARGUMENT, "tile_stride must be >= 1 when tile_rows is set");
    if (o.tile_first >= o.tile_stride) return fail(RT_ERR_BAD_ARGUMENT, "tile_first must be < tile_stride");
    rp.tile_rows = o.tile_rows, rp.tile_first = o.tile_first, rp.tile_stride = o.tile_stride;
    uint32_t rows = 0;
    for (uint64_t b = o.tile_first;; b += o.tile_stride) {  // bands b*tile_rows .. clipped to vsize
        uint64_t y0 = b * o.tile_rows;
        if (y0 >= cam->vsize) break;
        uint64_t y1 = y0 + o.tile_rows;
        rows += static_cast<uint32_t>((y1 > cam->vsize ? cam->vsize : y1) - y0);
    }
    rp.out_rows = rows;
    return RT_OK;
}

}  // namespace

// ------------------------------------------------------------------ scratch of rt_render_stats_noalloc
// The one entry point that allocates nothing keeps its kernel's scratch — the counter block and the
// per-tile ray map the instrumented kernel writes — in this file-scope array: every GPU the library
// runs on holds its own copy (RT_NOALLOC_MAX_TILES * 4 + 256 bytes) until the process ends.  The
// array's NAME is a host-side stand-in, not a device address: host code only ever uses the per-device
// pointer hipGetSymbolAddress returned with that device current (noalloc_scratch_of).
struct NoAllocScratch {
    unsigned long long counters[32];          // kCounters used; 256 bytes keep the map aligned
    uint32_t tile_rays[RT_NOALLOC_MAX_TILES];
};
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
NoAllocScratch* g_scratch_ptr[kMaxDevices] = {};

// The device's own copy of `scratch`; looked up once per device (rt_scene_create, with that device
// current) and kept.  Returns RT_OK or a negative rt_status.
int noalloc_scratch_of(int device, NoAllocScratch** out) {
    static_assert(kCounters <= 32, "counter block of NoAllocScratch");
    if (device < 0 || device >= kMaxDevices) return fail(RT_ERR_CAPACITY, "device ordinal beyond the scratch table (64 devices)");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int cur = -1;
        HIP_TRY(hipGetDevice(&cur));
        if (cur != device) return fail(RT_ERR_BAD_ARGUMENT, "scratch lookup needs its device current");
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) returned null");
        g_scratch_ptr[device] = static_cast<NoAllocScratch*>(p);
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}
}  // namespace

struct rt_scene {
    int device = 0;
    NoAllocScratch* scratch_ptr = nullptr;  // this device's copy of `scratch` (rt_render_stats_noalloc)
    bool exotic = false;  // some primitive is a cylinder or a triangle
    bool tree = false;    // the stream has hierarchy nodes or Group gates
    bool lane = false;    // a hierarchy long enough for lane-mode walks (rt_kernel.h LANE)
    bool area = false;    // some light is an area light
    bool spawn = true;    // some material is reflective or transparent (else: the kernels without path code)
    bool has_f64 = false; // the caller gave the f64 tables of the exact chain (rt_exact.h)
    KScene k{};
    std::vector<void*> owned;
    unsigned long long* counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // tile order (rt_kernel.h "Tile order"): scenes with transparent materials keep 20 bytes per tile of the largest
    // frame rendered so far — the cost of each tile in the last frame, and two sets of class words + order
    bool ordered = false;
    uint32_t* order_buf = nullptr;
    uint32_t order_cap = 0;                       // tiles the buffer holds
    uint32_t cost_tiles = 0, cost_tiles_x = 0;    // tiling of the frame whose costs the table holds (0: none yet)
    uint32_t order_flip = 0;
    // the device image of the host-output entry points (rt_render, rt_render_u8, rt_render_stats): kept between calls
    float* img = nullptr;
    size_t img_floats = 0;
    uint8_t* img8 = nullptr;
    size_t img8_bytes = 0;
};

namespace {

// Frames of fewer tiles than the chip holds waves at once (256 CUs x 4 SIMDs x 5) start all their tiles together anyway.
constexpr uint32_t kOrderMinTiles = 8192;

// The tile-cost table of a scene with glass for frames of up to n_tiles tiles (layout: launch_render below).
int reserve_order(rt_scene* s, uint32_t n_tiles, hipStream_t stream, bool may_wait) {
    if (!s->ordered || n_tiles < kOrderMinTiles || n_tiles <= s->order_cap) return RT_OK;
    if (may_wait) HIP_TRY(hipStreamSynchronize(stream));  // (kernels of earlier calls may still read the old table)
    if (s->order_buf) (void)hipFree(s->order_buf);
    s->order_buf = nullptr, s->order_cap = 0, s->cost_tiles = 0;
    const size_t words = 5u * static_cast<size_t>(n_tiles) + 2u * kOrderClasses;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->order_buf), words * sizeof(uint32_t)));
    s->order_cap = n_tiles;
    return RT_OK;
}
int reserve_image(rt_scene* s, size_t floats, bool want_u8) {
    if (floats > s->img_floats) {
        if (s->img) (void)hipFree(s->img);
        s->img = nullptr, s->img_floats = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->img), floats * sizeof(float)));
        s->img_floats = floats;
    }
    if (want_u8 && floats > s->img8_bytes) {
        if (s->img8) (void)hipFree(s->img8);
        s->img8 = nullptr, s->img8_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->img8), floats));
        s->img8_bytes = floats;
    }
    return RT_OK;
}

int launch_render(rt_scene* s, const rt_camera_desc* cam, const rt_options* opt, const RenderPlan& rp, float* d_out,
                  hipStream_t stream, bool stats, unsigned long long* counters = nullptr, uint32_t* tile_rays = nullptr,
                  bool allow_order = true) {
    KParams P{};
    P.sc = s->k;
    P.cam = pack_camera(*cam);
    const bool exact = s->has_f64 && cam->has_f64 != 0 && !rp.f32_only;
    if (exact) P.cam64 = pack_camera64(*cam);
    P.out = d_out;
    P.counters = counters ? counters : s->counters;
    P.tile_rays = tile_rays;
    P.tile_order = nullptr;
    P.tile_cost = nullptr;
    P.max_depth = rp.max_depth;
    P.jitter_mode = opt ? opt->jitter_mode : 0u;
    P.seed = opt ? opt->seed : 0u;
    P.tile_rows = rp.tile_rows, P.tile_first = rp.tile_first, P.tile_stride = rp.tile_stride;
    P.out_rows = rp.out_rows;
    P.global_rows = rp.global_rows ? 1u : 0u;
    P.tiles_x = (cam->hsize + 7u) / 8u;
    const uint32_t tiles_y = (rp.out_rows + 7u) / 8u;
    P.n_tiles = P.tiles_x * tiles_y;
    if (stats) HIP_TRY(hipMemsetAsync(P.counters, 0, kCounters * sizeof(unsigned long long), stream));
    if (P.n_tiles == 0) return RT_OK;
    if (s->ordered && allow_order && P.n_tiles >= kOrderMinTiles && !std::getenv("RT_AMD_NO_TILE_ORDER")) {
        // layout of order_buf: [cost: cap][2 x (counters + class words: 32 + cap)][2 x order: cap]
        if (P.n_tiles > s->order_cap) {  // first call with a frame this large and no rt_scene_reserve: (re)allocate, then keep
            const int rc = reserve_order(s, P.n_tiles, stream, true);
            if (rc != RT_OK) return rc;
        }
        const size_t cap = s->order_cap;
        uint32_t* cost = s->order_buf;
        if (s->cost_tiles == P.n_tiles && s->cost_tiles_x == P.tiles_x) {  // the last frame had this tiling: its costs order this one
            s->order_flip ^= 1u;  // (two sets: the kernels of the call before may still be reading theirs)
            uint32_t* cls = s->order_buf + cap + s->order_flip * (kOrderClasses + cap);
            uint32_t* order = s->order_buf + cap + 2u * (kOrderClasses + cap) + s->order_flip * cap;
            HIP_TRY(hipMemsetAsync(cls, 0, kOrderClasses * sizeof(uint32_t), stream));
            hipLaunchKernelGGL(tile_class_kernel, dim3((P.n_tiles + 1023u) / 1024u), dim3(1024), 0, stream, P.n_tiles, cost, cls);
            hipLaunchKernelGGL(tile_order_kernel, dim3((P.n_tiles + 255u) / 256u), dim3(256), 0, stream, P.n_tiles, cls, order);
            HIP_TRY(hipGetLastError());
            P.tile_order = order;
        }
        P.tile_cost = cost;
        s->cost_tiles = P.n_tiles, s->cost_tiles_x = P.tiles_x;
    }
    const uint32_t tiles_per_block = rtk::kBlockThreads / 64u;
    const uint32_t blocks = (P.n_tiles + tiles_per_block - 1u) / tiles_per_block;  // one wave per tile
    // 80 instantiations, compiled group by group (rt_kernels.hip): [f64 chain][lane mode][path code] x
    // variant = counters x cylinder/triangle code x hierarchy/gate code x area-light code
    const int variant = (stats ? 8 : 0) | (s->exotic ? 4 : 0) | (s->tree ? 2 : 0) | (s->area ? 1 : 0);
    const bool lane = s->lane && s->tree;
    const int group = (exact ? 4 : 0) | (lane ? 2 : 0) | ((s->spawn || lane) ? 1 : 0);  // lane mode always holds the path code
    KernelFn kernel = nullptr;
    switch (group) {
        case 0: kernel = kernel_group_0(variant); break;
        case 1: kernel = kernel_group_1(variant); break;
        case 3: kernel = kernel_group_3(variant); break;
        case 4: kernel = kernel_group_4(variant); break;
        case 5: kernel = kernel_group_5(variant); break;
        case 7: kernel = kernel_group_7(variant); break;
        default: break;
    }
    if (!kernel) return fail(RT_ERR_UNSUPPORTED, "no render kernel for this scene (library built with a reduced kernel set)");
    P.no_fix = std::getenv("RT_AMD_NO_FIX") ? 1u : 0u;  // (timing experiments)
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(rtk::kBlockThreads), 0, stream, P);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int render_host(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, float* rgb_out_host, uint8_t* rgb8_out_host,
                rt_stats* stats_out, bool stats) {
    if (!scene) return fail(RT_ERR_BAD_ARGUMENT, "scene is null");
    RenderPlan rp{};
    int rc = plan_render(camera, options, rp);
    if (rc != RT_OK) return rc;
    if (!rgb_out_host && !rgb8_out_host && !stats) return fail(RT_ERR_BAD_ARGUMENT, "output buffer is null");
    if (rp.global_rows) return fail(RT_ERR_BAD_ARGUMENT, "RT_FLAG_GLOBAL_ROWS is for rt_render_device only");
    HIP_TRY(hipSetDevice(scene->device));
    const size_t n = static_cast<size_t>(rp.out_rows) * camera->hsize * 3u;
    rc = reserve_image(scene, n ? n : 1u, rgb8_out_host != nullptr);
    if (rc != RT_OK) return rc;
    float* const d_out = scene->img;
    hipStream_t stream = nullptr;
    (void)hipEventRecord(scene->ev0, stream);
    rc = launch_render(scene, camera, options, rp, d_out, stream, stats);
    (void)hipEventRecord(scene->ev1, stream);
    if (rc == RT_OK && rgb8_out_host && n) rc = rt_quantize_u8_device(d_out, n, scene->img8, scene->device, stream);  // canvas.rs:31-45
    if (rc == RT_OK) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
    }
    if (rc == RT_OK && rgb_out_host && n) {
        hipError_t e = hipMemcpy(rgb_out_host, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(D2H)");
    }
    if (rc == RT_OK && rgb8_out_host && n) {
        hipError_t e = hipMemcpy(rgb8_out_host, scene->img8, n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(D2H, u8)");
    }
    if (rc == RT_OK && stats_out) {
        std::memset(stats_out, 0, sizeof(*stats_out));
        (void)hipEventElapsedTime(&stats_out->kernel_ms, scene->ev0, scene->ev1);
        stats_out->pixels = static_cast<uint64_t>(rp.out_rows) * camera->hsize;
        if (stats) {
            unsigned long long c[kCounters] = {0};
            hipError_t e = hipMemcpy(c, scene->counters, sizeof(c), hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(counters)");
            stats_out->primary_rays = c[0], stats_out->reflection_rays = c[1], stats_out->refraction_rays = c[2];
            stats_out->shadow_rays = c[3], stats_out->shaded_hits = c[4], stats_out->light_cells = c[5];
            stats_out->pattern_lookups = c[6], stats_out->container_passes = c[7], stats_out->shadow_rays_cast = c[8];
            stats_out->exact_nodes = c[9];
#ifdef RT_PHASE_CLOCK
            std::fprintf(stderr, "phase ticks (summed over waves): trace %llu prepare+containers %llu f64 %llu shade %llu children %llu | loop iterations %llu\n",
                         c[10], c[11], c[12], c[13], c[14], c[15]);
            std::fprintf(stderr, "second count: ticks (summed over waves) %llu, records %llu, walk steps %llu, the longest wave's ticks %llu\n", c[16], c[17], c[18], c[19]);
#endif
        }
    }
    return rc;
}

}  // namespace

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(int* count_out) {
    if (!count_out) return fail(RT_ERR_BAD_ARGUMENT, "count_out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count_out = 0;
        return fail(RT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count_out = n;
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** scene_out) {
    if (!scene_out) return fail(RT_ERR_BAD_ARGUMENT, "scene_out is null");
    *scene_out = nullptr;
    PackedScene ps;
    std::string err;
    const auto t_start = std::chrono::steady_clock::now();
    const bool staged_top = RT_LDS_TOP != 0 && !std::getenv("RT_AMD_NO_TOP");  // (N1, rt_kernel.h: an experiment; the variable: its A/B timings)
    int rc = RT_OK;
    try {  // (the build runs on host threads and sizes tables of hundreds of megabytes: nothing may unwind across the C ABI)
        rc = pack_scene(desc, ps, err, /*host_streams=*/false, staged_top);
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_CAPACITY, "out of host memory while building the scene tables");
    } catch (const std::exception& e) {
        return fail(RT_ERR_BAD_ARGUMENT, std::string("scene build failed: ") + e.what());
    }
    if (rc != RT_OK) return fail(rc, err);
    const auto t_packed = std::chrono::steady_clock::now();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(RT_ERR_BAD_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    rt_scene* s = new (std::nothrow) rt_scene;
    if (!s) return fail(RT_ERR_CAPACITY, "out of host memory");
    s->device = device;
    rc = noalloc_scratch_of(device, &s->scratch_ptr);  // set-up time, this device current
    if (rc == RT_OK) rc = upload(ps.recs, &s->k.recs, s->owned);
    if (rc == RT_OK) rc = upload(ps.order, &s->k.order, s->owned);
    if (rc == RT_OK) rc = upload(ps.gate, &s->k.gate, s->owned);
    if (rc == RT_OK) rc = upload(ps.bound, &s->k.bound, s->owned);
    if (rc == RT_OK && !ps.has_bvh) {
        rc = upload(ps.lnodes, &s->k.lnodes, s->owned);
        s->k.lcanon = s->k.lnodes, s->k.top = s->k.lnodes, s->k.top_k = 0u;
    }
    if (rc == RT_OK && ps.has_bvh) {
        // one allocation: the eight direction orders (laid out by the GPU), the canonical stream, the eight top regions
        const uint32_t n = static_cast<uint32_t>(ps.lnodes.size());
        const uint32_t* d_ret = nullptr;
        LNode* base = nullptr;
        if (ps.n_top >= n || (ps.lnodes[ps.n_top].ref >> kLRefShift) != REC_NODE || ps.top.size() != 8u * kTopK) rc = fail(RT_ERR_BAD_ARGUMENT, "internal: the hierarchy has no root record");
        if (rc == RT_OK) rc = upload(ps.portal_ret, &d_ret, s->owned);
        if (rc == RT_OK) {
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), (9 * static_cast<size_t>(n) + 8u * kTopK) * sizeof(LNode));
            if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(streams)");
            else s->owned.push_back(base);
        }
        if (rc == RT_OK) {
            LNode* canon = base + 8 * static_cast<size_t>(n);
            hipError_t e = hipMemcpy(canon, ps.lnodes.data(), static_cast<size_t>(n) * sizeof(LNode), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(canon + n, ps.top.data(), 8u * kTopK * sizeof(LNode), hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(lane_streams_kernel, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, canon, n, ps.n_top, d_ret, base);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) rc = hip_fail(e, "lane_streams_kernel");
            s->k.lnodes = base, s->k.lcanon = canon, s->k.top = canon + n, s->k.top_k = ps.top_k;
        }
    }
    if (rc == RT_OK) rc = upload(ps.gates, &s->k.gates, s->owned);
    if (rc == RT_OK) rc = upload(ps.extras, &s->k.extras, s->owned);
    if (rc == RT_OK) rc = upload(ps.mats, &s->k.mats, s->owned);
    if (rc == RT_OK) rc = upload(ps.pats, &s->k.pats, s->owned);
    if (rc == RT_OK) rc = upload(ps.lights, &s->k.lights, s->owned);
    if (rc == RT_OK) rc = upload(ps.xprims, &s->k.xprims, s->owned);
    if (rc == RT_OK) rc = upload(ps.xextras, &s->k.xextras, s->owned);
    if (rc == RT_OK) rc = upload(ps.xior, &s->k.xior, s->owned);
    if (rc == RT_OK) rc = upload(ps.xpats, &s->k.xpats, s->owned);
    if (rc == RT_OK) rc = upload(ps.xlights, &s->k.xlights, s->owned);
    if (rc == RT_OK) rc = upload(ps.xgates, &s->k.xgates, s->owned);
    s->has_f64 = ps.has_f64;
    s->k.n_items = ps.n_items;
    s->k.has_gates = ps.has_gates, s->k.has_bvh = ps.has_bvh;
    s->k.any_spawn = ps.any_spawn ? 1u : 0u;
    s->exotic = ps.exotic;
    s->tree = ps.has_bvh || ps.has_gates;
    // Lane-mode walks (rt_kernel.h LANE) pay where the rays of a tile have little in common or the hierarchy is deep:
    // measured (profiles/r02_lane_mode.txt) 1 M triangles 10.1 -> 1.8 ms, 100 k 5.2 -> 2.9 ms, C4 (1152 records, glass)
    // 11.3 -> 9.9 ms, but obj_mesh (1170 records, opaque) 0.40 -> 0.59 ms and 10 k triangles 1.6 -> 1.7 ms.
    s->spawn = ps.any_spawn;
    s->ordered = ps.any_glass || (std::getenv("RT_AMD_ORDER_ALL") && (ps.any_spawn || ps.has_area));
    s->lane = ps.has_bvh && (ps.n_items > RT_LANE_BIG_ITEMS || (ps.n_items > RT_LANE_MIN_ITEMS && ps.any_glass));
    if (const char* force = std::getenv("RT_AMD_LANE")) s->lane = ps.has_bvh && force[0] == '1';  // timing experiments
    s->area = ps.has_area;
    s->k.n_lights = ps.n_lights;
    if (rc == RT_OK) {
        void* c = nullptr;
        hipError_t e = hipMalloc(&c, kCounters * sizeof(unsigned long long));
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(counters)");
        else s->owned.push_back(c), s->counters = static_cast<unsigned long long*>(c);
    }
    if (rc == RT_OK && (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess))
        rc = fail(RT_ERR_HIP, "hipEventCreate failed");
    if (rc != RT_OK) {
        rt_scene_destroy(s);
        return rc;
    }
    if (std::getenv("RT_AMD_TIMING")) {  // where the time of a large scene goes (scripts/soup_perf.py)
        const auto t_end = std::chrono::steady_clock::now();
        std::fprintf(stderr, "rt_scene_create: %u records, host tables %.1f ms (%u threads), device allocations + uploads + stream layout %.1f ms\n", ps.n_items,
                     std::chrono::duration<double, std::milli>(t_packed - t_start).count(), pack_threads(),
                     std::chrono::duration<double, std::milli>(t_end - t_packed).count());
    }
    *scene_out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void* p : s->owned) (void)hipFree(p);
    if (s->order_buf) (void)hipFree(s->order_buf);
    if (s->img) (void)hipFree(s->img);
    if (s->img8) (void)hipFree(s->img8);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

int rt_scene_variant(const rt_scene* s, uint32_t* flags_out) {
    if (!s || !flags_out) return fail(RT_ERR_BAD_ARGUMENT, "rt_scene_variant: null argument");
    *flags_out = (s->area ? RT_VARIANT_AREA : 0u) | (s->tree ? RT_VARIANT_TREE : 0u) | (s->exotic ? RT_VARIANT_EXOTIC : 0u) |
                 (s->has_f64 ? RT_VARIANT_F64 : 0u) | (s->lane ? RT_VARIANT_LANE : 0u) | ((s->spawn || (s->lane && s->tree)) ? RT_VARIANT_SPAWN : 0u);
    return RT_OK;
}

int rt_scene_read_streams(const rt_scene* s, void* out, size_t cap, size_t* bytes_out) {
    if (!s || !bytes_out) return fail(RT_ERR_BAD_ARGUMENT, "rt_scene_read_streams: null argument");
    const size_t n_rec = std::max<uint32_t>(s->k.n_items, 1u);
    const size_t bytes = (s->k.has_bvh ? 9 * n_rec + 8u * kTopK : n_rec) * sizeof(LNode);
    *bytes_out = bytes;
    if (!out) return RT_OK;
    if (cap < bytes) return fail(RT_ERR_CAPACITY, "rt_scene_read_streams: the buffer is too small");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(out, s->k.lnodes, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_output_rows(const rt_camera_desc* camera, const rt_options* options, uint32_t* rows_out) {
    if (!rows_out) return fail(RT_ERR_BAD_ARGUMENT, "rows_out is null");
    RenderPlan rp{};
    int rc = plan_render(camera, options, rp);
    if (rc != RT_OK) return rc;
    *rows_out = rp.out_rows;
    return RT_OK;
}

int rt_render_device(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, float* rgb_out_device,
                     void* stream) {
    if (!scene) return fail(RT_ERR_BAD_ARGUMENT, "scene is null");
    if (!rgb_out_device) return fail(RT_ERR_BAD_ARGUMENT, "output buffer is null");
    RenderPlan rp{};
    int rc = plan_render(camera, options, rp);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    return launch_render(scene, camera, options, rp, rgb_out_device, static_cast<hipStream_t>(stream), false);
}

int rt_render(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, float* rgb_out_host,
              rt_stats* stats_out) {
    if (!rgb_out_host) return fail(RT_ERR_BAD_ARGUMENT, "output buffer is null");
    return render_host(scene, camera, options, rgb_out_host, nullptr, stats_out, false);
}

int rt_render_u8(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, uint8_t* rgb8_out_host,
                 rt_stats* stats_out) {
    if (!rgb8_out_host) return fail(RT_ERR_BAD_ARGUMENT, "output buffer is null");
    return render_host(scene, camera, options, nullptr, rgb8_out_host, stats_out, false);
}

int rt_scene_reserve(rt_scene* scene, uint32_t hsize, uint32_t rows) {
    if (!scene) return fail(RT_ERR_BAD_ARGUMENT, "scene is null");
    if (hsize == 0 || rows == 0) return fail(RT_ERR_BAD_ARGUMENT, "rt_scene_reserve: null dimensions");
    if (static_cast<uint64_t>(hsize) * rows > 0x7FFFFFFFull) return fail(RT_ERR_CAPACITY, "image too large");
    HIP_TRY(hipSetDevice(scene->device));
    HIP_TRY(hipDeviceSynchronize());  // set-up time: nothing of this scene may be in flight
    const uint32_t n_tiles = ((hsize + 7u) / 8u) * ((rows + 7u) / 8u);
    int rc = reserve_order(scene, n_tiles, nullptr, false);
    if (rc == RT_OK) rc = reserve_image(scene, static_cast<size_t>(hsize) * rows * 3u, true);
    return rc;
}

int rt_render_stats(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, float* rgb_out_host,
                    rt_stats* stats_out) {
    if (!stats_out) return fail(RT_ERR_BAD_ARGUMENT, "stats_out is null");
    return render_host(scene, camera, options, rgb_out_host, nullptr, stats_out, true);
}

int rt_render_stats_noalloc(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options,
                            float* rgb_out_device, void* stream, rt_stats* stats_out, uint32_t* tile_rays_out_host) {
    if (!scene) return fail(RT_ERR_BAD_ARGUMENT, "scene is null");
    if (!rgb_out_device) return fail(RT_ERR_BAD_ARGUMENT, "output buffer is null");
    if (!stats_out) return fail(RT_ERR_BAD_ARGUMENT, "stats_out is null");
    RenderPlan rp{};
    int rc = plan_render(camera, options, rp);
    if (rc != RT_OK) return rc;
    const uint64_t n_tiles = static_cast<uint64_t>((camera->hsize + 7u) / 8u) * ((rp.out_rows + 7u) / 8u);
    if (n_tiles > RT_NOALLOC_MAX_TILES)
        return fail(RT_ERR_CAPACITY, "rt_render_stats_noalloc: " + std::to_string(n_tiles) + " tiles of 8x8 pixels exceed RT_NOALLOC_MAX_TILES (" +
                                         std::to_string(RT_NOALLOC_MAX_TILES) + "); render row bands (rt_options.tile_*) or use rt_render_stats");
    NoAllocScratch* const sp = scene->scratch_ptr;  // looked up in rt_scene_create, this scene's device
    if (!sp) return fail(RT_ERR_HIP, "rt_render_stats_noalloc: the scene has no scratch pointer");
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the counters start at zero (launch_render clears them on the stream); every tile of the map is
    // written by its wave, so the last call's map needs no clearing
    HIP_TRY(hipEventRecord(scene->ev0, st));
    rc = launch_render(scene, camera, options, rp, rgb_out_device, st, true, sp->counters, sp->tile_rays, false);  // (no tile order: it would allocate)
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(scene->ev1, st));
    unsigned long long c[kCounters] = {0};
    HIP_TRY(hipMemcpyAsync(c, sp->counters, sizeof(c), hipMemcpyDeviceToHost, st));
    if (tile_rays_out_host && n_tiles)
        HIP_TRY(hipMemcpyAsync(tile_rays_out_host, sp->tile_rays, n_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::memset(stats_out, 0, sizeof(*stats_out));
    (void)hipEventElapsedTime(&stats_out->kernel_ms, scene->ev0, scene->ev1);
    stats_out->pixels = static_cast<uint64_t>(rp.out_rows) * camera->hsize;
    stats_out->primary_rays = c[0], stats_out->reflection_rays = c[1], stats_out->refraction_rays = c[2];
    stats_out->shadow_rays = c[3], stats_out->shaded_hits = c[4], stats_out->light_cells = c[5];
    stats_out->pattern_lookups = c[6], stats_out->container_passes = c[7], stats_out->shadow_rays_cast = c[8];
    stats_out->exact_nodes = c[9];
    return RT_OK;
}

int rt_quantize_u8_device(const float* rgb_device, size_t n, uint8_t* out_device, int device, void* stream) {
    if (n == 0) return RT_OK;
    if (!rgb_device || !out_device) return fail(RT_ERR_BAD_ARGUMENT, "buffer is null");
    HIP_TRY(hipSetDevice(device));
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(quantize_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rgb_device, n, out_device);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_quantize_u8(const float* rgb_host, size_t n, uint8_t* out_host) {
    if (n == 0) return RT_OK;
    if (!rgb_host || !out_host) return fail(RT_ERR_BAD_ARGUMENT, "buffer is null");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    float* d_in = nullptr;
    uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_in), n * sizeof(float)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), n);
    if (e != hipSuccess) {
        (void)hipFree(d_in);
        return hip_fail(e, "hipMalloc");
    }
    int rc = RT_OK;
    e = hipMemcpy(d_in, rgb_host, n * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(H2D)");
    if (rc == RT_OK) rc = rt_quantize_u8_device(d_in, n, d_out, dev, nullptr);
    if (rc == RT_OK) {
        e = hipMemcpy(out_host, d_out, n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(D2H)");
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

// ---------------------------------------------------------------- one frame shared across processes
static_assert(sizeof(hipIpcMemHandle_t) == sizeof(rt_ipc_handle), "rt_ipc_handle must hold a hipIpcMemHandle_t");

int rt_frame_create(int device, size_t bytes, void** frame_out, rt_ipc_handle* handle_out) {
    if (!frame_out || !handle_out || bytes == 0) return fail(RT_ERR_BAD_ARGUMENT, "rt_frame_create: null argument or zero size");
    HIP_TRY(hipSetDevice(device));
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return hip_fail(e, "hipIpcGetMemHandle");
    }
    std::memcpy(handle_out->bytes, &h, sizeof(h));
    *frame_out = p;
    return RT_OK;
}

int rt_frame_open(int device, int owner_device, const rt_ipc_handle* handle, void** frame_out) {
    if (!handle || !frame_out) return fail(RT_ERR_BAD_ARGUMENT, "rt_frame_open: null argument");
    HIP_TRY(hipSetDevice(device));
    if (owner_device != device) {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, device, owner_device));
        if (!can) return fail(RT_ERR_UNSUPPORTED, "rt_frame_open: no peer access between the two devices");
        hipError_t pe = hipDeviceEnablePeerAccess(owner_device, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return hip_fail(pe, "hipDeviceEnablePeerAccess");
        (void)hipGetLastError();  // "already enabled" is sticky otherwise
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle->bytes, sizeof(h));
    void* p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *frame_out = p;
    return RT_OK;
}

int rt_frame_close(int device, void* frame) {
    if (!frame) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipIpcCloseMemHandle(frame));
    return RT_OK;
}

int rt_frame_destroy(int device, void* frame) {
    if (!frame) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(frame));
    return RT_OK;
}

int rt_frame_download(int device, const void* frame, void* host_out, size_t bytes) {
    if (!frame || !host_out) return fail(RT_ERR_BAD_ARGUMENT, "rt_frame_download: null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, frame, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"
