// rt_amd.hip — gfx950 render kernels + the C ABI of include/rt_amd.h.
//
// One wave64 renders one 8x8 pixel tile; one lane per pixel (camera.rs:183-205 parallelises by
// row, this by tile).  The reference's recursion (world.rs:24-133) is a bounded iterative ray
// stack per lane; every phase of the loop runs in wave lockstep:
//
//   trace      closest hit of each lane's current ray.  The scene is walked in ONE wave-uniform
//              order (world.objects order, groups in depth-first pre-order with skip links), so
//              primitive / node records arrive through scalar loads (s_load_dwordx16) and the
//              per-primitive `switch` never diverges; a lane that fails a group's box test simply
//              sits out that subtree, and the wave jumps over a subtree no lane entered.
//   shade      Phong + shadows per light.  A point light is one any-hit shadow ray per lane.  An
//              AREA light is sampled wave-cooperatively: the shading points of the wave are taken
//              one (8x8 = 64 cells) or several (4x4: four) at a time, ONE LIGHT CELL PER LANE; the
//              64 shadow rays of a pass leave the same point (coherent, early-exit together), the
//              visibility count is a ballot + popcount and the Phong sums are butterfly-reduced.
//              Lanes whose own pixel has nothing to shade still work for their neighbours, which
//              is what keeps the wave busy on deep glass paths (>= 98 % of all rays are area-light
//              shadow rays).
//   containers the n1/n2 pass of intersection.rs:71-103 without a sorted list, only for lanes that
//              hit a transparent material.
//   spawn      reflection / refraction children (world.rs:94-133); one continues in the lane, the
//              other goes to the lane's stack (private memory, touched once per two-child node).
//
// FP32 throughout; no MFMA (the work is not GEMM-shaped), no LDS staging yet: every table is read
// either wave-uniformly through the scalar cache or as one 64-B record per lane.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_math.h"
#include "rt_pack.h"

using namespace rtk;

#ifndef RT_EXP
#define RT_EXP 0  // timing experiments only (scripts/exp_variants.sh): knock out one phase of the shading pass
#endif

namespace {

struct KScene {
    const DPrim* recs;  // the traversal stream (rt_math.h)
    const DExtra* extras;
    const DMaterial* mats;
    const DPattern* pats;
    const DLight* lights;
    uint32_t n_items, n_lights;
};
struct KParams {
    KScene sc;
    DCamera cam;
    float* out;
    unsigned long long* counters;  // STATS variant only: 8 x u64
    uint32_t max_depth, jitter_mode, seed;
    uint32_t tile_rows, tile_first, tile_stride;
    uint32_t out_rows, tiles_x, n_tiles;
};

constexpr int kStackCap = RT_MAX_DEPTH;

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ uint32_t shfl_u(uint32_t v, int src) { return static_cast<uint32_t>(__shfl(static_cast<int>(v), src, 64)); }
__device__ __forceinline__ V3 shfl_v3(V3 v, int src) { return V3{shfl_f(v.x, src), shfl_f(v.y, src), shfl_f(v.z, src)}; }

// Wave-uniform walk over the record stream.  `live` lanes take part; visit(index, record) returns
// whether the lane wants to continue.  Loop control depends only on ballots and on record fields,
// so the index and the whole record stay in SGPRs (one s_load_dwordx16 per record); the record that
// follows in the stream is requested before the current one is processed, so its latency hides
// behind the intersection arithmetic (it is wasted only when a group subtree is jumped).
template <class Visit>
__device__ __forceinline__ void traverse(const KScene& sc, const Ray& ray, bool live, Visit&& visit) {
    const uint32_t n = sc.n_items;
    if (n == 0u) return;
    uint32_t i = 0;
    uint32_t skip = 0;  // per lane: sit out records below this index (a group whose box it missed)
    DPrim rec = sc.recs[0];
    for (;;) {
        if (!wave_any(live)) break;
        const uint32_t seq = i + 1u;
        const DPrim nxt = sc.recs[seq < n ? seq : i];
        const bool part = live && i >= skip;
        uint32_t to = seq;
        if (rec.type == REC_NODE) {
            const bool enter = part && node_hit(rec, ray.o, ray.d);
            if (part && !enter) skip = rec.material;
            if (!wave_any(enter)) to = rec.material;
        } else if (part) {
            live = visit(i, rec);
        }
        if (to >= n) break;
        if (to == seq) rec = nxt;
        else rec = sc.recs[to];
        i = to;
    }
}

// world.rs:73-92 as an any-hit query: occluded iff some intersection lies in (0, dist).
__device__ __forceinline__ bool occluded(const KScene& sc, const Ray& ray, float dist, bool live) {
    bool occ = false;
    traverse(sc, ray, live, [&](uint32_t prim, const DPrim& p) {
        intersect_prim(p, sc.extras, ray, ray.self == prim, [&](int, float t, float, float) {
            if (t > 0.0f && t < dist) occ = true;
        });
        return !occ;
    });
    return occ;
}

struct StackEntry {
    V3 o, d, w, q, dq;
    uint32_t self, node;
};

__device__ __forceinline__ uint32_t level_of(uint32_t node) { return 31u - static_cast<uint32_t>(__clz(static_cast<int>(node + 1u))); }

template <bool STATS>
// 4 waves per SIMD (<= 128 VGPRs): the kernel waits on scalar loads, cross-lane traffic and
// transcendental latency, and measured 35 % faster at 4 waves/SIMD with a few spilled registers than
// at 2 without (5 and more waves spill too much; profiles/r01_occupancy_sweep.txt).
__global__ __launch_bounds__(256, 4) void render_kernel(const KParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = uni((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave >= P.n_tiles) return;
    const KScene& sc = P.sc;

    const uint32_t tile_y = wave / P.tiles_x, tile_x = wave - tile_y * P.tiles_x;
    const uint32_t x = tile_x * 8u + (lane & 7u);
    const uint32_t yl = tile_y * 8u + (lane >> 3);  // row in this call's packed output
    const uint32_t band = yl / P.tile_rows;
    const uint32_t gy = (P.tile_first + band * P.tile_stride) * P.tile_rows + (yl - band * P.tile_rows);  // global row
    const bool valid = x < P.cam.hsize && yl < P.out_rows && gy < P.cam.vsize;

    const uint32_t pixel_key = jitter_pixel_key(P.seed, gy * P.cam.hsize + x);
    const float nlights = static_cast<float>(sc.n_lights);

    // current path node of this lane
    Ray cur;
    camera_ray(P.cam, x, gy, cur.o, cur.d);
    cur.self = kNoPrim;
    cur.q = v3(0.0f, 0.0f, 0.0f);
    cur.dq = v3(0.0f, 0.0f, 0.0f);
    V3 w = v3(1.0f, 1.0f, 1.0f);
    uint32_t node = 0;
    bool active = valid && sc.n_lights != 0;  // world.rs:44: a fold over no lights is BLACK
    V3 acc = v3(0.0f, 0.0f, 0.0f);

    StackEntry stack[kStackCap];
    int sp = 0;

    unsigned long long c_primary = active ? 1ull : 0ull, c_reflect = 0, c_refract = 0, c_shadow = 0, c_shaded = 0,
                       c_cells = 0, c_pattern = 0, c_container = 0;

    while (wave_any(active)) {
        // ---------------------------------------------------------------- trace (world.rs:24-41)
        uint32_t hit_prim = kNoPrim;
        float hit_t = INFINITY, hit_u = 0.0f, hit_v = 0.0f;
        traverse(sc, cur, active, [&](uint32_t prim, const DPrim& p) {
            intersect_prim(p, sc.extras, cur, cur.self == prim, [&](int, float t, float u, float v) {
                if (closer_hit(t, hit_t)) hit_t = t, hit_prim = prim, hit_u = u, hit_v = v;  // intersection.rs:105-120
            });
            return true;
        });
        const bool has_hit = active && hit_prim != kNoPrim;

        // ---------------------------------------------------------------- prepare (intersection.rs:40-69)
        HitInfo h{};
        DMaterial m{};
        V3 over = v3(0.0f, 0.0f, 0.0f), pc = v3(0.0f, 0.0f, 0.0f);
        if (has_hit) {
            const DPrim p = sc.recs[hit_prim];
            m = sc.mats[p.material];
            h = prepare_hit(p, sc.extras, cur, cur.self == hit_prim, hit_t, hit_u, hit_v);
            over = h.p + h.n * kEps;
            pc = pattern_color(sc.pats[m.pattern], h.q + h.dq);  // pattern at over_point (material.rs:127)
            // the reference looks the pattern up once per light (material.rs:127); Solid costs nothing
            if (STATS) c_shaded++, c_pattern += (sc.pats[m.pattern].type != PAT_SOLID) ? sc.n_lights : 0u;
        }

        // ---------------------------------------------------------------- shade (world.rs:43-58)
        V3 surface = v3(0.0f, 0.0f, 0.0f);
        for (uint32_t li = 0; li < sc.n_lights; ++li) {
            const DLight l = sc.lights[li];
            const V3 lc = v3(l.color[0], l.color[1], l.color[2]);
            float lit = 0.0f, sum_d = 0.0f, sum_s = 0.0f;
            if (l.type == RT_LIGHT_POINT) {
                // light.rs:182-188 + material.rs:138-159 with a single cell
                const V3 lp = v3(l.pos[0], l.pos[1], l.pos[2]);
                const V3 tl = lp - over;
                const float dist2 = dot(tl, tl);
                const float inv_dist = rsq_f(dist2);
                const float dist = dist2 * inv_dist;
                const bool cast = has_hit && dist2 > kEps * kEps;
                Ray sr;
                sr.o = over, sr.d = tl * inv_dist, sr.self = hit_prim, sr.q = h.q, sr.dq = h.dq;
                const bool occ = occluded(sc, sr, dist, cast);
                lit = occ ? 0.0f : 1.0f;
                if (has_hit) phong_cell(lp, over, h.n, h.eye, m.shininess, sum_d, sum_s);
                if (STATS && cast) c_shadow++;
                if (STATS && has_hit) c_cells++;
            } else {
                // light.rs:192-216 + material.rs:133-161, one light cell per lane
                const uint32_t cells = l.usteps * l.vsteps;
                const uint32_t g = cells >= 64u ? 64u : (cells <= 1u ? 1u : (1u << (32 - __clz(static_cast<int>(cells - 1u)))));
                const uint32_t glog = 31u - static_cast<uint32_t>(__clz(static_cast<int>(g)));
                const uint32_t slots = 64u >> glog;
                const uint32_t chunks = (cells + 63u) >> 6;
                const uint32_t slot = lane >> glog, sub = lane & (g - 1u);
                const unsigned long long gmask = g == 64u ? ~0ull : ((1ull << g) - 1ull);
                // this lane's cell of chunk 0 and its centre do not depend on the shading point
                const uint32_t cv0 = sub / l.usteps, cu0 = sub - cv0 * l.usteps;
                const V3 centre0 = point_on_light(l, cu0, cv0, 0.5f, 0.5f);
                unsigned long long need = __ballot(has_hit);
                while (need) {
                    int owner = 0;
                    uint32_t my_slot = 0, taken = 0;
                    bool owner_now = false;
                    for (uint32_t k = 0; k < slots && need; ++k) {
                        const int o = __ffsll(static_cast<long long>(need)) - 1;
                        need &= need - 1ull;
                        if (slot == k) owner = o;
                        if (static_cast<int>(lane) == o) my_slot = k, owner_now = true;
                        taken = k + 1;
                    }
                    const bool lane_on = slot < taken;
                    // the shading point of this lane's slot
                    const V3 o_over = shfl_v3(over, owner), o_n = shfl_v3(h.n, owner), o_eye = shfl_v3(h.eye, owner);
                    Ray sr;
                    sr.o = o_over;
                    sr.self = shfl_u(hit_prim, owner);
                    sr.q = shfl_v3(h.q, owner);
                    sr.dq = shfl_v3(h.dq, owner);
                    const float o_shin = shfl_f(m.shininess, owner);
                    const uint32_t o_key = shfl_u(pixel_key, owner), o_node = shfl_u(node, owner);
                    unsigned int my_lit = 0;
                    float my_d = 0.0f, my_s = 0.0f;
                    for (uint32_t ch = 0; ch < chunks; ++ch) {
                        const uint32_t cell = ch * 64u + sub;
                        const bool con = lane_on && cell < cells;
                        uint32_t cv = cv0, cu = cu0;
                        V3 centre = centre0;
                        if (ch != 0u) {  // lights with more than 64 cells only
                            cv = cell / l.usteps, cu = cell - cv * l.usteps;
                            centre = point_on_light(l, cu, cv, 0.5f, 0.5f);
                        }
                        V3 lp = centre;
                        if (P.jitter_mode == RT_JITTER_COUNTER) {
                            float ju, jv;
                            counter_jitter(o_key, o_node, li, cell, ju, jv);
                            lp = point_on_light(l, cu, cv, ju, jv);
                        }
                        const V3 tl = lp - o_over;
                        const float dist2 = dot(tl, tl);
                        const float inv_dist = rsq_f(dist2);
                        const float dist = dist2 * inv_dist;
                        const bool cast = con && dist2 > kEps * kEps;  // null vector => lit (world.rs:77-81)
                        sr.d = tl * inv_dist;
#if RT_EXP & 4
                        const bool occ = cast && dist > 1e30f;
#else
                        const bool occ = occluded(sc, sr, dist, cast);
#endif
                        const unsigned long long b = __ballot(con && !occ);
                        float ldn = 0.0f, spec = 0.0f;
#if !(RT_EXP & 2)
                        if (con) phong_cell(centre, o_over, o_n, o_eye, o_shin, ldn, spec);
#endif
                        const int lead = static_cast<int>(my_slot << glog);
#if RT_EXP & 1
                        const float rd = ldn, rs = spec;
#else
                        for (uint32_t s = 1; s < g; s <<= 1) {  // butterfly inside the slot's lane group
                            ldn += __shfl_xor(ldn, static_cast<int>(s), 64);
                            spec += __shfl_xor(spec, static_cast<int>(s), 64);
                        }
                        const float rd = shfl_f(ldn, lead), rs = shfl_f(spec, lead);
#endif
                        if (owner_now) {
                            my_lit += static_cast<unsigned int>(__popcll((b >> lead) & gmask));
                            my_d += rd;
                            my_s += rs;
                        }
                        if (STATS && cast) c_shadow++;
                        if (STATS && con) c_cells++;
                    }
                    if (owner_now) lit = static_cast<float>(my_lit), sum_d = my_d, sum_s = my_s;
                }
                lit *= l.inv_samples;  // light.rs:208
            }
            if (has_hit) {  // material.rs:127-161
                const V3 ec = pc * lc;
                const V3 shade = ec * (m.diffuse * sum_d) + lc * (m.specular * sum_s);
                surface = surface + ec * m.ambient + shade * (l.inv_samples * lit);
            }
        }
        if (has_hit) acc = acc + w * surface;

        // ---------------------------------------------------------------- children (world.rs:59-69,94-133)
        const uint32_t depth = P.max_depth - level_of(node);  // remaining recursion depth of this node
        bool want_refl = has_hit && depth > 0u && !(fabsf(m.reflectivity) <= kEps);
        bool want_refr = has_hit && depth > 0u && !(fabsf(m.transparency) <= kEps);
        const bool use_schlick = has_hit && (m.reflectivity * m.transparency) > 0.0f;
        float n1 = 1.0f, n2 = 1.0f;
        const bool need_n = has_hit && depth > 0u && m.transparency > 0.0f;
        if (wave_any(need_n)) {
            ContainerScan cs;  // intersection.rs:71-103 without the sorted list
            traverse(sc, cur, need_n, [&](uint32_t prim, const DPrim& p) {
                Hits hh;
                hh.t[0] = hh.t[1] = hh.t[2] = hh.t[3] = RT_NOHIT;
                intersect_prim(p, sc.extras, cur, cur.self == prim, [&](int slot, float t, float, float) { hh.t[slot] = t; });
                cs.add(hh, prim, p.ior, hit_prim, hit_t);
                return true;
            });
            if (need_n) {
                cs.finish(m.ior, n1, n2);
                if (STATS) c_container++;
            }
        }
        V3 rdir = v3(0.0f, 0.0f, 0.0f);
        if (want_refr && !refract_dir(h.eye, h.n, n1, n2, rdir)) want_refr = false;  // total internal reflection
        const float fr = use_schlick ? schlick(h.eye, h.n, n1, n2) : 0.0f;
        // The reference adds reflected + refracted colour once PER LIGHT (world.rs:44-69): the
        // recursion is traced once here and weighted by the number of lights.
        const float k_refl = nlights * m.reflectivity * (use_schlick ? fr : 1.0f);
        const float k_refr = nlights * m.transparency * (use_schlick ? (1.0f - fr) : 1.0f);

        if (has_hit && want_refl && want_refr) {
            if (sp < kStackCap) {
                StackEntry e;
                e.o = h.p - h.n * kEps, e.d = rdir, e.w = w * k_refr, e.q = h.q, e.dq = neg(h.dq);
                e.self = hit_prim, e.node = 2u * node + 2u;
                stack[sp++] = e;
                if (STATS) c_refract++;
            }
        }
        if (has_hit && want_refl) {
            cur.o = over;
            cur.d = cur.d - h.n * (2.0f * dot(cur.d, h.n));  // tuple.rs:121-123
            cur.self = hit_prim, cur.q = h.q, cur.dq = h.dq;
            w = w * k_refl;
            node = 2u * node + 1u;
            if (STATS) c_reflect++;
        } else if (has_hit && want_refr) {
            cur.o = h.p - h.n * kEps;
            cur.d = rdir;
            cur.self = hit_prim, cur.q = h.q, cur.dq = neg(h.dq);
            w = w * k_refr;
            node = 2u * node + 2u;
            if (STATS) c_refract++;
        } else if (active) {
            if (sp > 0) {
                const StackEntry e = stack[--sp];
                cur.o = e.o, cur.d = e.d, cur.self = e.self, cur.q = e.q, cur.dq = e.dq;
                w = e.w;
                node = e.node;
            } else {
                active = false;
            }
        }
    }

    if (valid) {
        float* o = P.out + (static_cast<size_t>(yl) * P.cam.hsize + x) * 3u;
        o[0] = acc.x, o[1] = acc.y, o[2] = acc.z;
    }
    if (STATS) {
        unsigned long long v[8] = {c_primary, c_reflect, c_refract, c_shadow, c_shaded, c_cells, c_pattern, c_container};
        for (int k = 0; k < 8; ++k) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0 && s) atomicAdd(P.counters + k, s);
        }
    }
}

__global__ void quantize_kernel(const float* __restrict__ rgb, size_t n, uint8_t* __restrict__ out) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) out[i] = quantize_u8(rgb[i]);  // canvas.rs:37-39
}

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

template <class T>
int upload(const std::vector<T>& v, const T** out, std::vector<void*>& owned) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    owned.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RT_OK;
}

struct RenderPlan {
    uint32_t max_depth, tile_rows, tile_first, tile_stride, out_rows;
};

int plan_render(const rt_camera_desc* cam, const rt_options* opt, RenderPlan& rp) {
    if (!cam) return fail(RT_ERR_BAD_ARGUMENT, "camera is null");
    if (cam->hsize == 0 || cam->vsize == 0) return fail(RT_ERR_BAD_ARGUMENT, "camera cannot have null dimensions");
    if (static_cast<uint64_t>(cam->hsize) * cam->vsize > 0x7FFFFFFFull) return fail(RT_ERR_CAPACITY, "image too large");
    rt_options o{};
    if (opt) o = *opt;
    if (o.flags != 0) return fail(RT_ERR_BAD_ARGUMENT, "options.flags must be 0");
    if (o.jitter_mode > RT_JITTER_COUNTER) return fail(RT_ERR_UNSUPPORTED, "unknown jitter mode");
    rp.max_depth = o.max_depth ? o.max_depth : 5u;
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

struct rt_scene {
    int device = 0;
    KScene k{};
    std::vector<void*> owned;
    unsigned long long* counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

int launch_render(rt_scene* s, const rt_camera_desc* cam, const rt_options* opt, const RenderPlan& rp, float* d_out,
                  hipStream_t stream, bool stats) {
    KParams P{};
    P.sc = s->k;
    P.cam = pack_camera(*cam);
    P.out = d_out;
    P.counters = s->counters;
    P.max_depth = rp.max_depth;
    P.jitter_mode = opt ? opt->jitter_mode : 0u;
    P.seed = opt ? opt->seed : 0u;
    P.tile_rows = rp.tile_rows, P.tile_first = rp.tile_first, P.tile_stride = rp.tile_stride;
    P.out_rows = rp.out_rows;
    P.tiles_x = (cam->hsize + 7u) / 8u;
    const uint32_t tiles_y = (rp.out_rows + 7u) / 8u;
    P.n_tiles = P.tiles_x * tiles_y;
    if (P.n_tiles == 0) return RT_OK;
    const uint32_t blocks = (P.n_tiles + 3u) / 4u;  // 4 waves (tiles) per 256-thread block
    if (stats) {
        HIP_TRY(hipMemsetAsync(s->counters, 0, 8 * sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(render_kernel<true>, dim3(blocks), dim3(256), 0, stream, P);
    } else {
        hipLaunchKernelGGL(render_kernel<false>, dim3(blocks), dim3(256), 0, stream, P);
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int render_host(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, float* rgb_out_host,
                rt_stats* stats_out, bool stats) {
    if (!scene) return fail(RT_ERR_BAD_ARGUMENT, "scene is null");
    RenderPlan rp{};
    int rc = plan_render(camera, options, rp);
    if (rc != RT_OK) return rc;
    if (!rgb_out_host && !stats) return fail(RT_ERR_BAD_ARGUMENT, "output buffer is null");
    HIP_TRY(hipSetDevice(scene->device));
    const size_t n = static_cast<size_t>(rp.out_rows) * camera->hsize * 3u;
    float* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n ? n * sizeof(float) : sizeof(float)));
    hipStream_t stream = nullptr;
    (void)hipEventRecord(scene->ev0, stream);
    rc = launch_render(scene, camera, options, rp, d_out, stream, stats);
    (void)hipEventRecord(scene->ev1, stream);
    if (rc == RT_OK) {
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
    }
    if (rc == RT_OK && rgb_out_host && n) {
        hipError_t e = hipMemcpy(rgb_out_host, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(D2H)");
    }
    if (rc == RT_OK && stats_out) {
        std::memset(stats_out, 0, sizeof(*stats_out));
        (void)hipEventElapsedTime(&stats_out->kernel_ms, scene->ev0, scene->ev1);
        stats_out->pixels = static_cast<uint64_t>(rp.out_rows) * camera->hsize;
        if (stats) {
            unsigned long long c[8] = {0};
            hipError_t e = hipMemcpy(c, scene->counters, sizeof(c), hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(counters)");
            stats_out->primary_rays = c[0], stats_out->reflection_rays = c[1], stats_out->refraction_rays = c[2];
            stats_out->shadow_rays = c[3], stats_out->shaded_hits = c[4], stats_out->light_cells = c[5];
            stats_out->pattern_lookups = c[6], stats_out->container_passes = c[7];
        }
    }
    (void)hipFree(d_out);
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
    int rc = pack_scene(desc, ps, err);
    if (rc != RT_OK) return fail(rc, err);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(RT_ERR_BAD_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    rt_scene* s = new (std::nothrow) rt_scene;
    if (!s) return fail(RT_ERR_CAPACITY, "out of host memory");
    s->device = device;
    rc = upload(ps.recs, &s->k.recs, s->owned);
    if (rc == RT_OK) rc = upload(ps.extras, &s->k.extras, s->owned);
    if (rc == RT_OK) rc = upload(ps.mats, &s->k.mats, s->owned);
    if (rc == RT_OK) rc = upload(ps.pats, &s->k.pats, s->owned);
    if (rc == RT_OK) rc = upload(ps.lights, &s->k.lights, s->owned);
    s->k.n_items = ps.n_items;
    s->k.n_lights = ps.n_lights;
    if (rc == RT_OK) {
        void* c = nullptr;
        hipError_t e = hipMalloc(&c, 8 * sizeof(unsigned long long));
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(counters)");
        else s->owned.push_back(c), s->counters = static_cast<unsigned long long*>(c);
    }
    if (rc == RT_OK && (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess))
        rc = fail(RT_ERR_HIP, "hipEventCreate failed");
    if (rc != RT_OK) {
        rt_scene_destroy(s);
        return rc;
    }
    *scene_out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void* p : s->owned) (void)hipFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
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
    return render_host(scene, camera, options, rgb_out_host, stats_out, false);
}

int rt_render_stats(rt_scene* scene, const rt_camera_desc* camera, const rt_options* options, float* rgb_out_host,
                    rt_stats* stats_out) {
    if (!stats_out) return fail(RT_ERR_BAD_ARGUMENT, "stats_out is null");
    return render_host(scene, camera, options, rgb_out_host, stats_out, true);
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

}  // extern "C"
