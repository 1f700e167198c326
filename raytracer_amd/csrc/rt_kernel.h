// rt_kernel.h — the gfx950 device code of librt_amd.so (included by rt_amd.hip only).
//
// One wave64 renders one 8x8 pixel tile; one lane per pixel (camera.rs:183-205 parallelises by
// row, this by tile).  The reference's recursion (world.rs:24-133) is a bounded iterative ray
// stack per lane; every phase of the loop runs in wave lockstep:
//
//   trace      closest hit of each lane's current ray.  The scene is walked in ONE wave-uniform
//              order (the record stream of rt_math.h: world.objects order, groups in depth-first
//              pre-order with skip links), so records arrive through scalar loads and the
//              per-primitive `switch` never diverges; a lane that fails a group's box test sits
//              out that subtree, and the wave jumps over a subtree no lane entered.
//   prepare    hit point, normal, over point, pattern colour (intersection.rs:40-69).
//   containers the n1/n2 pass of intersection.rs:71-103 without a sorted list, only when a lane hit
//              a transparent material — done BEFORE shading so that the ray's origin and self data
//              are dead while the (register-hungry) shading loops run.
//   shade      per light (world.rs:43-58).  Point light: one any-hit shadow ray per lane.  AREA
//              light, three parts:
//                shafts       each lane tests the shaft from ITS shading point to the light
//                             (rt_math.h "Shafts") against the stream: records that provably stay
//                             outside cannot occlude whatever the jitter; a primitive that provably
//                             blocks the whole shaft settles the opposite answer.  A second walk
//                             over the records still flagged looks at the exact pyramid point ->
//                             parallelogram in the object space of each sphere or cube.  No possible
//                             blocker: fully lit without casting; blocked: lit = 0 without casting;
//                             whole light behind the surface: nothing to find out; otherwise the
//                             survivors form a record mask (streams of <= 64 records) for the passes.
//                Phong sums   over the cell CENTRES (material.rs:133-161), independent of the
//                             shadow rays: every lane for its own point, the cell quantities as
//                             forms in the cell indices (rt_math.h PhongGrid); the per-cell gates
//                             are dropped when bounds over the light show they cannot fire.
//                visibility   what is left (true penumbra, mostly).  A tile with many such points:
//                             every lane casts its own point's rays, cell after cell, the record
//                             under test in SGPRs (no cross-lane traffic, no memory access per ray).
//                             A tile with few of them is wave-cooperative: the wave's
//                             shading points are taken one (>= 64 cells) or several (16 cells: four)
//                             at a time, ONE LIGHT CELL PER LANE; the shadow rays of a pass leave the
//                             same point (coherent, they exit together), the count is a ballot +
//                             popcount.  Lanes whose own pixel has nothing to shade still work for
//                             their neighbours, which keeps the wave full on deep glass paths.
//   children   reflection / refraction rays (world.rs:94-133); one continues in the lane, the other
//              goes to the lane's stack (private memory, touched once per two-child node).
//
// FP32 throughout; no MFMA (nothing here is GEMM-shaped).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/rt_amd.h"
#include "rt_exact.h"
#include "rt_math.h"

#ifndef RT_EXP
#define RT_EXP 0  // timing experiments only (scripts/exp_variants.sh): knock out one phase of the shading pass
// bits: 4 no occlusion test in the cooperative passes | 16 spheres never occlude | 32 no shaft tests |
// 64 no container pass | 128 no point-light shadow rays | 256 per-lane traversal | 512 Phong sums over one
// cell | 1024 no shadow casting for area lights | 2048 no pyramid walk | 8192 no lane-local casting |
// 16384 no shared Phong sums.  Every one of them renders a WRONG image except 2048, 8192 and 16384.
#endif
#ifndef RT_XEXP
#define RT_XEXP 0  // timing experiments on the f64 chain (WRONG images): 1 no selection for spawning nodes | 2 no f64 hit
// preparation | 4 no f64 pattern lookup | 8 no f64 refraction
#endif
#ifndef RT_BLOCK_THREADS
// One tile (one wave) per workgroup: tiles differ in cost by an order of magnitude, and with four to a
// workgroup a new workgroup had to wait for a free wave slot on EVERY SIMD of a CU and for the LDS of a
// whole finished workgroup (profiles/r01_block_size_v45.txt: C4 10.8 -> 9.0 ms, cover_image_area
// 2.01 -> 1.76 ms with 64 threads instead of 256).
#define RT_BLOCK_THREADS 64
#endif
#ifndef RT_MIN_BLOCKS
#define RT_MIN_BLOCKS 4  // waves per SIMD (__launch_bounds__' second argument in HIP; occupancy sweeps override it)
#endif
#ifndef RT_LOCAL_CAST_MIN
#define RT_LOCAL_CAST_MIN 16  // points of a tile that need shadow rays: from here on each lane casts its own
#endif
#ifndef RT_SHARED_PHONG_MAX
#define RT_SHARED_PHONG_MAX 8  // up to this many shading points in a wave: the lanes share out their Phong sums
#endif
#ifndef RT_PREFETCH
#define RT_PREFETCH 0  // request record i+1 while record i is processed (costs 16 SGPR copies per record)
#endif

#ifndef RT_NO_EMIT
#define RT_NO_EMIT 0  // experiments: 1 = no FixRecords (what writing them costs the shading loops), 2 = nor their LDS rows and private list
#endif
#ifndef RT_FIX_PARTS
#define RT_FIX_PARTS 3  // experiments: bit 0 = the f64 over point of a refined node is kept for the records, bit 1 = the private record list and its flush exist
#endif
#ifndef RT_GATE_NEAR
// Group boxes clipped near an edge go to f64 (rt_math.h gates_pass_near): 1 = for the winner of the nearest-hit search,
// 2 = in the shadow walks, 4 = in the second count.  1 and 2 only in the EXOTIC kernels: in the others their mere code cost
// C4 3-5 %, cover_image 0-5 % and cover_image_area 11-13 % (profiles/r03_group_boxes.txt) for a case their scenes cannot
// have without a cube's corner on its Group's box — a sphere does not reach the edges of a box around it.
#define RT_GATE_NEAR 7
#endif
#ifndef RT_NO_NEAR
#define RT_NO_NEAR 0  // experiments: 1 = the plain any-hit tests (what noting the near things costs)
#endif

namespace rtk {

constexpr uint32_t kBlockThreads = RT_BLOCK_THREADS;

// One (path node, light) pair whose shadow rays met a near thing in FP32 (rt_math.h occludes_prim_near): what it takes to
// count that light's unoccluded cells again, with f64 where it matters, and to add the difference to the pixel
// (render_kernel "unsure shadow rays").  Lives in the lane's private memory until the wave's path loop is over.
struct FixRecord {
    uint32_t pix;        // the pixel's lane (ray stealing: not always the lane that holds the record)
    uint32_t light;      // light index | refined << 31
    uint32_t node, key;  // the jitter stream of the node (rt_math.h counter_jitter)
    float k[3];          // colour per unoccluded cell: weight x (diffuse and specular sums) / samples^2
    float lit;           // the count the shading loop used
    uint32_t hit_prim;
    float spare[6];
    uint32_t scratch[5]; // (private bookkeeping of the list, rt_kernel.h `pend`)
    double ray[6];       // a node that was not refined (and is not the camera ray's): its f64 ray (origin, direction);
                         // a refined node: ray[0..2] = its f64 over point, every bit of it — a point whose shadow rays
                         // hang on the reference's own rounding noise (XHit::qedge) is decided 1e-16 from the surface
};
static_assert(sizeof(FixRecord) == 128, "FixRecord layout");

struct KScene {
    const DPrim* recs;      // the traversal stream (rt_math.h, rt_pack.h)
    const uint32_t* order;  // per record: world.objects position of the primitive (tie order)
    const uint32_t* gate;   // per record: innermost enclosing Group of the caller's tree, or kNoGate
    const DBound* bound;    // per record: world-space bounding ball (shaft tests)
    const LNode* lnodes;    // per record: box + skip link in 32 bytes (lane-mode walks): the eight direction orders (one stream without a hierarchy)
    const LNode* lcanon;    // the same in canonical order (walks whose enter test follows no ray)
    const LNode* top;       // the staged top of each direction order (rt_math.h kTopK), 8 x kTopK records
    const DGate* gates;
    const DExtra* extras;
    const DMaterial* mats;
    const DPattern* pats;
    const DLight* lights;
    // the caller's f64 tables (EXACT variants only; rt_exact.h): per record, per DExtra slot, per material, per pattern
    const XPrim* xprims;
    const XExtra* xextras;
    const double* xior;
    const XPattern* xpats;
    const XLight* xlights;  // per light (always there: the caller's f64 geometry, or the f32 values widened)
    const XGate* xgates;    // per Group, parallel to `gates` (always there: the caller's f64 boxes, or the f32 values widened)
    uint32_t n_items, n_lights;
    uint32_t has_gates, has_bvh;
    uint32_t any_spawn;  // some material is reflective or transparent (no path has a second node otherwise)
    uint32_t top_k;      // records in use per staged top; 0 = none (the walks start at record 0 of the stream)
};
struct KParams {
    KScene sc;
    DCamera cam;
    XCamera cam64;  // EXACT variants only
    float* out;
    unsigned long long* counters;  // STATS variant only: kCounters x u64
    uint32_t* tile_rays;           // STATS variant only, may be null: rays cast per 8x8 tile, raster order of this call's rows
    const uint32_t* tile_order;    // may be null: the tile slot each wave renders (expensive tiles first), else the wave's own index
    uint32_t* tile_cost;           // may be null: per tile (raster index), 10 x rounds of its path loop + visibility passes (the tile order of the next frame)
    uint32_t max_depth, jitter_mode, seed;
    uint32_t tile_rows, tile_first, tile_stride;
    uint32_t out_rows, tiles_x, n_tiles;
    uint32_t global_rows;  // RT_FLAG_GLOBAL_ROWS: `out` is a full frame, rows go to their global position
    uint32_t no_fix;  // experiments (RT_AMD_NO_FIX): the unsure shadow rays keep their FP32 answers
};

constexpr int kStackCap = 8;  // RT_MAX_DEPTH
#ifdef RT_PHASE_CLOCK  // experiment: + iterations of the path loop and s_memtime ticks per phase, summed over waves
constexpr int kCounters = 20;  // + second count: ticks, records, walk steps, the longest wave's ticks
#define RT_TICK(k)                                             \
    if (STATS) {                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ph[k] += now_ - t_last;                                \
        t_last = now_;                                         \
    }
#else
#define RT_TICK(k)
constexpr int kCounters = 10;
#endif
constexpr int kCountersBase = 10;  // primary, reflect, refract, shadow, shaded, cells, pattern, container, shadow rays CAST, nodes refined in f64

// the builtin keeps the predicate as a lane mask (s_and with exec); __ballot() goes through a VGPR
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return wave_ballot(p) != 0ull; }
__device__ __forceinline__ bool wave_all(bool p) { return wave_ballot(!p) == 0ull; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ uint32_t shfl_u(uint32_t v, int src) { return static_cast<uint32_t>(__shfl(static_cast<int>(v), src, 64)); }
__device__ __forceinline__ V3 shfl_v3(V3 v, int src) { return V3{shfl_f(v.x, src), shfl_f(v.y, src), shfl_f(v.z, src)}; }
// broadcast from a WAVE-UNIFORM lane index: v_readlane_b32, no LDS round trip
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int src) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), src)); }
__device__ __forceinline__ float rl_f(float v, int src) { return __uint_as_float(rl_u(__float_as_uint(v), src)); }
__device__ __forceinline__ V3 rl_v3(V3 v, int src) { return V3{rl_f(v.x, src), rl_f(v.y, src), rl_f(v.z, src)}; }

// Wave-uniform walk over the record stream.  `live` lanes take part; visit(index, record, self)
// returns whether the lane wants to continue; enter(node record) says whether the lane's ray can
// meet anything inside a node's box.  Loop control depends only on ballots and on record fields, so
// the index and the whole record stay in SGPRs (scalar loads); a lane that misses a node's box sits
// out its subtree (`skip`), and the wave jumps a subtree no lane entered.  At the primitive a
// lane's ray STARTS on (`self_prim`, rt_math.h "self"; kNoPrim = none) the visitor is called with
// self = true in a block of its own: the lanes of a cooperative shadow pass share that primitive,
// so the two variants of the intersection code almost never run in the same iteration.  With 4
// waves per SIMD the load latency is covered by the other waves; requesting record i+1 ahead
// (RT_PREFETCH) measured slower because of the 16 SGPR copies per record.
//
// traverse_masked visits exactly the PRIMITIVE records a wave-uniform bit mask names (bit min(i, 63)
// stands for record i), nodes are never looked at: flat streams are walked this way (all ones), and
// so are the shadow passes of an area light, whose shaft tests (rt_math.h) have cleared the bits of
// every record that cannot block.
template <bool TREE, class Visit>
__device__ __forceinline__ void traverse_masked(const KScene& sc, uint32_t self_prim, bool live, unsigned long long mask, Visit&& visit) {
    const uint32_t n = sc.n_items;
    if (!TREE) {  // flat streams are short: a plain loop, 32 mask bits (bit 31 = every record from 31 on)
        const uint32_t m32 = static_cast<uint32_t>(mask);
        uint32_t m = m32 & (n >= 31u ? 0x7FFFFFFFu : ((1u << n) - 1u));
        uint32_t i = 0;
        for (;;) {
            if (m) {
                i = static_cast<uint32_t>(__builtin_ctz(m));
                m &= m - 1u;
            } else {  // records from 31 on share bit 31
                i = i < 31u ? 31u : i + 1u;
                if (!(m32 >> 31) || i >= n) break;
            }
            if (!wave_any(live)) break;
            const DPrim rec = sc.recs[i];
            const bool on_self = live && i == self_prim;
            if (wave_any(on_self)) {
                if (on_self) live = visit(i, rec, std::true_type{});
            }
            if (live && !on_self) live = visit(i, rec, std::false_type{});
        }
        return;
    }
    auto one = [&](uint32_t i) {
        const DPrim rec = sc.recs[i];
        if (TREE && rec.type == REC_NODE) return;  // all-ones masks only
        const bool on_self = live && i == self_prim;
        if (wave_any(on_self)) {
            if (on_self) live = visit(i, rec, std::true_type{});
        }
        if (live && !on_self) live = visit(i, rec, std::false_type{});
    };
    unsigned long long m = mask & 0x7FFFFFFFFFFFFFFFull;
    while (m) {
        const uint32_t i = static_cast<uint32_t>(__builtin_ctzll(m));
        m &= m - 1ull;
        if (i >= n || !wave_any(live)) return;
        one(i);
    }
    if (mask >> 63)
        for (uint32_t i = 63; i < n; ++i) {
            if (!wave_any(live)) return;
            one(i);
        }
}
// Lane mode: every lane walks the hierarchy ON ITS OWN (per-lane loads of the 32-byte LNode records, divergent
// control), in the order laid out for the sign pattern of ITS ray's direction (rt_math.h LNode: near child
// first).  The union walk below pays for every record ANY lane's ray can meet — a few dozen for the coherent
// rays of a tile (camera rays, shadow rays towards one light), nearly the whole stream for the rays of deep
// glass paths, which leave a tile in 64 directions.  Each round a lane steps through up to RT_LANE_WALK boxes
// until it stands on a primitive whose own box its ray meets; then the lanes that found one fetch that
// primitive's 64-byte record and run the visitor together, so the (long) intersection code is not interleaved
// with box tests.  `stream` = the lane's LNode stream, `i` = where it starts in it.
#ifndef RT_LANE_MIN_ITEMS
#define RT_LANE_MIN_ITEMS 96     // streams up to this long are always walked by the wave as one (rt_amd.hip picks the variant)
#define RT_LANE_BIG_ITEMS 32768  // longer ones always by the lanes; in between: by the lanes if a material is transparent
#endif
#ifndef RT_LANE_WALK
#define RT_LANE_WALK 8
#endif
//
// The staged top (N1; rt_math.h kTopK) — an experiment that stays switched off (RT_LDS_TOP = 0): measured on C4 it wins
// 3.7 % when on against off in the same binary and the code that carries it costs 2.8 %, and the kernels at 5 waves a SIMD
// lose a wave to its LDS (profiles/r03_lds_top.txt).  With it a walk in a ray's own order starts in the TOP REGION of that
// order: indices above n are records of the region, a portal whose box the ray meets leads to the subtree's first child
// in memory, and the links that leave that subtree lead back into the region.  TOP = 1: the region is read from memory
// like the stream; TOP = 2: from LDS, where the wave copied all eight regions when it started (RT_TOP_FIRST: the steps
// inside the region go first — while some lane stands in the region only those lanes step).  TOP = 0: a stream from
// record 0, no region.
#ifndef RT_LDS_TOP
#define RT_LDS_TOP 0  // 1 = regions in memory; 2 = in LDS for the kernels whose occupancy 4 KB more a wave does not lower
#endif
#ifndef RT_TOP_FIRST
#define RT_TOP_FIRST 0
#endif
__device__ __forceinline__ LNode* staged_top() {
    __shared__ LNode s_top[8 * kTopK];
    return s_top;
}
// Record j of the staged tops, read as LDS (two ds_read_b128).  Through a plain pointer the compiler merges this load
// with the stream's load of the other branch into ONE load through a generic pointer — a flat_load, which is what the
// first measurements of N1 ran on (profiles/r03_lds_top.txt).
__device__ __forceinline__ LNode staged_top_record(uint32_t j) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const U4 LdsU4;
    LdsU4* t = (LdsU4*)staged_top();
    const U4 a = t[2u * j], b = t[2u * j + 1u];
    LNode nd;
    nd.inv[0] = __uint_as_float(a.x), nd.inv[1] = __uint_as_float(a.y), nd.inv[2] = __uint_as_float(a.z), nd.inv[3] = __uint_as_float(a.w);
    nd.inv[4] = __uint_as_float(b.x), nd.inv[5] = __uint_as_float(b.y), nd.next = b.z, nd.ref = b.w;
    return nd;
#else  // (the host pass of the compiler only parses this)
    return staged_top()[j];
#endif
}
template <int TOP, class Enter, class Visit>
__device__ __forceinline__ void traverse_lane(const KScene& sc, const LNode* stream, [[maybe_unused]] const LNode* top, [[maybe_unused]] uint32_t top_at,
                                              uint32_t self_prim, bool live, uint32_t i, Enter&& enter_node, Visit&& visit) {
    const uint32_t n = sc.n_items;
    bool go = live && (TOP == 0 ? i < n : i != n);
    while (wave_any(go)) {
        uint32_t cand = kNoPrim;
#pragma unroll 1
        for (int k = 0; k < RT_LANE_WALK; ++k) {
            bool mine = go && cand == kNoPrim;
            LNode nd;
            if (TOP == 2 && RT_TOP_FIRST && wave_any(mine && i > n)) {
                mine = mine && i > n;
                if (mine) nd = staged_top_record(top_at + (i - n - 1u));
            } else if (TOP == 2 && !RT_TOP_FIRST) {
                if (mine && i > n) nd = staged_top_record(top_at + (i - n - 1u));
                else if (mine) nd = stream[i];
            } else if (mine) {
                nd = *((TOP == 1 && i > n) ? top + (i - n - 1u) : stream + i);
            }
            if (mine) {
                const bool in = enter_node(i, nd);
                const uint32_t type = nd.ref >> kLRefShift;
                if constexpr (TOP == 0) {
                    const bool is_node = type == REC_NODE;
                    cand = (in && !is_node) ? (nd.ref & kLRefMask) : kNoPrim;
                    i = (is_node && !in) ? nd.next : i + 1u;
                    go = i < n;
                } else {
                    cand = (in && type < REC_NODE) ? (nd.ref & kLRefMask) : kNoPrim;
                    i = (in && type == REC_NODE) ? i + 1u : ((in && type == REC_PORTAL) ? (nd.ref & kLRefMask) : nd.next);
                    go = i != n;
                }
            }
            if (!wave_any(go && cand == kNoPrim)) break;
        }
        const bool pend = cand != kNoPrim;
        if (wave_any(pend)) {
            bool keep = true;
            const bool on_self = pend && cand == self_prim;
            if (wave_any(on_self)) {
                if (on_self) keep = visit(cand, sc.recs[cand], std::true_type{});
            }
            if (pend && !on_self) keep = visit(cand, sc.recs[cand], std::false_type{});
            go = go && keep;
        }
    }
}
__device__ __forceinline__ uint32_t octant_of(V3 d) { return (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u); }
// LANE (a kernel variant, chosen per scene: a hierarchy of more than RT_LANE_MIN_ITEMS records): every walk is a
// lane-mode walk from the root — in the order of the ray's own direction `dir` when OWN_ORDER (closest-hit and
// any-hit queries), in the canonical order otherwise (walks whose enter test does not follow a ray: container pass,
// shaft tests).  Without LANE: the union walk, records through scalar loads.  (A wave that switches from one to the
// other at run time measured slower than either: both loops with their inlined visitors in one kernel cost the
// union walk 20 % through register pressure, profiles/r02_lane_mode.txt.)
template <bool TREE, int LANE, bool OWN_ORDER, class Enter, class Visit>
__device__ __forceinline__ void traverse(const KScene& sc, uint32_t self_prim, bool live, V3 dir, Enter&& enter_node, Visit&& visit) {
    const uint32_t n = sc.n_items;
    if (n == 0u) return;
    if (!TREE) {
        traverse_masked<false>(sc, self_prim, live, ~0ull, visit);
        return;
    }
    if constexpr (LANE != 0) {
        if constexpr (OWN_ORDER) {
            const uint32_t oct = octant_of(dir);
            constexpr int kTop = RT_LDS_TOP == 0 ? 0 : LANE;
            traverse_lane<kTop>(sc, sc.lnodes + oct * n, sc.top + oct * kTopK, oct * kTopK, self_prim, live, (kTop && sc.top_k) ? n + 1u : 0u, enter_node, visit);
        } else {
            traverse_lane<0>(sc, sc.lcanon, nullptr, 0u, self_prim, live, 0u, enter_node, visit);
        }
        return;
    }
    uint32_t i = 0;
    uint32_t skip = 0;  // per lane: sit out records below this index (a node whose box it missed)
    DPrim rec = sc.recs[0];
    for (;;) {
        if (!wave_any(live)) break;
        const uint32_t seq = i + 1u;
#if RT_PREFETCH
        const DPrim nxt = sc.recs[seq < n ? seq : i];
#endif
        const bool part = live && i >= skip;
        uint32_t to = seq;
        if (TREE && rec.type == REC_NODE) {
            const bool enter = part && enter_node(i, rec);
            if (part && !enter) skip = rec.material;
            if (!wave_any(enter)) to = rec.material;
        } else {
            const bool on_self = part && i == self_prim;
            if (wave_any(on_self)) {
                if (on_self) live = visit(i, rec, std::true_type{});
            }
            if (part && !on_self) live = visit(i, rec, std::false_type{});
        }
        if (to >= n) break;
#if RT_PREFETCH
        if (to == seq) rec = nxt;
        else
#endif
            rec = sc.recs[to];
        i = to;
    }
}

// world.rs:73-92 as an any-hit query: occluded iff some intersection lies in (0, dist).
// NEAR (the kernels with the f64 chain): `unsure` also reports whether some test on the way was a near thing — a grazed
// sphere, cube edge, cylinder wall or rim, triangle edge, a Group's box clipped at a corner, the own sphere at the
// terminator (rt_math.h occludes_prim_near, gates_pass_near): this light's count at this node
// is then worked out again by the fix kernel.
template <bool EXOTIC, bool TREE, int LANE, bool NEAR>
__device__ __forceinline__ bool occluded(const KScene& sc, const Ray& ray, float dist, bool live, bool masked, unsigned long long mask, bool& unsure) {
    bool occ = false;
    [[maybe_unused]] bool near = false;
    auto visit = [&](uint32_t prim, const DPrim& p, auto self) {
            bool hit = false;
            if constexpr (decltype(self)::value) {
                if constexpr (NEAR) hit = occludes_self_near<EXOTIC>(p, sc.extras, ray, dist, near);
                else intersect_prim<EXOTIC>(p, sc.extras, ray, true, [&](int, float t, float, float) { hit |= in_range(t, dist); });
            } else if constexpr (NEAR) {
                hit = occludes_prim_near<EXOTIC>(p, sc.extras, ray.o, ray.d, dist, near);
            } else {
#if RT_EXP & 16
                hit = p.type == PRIM_SPHERE ? false : occludes_prim<EXOTIC>(p, sc.extras, ray.o, ray.d, dist);
#else
                hit = occludes_prim<EXOTIC>(p, sc.extras, ray.o, ray.d, dist);
#endif
            }
            if ((TREE && sc.has_gates) && hit) {  // group.rs:107
                if constexpr (NEAR && EXOTIC && (RT_GATE_NEAR & 2) != 0) hit = gates_pass_near(sc.gates, sc.gate[prim], ray.o, ray.d, near);
                else hit = gates_pass(sc.gates, sc.gate[prim], ray.o, ray.d);
            }
            occ = hit;
            return !occ;
        };
    if (!TREE || masked) {
        traverse_masked<TREE>(sc, ray.self, live, mask, visit);
    } else {
        V3 inv_d = v3(0.0f, 0.0f, 0.0f);
        if (sc.has_bvh) inv_d = inv_dir(ray.d);
        traverse<TREE, LANE, true>(sc, ray.self, live, ray.d, [&](uint32_t, const auto& node) { return bvh_node_hit(node, ray.o, inv_d, 0.0f, dist); }, visit);
    }
    if constexpr (NEAR) unsure = near;
    return occ;
}

// The any-hit walk of the second count ("unsure shadow rays"): one lane, one ray, the canonical stream with its skip links.
// FP32 soft tests (rt_math.h occludes_prim_soft) where they are sure; the reference's own f64 arithmetic (rt_exact.h
// xoccludes: world.rs:73-92) from the f64 over point to the f64 light position where they are not, and always for the
// primitive the point stands on.
template <bool EXOTIC>
__device__ __forceinline__ bool fix_occluded(const KScene& sc, uint32_t self_prim, V3 o, V3 d, float dist, D3 o64, D3 lp64, unsigned long long* steps = nullptr) {
    const ShadowWin w = shadow_windows(dist);
    const V3 inv_d = inv_dir(d);
    bool occ = false;
    for (uint32_t i = 0; i < sc.n_items && !occ;) {
        if (steps) ++*steps;
        const DPrim p = sc.recs[i];
        if (p.type == REC_NODE) {
            i = bvh_node_hit(p, o, inv_d, w.lo_m, w.hi_m) ? i + 1u : p.material;
            continue;
        }
        ShadowTest st{true, false};
        if (i != self_prim) st = occludes_prim_soft<EXOTIC>(p, sc.extras, o, d, w);
        bool hit = st.maybe, gate_near = false;
        if (hit && sc.has_gates) hit = (RT_GATE_NEAR & 4) ? gates_pass_near(sc.gates, sc.gate[i], o, d, gate_near) : gates_pass(sc.gates, sc.gate[i], o, d);  // group.rs:107
        if (hit && !st.sure) hit = xoccludes<EXOTIC>(p.type, sc.xprims[i].inv, sc.xextras[p.extra].f, o64, lp64);
        if ((RT_GATE_NEAR & 4) && hit && gate_near) hit = xgates_pass_shadow(sc.gates, sc.xgates, sc.gate[i], o64, lp64);
        occ = hit;
        ++i;
    }
    return occ;
}

struct StackEntry {
    V3 o, d, w, q, dq;
    uint32_t self, node;
};

__device__ __forceinline__ uint32_t level_of(uint32_t node) { return 31u - static_cast<uint32_t>(__clz(static_cast<int>(node + 1u))); }

// The kernel waits on scalar loads, cross-lane traffic and transcendental latency: it measured 35 %
// faster at 4 waves/SIMD than at 2 early on (profiles/r01_occupancy_sweep.txt); what each
// instantiation runs at today is chosen by min_blocks_of below.
// EXOTIC = the scene holds cylinders or triangles (rt_math.h intersect_prim); TREE = its stream has
// hierarchy nodes and/or Group gates (rt_pack.h) — flat scenes run an instantiation without that code.
// The pixel of a lane: column x, row yl of this call's packed output, global row gy; false = outside.
__device__ __forceinline__ bool pixel_of_lane(const KParams& P, uint32_t order, uint32_t wave, uint32_t lane, uint32_t& x, uint32_t& yl, uint32_t& gy) {
    const uint32_t slot_y = wave / P.tiles_x, tile_x = wave - slot_y * P.tiles_x;
    // Tile rows are handed out from the middle of the image outwards (c, c-1, c+1, c-2, ...): the
    // expensive tiles (penumbrae, glass) are rarely at the top or bottom edge, and a tile whose 64
    // points all sit in two penumbrae runs ~0.1 ms on its own — started last it IS the frame time.
    uint32_t tile_y = slot_y;
    const uint32_t tiles_y = P.n_tiles / P.tiles_x;
    if (order == 1u) {
        const uint32_t c = (tiles_y - (tiles_y & 1u)) >> 1;
        tile_y = (slot_y & 1u) ? c - ((slot_y + 1u) >> 1) : c + (slot_y >> 1);
    } else if (order == 2u) {  // bottom row first
        tile_y = tiles_y - 1u - slot_y;
    } else if (order == 3u) {  // from the middle down, then from the middle up
        const uint32_t c = tiles_y >> 1;
        tile_y = slot_y < tiles_y - c ? c + slot_y : tiles_y - 1u - slot_y;
    }
    x = tile_x * 8u + (lane & 7u);
    yl = tile_y * 8u + (lane >> 3);
    const uint32_t band = yl / P.tile_rows;
    gy = (P.tile_first + band * P.tile_stride) * P.tile_rows + (yl - band * P.tile_rows);
    return x < P.cam.hsize && yl < P.out_rows && gy < P.cam.vsize;
}

// Waves per SIMD by instantiation, as measured (profiles/r01_occupancy_by_variant_v39.txt, _v43.txt):
// the kernels without area-light code take 5 (96 VGPRs, 12-31 spilled registers; C4 11.9 -> 10.6 ms,
// 6 waves are slower again); the area-light variants of flat streams and of streams with
// cylinders/triangles need about 140-170 VGPRs and run faster at 3 waves without spills than at 4 with
// 60-80 spilled registers; the area-light variant of plain hierarchies stays at 4.
#ifndef RT_MIN_BLOCKS_PLAIN
#define RT_MIN_BLOCKS_PLAIN 5
#endif
#ifndef RT_MIN_BLOCKS_PLAIN_X
#define RT_MIN_BLOCKS_PLAIN_X 4  // the same kernels with the f64 chain (EXACT): 128 VGPRs, fewer spills beside the f64 ray copies
// (profiles/r02_occupancy_exact.txt: C4 4K 8.4 -> 7.5 ms, cover_image 1.31 -> 1.20 ms; 3 and 6 are slower)
#endif
#ifndef RT_MIN_BLOCKS_FLAT_AREA
#define RT_MIN_BLOCKS_FLAT_AREA 3
#endif
constexpr int min_blocks_of(bool EXOTIC, bool TREE, bool AREA, bool EXACT, bool SPAWN = true) {
    // the flat area-light kernels of scenes without reflective or transparent materials (no stack, no children code)
    // fit 128 VGPRs with 5-31 spilled registers: 4 waves, C2 0.175 -> 0.159 ms (profiles/r02_nospawn.txt)
#ifndef RT_MIN_BLOCKS_NOSPAWN
#define RT_MIN_BLOCKS_NOSPAWN 4
#endif
    if (AREA && !TREE && !EXOTIC && !SPAWN) return RT_MIN_BLOCKS_NOSPAWN;
    // the area-light kernels of plain hierarchies with the f64 chain: 3 waves and 168 VGPRs (3 spilled) beat 4 waves and 128
    // (97 spilled) since the kernel arguments stopped taking spill lanes: cover_image_area 2.14 -> 2.05 ms
#ifndef RT_MIN_BLOCKS_TREE_AREA_X
#define RT_MIN_BLOCKS_TREE_AREA_X 3
#endif
    if (AREA && TREE && !EXOTIC && EXACT) return RT_MIN_BLOCKS_TREE_AREA_X;
    // the FP32-only flat area-light kernels with path code: a fourth wave pays there (C3 with RT_FLAG_F32_ONLY 0.379 -> 0.346 ms;
    // the same kernels WITH the f64 chain lose to it: C1 0.139 -> 0.150, profiles/r03_occupancy.txt)
    if (AREA && !TREE && !EXOTIC && !EXACT && SPAWN) return 4;
    return !AREA ? (EXACT ? RT_MIN_BLOCKS_PLAIN_X : RT_MIN_BLOCKS_PLAIN) : ((!TREE || EXOTIC) && RT_MIN_BLOCKS > RT_MIN_BLOCKS_FLAT_AREA ? RT_MIN_BLOCKS_FLAT_AREA : RT_MIN_BLOCKS);
}
// EXACT = the scene and the camera carry the caller's f64 tables: every path ray is carried a second time
// in f64 and the nodes that decide something are worked out again with the reference's formulas (rt_exact.h).
struct Ray64 {
    D3 o, d;
};
// LANE = every lane walks the hierarchy on its own (traverse above); needs TREE.
// SPAWN = some material is reflective or transparent; without it a path is its camera ray: no stack, no children,
// no container pass, one round.
#ifndef RT_XQ
#define RT_XQ 1  // experiments: 0 = rays leaving a refined hit keep the FP32 hit's object-space point
#endif
#ifndef RT_CONTAINER_F64
#define RT_CONTAINER_F64 1  // experiments: 0 = the container scan never asks f64 (before the stress fuzz family)
#endif
template <bool STATS, bool EXOTIC, bool TREE, bool AREA, bool EXACT, bool LANE = false, bool SPAWN = true>
__global__ __launch_bounds__(kBlockThreads, min_blocks_of(EXOTIC, TREE, AREA, EXACT, SPAWN)) void render_kernel(const KParams P) {
    static_assert(TREE || !LANE, "lane-mode walks need a hierarchy");
    // 2 = the staged tops live in LDS (RT_LDS_TOP = 2): the kernels with the f64 chain and no area-light code run 4 waves
    // a SIMD at 128 VGPRs with 6 KB of LDS a wave, and 4 KB more still fits 16 waves a CU; the others would lose a wave or
    // two a SIMD to it (the FP32-only hierarchy kernels 5 -> 3: 1 M triangles 1.88 -> 2.32 ms, profiles/r03_lds_top.txt)
    constexpr int kLane = LANE ? ((EXACT && !AREA && RT_LDS_TOP == 2) ? 2 : 1) : 0;
#ifdef RT_ORDER
    constexpr uint32_t kOrder0 = RT_ORDER;  // experiments
#else
    constexpr uint32_t kOrder0 = AREA ? 1u : 0u;
#endif
    const uint32_t kOrder = P.tile_order ? 0u : kOrder0;  // an explicit tile order names raster slots
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_id = uni((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave_id >= P.n_tiles) return;
    const uint32_t wave = P.tile_order ? uni(P.tile_order[wave_id]) : wave_id;  // the tile slot this wave renders
    uint32_t n_rounds = 0, n_visibility = 0;  // rounds of the path loop, points x light chunks that needed shadow rays: the tile's cost ("Tile order" below)
    const KScene& sc = P.sc;
    if constexpr (kLane == 2) {  // the staged tops of the eight direction orders: 4 KB, 64 bytes a lane
        static_assert(8 * kTopK * sizeof(LNode) == 64 * 4 * sizeof(uint4), "one wave copies the staged tops with four 16-byte loads a lane");
        if (sc.top_k) {
            uint4* dst = reinterpret_cast<uint4*>(staged_top());
            const uint4* src = reinterpret_cast<const uint4*>(sc.top);
            if (threadIdx.x < 64u) {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) dst[k * 64u + threadIdx.x] = src[k * 64u + threadIdx.x];
            }
            if (kBlockThreads > 64u) __syncthreads();
            else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }

    uint32_t x, yl, gy;
    const bool valid = pixel_of_lane(P, kOrder, wave, lane, x, yl, gy);

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
    // Unsure shadow rays (kernels with the f64 chain).  World::is_shadowed (world.rs:73-92) is a yes/no question, and FP32
    // cannot answer it for a ray that grazes an occluder, or that leaves a point whose place the reference's own rounding
    // decides (XHit::qedge); the f64 answer needs the node's f64 over point.  A shadow ray spawns nothing: its answer only
    // moves the count of unoccluded cells of ONE light at ONE node.  So the shading loops only NOTE that a light's rays met
    // a near thing (one compare per sphere test), and the lane then keeps a FixRecord for that (node, light) in its
    // private memory — the f64 over point, or the f64 ray to work it out from, and the colour one cell is worth.  When
    // the wave's path loop is over, the wave takes the records one by one: 64 lanes count that light's cells again, a
    // cell a lane, with soft tests + f64 (fix_occluded), and the difference goes to the pixel.  1e-4 .. 1e-3 of the nodes
    // leave a record.  (Forms that rendered the whole PIXEL again, in this kernel or in a second one, cost 30-140 % of
    // the frame; records in global memory for a second kernel cost 25-40 us a frame: profiles/r03_unsure_shadow_rays.txt.)
    // Ray stealing (kernels with path code): a lane whose pixel is finished takes over a waiting ray of another lane
    // of its wave (end of the path loop).  A lane then works for the pixel `pix` of another lane, so the colour of
    // a pixel is summed in LDS (rows 3..5 of the park area, indexed by the pixel's lane) instead of a register.
#ifndef RT_STEAL
#define RT_STEAL 1
#endif
    // Only the lane-mode kernels: in a union walk the mixed depths of stolen rays widen what the wave walks together
    // (cover_image 1.20 -> 1.65 ms with stealing), in lane mode every ray walks alone anyway (profiles/r02_ray_stealing.txt).
    constexpr bool kSteal = SPAWN && LANE && RT_STEAL != 0;
    static_assert(!kSteal || kBlockThreads == 64, "ray stealing exchanges through the wave's own park rows");
    uint32_t pix = lane;

    StackEntry stack[SPAWN ? kStackCap : 1];
    int sp = 0;
    // The f64 copies of the path rays live in private memory, written when a node spawns them and read
    // when a node is refined: [0] the child that continues in the lane, [1 + i] the ray of stack[i];
    // `cur_slot` says where the current ray's copy is.  The camera ray (node 0) is never stored: it is
    // worked out from the pixel when a refinement asks for it.  (Indexed dynamically, so the array never
    // competes for registers.)
    Ray64 r64[(EXACT && SPAWN) ? kStackCap + 1 : 1];
    uint32_t cur_slot = 0;
    // ... except slot 0 in the kernels with the f64 chain and no area-light code: they run 4 waves a SIMD with 24 park rows
    // and can afford 12 more (36 rows x 256 B x 16 waves = 147 KB a CU).  Every node whose child goes on in the lane writes
    // slot 0 and every refined node reads its slot: in private memory that was most of C4's 10.8 GB of L2<->fabric traffic.
#ifndef RT_R64_LDS
#define RT_R64_LDS 1
#endif
    // (Not the area-light kernels, although those at 3 waves a SIMD have the room — 47 rows x 256 B x 12 waves = 141 KB:
    // cover_image_area 2.05 -> 2.23 ms with it, 180 instead of 101 spilled SGPRs.)
    constexpr bool kR64Lds = EXACT && SPAWN && RT_R64_LDS != 0 && !AREA;
    constexpr int kR64Row = AREA ? 35 : 24;
    __shared__ float s_park[(kR64Row + (kR64Lds ? 12 : 0)) * kBlockThreads];
    float* const park = s_park + threadIdx.x;
    // (The LDS side goes through a pointer TYPED as LDS: with plain pointers the compiler folds "slot 0 ? LDS : private" into
    // one access through a generic pointer — a flat store inside the path loop, after which it no longer trusts the scene
    // tables to be unchanged and fetches every record of the union walks through vector loads: cover_image 1.33 -> 1.86 ms.)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) float LdsFloat;
    LdsFloat* const park64 = (LdsFloat*)park;
#else
    float* const park64 = park;
#endif
    auto r64_store = [&](uint32_t slot, D3 o, D3 d) {
        if (kR64Lds && slot == 0u) {
            const double v[6] = {o.x, o.y, o.z, d.x, d.y, d.z};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v[k]));
                park64[(kR64Row + 2 * k) * kBlockThreads] = __uint_as_float(static_cast<uint32_t>(b));
                park64[(kR64Row + 1 + 2 * k) * kBlockThreads] = __uint_as_float(static_cast<uint32_t>(b >> 32));
            }
        } else {
            r64[slot].o = o, r64[slot].d = d;
        }
    };
    auto r64_load = [&](uint32_t slot) {
        Ray64 r;
        if (kR64Lds && slot == 0u) {
            double v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const unsigned long long b = static_cast<unsigned long long>(__float_as_uint(park64[(kR64Row + 2 * k) * kBlockThreads])) |
                                             (static_cast<unsigned long long>(__float_as_uint(park64[(kR64Row + 1 + 2 * k) * kBlockThreads])) << 32);
                v[k] = __longlong_as_double(static_cast<long long>(b));
            }
            r.o = d3(v[0], v[1], v[2]), r.d = d3(v[3], v[4], v[5]);
        } else {
            r = r64[slot];
        }
        return r;
    };
    // EXACT: the FixRecords of this lane ("unsure shadow rays" above), until the wave's path loop is over;
    // pend[0].scratch[0] counts them (what does not fit keeps its FP32 count).  Private memory like the stack: indexed
    // dynamically, never in registers.
#ifndef RT_PEND
#define RT_PEND 2  // (not 1: a one-element array is indexed by a constant and moves into 32 registers — C4 7.4 -> 9.7 ms)
#endif
    constexpr uint32_t kPend = RT_PEND;
    FixRecord pend[(EXACT && RT_NO_EMIT < 2 && (RT_FIX_PARTS & 2)) ? kPend : 1];
    if (EXACT && RT_NO_EMIT < 2 && (RT_FIX_PARTS & 2)) pend[0].scratch[0] = 0u;
    // ... and, in six more words no record uses (pend[0].scratch[1..4], pend[1].scratch[0..1]), the f64 over point of the
    // current node if the node was refined (private as well: three LDS rows for its middle bits cost cover_image_area a
    // wave per CU, 2.0 -> 2.15 ms)
    // The f64 camera (30 dwords of kernel arguments) and the pointers of the f64 tables are read WHERE they are used —
    // refined nodes, second counts — through an address the compiler cannot move out of the path loop (n_rounds never reaches
    // 2^31, which it cannot know): read as plain `P.cam64`, `sc.xprims` their loads were hoisted to the kernel's entry, and
    // with no scalar registers to hold them they went straight into lanes of a spill VGPR (v_writelane at entry, v_readlane
    // at every use; C2 0.174 -> 0.165 ms for the camera alone).
#ifndef RT_LATE_CAM64
#define RT_LATE_CAM64 1
#endif
    auto args_here = [&]() -> const KParams& {  // the kernel arguments, read where the reader stands
        if (!RT_LATE_CAM64) return P;
        return *reinterpret_cast<const KParams*>(reinterpret_cast<const char*>(&P) + (n_rounds >> 31));
    };
    auto cam64_here = [&]() -> const XCamera& { return args_here().cam64; };
    auto cur64 = [&]() {
        Ray64 r;
        if (node == 0u) {
            // (No inline asm here, unlike the final store's recomputation: inside the path loop an
            // `asm volatile` counts as a possible write to memory, the compiler then no longer treats the
            // scene tables as unclobbered, and every record of the walks comes through a VECTOR load instead of
            // a scalar one — C4 9 -> 20 ms, profiles/r02_exact_chain_smem_regression.txt.)
            uint32_t px, pyl, pgy;
            pixel_of_lane(args_here(), kOrder, wave, threadIdx.x & 63u, px, pyl, pgy);
            xcamera_ray(cam64_here(), px, pgy, r.o, r.d);
        } else {
            r = r64_load(cur_slot);
        }
        return r;
    };
    // slots 24..33 (area-light variants): over point, (q, dq) and primitive of the hit, read back per
    // lane and, in the cooperative passes, from the owner's slots (an LDS broadcast)
    // slot 34: the lane's jitter key (constant for the whole kernel)
    constexpr bool kFix = EXACT && RT_NO_EMIT < 2;
    if (AREA) park[34 * kBlockThreads] = __uint_as_float(pixel_key);
    if (kSteal) park[3 * kBlockThreads] = 0.0f, park[4 * kBlockThreads] = 0.0f, park[5 * kBlockThreads] = 0.0f;
    float* const pixel_sum = s_park + (threadIdx.x & ~63u);  // + (3 + channel) * kBlockThreads + pix

    unsigned long long c_primary = active ? 1ull : 0ull, c_reflect = 0, c_refract = 0, c_shadow = 0, c_shaded = 0,
                       c_cells = 0, c_pattern = 0, c_container = 0, c_cast = 0, c_refined = 0;

#ifdef RT_PHASE_CLOCK
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    while (wave_any(active)) {
        ++n_rounds;
#ifdef RT_PHASE_CLOCK
        ph[5]++;
#endif
        // ---------------------------------------------------------------- trace (world.rs:24-41)
        uint32_t hit_prim = kNoPrim;
        float hit_t = INFINITY, hit_u = 0.0f, hit_v = 0.0f;
        V3 inv_d = v3(0.0f, 0.0f, 0.0f);
        if ((TREE && sc.has_bvh)) inv_d = inv_dir(cur.d);
        // EXACT: what the search cannot settle in FP32 — the winner's own test was a near thing, or another
        // primitive lies within the window of the winner (or of t = 0) — is settled in f64 right after it.
        bool hit_unsure = false, refined = false;
        uint32_t alt_prim = kNoPrim, excl0 = kNoPrim, excl1 = kNoPrim;
        float alt_t = INFINITY;
        double t64 = 0.0, u64 = 0.0, v64 = 0.0;
        // EXACT: the ray leaves a point where the reference's rounding noise decides whether it meets the primitive it stands
        // on again at t ~ 1e-15 (XHit::qedge of its parent node, handed down in dq: rt_math.h with_noise_flag).  f64 is asked
        // about that primitive first: a crossing within 1e-9 of the start is the hit (random_stress:180, random_field:113,
        // random_stress:39); anything else, and the search runs again.
        [[maybe_unused]] const bool self_noise = EXACT && active && cur.self != kNoPrim && noise_flag(cur.dq.x);
        // (The search and the f64 selection are two calls each, the second one rare.)
        auto do_trace = [&](bool live) {
            auto trace_enter = [&](uint32_t, const auto& nd) {
                return bvh_node_hit(nd, cur.o, inv_d, 0.0f, hit_t + (EXACT ? amb_window(hit_t) : tie_window(hit_t)));
            };
            auto trace_visit = [&](uint32_t prim, const DPrim& p, auto self) {
                    if (EXACT && (prim == excl0 || prim == excl1)) return true;
                    // a candidate that is not the winner but within the window of it (the winner only ever gets
                    // nearer, so a candidate outside the window now stays outside): keep the nearest one
                    auto note_alt = [&](float t) {
                        if (t <= hit_t + amb_window(hit_t) && t < alt_t && t > -amb_window(t) && prim != hit_prim && prim != cur.self &&
                            (!(TREE && sc.has_gates) || gates_pass(sc.gates, sc.gate[prim], cur.o, cur.d)))
                            alt_prim = prim, alt_t = t;
                    };
                    auto emit = [&](int slot, float t, float u, float v) {
                        // intersection.rs:105-120: smallest t > 0, world.objects order among approximately equal ones
                        bool better = closer_hit(t, hit_t);
                        if (!better && (TREE && sc.has_bvh) && hit_prim != kNoPrim && ties_hit(t, hit_t)) better = sc.order[prim] < sc.order[hit_prim];
                        if (better && (TREE && sc.has_gates)) better = gates_pass(sc.gates, sc.gate[prim], cur.o, cur.d);  // group.rs:107
                        if (better) {
                            if (EXACT && !(RT_XEXP & 16)) {
                                // the old winner is the nearest loser so far, if it is within the new winner's window
                                if (hit_prim != kNoPrim && hit_prim != prim && hit_t <= t + amb_window(t)) alt_prim = hit_prim, alt_t = hit_t;
                                else if (alt_prim != kNoPrim && !(alt_t <= t + amb_window(t))) alt_prim = kNoPrim, alt_t = INFINITY;
                                hit_unsure = (slot & kUnsure) != 0;
                            }
                            hit_t = t, hit_prim = prim, hit_u = u, hit_v = v;
                        } else if (EXACT && !(RT_XEXP & 16)) {
                            note_alt(t);
                        }
                    };
                    if (EXACT && !(RT_XEXP & 16)) intersect_prim<EXOTIC>(p, sc.extras, cur, decltype(self)::value, emit, note_alt);
                    else intersect_prim<EXOTIC>(p, sc.extras, cur, decltype(self)::value, emit);
                    return true;
                };
            traverse<TREE, kLane, true>(sc, cur.self, live, cur.d, trace_enter, trace_visit);
        };
        // Selection in f64 among what FP32 could not order, and for every node that spawns children.  Returns
        // whether f64 found none of the candidates in front of the ray (then they are left out of a second search).
        auto do_select = [&](bool live, bool may_retry) {
            if (alt_prim != kNoPrim && hit_prim != kNoPrim && !(alt_t <= hit_t + amb_window(hit_t))) alt_prim = kNoPrim;  // too far behind to matter
            bool spawns = false;
            if (SPAWN && !(RT_XEXP & 1) && sc.any_spawn && live && hit_prim != kNoPrim) {
                const DMaterial m = sc.mats[sc.recs[hit_prim].material];
                spawns = P.max_depth > level_of(node) && (!(fabsf(m.reflectivity) <= kEps) || !(fabsf(m.transparency) <= kEps));
            }
            bool retry = false;
            if (live && (hit_unsure || alt_prim != kNoPrim || spawns)) {
                const Ray64 r = cur64();
                const uint32_t a = hit_prim, b = alt_prim;  // entered in world.objects order
                const bool a_first = b == kNoPrim || (a != kNoPrim && (TREE ? sc.order[a] < sc.order[b] : a < b));
                const uint32_t first = a_first ? a : b, second = a_first ? b : a;  // first is never kNoPrim here
                const DPrim pa = sc.recs[first];
                const XHits ha = xintersect<EXOTIC>(pa.type, args_here().sc.xprims[first].inv, args_here().sc.xextras[pa.extra].f, r.o, r.d);
                XHits hb = ha;
                uint32_t type_b = REC_NODE;
                if (second != kNoPrim) {  // rare: a second candidate
                    const DPrim pb = sc.recs[second];
                    type_b = pb.type;
                    hb = xintersect<EXOTIC>(pb.type, args_here().sc.xprims[second].inv, args_here().sc.xextras[pb.extra].f, r.o, r.d);
                }
                uint32_t pr = kNoPrim;
                // (EXOTIC) a near miss in FRONT of the winner that f64 does not confirm may hide another: a ray through the edge
                // two triangles share can be a hair outside BOTH in FP32, and one candidate is all the search keeps
                // (random_mesh:2).  The search then runs again without the one f64 has cleared.
                const bool alt_front = EXOTIC && may_retry && a != kNoPrim && b != kNoPrim && a != b && alt_t < hit_t - amb_window(hit_t);  // (a == b: a cylinder's rim reported before its cap was hit)
                const bool found = xselect<EXOTIC>(ha, first, pa.type, hb, second, type_b, second != kNoPrim, t64, pr, u64, v64) &&
                                   !(may_retry && self_noise && !(t64 < 1e-9));
                const bool hidden = found && alt_front && pr == a;
                if (found && !hidden) {
                    refined = true;
                    hit_prim = pr, hit_t = static_cast<float>(t64), hit_u = static_cast<float>(u64), hit_v = static_cast<float>(v64);
                } else {
                    retry = may_retry;
                    excl0 = hidden ? b : (self_noise ? kNoPrim : a), excl1 = (hidden || self_noise) ? kNoPrim : b;  // (self_noise: the primitive's far side may be the hit)
                    hit_prim = kNoPrim, hit_t = INFINITY, alt_prim = kNoPrim, alt_t = INFINITY, hit_unsure = false;
                }
            }
            return retry;
        };
        do_trace(active);
        if (EXACT) {
            // A winner inside a Group whose box the ray clips within FP32 reach of an edge (rt_math.h gates_pass_near: the
            // FP32 boxes let in a superset): the reference's own test with the caller's own box (group.rs:107 in f64,
            // rt_exact.h xgates_pass) says whether the ray sees it; if not, the search runs again without it.
            bool gate_out = false;
            if constexpr (TREE && EXOTIC && (RT_GATE_NEAR & 1) != 0) {
                bool ask = false;  // (asked of the winner after the search: carried through it, the flag costs the walks registers)
                if (sc.has_gates && active && hit_prim != kNoPrim) gates_pass_near(sc.gates, sc.gate[hit_prim], cur.o, cur.d, ask);
                if (sc.has_gates && wave_any(ask)) {  // rare
                    if (ask) {
                        const Ray64 r = cur64();
                        gate_out = !xgates_pass(sc.gates, args_here().sc.xgates, sc.gate[hit_prim], r.o, r.d);
                    }
                    if (gate_out) excl0 = hit_prim, hit_prim = kNoPrim, hit_t = INFINITY, alt_prim = kNoPrim, alt_t = INFINITY, hit_unsure = false;
                }
            }
            if (self_noise && !gate_out) hit_prim = cur.self, hit_t = 0.0f, hit_u = 0.0f, hit_v = 0.0f, hit_unsure = true, alt_prim = kNoPrim, alt_t = INFINITY;
            const bool retry = do_select(active && !gate_out, true) || gate_out;
            if (wave_any(retry)) {  // rare: a grazing FP32 hit that f64 does not confirm, a winner its Group hides
                do_trace(retry);
                do_select(retry, false);
            }
        }
        RT_TICK(0)
        const bool has_hit = active && hit_prim != kNoPrim;
        const uint32_t depth = P.max_depth - level_of(node);  // remaining recursion depth of this node

        // ---------------------------------------------------------------- prepare (intersection.rs:40-69)
        // What survives into shading: over point, normal, object-space (q, dq) of the hit, pattern
        // colour, material index.  eye = -cur.d and p = over - n*eps are recomputed where needed.
        V3 over = v3(0.0f, 0.0f, 0.0f), nrm = v3(0.0f, 0.0f, 0.0f), hq = v3(0.0f, 0.0f, 0.0f), hdq = v3(0.0f, 0.0f, 0.0f);
        V3 pc = v3(0.0f, 0.0f, 0.0f);
        uint32_t mi = 0;
        float n1 = 1.0f, n2 = 1.0f;
        uint32_t n1_prim = kNoPrim, n2_prim = kNoPrim;  // EXACT: the records n1 / n2 came from
        bool need_n = false, hit_inside = false, late = false;
        float m_ior = 1.0f;
        if (has_hit) {
            const DPrim p = sc.recs[hit_prim];
            mi = p.material;
            const DMaterial m = sc.mats[mi];
            const bool patterned = EXACT && sc.pats[m.pattern].type != PAT_SOLID;  // (else h.eq is not needed)
            const HitInfo h = prepare_hit<EXOTIC>(p, sc.extras, cur, cur.self == hit_prim, hit_t, hit_u, hit_v, patterned);
            over = h.p + h.n * kEps, nrm = h.n, hq = h.q, hdq = h.dq, hit_inside = h.inside;
            pc = pattern_color(sc.pats[m.pattern], h.q + h.dq);  // pattern at over_point (material.rs:127)
            // EXACT: a pattern lookup within FP32 reach of a boundary, or a local normal that was a near
            // decision (the edge of a cube), is decided by the f64 hit of the same primitive
            late = EXACT && !(RT_XEXP & 64) && !refined && (pattern_unsure(sc.pats[m.pattern], h.q + h.dq, h.eq) || h.edge);
            need_n = SPAWN && depth > 0u && m.transparency > 0.0f;
            m_ior = m.ior;
            // the reference looks the pattern up once per light (material.rs:127); Solid costs nothing
            if (STATS) c_shaded++, c_pattern += (sc.pats[m.pattern].type != PAT_SOLID) ? sc.n_lights : 0u;
        }

        // ---------------------------------------------------------------- containers (intersection.rs:71-103)
        if (!(RT_EXP & 64) && wave_any(need_n)) {
            ContainerScan cs;
            cs.order = TREE ? sc.order : nullptr;
            [[maybe_unused]] uint32_t amb_prim = kNoPrim;
            const uint32_t ord_hit = (TREE && has_hit) ? sc.order[hit_prim] : hit_prim;
            const V3 hp = madd(cur.d, hit_t, cur.o);
            const float hm = container_margin(hit_t, hp);
            [[maybe_unused]] const float amb_w = container_amb_window(hit_t, fabsf(dot(nrm, cur.d)));  // EXACT: what FP32 cannot order against the hit
            traverse<TREE, kLane, false>(
                sc, cur.self, need_n, cur.d,
                [&](uint32_t, const auto& nd) {
                    // crossings behind the origin count too (sorted list), those beyond the hit never
                    if (EXOTIC) return bvh_node_hit(nd, cur.o, inv_d, -INFINITY, hit_t + tie_window(hit_t));
                    return node_holds_point(nd, hp, hm);  // rt_math.h: spheres and cubes only
                },
                [&](uint32_t prim, const DPrim& p, auto self) {
                    Hits hh;
                    hh.t[0] = hh.t[1] = hh.t[2] = hh.t[3] = RT_NOHIT;
                    bool any = false;
                    auto crossing = [&](int slot, float t, float, float) { hh.t[slot & 3] = t, any = true; };
                    if constexpr (EXOTIC && decltype(self)::value) {
                        // the triangle the ray starts on: its crossing behind the origin counts (the ray is inside the mesh)
                        // also when FP32 puts it a hair past an edge — the parent's hit on it stood (random_mesh:107)
                        intersect_prim<EXOTIC>(p, sc.extras, cur, true, crossing, [&](float t) {
                            if (p.type >= PRIM_TRIANGLE) hh.t[0] = t, any = true;
                        });
                    } else {
                        intersect_prim<EXOTIC>(p, sc.extras, cur, decltype(self)::value, crossing);
                    }
                    if (any && (TREE && sc.has_gates)) any = gates_pass(sc.gates, sc.gate[prim], cur.o, cur.d);
                    // EXACT: the first primitive with a crossing FP32 cannot order against the hit waits for f64
                    // (if constexpr: the other kernels' visitor must not even capture what this reads)
                    if constexpr (EXACT && RT_CONTAINER_F64 != 0) {
                        if (any && refined && prim != hit_prim && amb_prim == kNoPrim && crossing_near_hit(hh, hit_t, amb_w)) amb_prim = prim, any = false;
                    }
                    if (any) cs.add(hh, prim == hit_prim, TREE ? sc.order[prim] : prim, p.ior, ord_hit, hit_t, prim);
                    return true;
                });
            if constexpr (EXACT && RT_CONTAINER_F64 != 0) {  // rare: a face shared with the hit one, a body touching it
                if (wave_any(amb_prim != kNoPrim) && amb_prim != kNoPrim) {
                    const Ray64 r = cur64();
                    const DPrim pa = sc.recs[amb_prim];
                    const XHits hx = xintersect<EXOTIC>(pa.type, args_here().sc.xprims[amb_prim].inv, args_here().sc.xextras[pa.extra].f, r.o, r.d);
                    cs.add(xplace(hx, t64, hit_t), false, TREE ? sc.order[amb_prim] : amb_prim, pa.ior, ord_hit, hit_t, amb_prim);
                }
            }
            if (need_n) {
                cs.finish(m_ior, ord_hit, n1, n2);
                if (EXACT) cs.finish_prims(hit_prim, ord_hit, n1_prim, n2_prim);
                if (STATS) c_container++;
            }
        }
        RT_TICK(1)
        // ---------------------------------------------------------------- the f64 hit (rt_exact.h), refined nodes
        // The reference's own hit point, normal, inside flag, pattern cell and child rays; the children go
        // straight to private memory (r64[1] reflection, r64[2] refraction).
        bool tir64 = false;
        [[maybe_unused]] bool qedge = false;  // XHit::qedge: the reference's own rounding noise decides what leaves this point
        if (EXACT && !(RT_XEXP & 2) && has_hit && (refined || late)) {
            const DPrim p = sc.recs[hit_prim];
            const DMaterial m = sc.mats[mi];
            const double* xinv = args_here().sc.xprims[hit_prim].inv;
            const Ray64 r = cur64();
            if (!refined) {  // (late: the same primitive's f64 hit; if f64 finds none the FP32 hit stands)
                const XHits ha = xintersect<EXOTIC>(p.type, xinv, args_here().sc.xextras[p.extra].f, r.o, r.d);
                uint32_t pr = kNoPrim;
                refined = xselect<EXOTIC>(ha, hit_prim, p.type, ha, kNoPrim, REC_NODE, false, t64, pr, u64, v64);
            }
          if (refined) {
            const XHit xh = xprepare<EXOTIC>(p.type, xinv, args_here().sc.xextras[p.extra].f, r.o, r.d, t64, u64, v64);
            over = to_v3(xh.over), nrm = to_v3(xh.normal), hit_inside = xh.inside;
            qedge = xh.qedge;
            if constexpr (kFix && (RT_FIX_PARTS & 1)) {  // (the f64 over point travels to the second count whole)
                pend[0].scratch[1] = static_cast<uint32_t>(__double2loint(xh.over.x)), pend[0].scratch[2] = static_cast<uint32_t>(__double2hiint(xh.over.x));
                pend[0].scratch[3] = static_cast<uint32_t>(__double2loint(xh.over.y)), pend[0].scratch[4] = static_cast<uint32_t>(__double2hiint(xh.over.y));
                pend[1].scratch[0] = static_cast<uint32_t>(__double2loint(xh.over.z)), pend[1].scratch[1] = static_cast<uint32_t>(__double2hiint(xh.over.z));
            }
            // (the rays that leave this point start at the f64 point: their "self" form must stand on the same point.  The
            // FP32 hit point of a grazing ray is 1e-6 / |n . d| along the surface from it — 1e-4 in random_stress:162 —
            // and a child that itself grazes a face turns a 1e-6 offset into 1e-5 of t, random_stress:33.  Costs 1.5-3 %;
            // doing it for grazing hits only is no cheaper and misses the second case: profiles/r02_random_worlds.txt)
            if (RT_XQ) hq = xsnap_q(xh), hdq = xform_vector(p.inv, nrm) * kEps;
            const DPattern pat = sc.pats[m.pattern];
            if (!(RT_XEXP & 4) && pat.type != PAT_SOLID) {
                double frac;
                const bool is_a = xpattern_is_a(pat.type, xinv, args_here().sc.xpats[m.pattern].inv, xh.over, frac);
                const V3 ca = v3(pat.a[0], pat.a[1], pat.a[2]), cb = v3(pat.b[0], pat.b[1], pat.b[2]);
                pc = pat.type == PAT_GRADIENT ? ca + (cb - ca) * static_cast<float>(frac) : (is_a ? ca : cb);
            }
            // the children's f64 rays go to where the children phase below expects them: the one that
            // continues in the lane to r64[0], a refraction that waits on the stack to r64[1 + sp]
            const bool refl = SPAWN && depth > 0u && !(fabsf(m.reflectivity) <= kEps);
            bool refr = SPAWN && depth > 0u && !(fabsf(m.transparency) <= kEps);
            D3 rd = d3(0.0, 0.0, 0.0);
            if ((RT_XEXP & 8) && refr) rd = to_d3(v3(0.1f, 0.2f, 0.3f));
            if (!(RT_XEXP & 8) && refr) {  // world.rs:109-130 with the f64 indices of the records the container pass named
                const double x1 = n1_prim == kNoPrim ? 1.0 : args_here().sc.xior[sc.recs[n1_prim].material];
                const double x2 = n2_prim == kNoPrim ? 1.0 : args_here().sc.xior[sc.recs[n2_prim].material];
                tir64 = !xrefract(xh, x1, x2, rd);
                refr = !tir64;
            }
            if (refr) r64_store((refl && sp < kStackCap) ? 1u + static_cast<uint32_t>(sp) : 0u, xh.under, rd);
            if (refl) r64_store(0u, xh.over, xh.reflectv);
            if (STATS) c_refined++;
          }
        }
        RT_TICK(2)
        // from here on the ray's origin and self data are dead; its direction is kept (eye, reflection)
        // normal, eye vector and shininess are only needed by the Phong sums: parked as well
        auto ld3 = [&](int k) { return v3(park[k * kBlockThreads], park[(k + 1) * kBlockThreads], park[(k + 2) * kBlockThreads]); };
        park[17 * kBlockThreads] = nrm.x, park[18 * kBlockThreads] = nrm.y, park[19 * kBlockThreads] = nrm.z;
        park[20 * kBlockThreads] = -cur.d.x, park[21 * kBlockThreads] = -cur.d.y, park[22 * kBlockThreads] = -cur.d.z;

        // ---------------------------------------------------------------- shade (world.rs:43-58)
        float m_ambient = 0.0f, m_diffuse = 0.0f, m_specular = 0.0f, m_shininess = 1.0f;
        if (has_hit) {
            const DMaterial m = sc.mats[mi];
            m_ambient = m.ambient, m_diffuse = m.diffuse, m_specular = m.specular, m_shininess = m.shininess;
        }
        park[23 * kBlockThreads] = m_shininess;
        // material.rs:146-159: diffuse and specular are products with these two coefficients; when both
        // are zero no light sample and no shadow ray can change the pixel (ambient only)
        const bool shades = has_hit && (m_diffuse != 0.0f || m_specular != 0.0f);
        // What the light loops do not touch waits in LDS (slot k of this lane at park[k * kBlockThreads]): the
        // shading loops need every register they can get at 4 waves/SIMD, and a value parked here
        // costs no memory traffic, unlike a compiler spill to scratch.
        park[0 * kBlockThreads] = w.x, park[1 * kBlockThreads] = w.y, park[2 * kBlockThreads] = w.z;
        if (!kSteal) park[3 * kBlockThreads] = acc.x, park[4 * kBlockThreads] = acc.y, park[5 * kBlockThreads] = acc.z;
        park[6 * kBlockThreads] = n1, park[7 * kBlockThreads] = n2;
        park[8 * kBlockThreads] = pc.x, park[9 * kBlockThreads] = pc.y, park[10 * kBlockThreads] = pc.z;
        park[11 * kBlockThreads] = m_ambient, park[12 * kBlockThreads] = m_diffuse, park[13 * kBlockThreads] = m_specular;
        park[14 * kBlockThreads] = 0.0f, park[15 * kBlockThreads] = 0.0f, park[16 * kBlockThreads] = 0.0f;  // surface colour so far
        if constexpr (EXACT) hdq.x = with_noise_flag(hdq.x, qedge);  // (for the rays that leave this point: "self_noise")
        if (AREA) {
            park[24 * kBlockThreads] = over.x, park[25 * kBlockThreads] = over.y, park[26 * kBlockThreads] = over.z;
            park[27 * kBlockThreads] = hq.x, park[28 * kBlockThreads] = hq.y, park[29 * kBlockThreads] = hq.z;
            park[30 * kBlockThreads] = hdq.x, park[31 * kBlockThreads] = hdq.y, park[32 * kBlockThreads] = hdq.z;
            park[33 * kBlockThreads] = __uint_as_float(hit_prim);
        }
        // the hit's data as the light loops see it
        auto h_over = [&] { return AREA ? ld3(24) : over; };
        auto h_q = [&] { return AREA ? ld3(27) : hq; };
        auto h_dq = [&] { return AREA ? ld3(30) : hdq; };
        auto h_prim = [&] { return AREA ? __float_as_uint(park[33 * kBlockThreads]) : hit_prim; };
        // A FixRecord for a lane whose rays towards light `li` met a near thing.  It waits in the lane's PRIVATE memory until
        // the wave ends: a store to global memory inside the path loop would make the compiler reload the scene tables
        // after it, through vector loads (the trap of profiles/r02_exact_chain_smem_regression.txt: cover_image
        // 1.2 -> 2.2 ms with the records written here).
        [[maybe_unused]] auto emit_fix = [&](bool f, uint32_t li, float lit_count, float sum_d, float sum_s, V3 lc, float inv_samples) {
#ifndef RT_EMIT_HINT
#define RT_EMIT_HINT 1  // the record code is laid out as the unlikely side (cover_image_area 2.61 -> 2.40 ms)
#endif
            if constexpr (EXACT && !RT_NO_EMIT) {
                if (RT_EMIT_HINT ? __builtin_expect(f, false) : f) {
                    const uint32_t n = pend[0].scratch[0];
                    if (n < kPend) {
                        FixRecord& r = pend[n];
                        r.pix = pix;
                        r.light = li | (refined ? 0x80000000u : 0u);
                        r.node = node, r.key = AREA ? __float_as_uint(park[34 * kBlockThreads]) : 0u;
                        const V3 ec = v3(park[8 * kBlockThreads], park[9 * kBlockThreads], park[10 * kBlockThreads]) * lc;
                        const V3 shade = ec * (park[12 * kBlockThreads] * sum_d) + lc * (park[13 * kBlockThreads] * sum_s);
                        const V3 k = v3(park[0 * kBlockThreads], park[1 * kBlockThreads], park[2 * kBlockThreads]) * shade * (inv_samples * inv_samples);
                        r.k[0] = k.x, r.k[1] = k.y, r.k[2] = k.z, r.lit = lit_count;
                        r.hit_prim = h_prim();
                        if (refined) {
                            r.ray[0] = __hiloint2double(static_cast<int>(pend[0].scratch[2]), static_cast<int>(pend[0].scratch[1]));
                            r.ray[1] = __hiloint2double(static_cast<int>(pend[0].scratch[4]), static_cast<int>(pend[0].scratch[3]));
                            r.ray[2] = __hiloint2double(static_cast<int>(pend[1].scratch[1]), static_cast<int>(pend[1].scratch[0]));
                        } else if (node != 0u) {  // (its ray is still where it was: only a refined node writes its children's rays early;
                            const Ray64 rr = r64_load(cur_slot);  //  the camera ray is worked out from the pixel when the record leaves)
                            r.ray[0] = rr.o.x, r.ray[1] = rr.o.y, r.ray[2] = rr.o.z, r.ray[3] = rr.d.x, r.ray[4] = rr.d.y, r.ray[5] = rr.d.z;
                        }
                        pend[0].scratch[0] = n + 1u;
                    }
                }
            }
        };
        for (uint32_t li = 0; li < sc.n_lights; ++li) {
            const DLight l = sc.lights[li];
            const V3 lc = v3(l.color[0], l.color[1], l.color[2]);
            float lit = 0.0f, sum_d = 0.0f, sum_s = 0.0f;
            [[maybe_unused]] bool fix_li = false;  // EXACT: this lane's rays towards this light met a near thing
            if (!AREA || l.type == RT_LIGHT_POINT) {
                // light.rs:182-188 + material.rs:138-159 with a single cell
                const V3 lp = v3(l.pos[0], l.pos[1], l.pos[2]);
                const V3 p_over = h_over();
                const V3 tl = lp - p_over;
                const float dist2 = dot(tl, tl);
                const float inv_dist = rsq_f(dist2);
                const float dist = dist2 * inv_dist;
                const bool cast = has_hit && dist2 > kEps * kEps;  // null vector => lit (world.rs:77-81)
                if (shades) phong_cell(phong_point(p_over, ld3(17), ld3(20), park[23 * kBlockThreads]), lp, sum_d, sum_s);
                // a light behind the surface contributes 0 x intensity (material.rs:145,161): its shadow
                // ray cannot change the pixel and is answered without tracing
                const bool matters = sum_d != 0.0f || sum_s != 0.0f;
                Ray sr;
                sr.o = p_over, sr.d = tl * inv_dist, sr.self = h_prim(), sr.q = h_q(), sr.dq = h_dq();
                bool unsure = false;
                const bool occ = occluded<EXOTIC, TREE, kLane, EXACT && !RT_NO_NEAR>(sc, sr, dist, cast && matters && !(RT_EXP & 128), false, ~0ull, unsure);
                lit = occ ? 0.0f : 1.0f;
                if constexpr (EXACT) emit_fix((unsure || qedge) && cast && matters, li, lit, sum_d, sum_s, lc, 1.0f);
                if (STATS && cast) c_shadow++;
                if (STATS && cast && matters) c_cast++;
                if (STATS && has_hit) c_cells++;
            } else {
                // light.rs:192-216 + material.rs:133-161
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
                unsigned long long need = __ballot(shades);
                if (STATS && has_hit && !shades) c_shadow += cells, c_cells += cells;  // the reference casts them all the same

                // Shaft tests (rt_math.h): which records could occlude ANY ray from this lane's point
                // to this light, whatever the jitter.  Streams of up to 64 records: one bit per
                // primitive record; longer ones: nonzero = something might.  A point with none is
                // fully lit without casting.
                // (flat streams are short: a 32-bit mask per lane, widened for the walk, saves registers)
                using mask_t = typename std::conditional<TREE, unsigned long long, uint32_t>::type;
                constexpr uint32_t kTopBit = TREE ? 63u : 31u;
                const bool by_mask = !TREE || sc.n_items <= 64u;
                mask_t blockers = shades ? ~mask_t(0) : mask_t(0);
                bool umbra = false;  // some primitive occludes every ray from this lane's point to this light
                bool far_front = false, no_spec = false;  // what holds for ALL cells of this light (PhongBounds)
                unsigned long long behind_mask = 0ull;  // lanes whose light lies wholly below their tangent plane
                if (!(RT_EXP & 32) && wave_any(shades)) {
                    V3 lcen;
                    float lrad;
                    light_ball(l, lcen, lrad);
                    const V3 p_over = h_over();
                    const Shaft sh = make_shaft(p_over, lcen, lrad);
                    const int side = shaft_side_of_surface(sh, ld3(17), l);
                    const bool self_clear = !hit_inside && side > 0;
                    const bool walks = shades && side >= 0;  // else: Phong sums exactly zero, nothing to find out
                    behind_mask = wave_ballot(shades && side < 0);
                    const float p_dqy = h_dq().y;
                    const bool gateless = !(TREE && sc.has_gates);  // a Group's box may hide a primitive from a ray
                    {
                        const PhongBounds pb = phong_bounds(phong_point(p_over, ld3(17), ld3(20), park[23 * kBlockThreads]), sh.D, sh.dlen, sh.lr, sh.cosk, sh.wide);
                        far_front = pb.far_front, no_spec = pb.no_spec || park[13 * kBlockThreads] == 0.0f;  // no specular term at all
                    }
                    blockers = 0;
                    traverse<TREE, kLane, false>(
                        sc, h_prim(), walks, sh.D,
                        [&](uint32_t i, const auto&) {  // (lane mode asks for primitives too: the same ball test as the visitor's)
                            const DBound b = sc.bound[i];
                            return shaft_meets_ball(sh, v3(b.c[0], b.c[1], b.c[2]), b.r);
                        },
                        [&](uint32_t prim, const DPrim& p, auto self) {
                            bool mb;
                            if (p.type == PRIM_PLANE) {
                                // the origin's height exactly as the exact test sees it (its own plane: dq.y)
                                const float oy = decltype(self)::value
                                                     ? p_dqy
                                                     : fmaf(p.inv[4], p_over.x, fmaf(p.inv[5], p_over.y, fmaf(p.inv[6], p_over.z, p.inv[7])));
                                mb = shaft_meets_plane(sh, p.inv, oy);
                                if (!decltype(self)::value && mb && gateless) umbra |= shaft_blocked_by_plane(sh, p.inv, oy);
                            } else if (decltype(self)::value) {
                                mb = p.type >= PRIM_CYLINDER || !self_clear;  // convex shapes only
                                // a point inside its own sphere or cube (hit from within), the light outside it
                                if (hit_inside && p.type < PRIM_CYLINDER && gateless) {
                                    const DBound b = sc.bound[prim];
                                    umbra |= shaft_leaves_solid(sh, v3(b.c[0], b.c[1], b.c[2]), b.r);
                                }
                            } else {
                                const DBound b = sc.bound[prim];
                                mb = shaft_meets_ball(sh, v3(b.c[0], b.c[1], b.c[2]), b.r);
                                if (mb && p.type == PRIM_CUBE) mb = shaft_meets_cube(sh, p.inv);  // its three slabs
                                // ... or does it occlude EVERY ray of the shaft (the point is in its umbra)?
                                if (mb && gateless && b.r_in > 0.0f) umbra |= shaft_blocked_by_ball(sh, v3(b.c[0], b.c[1], b.c[2]), b.r_in, b.r);
                            }
                            if (mb) blockers |= mask_t(1) << (prim < kTopBit ? prim : kTopBit);
                            return by_mask || !mb;  // without a mask the first possible blocker settles it
                        });
                    // What the balls leave undecided: the pyramid from the point to the light's
                    // parallelogram itself, in the object space of each sphere or cube still flagged
                    // (rt_math.h ShaftObj) — a walk of its own, over the flagged records only.
                    if (by_mask && !(RT_EXP & 2048)) {
                        const uint32_t self_prim = h_prim();
                        // this lane's flagged records, without its own primitive and the shared top bit
                        mask_t rem = blockers & ~(mask_t(1) << kTopBit);
                        if (self_prim < kTopBit) rem &= ~(mask_t(1) << self_prim);
                        for (;;) {
                            if (umbra) rem = 0;
                            const unsigned long long todo = wave_ballot(rem != 0);
                            if (!todo) break;
                            const int first = __ffsll(static_cast<long long>(todo)) - 1;  // some lane with work: its lowest record
                            uint32_t i;
                            if (TREE) {
                                const uint32_t lo = rl_u(static_cast<uint32_t>(rem), first);
                                const uint32_t hi = rl_u(static_cast<uint32_t>(static_cast<unsigned long long>(rem) >> 32), first);
                                i = lo ? static_cast<uint32_t>(__builtin_ctz(lo)) : 32u + static_cast<uint32_t>(__builtin_ctz(hi));
                            } else {
                                i = static_cast<uint32_t>(__builtin_ctz(rl_u(static_cast<uint32_t>(rem), first)));
                            }
                            const bool mine = ((rem >> i) & mask_t(1)) != 0;
                            rem &= ~(mask_t(1) << i);
                            const DBound b = sc.bound[i];
                            if (!(b.margin > 0.0f)) continue;  // spheres and cubes only
                            const DPrim p = sc.recs[i];
                            if (mine) {
                                const ShaftObj so = shaft_in_object_space(p.inv, h_over(), l);
                                if (pyramid_misses(so, p.type == PRIM_CUBE, b.margin)) blockers &= ~(mask_t(1) << i);
                                else if (gateless) umbra |= pyramid_blocked(so, p.type == PRIM_CUBE, b.margin);
                            }
                        }
                    }
                }
                const bool open = shades && blockers == 0;
                if (open) lit = static_cast<float>(cells);

                // Phong sums over the cell centres, every lane for its own point, cells in the
                // reference's order (material.rs:133-136).
                const unsigned long long summing = wave_ballot(shades) & ~behind_mask;
                const uint32_t n_summing = static_cast<uint32_t>(__popcll(summing));
                // (only kernels with path code: without children no wave is down to a few deep-path points, and the headline
                // kernel is 2.5 % faster without this branch, 0.158 -> 0.154 ms)
                if (AREA && SPAWN && n_summing != 0u && n_summing <= RT_SHARED_PHONG_MAX && cells >= 16u && !(RT_EXP & 16384)) {
                    // Few points in the wave (deep glass paths): the lanes share them out.  Each point
                    // gets 64 / 2^ceil(log2 n) lanes, each of which sums a run of its cells from the
                    // point's parked data (an LDS broadcast); a butterfly adds the runs up.
                    uint32_t plog = 0;
                    while ((1u << plog) < n_summing) ++plog;
                    const uint32_t glog2 = 6u - plog, gl = 1u << glog2;  // lanes per point
                    const uint32_t rank = lane >> glog2, part = lane & (gl - 1u);
                    int whose = -1;
                    {
                        unsigned long long a = summing;
                        for (uint32_t k = 0; a; ++k) {
                            const int o = __ffsll(static_cast<long long>(a)) - 1;
                            a &= a - 1ull;
                            if (rank == k) whose = o;
                        }
                    }
                    const float* const op = s_park + (threadIdx.x & ~63u) + static_cast<uint32_t>(whose < 0 ? 0 : whose);
                    auto od3 = [&](int k) { return v3(op[k * kBlockThreads], op[(k + 1) * kBlockThreads], op[(k + 2) * kBlockThreads]); };
                    const PhongPoint pp = phong_point(od3(24), od3(17), od3(20), op[23 * kBlockThreads]);
                    const uint32_t per = (cells + gl - 1u) >> glog2;
                    uint32_t c = part * per;
                    uint32_t cv = c / l.usteps, cu = c - cv * l.usteps;
                    float part_d = 0.0f, part_s = 0.0f;
                    for (uint32_t j = 0; j < per; ++j, ++c) {
                        if (whose >= 0 && c < cells) {
                            float ldn, spec;
                            phong_cell(pp, point_on_light(l, cu, cv, 0.5f, 0.5f), ldn, spec);
                            part_d += ldn, part_s += spec;
                        }
                        if (++cu == l.usteps) cu = 0u, ++cv;
                    }
                    for (uint32_t st = 1; st < gl; st <<= 1) {
                        part_d += __shfl_xor(part_d, static_cast<int>(st), 64);
                        part_s += __shfl_xor(part_s, static_cast<int>(st), 64);
                    }
                    const uint32_t my_rank = static_cast<uint32_t>(__popcll(summing & ((1ull << lane) - 1ull)));
                    const float got_d = shfl_f(part_d, static_cast<int>(my_rank << glog2)), got_s = shfl_f(part_s, static_cast<int>(my_rank << glog2));
                    if ((summing >> lane) & 1ull) sum_d = got_d, sum_s = got_s;
                } else if (summing) {  // (a tile of faces turned away from the light has nothing to sum)
                    PhongPoint pp = phong_point(h_over(), ld3(17), ld3(20), park[23 * kBlockThreads]);
                    if (!shades) pp.gate = INFINITY;  // lanes without a shading point run along and are zeroed below
                    const PhongGrid pg = phong_grid(pp, l);
                    // the per-cell gates are dropped when they are known to pass (or fail) in every lane
                    const uint32_t pus = (RT_EXP & 512) ? 1u : l.usteps, pvs = (RT_EXP & 512) ? 1u : l.vsteps;
                    if (!wave_all(!shades || no_spec))
                        phong_grid_sums<true, true>(pp, pg, pus, pvs, sum_d, sum_s);
                    else if (!wave_all(!shades || far_front))
                        phong_grid_sums<true, false>(pp, pg, pus, pvs, sum_d, sum_s);
                    else
                        phong_grid_sums<false, false>(pp, pg, pus, pvs, sum_d, sum_s);
                    if (!shades || ((behind_mask >> lane) & 1ull)) sum_d = 0.0f, sum_s = 0.0f;
                }
                if (STATS && shades) c_cells += cells;
                {
                    // Every cell behind the surface (material.rs:145 gates both Phong terms): the sums
                    // are exactly zero and multiply the intensity (material.rs:161), so the shadow rays
                    // of this point cannot change the pixel.  They still count as rays of the
                    // reference's accounting (light.rs:196-208 casts them), answered without tracing.
                    const bool dark = shades && sum_d == 0.0f && sum_s == 0.0f;
                    // likewise the rays of a point whose shaft to the light is empty (`open`: lit = cells)
                    // or wholly blocked by one primitive (`umbra`: lit = 0)
                    if (STATS && (dark || open || umbra)) c_shadow += cells;
                    need &= ~__ballot(dark || open || umbra);
                }
                n_visibility += static_cast<uint32_t>(__popcll(need)) * chunks;  // (counted whether or not a cost table asks: no pointer to keep)

                if (RT_EXP & 1024) need = 0;
                // A tile with many such points (it lies in a penumbra): every lane casts its OWN
                // point's rays, cell after cell, record after record of the lanes' masks.  An
                // iteration has no cross-lane traffic and no memory access — the record under test
                // stays in SGPRs for all cells — so the 64 + 16 dependent passes of a tile inside two
                // penumbrae (~1 us each: LDS broadcast, jitter, scalar record load, test, ballot)
                // become 80 short iterations.  That chain of ONE wave was a fixed ~0.08 ms of a frame
                // however small the frame (profiles/r01_c2_size_sweep_v40.txt).  Few points: the
                // cooperative passes below cost less, they only run for the points that need them.
                if (by_mask && cells <= 64u && static_cast<uint32_t>(__popcll(need)) >= RT_LOCAL_CAST_MIN && !(RT_EXP & 8192)) {
                    const bool mine = ((need >> lane) & 1ull) != 0ull;
                    mask_t rem = mine ? blockers : mask_t(0);
                    if (!wave_any(((rem >> kTopBit) & mask_t(1)) != 0)) {  // (the shared top bit: cooperative passes)
                        Ray sr;
                        sr.o = h_over(), sr.q = h_q(), sr.dq = h_dq(), sr.self = h_prim();
                        sr.d = v3(0.0f, 0.0f, 0.0f);
                        const uint32_t key = __float_as_uint(park[34 * kBlockThreads]);
                        unsigned long long occ_bits = 0ull;  // this lane's cells found occluded so far
                        bool first = true;
                        [[maybe_unused]] bool band = false;  // EXACT: some cell met a near thing
                        for (;;) {
                            const unsigned long long todo = wave_ballot(rem != 0);
                            if (!todo) break;
                            const int lead = __ffsll(static_cast<long long>(todo)) - 1;  // some lane with work: its lowest record
                            uint32_t i;
                            if (TREE) {
                                const uint32_t lo = rl_u(static_cast<uint32_t>(rem), lead);
                                const uint32_t hi = rl_u(static_cast<uint32_t>(static_cast<unsigned long long>(rem) >> 32), lead);
                                i = lo ? static_cast<uint32_t>(__builtin_ctz(lo)) : 32u + static_cast<uint32_t>(__builtin_ctz(hi));
                            } else {
                                i = static_cast<uint32_t>(__builtin_ctz(rl_u(static_cast<uint32_t>(rem), lead)));
                            }
                            const bool has = ((rem >> i) & mask_t(1)) != 0;
                            rem &= ~(mask_t(1) << i);
                            const DPrim rec = sc.recs[i];
                            const bool on_self = i == sr.self;
                            uint32_t cu = 0, cv = 0;
                            for (uint32_t cell = 0; cell < cells; ++cell) {
                                float ju = 0.5f, jv = 0.5f;
                                if (P.jitter_mode == RT_JITTER_COUNTER) counter_jitter(key, node, li, cell, ju, jv);
                                const V3 tl = point_on_light(l, cu, cv, ju, jv) - sr.o;
                                const float dist2 = dot(tl, tl);
                                const float inv_dist = rsq_f(dist2);
                                const float dist = dist2 * inv_dist;
                                const bool ray = dist2 > kEps * kEps;  // null vector => lit (world.rs:77-81)
                                sr.d = tl * inv_dist;
                                bool hit = false;
                                if (has && ray && !((occ_bits >> cell) & 1ull)) {
                                    if (on_self) {
                                        if constexpr (EXACT && !RT_NO_NEAR) hit = occludes_self_near<EXOTIC>(rec, sc.extras, sr, dist, band);
                                        else intersect_prim<EXOTIC>(rec, sc.extras, sr, true, [&](int, float t, float, float) { hit |= in_range(t, dist); });
                                    } else if constexpr (EXACT && !RT_NO_NEAR) {
                                        hit = occludes_prim_near<EXOTIC>(rec, sc.extras, sr.o, sr.d, dist, band);
                                    } else {
                                        hit = occludes_prim<EXOTIC>(rec, sc.extras, sr.o, sr.d, dist);
                                    }
                                    if ((TREE && sc.has_gates) && hit) {  // group.rs:107
                                        if constexpr (EXACT && EXOTIC && !RT_NO_NEAR && (RT_GATE_NEAR & 2) != 0) hit = gates_pass_near(sc.gates, sc.gate[i], sr.o, sr.d, band);
                                        else hit = gates_pass(sc.gates, sc.gate[i], sr.o, sr.d);
                                    }
                                }
                                if (hit) occ_bits |= 1ull << cell;
                                if (STATS && first && mine && ray) c_shadow++, c_cast++;
                                if (++cu == l.usteps) cu = 0u, ++cv;
                            }
                            first = false;
                        }
                        if (mine) lit = static_cast<float>(cells - static_cast<uint32_t>(__popcll(occ_bits)));
                        if constexpr (EXACT) fix_li |= mine && band;
                        need = 0ull;
                    }
                }
                while (need) {
                    // the shading points ("owners") of this pass: one per slot of g lanes
                    int owner = 0;
                    uint32_t my_slot = 0, taken = 0;
                    mask_t own_mask = 0;
                    bool owner_now = false;
                    for (uint32_t k = 0; k < slots && need; ++k) {
                        const int o = __ffsll(static_cast<long long>(need)) - 1;
                        need &= need - 1ull;
                        if (TREE)
                            own_mask |= static_cast<mask_t>(rl_u(static_cast<uint32_t>(blockers), o)) |
                                        static_cast<mask_t>(static_cast<unsigned long long>(rl_u(static_cast<uint32_t>(static_cast<unsigned long long>(blockers) >> 32), o)) << 32);
                        else
                            own_mask |= static_cast<mask_t>(rl_u(static_cast<uint32_t>(blockers), o));
                        if (slot == k) owner = o;
                        if (static_cast<int>(lane) == o) my_slot = k, owner_now = true;
                        taken = k + 1;
                    }
                    const unsigned long long pass_mask = static_cast<unsigned long long>(own_mask);
                    const bool lane_on = slot < taken;
                    Ray sr;
                    uint32_t o_key, o_node;
                    {  // the owner's hit data: its LDS slots, read by every lane of its group (a broadcast)
                        const float* const op = s_park + (threadIdx.x & ~63u) + static_cast<uint32_t>(owner);
                        auto od3 = [&](int k) { return v3(op[k * kBlockThreads], op[(k + 1) * kBlockThreads], op[(k + 2) * kBlockThreads]); };
                        sr.o = od3(24), sr.q = od3(27), sr.dq = od3(30), sr.self = __float_as_uint(op[33 * kBlockThreads]);
                        o_key = __float_as_uint(op[34 * kBlockThreads]);
                    }
                    if (slots == 1u) {  // one owner for the whole wave: scalar broadcasts
                        const int os = __builtin_amdgcn_readfirstlane(owner);
                        o_node = rl_u(node, os);
                    } else {
                        o_node = shfl_u(node, owner);
                    }
                    unsigned int my_lit = 0;
                    [[maybe_unused]] bool owner_unsure = false;
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
                            counter_jitter(o_key, o_node, li, cell, ju, jv);  // first draw goes to u (light.rs:215)
                            lp = point_on_light(l, cu, cv, ju, jv);
                        }
                        const V3 tl = lp - sr.o;
                        const float dist2 = dot(tl, tl);
                        const float inv_dist = rsq_f(dist2);
                        const float dist = dist2 * inv_dist;
                        const bool cast = con && dist2 > kEps * kEps;  // null vector => lit (world.rs:77-81)
                        sr.d = tl * inv_dist;
#if RT_EXP & 4
                        const bool occ = cast && dist > 1e30f;
#else
                        bool unsure = false;
                        const bool occ = occluded<EXOTIC, TREE, kLane, EXACT && !RT_NO_NEAR>(sc, sr, dist, cast, by_mask, pass_mask, unsure);
#endif
                        const unsigned long long b = __ballot(con && !occ);
                        const int lead = static_cast<int>(my_slot << glog);
                        if (owner_now) my_lit += static_cast<unsigned int>(__popcll((b >> lead) & gmask));
                        if constexpr (EXACT) {
                            const unsigned long long bu = wave_ballot(unsure);
                            if (bu && owner_now) owner_unsure |= ((bu >> lead) & gmask) != 0ull;
                        }
                        if (STATS && cast) c_shadow++, c_cast++;
                    }
                    if (owner_now) lit = static_cast<float>(my_lit);
                    if constexpr (EXACT) fix_li |= owner_now && owner_unsure;
                }
                // (a qedge point also when its shaft was found empty or wholly blocked and nothing was cast: the shaft tests
                // reason about the point OUTSIDE its own convex body, and this one stands 1e-15 inside — the corner line of
                // cover_image_area's room, lit by the tests and shadowed by its own face in the reference)
                if constexpr (EXACT) emit_fix(fix_li || (qedge && shades && (sum_d != 0.0f || sum_s != 0.0f)), li, lit, sum_d, sum_s, lc, l.inv_samples);
                lit *= l.inv_samples;  // light.rs:208
            }
            if (has_hit) {  // material.rs:127-161
                const V3 ec = v3(park[8 * kBlockThreads], park[9 * kBlockThreads], park[10 * kBlockThreads]) * lc;
                const V3 shade = ec * (park[12 * kBlockThreads] * sum_d) + lc * (park[13 * kBlockThreads] * sum_s);
                const V3 add = ec * park[11 * kBlockThreads] + shade * (l.inv_samples * lit);
                park[14 * kBlockThreads] += add.x, park[15 * kBlockThreads] += add.y, park[16 * kBlockThreads] += add.z;
            }
        }
        w = v3(park[0 * kBlockThreads], park[1 * kBlockThreads], park[2 * kBlockThreads]);
        if (!kSteal) acc = v3(park[3 * kBlockThreads], park[4 * kBlockThreads], park[5 * kBlockThreads]);
        n1 = park[6 * kBlockThreads], n2 = park[7 * kBlockThreads];
        nrm = ld3(17);
        if (AREA) over = ld3(24), hq = ld3(27), hdq = ld3(30), hit_prim = __float_as_uint(park[33 * kBlockThreads]);
        const V3 eye = ld3(20);
        if (has_hit) {
            const V3 add = w * v3(park[14 * kBlockThreads], park[15 * kBlockThreads], park[16 * kBlockThreads]);
            if (kSteal) {  // (LDS float add; lanes of one instruction are served in lane order: the sum is reproducible)
                atomicAdd(pixel_sum + 3 * kBlockThreads + pix, add.x);
                atomicAdd(pixel_sum + 4 * kBlockThreads + pix, add.y);
                atomicAdd(pixel_sum + 5 * kBlockThreads + pix, add.z);
            } else {
                acc = acc + add;
            }
        }

        RT_TICK(3)
        // ---------------------------------------------------------------- children (world.rs:59-69,94-133)
        float m_reflectivity = 0.0f, m_transparency = 0.0f;
        if (has_hit) {
            const DMaterial m = sc.mats[mi];
            m_reflectivity = m.reflectivity, m_transparency = m.transparency;
        }
        bool want_refl = SPAWN && has_hit && depth > 0u && !(fabsf(m_reflectivity) <= kEps);
        bool want_refr = SPAWN && has_hit && depth > 0u && !(fabsf(m_transparency) <= kEps);
        const bool use_schlick = has_hit && (m_reflectivity * m_transparency) > 0.0f;
        V3 rdir = v3(0.0f, 0.0f, 0.0f);
        const bool chain = EXACT && refined && !(RT_XEXP & 2);  // the children's f64 rays wait in r64[]
        if (want_refr && (chain ? tir64 : !refract_dir(eye, nrm, n1, n2, rdir))) want_refr = false;  // total internal reflection
        const float fr = use_schlick ? schlick(eye, nrm, n1, n2) : 0.0f;
        // The reference adds reflected + refracted colour once PER LIGHT (world.rs:44-69): the
        // recursion is traced once here and weighted by the number of lights.
        const float k_refl = nlights * m_reflectivity * (use_schlick ? fr : 1.0f);
        const float k_refr = nlights * m_transparency * (use_schlick ? (1.0f - fr) : 1.0f);
        V3 under = over - nrm * (2.0f * kEps);  // point - n*eps (intersection.rs:53)
        const bool push = has_hit && want_refl && want_refr && sp < kStackCap;
        if (EXACT && !(RT_XEXP & 32) && has_hit && want_refr) {
            // the refraction ray's f64 copy: written by the refined node above, else (no f64 hit) the chain
            // restarts from the FP32 ray
            const uint32_t slot = push ? 1u + static_cast<uint32_t>(sp) : 0u;
            if (chain) {
                const Ray64 kept = r64_load(slot);
                under = to_v3(kept.o), rdir = to_v3(kept.d);
            } else {
                r64_store(slot, to_d3(under), to_d3(rdir));
            }
        }

        if (push) {
            StackEntry e;
            e.o = under, e.d = rdir, e.w = w * k_refr, e.q = hq, e.dq = neg(hdq);
            e.self = hit_prim, e.node = 2u * node + 2u;
            stack[sp++] = e;
            if (STATS) c_refract++;
        }
        if (has_hit && want_refl) {
            cur.o = over;
            const V3 din = neg(eye);                     // the incoming direction, back from its parked negation
            cur.d = din - nrm * (2.0f * dot(din, nrm));  // tuple.rs:121-123
            if (EXACT && !(RT_XEXP & 32)) {
                if (chain) cur.d = to_v3(r64_load(0u).d);
                else r64_store(0u, to_d3(cur.o), to_d3(cur.d));
                cur_slot = 0;
            }
            cur.self = hit_prim, cur.q = hq, cur.dq = hdq;
            w = w * k_refl;
            node = 2u * node + 1u;
            if (STATS) c_reflect++;
        } else if (has_hit && want_refr) {
            cur.o = under;
            cur.d = rdir;
            if (EXACT) cur_slot = 0;
            cur.self = hit_prim, cur.q = hq, cur.dq = neg(hdq);
            w = w * k_refr;
            node = 2u * node + 2u;
            if (STATS) c_refract++;
        } else if (active) {
            if (sp > 0) {
                const StackEntry e = stack[--sp];
                cur.o = e.o, cur.d = e.d, cur.self = e.self, cur.q = e.q, cur.dq = e.dq;
                if (EXACT) cur_slot = 1u + static_cast<uint32_t>(sp);
                w = e.w;
                node = e.node;
            } else {
                active = false;
            }
        }
        if (kSteal) {
            // Idle lanes (pixel finished, or outside the image) take the newest waiting ray of lanes that have one:
            // the k-th idle lane from the k-th such lane, through the park area (rows 6.. are dead here).  A round
            // costs a wave the same with 5 busy lanes as with 64 — C4's rounds ran 43 % full — so the rounds of a
            // tile go from its deepest pixel's node count towards its total node count / 64.
#ifndef RT_STEAL_PASSES
#define RT_STEAL_PASSES 3  // a lane with several waiting rays serves several idle lanes in one round
#endif
#pragma unroll 1
            for (int pass = 0; pass < RT_STEAL_PASSES; ++pass) {
            const bool idle = !active;
            const bool donor = active && sp > 0;
            const unsigned long long m_idle = wave_ballot(idle), m_donor = wave_ballot(donor);
            const uint32_t n_idle = static_cast<uint32_t>(__popcll(m_idle)), n_donor = static_cast<uint32_t>(__popcll(m_donor));
            const uint32_t n_pairs = n_idle < n_donor ? n_idle : n_donor;
            if (n_pairs == 0u) break;
            {
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t r_idle = static_cast<uint32_t>(__popcll(m_idle & below)), r_donor = static_cast<uint32_t>(__popcll(m_donor & below));
                float* const ex = s_park + (threadIdx.x & ~63u) + 6 * kBlockThreads;  // field f of pair r at ex[f * 32 + r]
                auto put = [&](int f, uint32_t r, float v) { ex[f * 32 + static_cast<int>(r)] = v; };
                auto get = [&](int f, uint32_t r) { return ex[f * 32 + static_cast<int>(r)]; };
                if (donor && r_donor < n_pairs) {
                    const StackEntry e = stack[--sp];  // (the newest; the oldest — a ring stack — measured the same)
                    const float f[19] = {e.o.x, e.o.y, e.o.z, e.d.x, e.d.y, e.d.z, e.w.x, e.w.y, e.w.z, e.q.x, e.q.y, e.q.z, e.dq.x, e.dq.y, e.dq.z,
                                         __uint_as_float(e.self), __uint_as_float(e.node), __uint_as_float(pix),
                                         AREA ? park[34 * kBlockThreads] : 0.0f};
                    for (int k = 0; k < 19; ++k) put(k, r_donor, f[k]);
                    if (EXACT) {
                        const Ray64 r = r64[1 + sp];
                        const double d6[6] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z};
                        for (int k = 0; k < 6; ++k) {
                            const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(d6[k]));
                            put(19 + 2 * k, r_donor, __uint_as_float(static_cast<uint32_t>(b)));
                            put(20 + 2 * k, r_donor, __uint_as_float(static_cast<uint32_t>(b >> 32)));
                        }
                    }
                }
                // (same wave: LDS operations complete in order; the fence keeps the compiler from moving the reads up)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (idle && r_idle < n_pairs) {
                    cur.o = v3(get(0, r_idle), get(1, r_idle), get(2, r_idle));
                    cur.d = v3(get(3, r_idle), get(4, r_idle), get(5, r_idle));
                    w = v3(get(6, r_idle), get(7, r_idle), get(8, r_idle));
                    cur.q = v3(get(9, r_idle), get(10, r_idle), get(11, r_idle));
                    cur.dq = v3(get(12, r_idle), get(13, r_idle), get(14, r_idle));
                    cur.self = __float_as_uint(get(15, r_idle));
                    node = __float_as_uint(get(16, r_idle));
                    pix = __float_as_uint(get(17, r_idle));
                    const float key = get(18, r_idle);
                    if (EXACT) {
                        double d6[6];
                        for (int k = 0; k < 6; ++k) {
                            const unsigned long long b = static_cast<unsigned long long>(__float_as_uint(get(19 + 2 * k, r_idle))) |
                                                         (static_cast<unsigned long long>(__float_as_uint(get(20 + 2 * k, r_idle))) << 32);
                            d6[k] = __longlong_as_double(static_cast<long long>(b));
                        }
                        r64_store(0u, d3(d6[0], d6[1], d6[2]), d3(d6[3], d6[4], d6[5]));
                        cur_slot = 0;
                    }
                    if (AREA) park[34 * kBlockThreads] = key;
                    active = true;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            }
        }
        RT_TICK(4)
    }

    if constexpr (kFix && (RT_FIX_PARTS & 2)) {
        // The second count of the lights whose rays met a near thing.  A pass takes the light of the first waiting lane's
        // newest record and, from every lane whose newest record is of that light, that record — up to kFixSlots of them,
        // lower lanes first (one order, whatever happens around the wave).  They go through LDS rows (the park rows are
        // free now) to slots of `width` lanes each, width = the light's cell count rounded up to a power of two: a slot
        // counts its record's cells, a cell a lane.  (One record a pass, all 64 lanes on it, left 63 lanes idle for a point
        // light: a tile with 27 records of obj_mesh's two point lights kept its wave 1.4 ms after the rest of the frame,
        // profiles/r03_group_boxes.txt.)
        constexpr uint32_t kFixSlots = 16;  // 16 records x 32 words = park rows 6..13 (rows 3..5 may hold the pixel sums)
        uint32_t n_mine = P.no_fix ? 0u : pend[0].scratch[0];
#ifdef RT_PHASE_CLOCK
        const unsigned long long t_fix = __builtin_amdgcn_s_memtime();
#endif
        float* const rows = s_park + (threadIdx.x & ~63u);
        auto ex_at = [&](uint32_t slot_, uint32_t k) -> float& {
            const uint32_t w = slot_ * 32u + k;
            return rows[(6u + (w >> 6)) * kBlockThreads + (w & 63u)];
        };
#pragma unroll 1
        for (;;) {
            const unsigned long long mm = wave_ballot(n_mine != 0u);
            if (mm == 0ull) break;
            const uint32_t my_light = n_mine != 0u ? (pend[(n_mine - 1u) < kPend ? n_mine - 1u : 0u].light & 0x7FFFFFFFu) : 0xFFFFFFFFu;
            const uint32_t li = rl_u(my_light, __ffsll(static_cast<long long>(mm)) - 1);
            const bool light_ok = li < sc.n_lights;
            const DLight l = sc.lights[light_ok ? li : 0u];
            const XLight xl = args_here().sc.xlights[light_ok ? li : 0u];
            const uint32_t cells = light_ok ? l.usteps * l.vsteps : 1u;
            // width of a slot, log2: 0 for a point light .. 6 for 33 cells and more
            const uint32_t wlog = cells >= 33u ? 6u : (cells <= 1u ? 0u : 32u - static_cast<uint32_t>(__clz(static_cast<int>(cells - 1u))));
            const uint32_t width = 1u << wlog;
            const uint32_t room = (64u >> wlog) < kFixSlots ? (64u >> wlog) : kFixSlots;
            const unsigned long long same = wave_ballot(n_mine != 0u && my_light == li);
            const uint32_t rank = static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1ull)));
            const uint32_t waiting = static_cast<uint32_t>(__popcll(same));
            const uint32_t taken = waiting < room ? waiting : room;
#ifdef RT_PHASE_CLOCK
            ph[7] += taken;
#endif
            if (((same >> lane) & 1ull) != 0ull && rank < taken) {
                --n_mine;
                const FixRecord r = pend[n_mine < kPend ? n_mine : 0u];
                const uint32_t* wds = reinterpret_cast<const uint32_t*>(&r);
                for (uint32_t k = 0; k < 32u; ++k) ex_at(rank, k) = __uint_as_float(wds[k]);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const uint32_t slot = lane >> wlog, sub = lane & (width - 1u);
            const bool on = slot < taken;
            const uint32_t rs = on ? slot : 0u;
            auto wd = [&](uint32_t k) { return __float_as_uint(ex_at(rs, k)); };
            auto fl = [&](uint32_t k) { return ex_at(rs, k); };
            auto db = [&](uint32_t k) {
                return __longlong_as_double(static_cast<long long>(static_cast<unsigned long long>(wd(k)) | (static_cast<unsigned long long>(wd(k + 1u)) << 32)));
            };
            const uint32_t pl = wd(0) & 63u, at_node = wd(2), key = wd(3), self_prim = wd(8);
            const bool was_refined = (wd(1) >> 31) != 0u;
            D3 over64 = d3(db(20), db(22), db(24));  // (a refined node's)
            bool ok = on && light_ok && self_prim < sc.n_items;
#ifdef RT_FIX_STUB  // experiment: the second count without its arithmetic (what the block's mere presence costs)
            ok = false;
#endif
            if (ok && !was_refined) {  // the node's f64 hit, worked out again from its f64 ray (intersection.rs:40-69)
                D3 ro = d3(db(20), db(22), db(24)), rd = d3(db(26), db(28), db(30));
                if (at_node == 0u) {
                    uint32_t px, pyl, pgy;
                    pixel_of_lane(args_here(), kOrder, wave, pl, px, pyl, pgy);
                    xcamera_ray(cam64_here(), px, pgy, ro, rd);
                }
                const DPrim p = sc.recs[self_prim];
                const double* xinv = args_here().sc.xprims[self_prim].inv;
                const XHits ha = xintersect<EXOTIC>(p.type, xinv, args_here().sc.xextras[p.extra].f, ro, rd);
                double ft = 0.0, fu = 0.0, fv = 0.0;
                uint32_t pr = kNoPrim;
                ok = xselect<EXOTIC>(ha, self_prim, p.type, ha, kNoPrim, REC_NODE, false, ft, pr, fu, fv);
                if (ok) over64 = xprepare<EXOTIC>(p.type, xinv, args_here().sc.xextras[p.extra].f, ro, rd, ft, fu, fv).over;
            }
            const V3 o = to_v3(over64);
            const unsigned long long slot_mask = (wlog == 6u ? ~0ull : ((1ull << width) - 1ull)) << ((slot << wlog) & 63u);
            uint32_t lit = 0;
#pragma unroll 1
            for (uint32_t c0 = 0; c0 < cells; c0 += width) {
                const uint32_t cell = c0 + sub;
                bool unocc = false;
                if (ok && cell < cells) {
                    V3 lp = v3(l.pos[0], l.pos[1], l.pos[2]);
                    D3 lp64 = d3(xl.pos[0], xl.pos[1], xl.pos[2]);
                    if (l.type != RT_LIGHT_POINT) {  // light.rs:211-216
                        const uint32_t cv = cell / l.usteps, cu = cell - cv * l.usteps;
                        float ju = 0.5f, jv = 0.5f;
                        double ju64 = 0.5, jv64 = 0.5;
                        if (P.jitter_mode == RT_JITTER_COUNTER) {
                            counter_jitter(key, at_node, li, cell, ju, jv);
                            xcounter_jitter(key, at_node, li, cell, ju64, jv64);
                        }
                        lp = point_on_light(l, cu, cv, ju, jv), lp64 = xpoint_on_light(xl, cu, cv, ju64, jv64);
                    }
                    const V3 tl = lp - o;
                    const float dist2 = dot(tl, tl);
                    unocc = true;
                    if (dist2 > 0.0f) {  // (a null vector is lit, world.rs:77-81: xoccludes decides that in f64)
                        const float inv_dist = rsq_f(dist2);
#ifdef RT_PHASE_CLOCK
                        unocc = !fix_occluded<EXOTIC>(args_here().sc, self_prim, o, tl * inv_dist, dist2 * inv_dist, over64, lp64, &ph[8]);
#else
                        unocc = !fix_occluded<EXOTIC>(args_here().sc, self_prim, o, tl * inv_dist, dist2 * inv_dist, over64, lp64);
#endif
                    }
                }
                lit += static_cast<uint32_t>(__popcll(wave_ballot(unocc) & slot_mask));
            }
            const float delta = ok ? static_cast<float>(lit) - fl(7) : 0.0f;
            const V3 add = v3(fl(4), fl(5), fl(6)) * delta;
            // into the pixels, slot after slot: the lane of the record's pixel adds (with ray stealing: to its LDS sums)
#pragma unroll 1
            for (uint32_t s_ = 0; s_ < taken; ++s_) {
                const int from = static_cast<int>(s_ << wlog);
                const uint32_t to = rl_u(pl, from);
                const V3 a = rl_v3(add, from);
                if (lane == to && rl_f(delta, from) != 0.0f) {
                    if (kSteal) {
                        pixel_sum[3 * kBlockThreads + lane] += a.x, pixel_sum[4 * kBlockThreads + lane] += a.y, pixel_sum[5 * kBlockThreads + lane] += a.z;
                    } else {
                        acc = acc + a;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
#ifdef RT_PHASE_CLOCK
        if (STATS) ph[6] = __builtin_amdgcn_s_memtime() - t_fix;
#endif
    }
    {
        // the pixel's coordinates are worked out again rather than kept in registers (or spilled to
        // scratch) for the whole kernel; the empty asm keeps the compiler from merging the two copies
        uint32_t lane_again = threadIdx.x & 63u;
        asm volatile("" : "+v"(lane_again));
        uint32_t ox, oyl, ogy;
        const KParams& PE = args_here();  // (the tiling and the camera's size, read again: not kept across the path loop)
        if (pixel_of_lane(PE, kOrder, wave, lane_again, ox, oyl, ogy)) {
            float* o = PE.out + (static_cast<size_t>(PE.global_rows ? ogy : oyl) * PE.cam.hsize + ox) * 3u;
            if (kSteal) acc = v3(park[3 * kBlockThreads], park[4 * kBlockThreads], park[5 * kBlockThreads]);
            o[0] = acc.x, o[1] = acc.y, o[2] = acc.z;
        }
    }
    if (args_here().tile_cost) {
        const KParams& PE = args_here();
        uint32_t tx, tyl, tgy;
        pixel_of_lane(PE, kOrder, wave, 0u, tx, tyl, tgy);  // lane 0 = the tile's top-left pixel
        if (lane == 0) PE.tile_cost[(tyl >> 3) * PE.tiles_x + (tx >> 3)] = 10u * n_rounds + n_visibility;  // (a round ~ ten visibility passes)
    }
    if (STATS) {
        unsigned long long v[kCounters] = {c_primary, c_reflect, c_refract, c_shadow, c_shaded, c_cells, c_pattern, c_container, c_cast, c_refined};
#ifdef RT_PHASE_CLOCK
        for (int k = 0; k < 9; ++k) v[kCountersBase + k] = (lane == 0 || k == 8) ? ph[k] : 0ull;
        v[kCountersBase + 9] = 0ull;
        if (lane == 0) atomicMax(P.counters + kCountersBase + 9, ph[6]);
#endif
        unsigned long long cast_rays = 0;  // primary + reflection + refraction + shadow rays really traced
        for (int k = 0; k < kCounters; ++k) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0 && s) atomicAdd(P.counters + k, s);
            if (k < 3 || k == 8) cast_rays += s;
        }
        if (P.tile_rays) {
            uint32_t tx, tyl, tgy;
            pixel_of_lane(P, kOrder, wave, 0u, tx, tyl, tgy);  // lane 0 = the tile's top-left pixel
            if (lane == 0) P.tile_rays[(tyl >> 3) * P.tiles_x + (tx >> 3)] = static_cast<uint32_t>(cast_rays);
        }
    }
}

}  // namespace rtk
