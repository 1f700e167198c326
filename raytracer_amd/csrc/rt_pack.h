// rt_pack.h — host side of rt_scene_create: validates an rt_scene_desc and compiles it into the
// device tables of rt_math.h.  Pure C++ (no HIP), so the test-only host simulator reuses exactly the
// same tables.
//
// The traversal stream.  The caller's scene is a list of objects, some of them Groups with a box
// (world.rs:32-41, group.rs:106-119); the reference tests every object, and a Group's children when
// the ray's line hits the Group's box.  What a ray can see is therefore: every primitive whose
// enclosing Groups' box tests all pass.  The device keeps that rule (the Groups become GATES,
// rt_math.h) but does not walk the caller's tree, whose leaves may hold hundreds of children
// (Group::divide(256) in examples/multiple_glass_and_metallic_spheres.rs:104).  It walks its own
// bounding-volume hierarchy instead: bounded primitives (spheres, cubes, bounded cylinders,
// triangles) are sorted into a binary tree of world-space boxes with at most two primitives per
// leaf, unbounded ones (planes, infinite cylinders) stay at the top, and the whole thing is laid out
// as ONE array of 64-byte records in depth-first pre-order with skip links.  Scenes with a handful
// of primitives and no Groups are laid out flat, in world.objects order.
#pragma once

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_exact.h"
#include "rt_math.h"

namespace rtk {

// The per-record tables of a large scene are hundreds of megabytes that pack_scene's threads fill record by record:
// a vector that does not zero them first (resize() leaves trivially constructible elements as they are).
template <class T>
struct LeaveAlone : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = LeaveAlone<U>;
    };
    template <class U>
    void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new (static_cast<void*>(p)) U;
    }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
template <class T>
using PodVec = std::vector<T, LeaveAlone<T>>;

struct PackedScene {
    PodVec<DPrim> recs;       // the traversal stream
    PodVec<uint32_t> order;   // per record: the primitive's position in world.objects order (nodes: 0)
    PodVec<uint32_t> gate;    // per record: innermost enclosing Group (index into gates) or kNoGate
    PodVec<DBound> bound;     // per record: world-space bounding ball (shaft tests)
    PodVec<LNode> lnodes;     // per record: the compact twin of the stream, canonical order (lane-mode walks)
    // the eight direction orders of it (8 x records; a scene with a hierarchy, and only if the caller of pack_scene asked
    // for them — host_streams; rt_scene_create lets the GPU lay them out), and the staged top (rt_math.h kTopK)
    PodVec<LNode> lstreams;
    std::vector<LNode> top;           // 8 x kTopK
    std::vector<uint32_t> portal_ret;  // 8 per portal
    uint32_t top_k = 0;               // records in use per top region; 0 = no staged top
    uint32_t n_top = 0;           // records in front of the hierarchy's root (unbounded primitives)
    std::vector<DGate> gates;     // the caller's Groups
    std::vector<XGate> xgates;    // ... and their boxes in the caller's own f64 (node_box64), or the f32 values widened
    PodVec<DExtra> extras;
    std::vector<DMaterial> mats;
    std::vector<DPattern> pats;
    std::vector<DLight> lights;
    // the caller's f64 tables for the exact chain (rt_exact.h), laid out like their FP32 twins: per stream
    // record, per DExtra slot, per material, per pattern.  Empty when the caller did not provide them.
    PodVec<XPrim> xprims;
    PodVec<XExtra> xextras;
    std::vector<double> xior;
    std::vector<XPattern> xpats;
    std::vector<XLight> xlights;  // always filled: the caller's f64 light geometry, or the f32 values widened
    bool has_f64 = false;
    uint32_t n_items = 0, n_lights = 0;  // real counts (the vectors are padded to be non-empty)
    uint32_t max_cells = 1;              // largest usteps*vsteps over the lights
    bool has_gates = false, has_bvh = false, exotic = false, has_area = false;
    bool any_spawn = false;  // some material is reflective or transparent
    bool any_glass = false;  // some material is transparent (paths branch: rays of a tile soon have little in common)
};

constexpr uint32_t kFlatLimit = 8;  // at most this many primitives and no Groups: no hierarchy

// Widen a box bound outwards by a relative 4e-6 (+ a tiny absolute term): the slab tests on the
// device run in FP32, and nothing that intersects in f64 may be culled because a box test rounded
// the other way.
inline float widen(float v, bool up) {
    if (std::isinf(v) || std::isnan(v)) return v;
    float pad = std::fabs(v) * 4e-6f + 1e-6f;
    return up ? v + pad : v - pad;
}

struct HostBox {
    double mn[3], mx[3];
    bool bounded;
    void add(const double p[3]) {
        for (int k = 0; k < 3; ++k) mn[k] = std::min(mn[k], p[k]), mx[k] = std::max(mx[k], p[k]);
    }
    void merge(const HostBox& o) {
        for (int k = 0; k < 3; ++k) mn[k] = std::min(mn[k], o.mn[k]), mx[k] = std::max(mx[k], o.mx[k]);
    }
    static HostBox empty() {
        HostBox b;
        for (int k = 0; k < 3; ++k) b.mn[k] = INFINITY, b.mx[k] = -INFINITY;
        b.bounded = true;
        return b;
    }
};

// Inverse of a 3x4 affine (rows) in f64; false if singular.
inline bool invert_affine(const float* m, double out[12]) {
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    if (!(std::fabs(det) > 1e-300) || !std::isfinite(det)) return false;
    const double r = 1.0 / det;
    double L[9] = {A * r, -(b * i - c * h) * r, (b * f - c * e) * r, B * r, (a * i - c * g) * r, -(a * f - c * d) * r,
                   C * r, -(a * h - b * g) * r, (a * e - b * d) * r};
    const double t[3] = {m[3], m[7], m[11]};
    for (int k = 0; k < 3; ++k) {
        out[4 * k + 0] = L[3 * k + 0], out[4 * k + 1] = L[3 * k + 1], out[4 * k + 2] = L[3 * k + 2];
        out[4 * k + 3] = -(L[3 * k + 0] * t[0] + L[3 * k + 1] * t[1] + L[3 * k + 2] * t[2]);
    }
    return true;
}

// World-space box of one primitive, from the SAME f32 transform_inverse the device intersects with.
inline HostBox prim_world_box(uint32_t type, const float* inv, const float* param) {
    HostBox box = HostBox::empty();
    double fwd[12];
    if (type == RT_PRIM_PLANE || !invert_affine(inv, fwd)) {
        box.bounded = false;
        return box;
    }
    auto put = [&](double x, double y, double z) {
        double p[3];
        for (int k = 0; k < 3; ++k) p[k] = fwd[4 * k] * x + fwd[4 * k + 1] * y + fwd[4 * k + 2] * z + fwd[4 * k + 3];
        box.add(p);
    };
    if (type == RT_PRIM_TRIANGLE || type == RT_PRIM_SMOOTH_TRIANGLE) {
        const float* v0 = param;
        const float* e0 = param + 3;
        const float* e1 = param + 6;
        put(v0[0], v0[1], v0[2]);
        put(double(v0[0]) + e0[0], double(v0[1]) + e0[1], double(v0[2]) + e0[2]);  // triangle.rs: e0 = v1 - v0
        put(double(v0[0]) + e1[0], double(v0[1]) + e1[1], double(v0[2]) + e1[2]);
    } else {
        double y0 = -1.0, y1 = 1.0;
        if (type == RT_PRIM_CYLINDER) {
            y0 = param[0], y1 = param[1];
            if (!std::isfinite(y0) || !std::isfinite(y1)) {
                box.bounded = false;
                return box;
            }
        }
        for (int c = 0; c < 8; ++c) put((c & 1) ? 1.0 : -1.0, (c & 2) ? y1 : y0, (c & 4) ? 1.0 : -1.0);
    }
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(box.mn[k]) || !std::isfinite(box.mx[k])) {
            box.bounded = false;
            return box;
        }
        // f64 -> f32 outwards, plus room for the FP32 intersection arithmetic
        const double pad = 1e-5 * std::max(std::fabs(box.mn[k]), std::fabs(box.mx[k])) + 1e-5;
        box.mn[k] -= pad, box.mx[k] += pad;
    }
    return box;
}

// World-space bounding ball of one primitive (shaft tests, rt_math.h), from the same f32
// transform_inverse.  Unbounded or degenerate: r = +inf ("may always block").
inline DBound prim_world_ball(uint32_t type, const float* inv, const float* param) {
    DBound out{{0.0f, 0.0f, 0.0f}, INFINITY};
    double fwd[12];
    if (type == RT_PRIM_PLANE || !invert_affine(inv, fwd)) return out;
    double pts[8][3];
    int n = 0;
    if (type == RT_PRIM_TRIANGLE || type == RT_PRIM_SMOOTH_TRIANGLE) {
        for (int k = 0; k < 3; ++k) {
            pts[0][k] = param[k];
            pts[1][k] = double(param[k]) + param[3 + k];
            pts[2][k] = double(param[k]) + param[6 + k];
        }
        n = 3;
    } else {
        double y0 = -1.0, y1 = 1.0;
        if (type == RT_PRIM_CYLINDER) y0 = param[0], y1 = param[1];
        if (!std::isfinite(y0) || !std::isfinite(y1)) return out;
        for (int c = 0; c < 8; ++c) pts[c][0] = (c & 1) ? 1.0 : -1.0, pts[c][1] = (c & 2) ? y1 : y0, pts[c][2] = (c & 4) ? 1.0 : -1.0;
        n = 8;
    }
    double cen[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) cen[k] += pts[i][k] / n;
    double r2 = 0.0;
    if (type == RT_PRIM_SPHERE) {
        // |F x| <= sqrt(lambda_max(F^T F)) for |x| = 1; Gershgorin row sums bound lambda_max
        for (int i = 0; i < 3; ++i) {
            double row = 0.0;
            for (int j = 0; j < 3; ++j) {
                double g = 0.0;
                for (int k = 0; k < 3; ++k) g += fwd[4 * k + i] * fwd[4 * k + j];
                row += std::fabs(g);
            }
            r2 = std::max(r2, row);
        }
    } else {
        for (int i = 0; i < n; ++i) {
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double v = fwd[4 * k] * (pts[i][0] - cen[0]) + fwd[4 * k + 1] * (pts[i][1] - cen[1]) + fwd[4 * k + 2] * (pts[i][2] - cen[2]);
                d2 += v * v;
            }
            r2 = std::max(r2, d2);
        }
    }
    double c[3], cmax = 0.0;
    for (int k = 0; k < 3; ++k) {
        c[k] = fwd[4 * k] * cen[0] + fwd[4 * k + 1] * cen[1] + fwd[4 * k + 2] * cen[2] + fwd[4 * k + 3];
        cmax = std::max(cmax, std::fabs(c[k]));
    }
    const double r = std::sqrt(r2) * (1.0 + 1e-5) + 1e-5 * cmax + 1e-5;
    if (!std::isfinite(r) || !std::isfinite(cmax)) return out;
    out.c[0] = static_cast<float>(c[0]), out.c[1] = static_cast<float>(c[1]), out.c[2] = static_cast<float>(c[2]);
    out.r = static_cast<float>(r);
    // Inscribed ball around the same centre (spheres and cubes: the full-occlusion test of rt_math.h).
    // Cube: the faces are the planes row_k . x = +-1, at distance 1 / |row_k| from the centre.
    // Sphere (an ellipsoid in world space): smallest semi-axis = 1 / sigma_max(inv), bounded through
    // Gershgorin on inv inv^T.
    if (type == RT_PRIM_CUBE || type == RT_PRIM_SPHERE) {
        double g[3][3], worst = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) g[i][j] = double(inv[4 * i]) * inv[4 * j] + double(inv[4 * i + 1]) * inv[4 * j + 1] + double(inv[4 * i + 2]) * inv[4 * j + 2];
        for (int i = 0; i < 3; ++i) worst = std::max(worst, type == RT_PRIM_CUBE ? g[i][i] : std::fabs(g[i][0]) + std::fabs(g[i][1]) + std::fabs(g[i][2]));
        const double rin = (worst > 0.0 ? 1.0 / std::sqrt(worst) : 0.0) * (1.0 - 1e-4);
        if (std::isfinite(rin) && rin > 0.0 && rin < r) out.r_in = static_cast<float>(rin);
        // margin of the object-space pyramid tests (rt_math.h ShaftObj); 0 = do not use them
        const double margin = 1e-3 + 1e-4 * std::sqrt(g[0][0] + g[1][1] + g[2][2]);
        if (std::isfinite(margin) && margin < 0.25) out.margin = static_cast<float>(margin);
    }
    return out;
}
// Ball around a hierarchy node's box.
inline DBound node_ball(const DPrim& node) {
    DBound b{};
    double r2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        b.c[k] = 0.5f * (node.inv[k] + node.inv[3 + k]);
        const double h = 0.5 * (double(node.inv[3 + k]) - node.inv[k]);
        r2 += h * h;
    }
    b.r = static_cast<float>(std::sqrt(r2) * (1.0 + 1e-5) + 1e-5);
    if (!std::isfinite(b.r) || !std::isfinite(b.c[0]) || !std::isfinite(b.c[1]) || !std::isfinite(b.c[2])) b = DBound{{0.0f, 0.0f, 0.0f}, INFINITY};
    return b;
}

// ---- worker threads of rt_scene_create -------------------------------------------------------------------------------
// RT_AMD_THREADS, else OMP_NUM_THREADS (the GPU boxes set it to their CPU share), else the machine's count; at most 32.
inline unsigned pack_threads() {
    static const unsigned n = [] {
        long v = 0;
        for (const char* name : {"RT_AMD_THREADS", "OMP_NUM_THREADS"}) {
            const char* e = std::getenv(name);
            if (e && *e && (v = std::strtol(e, nullptr, 10)) > 0) break;
            v = 0;
        }
        if (v <= 0) v = static_cast<long>(std::thread::hardware_concurrency());
        return static_cast<unsigned>(v < 1 ? 1 : (v > 32 ? 32 : v));
    }();
    return n;
}
// fn(chunk, lo, hi) over [0, n) cut into at most pack_threads() contiguous chunks of at least `grain` elements; returns the
// number of chunks.  The chunk boundaries depend on n and the thread count only; every user below is written so that
// its RESULT does not depend on them (min/max merges, stable scatters, disjoint stores).
template <class F>
inline unsigned parallel_chunks(size_t n, size_t grain, F&& fn) {
    size_t chunks = std::min<size_t>(pack_threads(), grain ? (n + grain - 1) / grain : 1);
    if (chunks <= 1) {
        if (n) fn(0u, size_t{0}, n);
        return n ? 1u : 0u;
    }
    std::vector<std::thread> th;
    th.reserve(chunks - 1);
    size_t started = 1;  // chunk 0 is the caller's
    try {
        for (; started < chunks; ++started)
            th.emplace_back([&fn, c = started, n, chunks] { fn(static_cast<unsigned>(c), c * n / chunks, (c + 1) * n / chunks); });
    } catch (const std::system_error&) {
        // no more threads to be had: the caller takes the chunks that got none
    }
    fn(0u, size_t{0}, n / chunks);
    for (size_t c = started; c < chunks; ++c) fn(static_cast<unsigned>(c), c * n / chunks, (c + 1) * n / chunks);
    for (auto& t : th) t.join();
    return static_cast<unsigned>(chunks);
}

// The 64-byte record of one of the caller's primitives, made where it is needed (no intermediate copies of a million of
// them): `extra_of[i]` = the DExtra slot of primitive i (cylinders and triangles).
struct PrimSource {
    const rt_scene_desc* d = nullptr;
    const uint32_t* extra_of = nullptr;
    const DMaterial* mats = nullptr;
    DPrim record(uint32_t i) const {
        DPrim p{};
        std::memcpy(p.inv, d->prim_inv + 12 * static_cast<size_t>(i), sizeof(p.inv));
        p.type = d->prim_type[i];
        p.material = d->prim_material[i];
        p.ior = mats[p.material].ior;
        p.extra = p.type >= RT_PRIM_CYLINDER ? extra_of[i] : 0u;
        return p;
    }
};

// The device's own hierarchy over the bounded primitives: a binary tree built top-down with the surface-area
// heuristic (16 bins per axis over the centroids; expected cost of a split = area(left) * n(left) + area(right) *
// n(right); the median along the widest axis when the bins cannot separate anything), at most two primitives
// per leaf (it stands in for Group::divide, group.rs:168-231, which the reference runs once per scene on one thread).
//
// A subtree over n primitives takes exactly 2n - 1 records (a bare record for a single primitive, a box record + two
// primitives for a pair, 1 + left + right above that), so the place of every record in the canonical stream (pre-order,
// low side of every split first) is known on the way DOWN.  The build therefore has no tree to keep and no emission
// pass: each subtree writes its records where they belong, subtrees are independent, and a pool of host threads takes
// them from a shared stack (big nodes also bin and partition their range with all threads).  The result does not
// depend on the number of threads or on who built what.  (The direction-ordered streams of the lane-mode walks are
// laid out from the canonical one afterwards: lane_place, rt_math.h — by the GPU in rt_scene_create.)
struct BvhBuild {
    static constexpr int kBins = 16;
    static constexpr uint32_t kSpawn = 4096;     // subtrees above this many primitives are handed to the pool
    static constexpr uint32_t kWide = 1u << 17;  // nodes above this many primitives use all threads for their own passes
    // A primitive as the build moves it around: its box travels with it, so every pass over a range reads memory in
    // order (a permutation of indices into a box table made the build wait on a cache miss per primitive per level).
    struct Item {
        double mn[3], mx[3];
        uint32_t k;  // index into `ids`: the primitive's rank in world.objects order among the bounded ones
    };
    struct Task {
        uint32_t lo, hi;  // items [lo, hi) ...
        uint32_t side;    // ... of buf[side]
        int depth;
        uint32_t pos;     // first record of the subtree in the canonical stream
    };
    PodVec<Item> buf[2];        // the partitions scatter a range from one into the same range of the other
    std::vector<uint32_t> ids;  // item index of each bounded primitive
    PackedScene* out = nullptr;
    const PrimSource* src = nullptr;
    const std::vector<uint32_t>* item_gate = nullptr;

    // the pool
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Task> stack;
    size_t open_tasks = 0;  // queued or running

    static double area(const HostBox& b) {
        const double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    static void merge(HostBox& into, const Item& it) {
        for (int k = 0; k < 3; ++k) into.mn[k] = std::min(into.mn[k], it.mn[k]), into.mx[k] = std::max(into.mx[k], it.mx[k]);
    }
    static LNode prim_lnode(bool bounded, const double* mn, const double* mx, uint32_t type, uint32_t rec, uint32_t next) {
        LNode ln{};
        for (int k = 0; k < 3; ++k) {
            ln.inv[k] = bounded ? widen(std::nextafter(static_cast<float>(mn[k]), -INFINITY), false) : -INFINITY;
            ln.inv[3 + k] = bounded ? widen(std::nextafter(static_cast<float>(mx[k]), INFINITY), true) : INFINITY;
        }
        ln.next = next;
        ln.ref = (type << kLRefShift) | rec;
        return ln;
    }
    void put_prim(const Item& it, uint32_t at) {
        const uint32_t item = ids[it.k];
        const DPrim r = src->record(src->d->item_ref[item]);
        out->recs[at] = r, out->order[at] = item, out->gate[at] = (*item_gate)[item];
        out->lnodes[at] = prim_lnode(true, it.mn, it.mx, r.type, at, at + 1u);
    }
    // axis: the split axis, or kLeafAxis for a pair of primitives (whose order no direction changes)
    void put_node(const HostBox& box, uint32_t n, uint32_t axis, uint32_t pos) {
        DPrim node{};
        for (int k = 0; k < 3; ++k) {
            node.inv[k] = widen(std::nextafter(static_cast<float>(box.mn[k]), -INFINITY), false);
            node.inv[3 + k] = widen(std::nextafter(static_cast<float>(box.mx[k]), INFINITY), true);
        }
        node.type = REC_NODE;
        node.material = pos + 2u * n - 1u;  // skip link
        out->recs[pos] = node, out->order[pos] = 0, out->gate[pos] = kNoGate;
        LNode ln{};
        for (int k = 0; k < 6; ++k) ln.inv[k] = node.inv[k];
        ln.ref = (REC_NODE << kLRefShift) | axis;
        ln.next = node.material;
        out->lnodes[pos] = ln;
    }
    struct Bins {
        HostBox box[3][kBins];
        uint32_t n[3][kBins];
        Bins() {
            for (auto& a : box)
                for (auto& b : a) b = HostBox::empty();
            std::memset(n, 0, sizeof(n));
        }
    };
    static int bin_of(const Item& b, int axis, double c0, double ext) {
        const int bi = static_cast<int>((0.5 * (b.mn[axis] + b.mx[axis]) - c0) / ext * kBins);
        return bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
    }
    // One subtree: writes its records, hands big children to the pool (push), descends into the rest itself.
    template <class Push>
    void subtree(Task t, Push&& push) {
        const uint32_t lo = t.lo, hi = t.hi, n = hi - lo;
        Item* items = buf[t.side].data();
        if (n == 1) {
            put_prim(items[lo], t.pos);
            return;
        }
        const size_t grain = n > kWide ? size_t{1} << 15 : size_t{n};  // (the top few nodes: the pool has little else to do yet)
        // the node's box and the box of the centroids
        HostBox box = HostBox::empty(), cen = HostBox::empty();
        {
            HostBox pb[32], pc[32];
            const unsigned chunks = parallel_chunks(n, grain, [&](unsigned c, size_t a, size_t e) {
                HostBox b0 = HostBox::empty(), c0 = HostBox::empty();
                for (size_t k = lo + a; k < lo + e; ++k) {
                    const Item& b = items[k];
                    merge(b0, b);
                    const double ctr[3] = {0.5 * (b.mn[0] + b.mx[0]), 0.5 * (b.mn[1] + b.mx[1]), 0.5 * (b.mn[2] + b.mx[2])};
                    c0.add(ctr);
                }
                pb[c] = b0, pc[c] = c0;
            });
            for (unsigned c = 0; c < chunks; ++c) box.merge(pb[c]), cen.merge(pc[c]);
        }
        if (n == 2) {
            put_node(box, n, kLeafAxis, t.pos);
            put_prim(items[lo], t.pos + 1u), put_prim(items[lo + 1], t.pos + 2u);
            return;
        }
        // the bins of the three axes in one pass
        double c0[3], ext[3];
        bool use[3];
        for (int axis = 0; axis < 3; ++axis) {
            c0[axis] = cen.mn[axis], ext[axis] = cen.mx[axis] - cen.mn[axis];
            use[axis] = ext[axis] > 0.0 && std::isfinite(ext[axis]);
        }
        Bins bins;
        auto fill = [&](Bins& w, size_t a, size_t e) {
            for (size_t k = lo + a; k < lo + e; ++k) {
                const Item& b = items[k];
                for (int axis = 0; axis < 3; ++axis) {
                    if (!use[axis]) continue;
                    const int bi = bin_of(b, axis, c0[axis], ext[axis]);
                    merge(w.box[axis][bi], b), w.n[axis][bi]++;
                }
            }
        };
        if (grain >= n) {
            fill(bins, 0, n);
        } else {
            std::vector<Bins> part(pack_threads());
            const unsigned chunks = parallel_chunks(n, grain, [&](unsigned c, size_t a, size_t e) { fill(part[c], a, e); });
            for (unsigned c = 0; c < chunks; ++c)
                for (int axis = 0; axis < 3; ++axis)
                    for (int i = 0; i < kBins; ++i)
                        if (part[c].n[axis][i]) bins.box[axis][i].merge(part[c].box[axis][i]), bins.n[axis][i] += part[c].n[axis][i];
        }
        int best_axis = -1, best_bin = 0;
        double best_cost = INFINITY;
        for (int axis = 0; axis < 3; ++axis) {
            if (!use[axis]) continue;
            double right_area[kBins];
            uint32_t right_n[kBins];
            HostBox acc = HostBox::empty();
            uint32_t cnt = 0;
            double a_acc = area(acc);  // (an empty bin changes neither the box nor its area)
            for (int i = kBins - 1; i > 0; --i) {
                if (bins.n[axis][i]) acc.merge(bins.box[axis][i]), a_acc = area(acc);
                cnt += bins.n[axis][i];
                right_area[i] = a_acc, right_n[i] = cnt;
            }
            acc = HostBox::empty(), cnt = 0, a_acc = area(acc);
            for (int i = 0; i + 1 < kBins; ++i) {  // split between bin i and bin i + 1
                if (bins.n[axis][i]) acc.merge(bins.box[axis][i]), a_acc = area(acc);
                cnt += bins.n[axis][i];
                if (cnt == 0 || right_n[i + 1] == 0) continue;
                const double cost = a_acc * cnt + right_area[i + 1] * right_n[i + 1];
                if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = i;
            }
        }
        uint32_t mid = lo + n / 2, side = t.side;
        int axis = 0;
        if (best_axis >= 0 && t.depth < 40) {  // (deeper than that the splits are lopsided: halve instead)
            axis = best_axis;
            // stable partition by scattering into the other buffer: each side keeps its order
            uint32_t n_low[32];
            auto low = [&](const Item& a) { return bin_of(a, axis, c0[axis], ext[axis]) <= best_bin; };
            const unsigned chunks = parallel_chunks(n, grain, [&](unsigned c, size_t a, size_t e) {
                uint32_t m = 0;
                for (size_t k = lo + a; k < lo + e; ++k) m += low(items[k]) ? 1u : 0u;
                n_low[c] = m;
            });
            uint32_t all_low = 0;
            for (unsigned c = 0; c < chunks; ++c) all_low += n_low[c];
            mid = lo + all_low;
            if (mid != lo && mid != hi) {
                Item* dst = buf[side ^ 1u].data();
                parallel_chunks(n, grain, [&](unsigned c, size_t a, size_t e) {
                    uint32_t l = lo, h = mid + static_cast<uint32_t>(a);
                    for (unsigned q = 0; q < c; ++q) l += n_low[q];
                    h -= l - lo;  // the high elements in front of this chunk
                    for (size_t k = lo + a; k < lo + e; ++k) {
                        if (low(items[k])) dst[l++] = items[k];
                        else dst[h++] = items[k];
                    }
                });
                side ^= 1u;
            }
        }
        if (best_axis < 0 || t.depth >= 40 || mid == lo || mid == hi) {  // nothing to separate by: the median along the widest axis
            axis = 0;
            double widest = -1.0;
            for (int k = 0; k < 3; ++k)
                if (cen.mx[k] - cen.mn[k] > widest) widest = cen.mx[k] - cen.mn[k], axis = k;
            mid = lo + n / 2;
            std::nth_element(items + lo, items + mid, items + hi, [&](const Item& a, const Item& b) {
                const double ca = a.mn[axis] + a.mx[axis], cb = b.mn[axis] + b.mx[axis];
                return ca < cb || (ca == cb && a.k < b.k);
            });
        }
        put_node(box, n, static_cast<uint32_t>(axis), t.pos);
        const uint32_t nl = mid - lo;
        const Task l{lo, mid, side, t.depth + 1, t.pos + 1u}, r{mid, hi, side, t.depth + 1, t.pos + 2u * nl};
        if (n > kSpawn) {
            push(r);
            push(l);
            return;
        }
        subtree(l, push);
        subtree(r, push);
    }
    // Builds the tree over buf[0][0, n) with its first record at `first`.
    void run(uint32_t first) {
        const Task root{0, static_cast<uint32_t>(buf[0].size()), 0u, 0, first};
        buf[1].resize(buf[0].size());
        stack.push_back(root);
        open_tasks = 1;
        auto worker = [this] {
            auto push = [this](const Task& t) {
                {
                    std::lock_guard<std::mutex> g(mu);
                    stack.push_back(t);
                    ++open_tasks;
                }
                cv.notify_one();
            };
            for (;;) {
                Task t;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [this] { return !stack.empty() || open_tasks == 0; });
                    if (stack.empty()) return;
                    t = stack.back();
                    stack.pop_back();
                }
                subtree(t, push);
                {
                    std::lock_guard<std::mutex> g(mu);
                    --open_tasks;
                }
                cv.notify_all();
            }
        };
        const unsigned n_threads = buf[0].size() > kSpawn ? pack_threads() : 1u;
        std::vector<std::thread> th;
        try {
            for (unsigned i = 1; i < n_threads; ++i) th.emplace_back(worker);
        } catch (const std::system_error&) {
            // fewer workers than asked for: the pool still drains (this thread is one of them)
        }
        worker();
        for (auto& t : th) t.join();
    }
};

// The staged top of the hierarchy (rt_math.h kTopK): which nodes are opened — the one with the most records below it
// next, the earlier one among equals — until the region is full, then the region of each direction order in that order's
// pre-order.  Marks the portal roots in the canonical stream (canon[p].ref).  k = 0: no staging (no hierarchy, or more
// primitives in front of the root than a region holds).
struct TopRegions {
    std::vector<LNode> recs;        // 8 x kTopK (unused records: zero)
    std::vector<uint32_t> ret;      // 8 per portal: where the walks of each order go on after that portal's subtree
    uint32_t k = 0;                 // records in use per region
};
inline TopRegions build_top(LNode* canon, uint32_t n, uint32_t n_top, bool has_bvh) {
    TopRegions t;
    t.recs.assign(8 * kTopK, LNode{});
    t.ret.assign(8, 0u);
    if (!has_bvh || n_top + 3u > kTopK || n_top >= n) return t;
    auto is_node = [&](uint32_t p) { return (canon[p].ref >> kLRefShift) == REC_NODE; };
    auto span = [&](uint32_t p) { return is_node(p) ? canon[p].next - p : 1u; };
    std::vector<uint32_t> open;  // nodes opened (canonical positions)
    std::vector<uint32_t> edge{n_top};
    uint32_t count = n_top + 1u;
    for (;;) {
        size_t best = edge.size();
        for (size_t e = 0; e < edge.size(); ++e)
            if (is_node(edge[e]) && (best == edge.size() || span(edge[e]) > span(edge[best]) || (span(edge[e]) == span(edge[best]) && edge[e] < edge[best]))) best = e;
        if (best == edge.size() || count + 2u > kTopK) break;
        const uint32_t p = edge[best], l = p + 1u, r = l + span(l);
        open.push_back(p);
        edge[best] = l, edge.push_back(r);
        count += 2u;
    }
    std::sort(edge.begin(), edge.end());
    uint32_t n_portals = 0;
    for (uint32_t p : edge)
        if (is_node(p)) canon[p].ref |= kPortalFlag | (n_portals++ << kPortalShift);
    t.ret.assign(8 * std::max(1u, n_portals), 0u);
    t.k = count;
    auto idx = [&](uint32_t j) { return j >= count ? n : n + 1u + j; };
    for (uint32_t o = 0; o < 8u; ++o) {
        LNode* reg = t.recs.data() + o * kTopK;
        uint32_t j = 0;
        for (; j < n_top; ++j) reg[j] = canon[j], reg[j].next = idx(j + 1u);
        // pre-order of the opened part in this order; pos = the record's place in this order's stream in memory
        struct Frame {
            uint32_t p, pos;
        };
        std::vector<Frame> todo{{n_top, n_top}};
        std::vector<uint32_t> close;  // region records of opened nodes
        while (!todo.empty()) {
            const Frame f = todo.back();
            todo.pop_back();
            const bool opened = std::find(open.begin(), open.end(), f.p) != open.end();
            LNode rec = canon[f.p];
            if (opened) {
                rec.ref = REC_NODE << kLRefShift;
                const uint32_t axis = canon[f.p].ref & 3u, l = f.p + 1u, r = l + span(l);
                const bool flip = ((o >> axis) & 1u) != 0u;
                const uint32_t first = flip ? r : l, second = flip ? l : r;
                todo.push_back({second, f.pos + 1u + span(first)});
                todo.push_back({first, f.pos + 1u});
            } else if (is_node(f.p)) {
                rec.ref = (REC_PORTAL << kLRefShift) | (f.pos + 1u);  // the subtree's first child, in this order's stream
                t.ret[8u * ((canon[f.p].ref & kLRefMask) >> kPortalShift) + o] = idx(j + 1u);
            }
            rec.next = idx(j + 1u);  // (opened nodes: set below, once the size of what is under them is known)
            reg[j] = rec;
            if (opened) close.push_back(j);
            ++j;
        }
        // skip links of the opened nodes: an opened node is followed by its two subtrees, whose leaves are the primitives
        // and portals at the region's edge
        for (const uint32_t at : close) {
            uint32_t need = 1, q = at;
            while (need) {
                need = (reg[q].ref >> kLRefShift) == REC_NODE ? need + 1u : need - 1u;
                ++q;
            }
            reg[at].next = idx(q);
        }
    }
    return t;
}
// All eight direction-ordered streams from the canonical one, on the host (the scalar spec of the tests; rt_scene_create
// lets the GPU do it): streams = 8 x n records.
inline void place_streams_on_host(const LNode* canon, uint32_t n, uint32_t n_top, const uint32_t* portal_ret, LNode* streams) {
    parallel_chunks(n, 1u << 12, [&](unsigned, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) lane_place(canon, n, n_top, portal_ret, static_cast<uint32_t>(i), streams);
    });
}

#ifdef RT_PACK_TIMING  // (scripts/pack_time.cpp: where the time of a large scene goes)
#include <chrono>
inline double g_pack_marks[16];
#define RT_PACK_MARK(i) g_pack_marks[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()
#else
#define RT_PACK_MARK(i) ((void)0)
#endif
// Returns RT_OK or a negative rt_status; `err` receives the message.
// host_streams: lay the seven other direction-ordered LNode streams out here as well (the host simulator of the tests);
// rt_scene_create makes them on the GPU from the canonical one (lane_streams_kernel, rt_aux_kernels.h).
// staged_top: pick the top regions of the lane-mode walks (rt_math.h kTopK); false = plain streams (experiments).
inline int pack_scene(const rt_scene_desc* d, PackedScene& out, std::string& err, bool host_streams = true, bool staged_top = true) {
    auto bad = [&](const char* m) {
        err = m;
        return static_cast<int>(RT_ERR_BAD_ARGUMENT);
    };
    if (!d) return bad("scene description is null");
    if (d->abi_version != RT_ABI_VERSION) return bad("abi_version mismatch");
    if (d->n_prims && (!d->prim_type || !d->prim_inv || !d->prim_material)) return bad("primitive arrays are null");
    if (d->n_nodes && (!d->node_min || !d->node_max)) return bad("node arrays are null");
    if (d->n_items && (!d->item_ref || !d->item_next)) return bad("item arrays are null");
    if (d->n_materials && (!d->mat_param || !d->mat_pattern)) return bad("material arrays are null");
    if (d->n_patterns && (!d->pat_type || !d->pat_color_a || !d->pat_color_b || !d->pat_inv)) return bad("pattern arrays are null");
    if (d->n_lights && (!d->light_type || !d->light_pos || !d->light_uvec || !d->light_vvec || !d->light_usteps ||
                        !d->light_vsteps || !d->light_color))
        return bad("light arrays are null");
    if (d->n_lights > RT_MAX_LIGHTS) {
        err = "too many lights (RT_MAX_LIGHTS)";
        return RT_ERR_CAPACITY;
    }
    if (d->n_prims >= 0x7FFFFFFFu || d->n_items >= 0x3FFFFFFFu) {
        err = "too many primitives";
        return RT_ERR_CAPACITY;
    }

    RT_PACK_MARK(0);
    out = PackedScene{};
    const int n64 = (d->prim_inv64 ? 1 : 0) + (d->prim_param64 ? 1 : 0) + (d->mat_ior64 ? 1 : 0) + (d->pat_inv64 ? 1 : 0);
    if (n64 != 0 && n64 != 4) return bad("the f64 tables (prim_inv64, prim_param64, mat_ior64, pat_inv64) come all four or not at all");
    out.has_f64 = n64 == 4;
    for (uint32_t i = 0; i < d->n_patterns; ++i) {
        if (d->pat_type[i] > RT_PAT_CHECKER) {
            err = "unknown pattern tag";
            return RT_ERR_UNSUPPORTED;
        }
        DPattern p{};
        std::memcpy(p.inv, d->pat_inv + 12 * i, sizeof(p.inv));
        std::memcpy(p.a, d->pat_color_a + 3 * i, sizeof(p.a));
        std::memcpy(p.b, d->pat_color_b + 3 * i, sizeof(p.b));
        p.type = d->pat_type[i];
        out.pats.push_back(p);
        if (out.has_f64) {
            XPattern xp{};
            std::memcpy(xp.inv, d->pat_inv64 + 12 * i, sizeof(xp.inv));
            out.xpats.push_back(xp);
        }
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->mat_pattern[i] >= d->n_patterns) return bad("material pattern index out of range");
        const float* m = d->mat_param + 8 * i;
        out.mats.push_back(DMaterial{m[0], m[1], m[2], m[3], m[4], m[5], m[6], d->mat_pattern[i]});
        out.any_spawn |= !(std::fabs(m[5]) <= kEps) || !(std::fabs(m[6]) <= kEps);
        out.any_glass |= !(std::fabs(m[6]) <= kEps);
        if (out.has_f64) out.xior.push_back(d->mat_ior64[i]);
    }
    RT_PACK_MARK(1);
    // by primitive index: validation and the DExtra slots (a running count), then the slots filled by all threads
    std::vector<uint32_t> extra_of(d->n_prims, 0u);
    uint32_t n_extras = 0;
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        if (d->prim_type[i] > RT_PRIM_SMOOTH_TRIANGLE) {
            err = "unknown primitive tag";
            return RT_ERR_UNSUPPORTED;
        }
        if (d->prim_material[i] >= d->n_materials) return bad("primitive material index out of range");
        if (d->prim_type[i] >= RT_PRIM_CYLINDER) extra_of[i] = n_extras++;
    }
    if (n_extras && !d->prim_param) return bad("prim_param is null but a cylinder/triangle needs it");
    out.extras.resize(n_extras);
    if (out.has_f64) out.xextras.resize(n_extras);
    parallel_chunks(n_extras ? d->n_prims : 0, 1u << 14, [&](unsigned, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            if (d->prim_type[i] < RT_PRIM_CYLINDER) continue;
            DExtra e{};
            std::memcpy(e.f, d->prim_param + 24 * i, 24 * sizeof(float));
            out.extras[extra_of[i]] = e;
            if (out.has_f64) {
                XExtra xe{};
                std::memcpy(xe.f, d->prim_param64 + 24 * i, sizeof(xe.f));
                out.xextras[extra_of[i]] = xe;
            }
        }
    });
    const PrimSource src{d, extra_of.data(), out.mats.data()};
    if (out.extras.empty()) out.extras.push_back(DExtra{});  // never hand a null table to the kernel
    if (out.xextras.empty()) out.xextras.push_back(XExtra{});

    RT_PACK_MARK(2);
    // The caller's item tree: pre-order with forward skip links.  i < next <= n_items and proper
    // nesting are checked; Groups become gates, primitives remember their innermost Group.
    std::vector<uint32_t> item_gate(d->n_items, kNoGate);
    std::vector<uint32_t> prim_items;  // items that are primitives, in world.objects order
    {
        std::vector<uint32_t> open_end, open_gate;  // enclosing subtrees: their ends and gates
        std::vector<uint8_t> prim_seen(d->n_prims, 0);
        for (uint32_t i = 0; i < d->n_items; ++i) {
            while (!open_end.empty() && open_end.back() == i) open_end.pop_back(), open_gate.pop_back();
            const uint32_t ref = d->item_ref[i], next = d->item_next[i];
            if (next <= i || next > d->n_items) return bad("item_next out of range");
            if (!open_end.empty() && next > open_end.back()) return bad("item subtree crosses its parent's end");
            const uint32_t enclosing = open_gate.empty() ? kNoGate : open_gate.back();
            if (ref & RT_ITEM_NODE) {
                const uint32_t n = ref & ~RT_ITEM_NODE;
                if (n >= d->n_nodes) return bad("item node index out of range");
                DGate g{};
                for (int k = 0; k < 3; ++k) {
                    g.mn[k] = d->node_min[3 * n + k], g.mx[k] = d->node_max[3 * n + k];
                    // a box of zero thickness is never entered (tmin < tmax is strict):
                    // that quirk is part of the reference's output, so such an axis is not widened
                    if (g.mn[k] != g.mx[k]) g.mn[k] = widen(g.mn[k], false), g.mx[k] = widen(g.mx[k], true);
                }
                g.parent = enclosing;
                if (next > i + 1) {
                    open_end.push_back(next);
                    open_gate.push_back(static_cast<uint32_t>(out.gates.size()));
                }
                out.gates.push_back(g);
                XGate xg{};
                for (int k = 0; k < 3; ++k) {
                    xg.mn[k] = d->node_box64 ? d->node_box64[6 * n + k] : static_cast<double>(d->node_min[3 * n + k]);
                    xg.mx[k] = d->node_box64 ? d->node_box64[6 * n + 3 + k] : static_cast<double>(d->node_max[3 * n + k]);
                }
                out.xgates.push_back(xg);
            } else {
                if (ref >= d->n_prims) return bad("item primitive index out of range");
                if (next != i + 1) return bad("a primitive item must link to the next item");
                if (prim_seen[ref]) return bad("primitive referenced by more than one item");
                prim_seen[ref] = 1;
                item_gate[i] = enclosing;
                prim_items.push_back(i);
                out.exotic |= d->prim_type[ref] >= RT_PRIM_CYLINDER;
                out.has_gates |= enclosing != kNoGate;
            }
        }
    }
    if (out.gates.empty()) out.gates.push_back(DGate{}), out.xgates.push_back(XGate{});

    RT_PACK_MARK(3);
    // Lay the stream out: unbounded primitives (and everything, in a flat scene) in front, in world.objects order, then the
    // hierarchy over the bounded ones.
    BvhBuild b;
    std::vector<uint32_t> front;  // items in front of the tree
    {
        const bool flat = prim_items.size() <= kFlatLimit && !out.has_gates;
        std::vector<HostBox> box(flat ? 0 : prim_items.size());
        parallel_chunks(box.size(), 1u << 14, [&](unsigned, size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                const uint32_t ref = d->item_ref[prim_items[k]];
                box[k] = prim_world_box(d->prim_type[ref], d->prim_inv + 12 * ref, d->prim_param ? d->prim_param + 24 * ref : nullptr);
            }
        });
        // A primitive strictly inside the box of every Group around it needs no gate: a line that meets it anywhere runs
        // through the interior of those boxes, tmin < tmax (group.rs:107) holds in the reference and here.  What keeps its
        // gates: the few primitives a Group's box is drawn around (they touch it), unbounded ones, and primitives that lie
        // OUTSIDE the boxes around them (shapes_zoo's cube, profiles/r03_group_boxes.txt).
        if (out.has_gates) {
            parallel_chunks(box.size(), 1u << 14, [&](unsigned, size_t lo, size_t hi) {
                for (size_t k = lo; k < hi; ++k) {
                    if (!box[k].bounded) continue;
                    bool inside = true;
                    for (uint32_t g = item_gate[prim_items[k]]; g != kNoGate && inside; g = out.gates[g].parent) {
                        const XGate& xg = out.xgates[g];
                        for (int c = 0; c < 3; ++c) {
                            const double m = 1e-5 * (1.0 + std::fabs(box[k].mn[c]) + std::fabs(box[k].mx[c]));
                            inside = inside && xg.mn[c] + m < box[k].mn[c] && box[k].mx[c] + m < xg.mx[c];
                        }
                    }
                    if (inside) item_gate[prim_items[k]] = kNoGate;
                }
            });
            out.has_gates = false;
            for (uint32_t item : prim_items) out.has_gates |= item_gate[item] != kNoGate;
        }
        size_t n_bounded = 0;
        for (const HostBox& x : box) n_bounded += x.bounded ? 1 : 0;
        if (flat || n_bounded <= 2) {
            front = prim_items;
        } else {
            b.buf[0].reserve(n_bounded), b.ids.reserve(n_bounded);
            for (size_t k = 0; k < box.size(); ++k) {
                if (!box[k].bounded) {
                    front.push_back(prim_items[k]);
                    continue;
                }
                BvhBuild::Item it;
                for (int c = 0; c < 3; ++c) it.mn[c] = box[k].mn[c], it.mx[c] = box[k].mx[c];
                it.k = static_cast<uint32_t>(b.ids.size());
                b.buf[0].push_back(it), b.ids.push_back(prim_items[k]);
            }
            out.has_bvh = true;
        }
    }
    RT_PACK_MARK(4);
    const size_t n_top = front.size();
    const size_t total = std::max<size_t>(1, n_top + (out.has_bvh ? 2 * b.ids.size() - 1 : 0));
    if (total >= (1u << kLRefShift)) {
        err = "too many records for the compact streams";
        return RT_ERR_CAPACITY;
    }
    out.n_items = static_cast<uint32_t>(n_top + (out.has_bvh ? 2 * b.ids.size() - 1 : 0));
    out.n_top = static_cast<uint32_t>(n_top);
    out.recs.resize(total), out.order.resize(total), out.gate.resize(total);
    out.lnodes.resize(total);
    if (out.n_items == 0) {  // the padding record of an empty scene (never walked)
        out.recs[0] = DPrim{}, out.order[0] = 0u, out.gate[0] = kNoGate, out.lnodes[0] = BvhBuild::prim_lnode(false, nullptr, nullptr, 0u, 0u, 1u);
    }
    for (size_t i = 0; i < n_top; ++i) {
        const uint32_t item = front[i];
        const DPrim r = src.record(d->item_ref[item]);
        out.recs[i] = r, out.order[i] = item, out.gate[i] = item_gate[item];
        const HostBox pb = prim_world_box(r.type, r.inv, out.extras[r.extra].f);
        const LNode ln = BvhBuild::prim_lnode(pb.bounded, pb.mn, pb.mx, r.type, static_cast<uint32_t>(i), static_cast<uint32_t>(i) + 1u);
        out.lnodes[i] = ln;
    }
    RT_PACK_MARK(5);
    if (out.has_bvh) {
        b.out = &out, b.src = &src, b.item_gate = &item_gate;
        b.run(static_cast<uint32_t>(n_top));
    }
    {
        TopRegions top = build_top(out.lnodes.data(), static_cast<uint32_t>(total), out.n_top, out.has_bvh && staged_top);
        out.top = std::move(top.recs), out.portal_ret = std::move(top.ret), out.top_k = top.k;
    }
    if (out.has_bvh && host_streams) {
        out.lstreams.resize(8 * total);
        place_streams_on_host(out.lnodes.data(), static_cast<uint32_t>(total), out.n_top, out.portal_ret.data(), out.lstreams.data());
    }
    RT_PACK_MARK(6);
    out.bound.resize(total);
    parallel_chunks(total, 1u << 14, [&](unsigned, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const DPrim& r = out.recs[i];
            out.bound[i] = r.type == REC_NODE ? node_ball(r) : prim_world_ball(r.type, r.inv, out.extras[r.extra].f);
        }
    });
    RT_PACK_MARK(7);
    if (out.has_f64) {  // record i of the stream is the primitive of item order[i]
        out.xprims.resize(out.recs.size());
        if (out.n_items == 0) out.xprims[0] = XPrim{};
        parallel_chunks(out.n_items, 1u << 14, [&](unsigned, size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                if (out.recs[i].type == REC_NODE) {
                    out.xprims[i] = XPrim{};  // (never read; the table is not zeroed when it is sized)
                    continue;
                }
                const uint32_t ref = d->item_ref[out.order[i]];
                std::memcpy(out.xprims[i].inv, d->prim_inv64 + 12 * static_cast<size_t>(ref), sizeof(out.xprims[i].inv));
            }
        });
    }
    if (out.xprims.empty()) out.xprims.push_back(XPrim{});
    if (out.xpats.empty()) out.xpats.push_back(XPattern{});
    if (out.xior.empty()) out.xior.push_back(1.0);

    for (uint32_t i = 0; i < d->n_lights; ++i) {
        if (d->light_type[i] > RT_LIGHT_AREA) {
            err = "unknown light tag";
            return RT_ERR_UNSUPPORTED;
        }
        DLight l{};
        std::memcpy(l.pos, d->light_pos + 3 * i, sizeof(l.pos));
        std::memcpy(l.uvec, d->light_uvec + 3 * i, sizeof(l.uvec));
        std::memcpy(l.vvec, d->light_vvec + 3 * i, sizeof(l.vvec));
        std::memcpy(l.color, d->light_color + 3 * i, sizeof(l.color));
        l.type = d->light_type[i];
        out.has_area |= l.type == RT_LIGHT_AREA;
        l.usteps = l.type == RT_LIGHT_AREA ? d->light_usteps[i] : 1u;
        l.vsteps = l.type == RT_LIGHT_AREA ? d->light_vsteps[i] : 1u;
        if (l.usteps == 0 || l.vsteps == 0) return bad("area light with zero cells");
        uint64_t cells = static_cast<uint64_t>(l.usteps) * l.vsteps;
        if (cells > RT_MAX_LIGHT_CELLS) {
            err = "area light has too many cells (RT_MAX_LIGHT_CELLS)";
            return RT_ERR_CAPACITY;
        }
        l.inv_samples = 1.0f / static_cast<float>(cells);
        if (cells > out.max_cells) out.max_cells = static_cast<uint32_t>(cells);
        out.lights.push_back(l);
        XLight xl{};
        for (int k = 0; k < 3; ++k) {
            xl.pos[k] = d->light_geom64 ? d->light_geom64[9 * i + k] : static_cast<double>(l.pos[k]);
            xl.uvec[k] = d->light_geom64 ? d->light_geom64[9 * i + 3 + k] : static_cast<double>(l.uvec[k]);
            xl.vvec[k] = d->light_geom64 ? d->light_geom64[9 * i + 6 + k] : static_cast<double>(l.vvec[k]);
        }
        out.xlights.push_back(xl);
    }
    RT_PACK_MARK(8);
    out.n_lights = d->n_lights;
    if (out.lights.empty()) out.lights.push_back(DLight{}), out.xlights.push_back(XLight{});
    return RT_OK;
}

// rt_options.max_depth as the kernels take it (world.rs:11,24): 0 in a zeroed struct is RECURSION_DEPTH = 5,
// RT_FLAG_DEPTH_ZERO is depth 0 itself.  false: the flag together with a nonzero max_depth.
inline bool options_max_depth(const rt_options* o, uint32_t& depth) {
    depth = 5u;
    if (!o) return true;
    if (o->flags & RT_FLAG_DEPTH_ZERO) {
        depth = 0u;
        return o->max_depth == 0u;
    }
    if (o->max_depth) depth = o->max_depth;
    return true;
}

inline XCamera pack_camera64(const rt_camera_desc& c) {
    XCamera x{};
    x.pixel_size = c.pixel_size64, x.half_width = c.half_width64, x.half_height = c.half_height64;
    std::memcpy(x.inv, c.inv64, sizeof(x.inv));
    return x;
}
inline DCamera pack_camera(const rt_camera_desc& c) {
    DCamera d{};
    d.pixel_size = c.pixel_size, d.half_width = c.half_width, d.half_height = c.half_height;
    d.hsize = c.hsize, d.vsize = c.vsize;
    std::memcpy(d.inv, c.inv, sizeof(d.inv));
    return d;
}

}  // namespace rtk
