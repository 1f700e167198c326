// rt_math.h — per-lane FP32 arithmetic of the gfx950 render kernels: ray/primitive intersection,
// hit preparation, patterns, Phong terms, Schlick/refraction, the jitter hash.
//
// Every function is `RT_HD` (host + device) so that the SAME arithmetic can be compiled by g++
// into tests/cpp/hostsim.cpp (a test-only executable spec used to debug f32-vs-f64 parity in a
// container without a GPU).  The product library only ever runs these on the GPU.
//
// Numerical design (DESIGN.md §"FP32 against an f64 reference with EPSILON = 1e-5"):
//  * the reference offsets secondary-ray origins by normal * 1e-5 (intersection.rs:52-53), which
//    is below FP32 resolution for |coordinate| >~ 100 and comparable to FP32 error at ~10.  A ray
//    that starts on primitive P therefore carries P's id, the hit point SNAPPED onto P's surface
//    in P's object space (`q`) and the object-space offset (`dq`); the intersection with P itself
//    is computed from (q, dq) analytically, so the sign of the near-zero root — the thing that
//    decides self-shadowing — is exact, and patterns are evaluated at q + dq;
//  * the sphere discriminant uses the perpendicular-distance form, not b^2 - 4ac (sphere.rs:49),
//    which cancels catastrophically in FP32 for far ray origins.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtk {

constexpr float kEps = 1e-5f;  // float.rs:3

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32).  `a / b` and sqrtf()
// would expand to the ~10-instruction correctly-rounded sequences; nothing on this path needs the
// last ulp (every ulp-sensitive step is formulated so that it does not cancel, see the header), and
// x * rcp(0) keeps the IEEE +-inf / NaN outcomes the slab tests rely on (cube.rs:83).
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }
RT_HD float sqrt_f(float x) { return __builtin_amdgcn_sqrtf(x); }
RT_HD float rsq_f(float x) { return __builtin_amdgcn_rsqf(x); }
RT_HD float fast_log2(float x) { return __builtin_amdgcn_logf(x); }  // v_log_f32; gates only, never a result
RT_HD float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32; gates only, never a result
#else
RT_HD float fast_log2(float x) { return log2f(x); }
RT_HD float fast_exp2(float x) { return exp2f(x); }
#ifdef RT_HOST_ULP_NOISE  // tests/cpp/hostsim.cpp only: the host's exact 1/x, sqrt, 1/sqrt moved by -1, 0 or +1 ulp, chosen by a hash
                          // of the argument's bits and a seed — what the hardware's v_rcp / v_sqrt / v_rsq may do to any decision
extern unsigned rt_host_ulp_noise;  // 0 = off
inline float rt_ulp_noise(float r, float x) {
    if (!rt_host_ulp_noise || !(r == r) || r == INFINITY || r == -INFINITY) return r;
    unsigned b;
    __builtin_memcpy(&b, &x, 4);
    b = (b ^ rt_host_ulp_noise) * 0x9E3779B1u;
    b ^= b >> 15;
    const unsigned k = b % 3u;  // 0: as it is, 1: one ulp up, 2: one ulp down
    return k == 0u ? r : nextafterf(r, k == 1u ? INFINITY : -INFINITY);
}
RT_HD float rcp_f(float x) { return rt_ulp_noise(1.0f / x, x); }
RT_HD float sqrt_f(float x) { return rt_ulp_noise(sqrtf(x), x); }
RT_HD float rsq_f(float x) { return rt_ulp_noise(1.0f / sqrtf(x), x); }
#else
RT_HD float rcp_f(float x) { return 1.0f / x; }
RT_HD float sqrt_f(float x) { return sqrtf(x); }
RT_HD float rsq_f(float x) { return 1.0f / sqrtf(x); }
#endif
#endif
RT_HD float div_f(float a, float b) { return a * rcp_f(b); }

// Device-side records (built by rt_scene_create from rt_scene_desc; see DESIGN.md "HBM layout").
// The scene is ONE stream of 64-byte records in traversal order (world.objects order, groups in
// depth-first pre-order): record i is a primitive, or a group node whose subtree is the records
// (i, next).  A primitive is identified everywhere on the device by its position in the stream.
struct DPrim {  // 64 B — one s_load_dwordx16 when the index is wave-uniform
    float inv[12];      // primitive: transform_inverse rows.  node: box min = inv[0..2], max = inv[3..5]
    uint32_t type;      // PRIM_* or REC_NODE
    uint32_t material;  // primitive: material index.  node: `next`, the record after its subtree
    float ior;          // copy of the material's index_of_refraction (container pass)
    uint32_t extra;     // index into DExtra (cylinder, triangles)
};
// The walk order of the lanes that traverse ON THEIR OWN (rt_kernel.h, lane mode): the same hierarchy laid out
// EIGHT times, once per sign pattern of a ray's direction, each time in depth-first pre-order with the child
// that such a ray reaches first in front (a node split along x puts its low-x child first for d.x >= 0, its
// high-x child first for d.x < 0).  A closest-hit walk in its ray's own order meets the near things first and
// the box tests cull most of what lies behind them.  32 bytes a record: a node's box and skip link, or a
// primitive's own world-space box (the same widening as the node boxes; +-inf for planes and unbounded cylinders)
// and the position of its 64-byte record in the canonical stream — a lane only fetches the record of a
// primitive whose box its ray meets.  Stream k starts at lnodes[k * n_items]; stream 0 is the canonical order.
struct alignas(32) LNode {
    float inv[6];   // box min = inv[0..2], max = inv[3..5] (the field names of DPrim: the box tests take either)
    uint32_t next;  // node: the position after its subtree IN THIS STREAM.  primitive: its position + 1
    uint32_t ref;   // primitive: type << kLRefShift | record index in the canonical stream.  node: REC_NODE << kLRefShift | split axis
};
constexpr uint32_t kLRefShift = 29u;
constexpr uint32_t kLRefMask = (1u << kLRefShift) - 1u;
// (a node's low bits: the axis it was split along, 0..2, or kLeafAxis for a node over two primitives, whose order is the
// same in every stream.  The walks never look at them; the kernel that lays the direction-ordered streams out does.)
constexpr uint32_t kLeafAxis = 3u;

// The staged top of the hierarchy (N1).  The hierarchy walks of the lane-mode kernels wait on dependent loads, and what
// most of their steps look at is the same few records: the primitives in front of the root and the top of the tree.  The
// scene therefore carries, per direction order, a TOP REGION of at most kTopK records that every wave copies into LDS
// when it starts: those primitives, the largest nodes of the tree in that order's pre-order (rt_pack.h build_top picks
// them: the node with the most primitives below it is opened next), and where the region ends a PORTAL per subtree left
// out — the subtree root's box, and the place of its first child in that order's stream in memory.  A walk starts in
// the region; a lane whose ray meets a portal's box goes on in memory, and the skip links that would leave that subtree
// lead back to the record after the portal (lane_place below writes them).  Indices: 0..n-1 = the stream in memory,
// n = the end of the walk, n + 1 + j = record j of the region.
constexpr uint32_t kTopK = 16u;        // 8 orders x 16 records x 32 bytes = 4 KB of LDS a wave (what 16 waves a CU leave)
constexpr uint32_t kPortalFlag = 4u;   // canonical stream, a node's ref: bit 2 = a portal leads here, bits 3.. = which
constexpr uint32_t kPortalShift = 3u;
constexpr uint32_t kNoPortal = 0xFFFFFFFFu;

struct DExtra {  // 128 B: cylinder {min,max,closed}; triangle {v0,e0,e1,normal,n0,n1,n2}
    float f[32];
};
struct DMaterial {  // 32 B
    float ambient, diffuse, specular, shininess, ior, reflectivity, transparency;
    uint32_t pattern;
};
struct DPattern {  // 80 B
    float inv[12];
    float a[3];
    uint32_t type;
    float b[3];
    float pad;
};
struct DLight {  // 64 B
    float pos[3];
    uint32_t type;
    float uvec[3];
    uint32_t usteps;
    float vvec[3];
    uint32_t vsteps;
    float color[3];
    float inv_samples;
};
struct DCamera {
    float pixel_size, half_width, half_height;
    uint32_t hsize, vsize;
    float inv[12];
};

enum : uint32_t { PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_CUBE = 2, PRIM_CYLINDER = 3, PRIM_TRIANGLE = 4, PRIM_SMOOTH = 5,
                  REC_NODE = 6,
                  REC_PORTAL = 7 };  // (LNode records of a staged top only, below)
enum : uint32_t { PAT_SOLID = 0, PAT_STRIPE = 1, PAT_GRADIENT = 2, PAT_RING = 3, PAT_CHECKER = 4 };
constexpr uint32_t kItemNode = 0x80000000u;
constexpr uint32_t kNoPrim = 0xFFFFFFFFu;

// Lays record i of the canonical stream out in the eight direction-ordered streams (streams[o * n + ...]): walks down
// from the root to the record — a subtree is the span [p, next(p)) of the canonical stream, its low child starts at p + 1
// and its high child right after the low child's span — adding up where that path leads in each order (far child first
// where bit `axis` of the order's number is set).  portal_ret[8 * id + o] = where the walks of order o go on after the
// subtree of portal `id`.  One thread of lane_streams_kernel (rt_aux_kernels.h) per record, or a host loop (rt_pack.h).
RT_HD void lane_place(const LNode* canon, uint32_t n, uint32_t n_top, const uint32_t* portal_ret, uint32_t i, LNode* streams) {
    const LNode me = canon[i];
    if (i < n_top) {  // in front of the root: the same place in every stream
        for (uint32_t o = 0; o < 8u; ++o) streams[static_cast<size_t>(o) * n + i] = me;
        return;
    }
    const bool me_node = (me.ref >> kLRefShift) == REC_NODE;
    const uint32_t my_span = me_node ? me.next - i : 1u;
    uint32_t pos[8], sub_end[8];
    for (uint32_t o = 0; o < 8u; ++o) pos[o] = n_top, sub_end[o] = 0u;
    uint32_t portal = kNoPortal;
    uint32_t p = n_top;
    for (int guard = 0; guard < 512; ++guard) {
        const uint32_t p_ref = canon[p].ref, p_end = canon[p].next;
        const bool p_node = (p_ref >> kLRefShift) == REC_NODE;
        if (p_node && (p_ref & kPortalFlag) && portal == kNoPortal) {
            portal = (p_ref & kLRefMask) >> kPortalShift;
            for (uint32_t o = 0; o < 8u; ++o) sub_end[o] = pos[o] + (p_end - p);
        }
        if (p == i || !p_node) break;
        const uint32_t axis = p_ref & 3u, l = p + 1u;
        if (l >= n) break;
        const LNode low = canon[l];
        const uint32_t span_l = (low.ref >> kLRefShift) == REC_NODE ? low.next - l : 1u;
        const uint32_t r = l + span_l, span_r = p_end - r;
        const bool high = i >= r;
        for (uint32_t o = 0; o < 8u; ++o) {
            const bool flip = ((o >> axis) & 1u) != 0u;
            pos[o] += 1u + (high ? (flip ? 0u : span_l) : (flip ? span_r : 0u));
        }
        p = high ? r : l;
    }
    if (p != i) return;  // (links that do not nest: pack_scene never makes them)
    for (uint32_t o = 0; o < 8u; ++o) {
        if (pos[o] >= n || my_span > n - pos[o]) continue;
        LNode out = me;
        out.next = pos[o] + my_span;
        if (portal != kNoPortal && out.next == sub_end[o]) out.next = portal_ret[8u * portal + o];
        if (me_node) out.ref = (REC_NODE << kLRefShift) | (me.ref & 3u);
        streams[static_cast<size_t>(o) * n + pos[o]] = out;
    }
}

struct V3 {
    float x, y, z;
};
RT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT_HD float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RT_HD V3 cross(V3 a, V3 b) {
    return V3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
// a*b - c*d with one rounding's worth of error (Kahan): used where nearly parallel vectors cancel.
RT_HD float diff_of_products(float a, float b, float c, float d) {
    float w = c * d;
    float e = fmaf(c, d, -w);  // exact rounding error of w
    float f = fmaf(a, b, -w);
    return f - e;
}
RT_HD V3 cross_exact(V3 a, V3 b) {
    return V3{diff_of_products(a.y, b.z, a.z, b.y), diff_of_products(a.z, b.x, a.x, b.z), diff_of_products(a.x, b.y, a.y, b.x)};
}
RT_HD V3 madd(V3 d, float t, V3 o) { return V3{fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z)}; }  // o + d*t
RT_HD V3 normalize(V3 a) {
    return a * rsq_f(dot(a, a));
}

// 3x4 affine (rows) applied to a point / a vector / transposed-linear applied to a normal.
RT_HD V3 xform_point(const float* m, V3 p) {
    return V3{fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[3]))), fmaf(m[4], p.x, fmaf(m[5], p.y, fmaf(m[6], p.z, m[7]))),
              fmaf(m[8], p.x, fmaf(m[9], p.y, fmaf(m[10], p.z, m[11])))};
}
RT_HD V3 xform_vector(const float* m, V3 v) {
    return V3{fmaf(m[0], v.x, fmaf(m[1], v.y, m[2] * v.z)), fmaf(m[4], v.x, fmaf(m[5], v.y, m[6] * v.z)),
              fmaf(m[8], v.x, fmaf(m[9], v.y, m[10] * v.z))};
}
RT_HD V3 xform_normal(const float* m, V3 n) {  // transform_inverse.transpose() * n, w dropped (shape.rs:89-90)
    return V3{fmaf(m[0], n.x, fmaf(m[4], n.y, m[8] * n.z)), fmaf(m[1], n.x, fmaf(m[5], n.y, m[9] * n.z)),
              fmaf(m[2], n.x, fmaf(m[6], n.y, m[10] * n.z))};
}

// float.rs:5-37 in f32
RT_HD bool approx(float a, float b) {
    if (isinf(a) && isinf(b)) return a == b;
    return fabsf(a - b) <= kEps;
}
// partial_cmp(a,b) == Greater  (NaN => Greater, float.rs:29-37)
// (Without approx()'s equal-infinities case: max_by3/min_by3 only use the comparison to pick one of
// two VALUES, and two infinities of the same sign are the same value whichever is picked.)
RT_HD bool cmp_greater(float a, float b) { return !(fabsf(a - b) <= kEps) & !(a < b); }
// cube.rs:63-73: max_by keeps the last of approx-equal maxima, min_by the first of approx-equal minima
RT_HD float max_by3(float a, float b, float c) {
    float m = cmp_greater(a, b) ? a : b;
    return cmp_greater(m, c) ? m : c;
}
RT_HD float min_by3(float a, float b, float c) {
    float m = cmp_greater(a, b) ? b : a;
    return cmp_greater(m, c) ? c : m;
}

// Nearest-hit selection with the reference's tie rule.  Intersection::sort (intersection.rs:105-115)
// calls two t "Equal" when |dt| <= 1e-5 and the (insertion) sort keeps equal elements in the order
// they were produced, i.e. world.objects order; Intersection::hit then takes the first t > 0.0.
// Primitives are visited in that same order here, so a later candidate replaces the current best
// only if it is smaller AND not approximately equal (e.g. the edge where two walls meet).
//
// FP32 cannot resolve |dt| below ~1e-6 * t, so a tie that is EXACT in the reference (a ray hitting
// the edge where two planes meet — every example scene has such edges, and cameras on their
// symmetry planes) would be decided by rounding noise once t > ~10.  The tie window is therefore
// 1e-5 * max(1, t/4): the reference's own window wherever FP32 can resolve it, wider only where it
// cannot.
RT_HD float tie_window(float t) { return kEps * fmaxf(1.0f, 0.25f * fabsf(t)); }
// Two candidates closer than this (or a candidate this close to t = 0) are more than FP32 can order
// against the reference's 1e-5 tie rule: the exact chain (rt_exact.h) settles them in f64.
// (64 tie windows since the stress fuzz family: a ray running within 0.002 rad of a face two cubes share knows each
// crossing to 1e-7 / |d_k| = 5e-5 only; 4 windows missed that pair, 64 cost nothing measurable — C4 6.87 vs 6.87 ms,
// cover_image 1.22 vs 1.22, profiles/r02_random_worlds.txt)
#ifndef RT_AMB_FACTOR
#define RT_AMB_FACTOR 64.0f
#endif
RT_HD float amb_window(float t) { return RT_AMB_FACTOR * kEps * fmaxf(1.0f, 0.25f * fabsf(t)); }
RT_HD bool closer_hit(float t, float best) { return t > 0.0f && t < best - tie_window(t); }  // best may be +inf
// Candidates do not always arrive in world.objects order (the device walks its own hierarchy): an
// approximately equal t is a TIE, which the caller resolves by the primitives' order keys.
RT_HD bool ties_hit(float t, float best) { return t > 0.0f && fabsf(t - best) <= tie_window(t); }
// Does an intersection of a primitive come before the hit (t_hit) in that sorted list?  `ord` /
// `ord_hit` are the primitives' positions in world.objects order (DESIGN.md "order keys").
RT_HD bool precedes_hit(float t, float t_hit, uint32_t ord, uint32_t ord_hit) {
    return (fabsf(t - t_hit) <= tie_window(t_hit)) ? (ord < ord_hit) : (t < t_hit);
}

// A ray in world space plus what it knows about the primitive it starts on.
struct Ray {
    V3 o, d;
    uint32_t self;  // primitive the origin sits on, or kNoPrim
    V3 q;           // snapped object-space point on `self`
    V3 dq;          // object-space offset: origin_obj = q + dq  (= +-1e-5 * inv_linear * world normal)
};

// The intersections of one primitive are handed to an `emit(slot, t, u, v)` functor as they are
// found: slot is a compile-time constant at every call site (0/1 = the two roots or slab
// distances, 2/3 = cylinder caps), so a visitor that stores them indexes registers, and a visitor
// that only wants the nearest hit or "any hit before the light" keeps no list at all.
struct Hits {  // slot store used by the container pass (intersection.rs:71-103); absent = NaN
    float t[4];
};
#define RT_NOHIT __builtin_nanf("")

// ---------------------------------------------------------------------------------------------
// Foot of the perpendicular from the unit sphere's centre onto the line o + t d, in FP32 without
// the cancellation of b^2 - 4ac: tc = -(d.o)/a is refined by one Gram-Schmidt step (the residual
// l.d is tiny and therefore known to full relative accuracy), after which l is perpendicular to d
// to ~1e-7 and the roots are tc -+ half with half = sqrt((1 - |l|^2)/a).  The hit POINT is then
// l -+ half*d (prepare_hit), whose error is ~1e-7 object units however far away the ray started;
// o + t*d would carry the full rounding error of t (~1e-7 * distance) into the normal.
RT_HD bool sphere_foot(V3 o, V3 d, float a, float& tc, V3& l, float& half) {
    float inv_a = rcp_f(a);
    float tc0 = -dot(d, o) * inv_a;
    V3 l0 = madd(d, tc0, o);
    float e = dot(l0, d) * inv_a;
    tc = tc0 - e;
    l = madd(d, -e, l0);
    float m = 1.0f - dot(l, l);
    half = sqrt_f(fmaxf(m, 0.0f) * inv_a);
    return !(m < 0.0f);
}

// Roots of a t^2 + 2 h t + c = 0 for a ray that STARTS on the quadric (c ~ 1e-5, no cancellation
// left in it): Vieta for the near root.  false = discriminant < 0.
RT_HD bool on_surface_roots(float a, float h, float c, float& t0, float& t1) {
    float disc = fmaf(h, h, -(a * c));
    if (disc < 0.0f) return false;
    float sq = sqrt_f(disc);
    float qq = -(h + copysignf(sq, h));
    float ta = div_f(qq, a);
    float tb = (qq != 0.0f) ? div_f(c, qq) : ta;
    t0 = fminf(ta, tb), t1 = fmaxf(ta, tb);
    return true;
}

// sphere.rs:38-72.  a = d.d, h = d.o (= b/2), c = o.o - 1.  Always two roots when it hits.
// Candidates the FP32 tests cannot settle (the exact chain, rt_exact.h, works them out in f64): a hit
// whose own test was a near thing is emitted with kUnsure added to its slot; a MISS that was a near
// thing is reported to `soft(t)` with the parameter where the hit would be.  Thresholds are in object
// units and generous (FP32 error of an object-space ray position is ~1e-7 |translation| / scale):
// the band of rays they catch is ~1e-4 of a shape's projected area.
constexpr int kUnsure = 8;
constexpr float kGraze = 1e-4f;
struct NoSoft {
    RT_HD void operator()(float) const {}
};
template <class Emit, class Soft>
RT_HD void isect_sphere(V3 o, V3 d, bool self, V3 q, V3 dq, Emit&& emit, Soft&& soft) {
    float a = dot(d, d);
    float t0, t1;
    int unsure = 0;
    if (self) {
        // |q| == 1 exactly by construction, so c = |q + dq|^2 - 1 = 2 q.dq + dq.dq without cancellation
        const float h = dot(d, q + dq), c = fmaf(2.0f, dot(q, dq), dot(dq, dq));
        // A ray that leaves the sphere almost tangentially (a shadow ray at the terminator) passes the sphere it stands
        // on at a height of the 1e-5 offset: the discriminant h^2 - a c = (|h| - s)(|h| + s), s = sqrt(a c), changes
        // sign where |h| = s ~ 4.5e-3 |d|, and q is the true point to ~3e-7 only: within 2e-6 |d| of that it is unsure.
        // (|disc| < 2e-6 |d| (|h| + s) with |h| ~ s there, squared: no square root on this path)
        const float disc = fmaf(h, h, -(a * c));
        const bool graze = disc * disc < 64e-12f * a * (h * h);
        if (!on_surface_roots(a, h, c, t0, t1)) {
            if (graze) soft(-h * rcp_f(a));
            return;
        }
        unsure = graze ? kUnsure : 0;
    } else {
        float tc, half;
        V3 l;
        const bool ok = sphere_foot(o, d, a, tc, l, half);  // discriminant < 0  (sphere.rs:51)
        const float m = 1.0f - dot(l, l);
        if (!ok) {
            if (m > -kGraze) soft(tc);
            return;
        }
        unsure = m < kGraze ? kUnsure : 0;
        t0 = tc - half, t1 = tc + half;
    }
    emit(0 | unsure, t0, 0.0f, 0.0f);
    emit(1 | unsure, t1, 0.0f, 0.0f);
}

// cube.rs:39-122 (the same slab code serves Group boxes: bounding_box.rs:81-84).
// num_lo / num_hi are (min - origin) and (max - origin) per axis.
RT_HD void slab_axis(float num_lo, float num_hi, float d, float& tmin, float& tmax) {
    // cube.rs:83: `ge(|d|, EPSILON)` is approximate and therefore true for every finite d, so the
    // IEEE division (x/0 = +-inf, 0/0 = NaN) is what the reference always executes.
    float inv_d = rcp_f(d);
    float a = num_lo * inv_d, b = num_hi * inv_d;
    bool sw = a > b;
    tmin = sw ? b : a;
    tmax = sw ? a : b;
}
RT_HD void isect_box(V3 num_lo, V3 num_hi, V3 d, float& tmin, float& tmax) {
    float x0, x1, y0, y1, z0, z1;
    slab_axis(num_lo.x, num_hi.x, d.x, x0, x1);
    slab_axis(num_lo.y, num_hi.y, d.y, y0, y1);
    slab_axis(num_lo.z, num_hi.z, d.z, z0, z1);
    tmin = max_by3(x0, y0, z0);
    tmax = min_by3(x1, y1, z1);
}
template <class Emit, class Soft>
RT_HD void isect_cube(V3 o, V3 d, bool self, V3 q, V3 dq, Emit&& emit, Soft&& soft) {
    V3 lo, hi;
    if (self) {  // the snapped coordinate of q is exactly +-1, so (bound - q) is exactly 0 or +-2
        lo = (v3(-1.0f, -1.0f, -1.0f) - q) - dq;
        hi = (v3(1.0f, 1.0f, 1.0f) - q) - dq;
    } else {
        lo = v3(-1.0f, -1.0f, -1.0f) - o;
        hi = v3(1.0f, 1.0f, 1.0f) - o;
    }
    float tmin, tmax;
    isect_box(lo, hi, d, tmin, tmax);
    const float room = kGraze * fmaxf(1.0f, fabsf(tmax));  // the ray clips an edge of the cube
    if (tmin > tmax) {  // cube.rs:104
        if (tmin - tmax <= room) soft(tmax);
        return;
    }
    const int unsure = (tmax - tmin <= room) ? kUnsure : 0;
    emit(0 | unsure, tmin, 0.0f, 0.0f);
    emit(1 | unsure, tmax, 0.0f, 0.0f);
}

// cylinder.rs:126-225
// Near things of a cylinder (as for the sphere above: kUnsure on a hit, `soft` for a miss): a ray that grazes the wall,
// a wall crossing at the height of a rim (cylinder.rs:171-181, strict ymin < y < ymax), a cap crossing at the rim's
// radius (cylinder.rs:200-215).
RT_HD float cyl_cap_rr(float ox, float oz, float dx, float dz, float t) {
    float x = fmaf(t, dx, ox), z = fmaf(t, dz, oz);
    return fmaf(x, x, z * z);  // le(x^2 + z^2, 1) follows
}
template <class Emit, class Soft>
RT_HD void isect_cylinder(V3 o, V3 d, bool self, V3 q, V3 dq, float ymin, float ymax, bool closed, Emit&& emit, Soft&& soft) {
    V3 os = self ? q + dq : o;
    float a = fmaf(d.x, d.x, d.z * d.z);
    if (!approx(a, 0.0f)) {
        float t0, t1;
        int unsure = 0;
        if (self && q.y != ymin && q.y != ymax && fabsf(fmaf(q.x, q.x, q.z * q.z) - 1.0f) < 1e-5f) {
            // origin on the wall (prepare_hit normalised q there; a point it snapped onto a CAP has q.y == ymin / ymax
            // and may lie anywhere inside the rim, 1e-4 from it included): q_x^2 + q_z^2 == 1, c has no cancellation
            float h = fmaf(os.x, d.x, os.z * d.z);  // b/2
            float c = fmaf(2.0f, fmaf(q.x, dq.x, q.z * dq.z), fmaf(dq.x, dq.x, dq.z * dq.z));
            const float disc = fmaf(h, h, -(a * c));
            const bool graze = disc * disc < 64e-12f * a * (h * h);  // (isect_sphere: the ray leaves its own wall tangentially)
            if (!on_surface_roots(a, h, c, t0, t1)) {  // cylinder.rs:140-142 returns WITHOUT testing the caps
                if (graze) soft(-h * rcp_f(a));
                return;
            }
            unsure = graze ? kUnsure : 0;
        } else {
            // the sphere_foot construction in the xz plane
            float tc, half;
            V3 l;
            const bool ok = sphere_foot(v3(os.x, 0.0f, os.z), v3(d.x, 0.0f, d.z), a, tc, l, half);
            const float m = 1.0f - dot(l, l);
            if (!ok) {
                if (m > -kGraze) soft(tc);
                return;
            }
            unsure = m < kGraze ? kUnsure : 0;
            t0 = tc - half, t1 = tc + half;
        }
        auto wall = [&](int slot, float t) {
            const float y = fmaf(t, d.y, os.y);
            const float room = kGraze * fmaxf(1.0f, fabsf(y));
            const bool rim = fabsf(y - ymin) < room || fabsf(y - ymax) < room;  // (an open end at +-inf is never near)
            if (ymin < y && y < ymax) emit(slot | unsure | (rim ? kUnsure : 0), t, 0.0f, 0.0f);
            else if (rim) soft(t);
        };
        wall(0, t0);
        wall(1, t1);
    }
    if (!closed || approx(d.y, 0.0f)) return;  // cylinder.rs:192
    float nlo, nhi;
    if (self) {
        nlo = (ymin - q.y) - dq.y;
        nhi = (ymax - q.y) - dq.y;
    } else {
        nlo = ymin - o.y;
        nhi = ymax - o.y;
    }
    float inv_dy = rcp_f(d.y);
    auto cap = [&](int slot, float t) {
        const float rr = cyl_cap_rr(os.x, os.z, d.x, d.z, t);
        const bool rim = fabsf(rr - 1.0f) < kGraze;
        if (rr <= 1.0f + kEps) emit(slot | (rim ? kUnsure : 0), t, 0.0f, 0.0f);
        else if (rim) soft(t);
    };
    cap(2, nlo * inv_dy);
    cap(3, nhi * inv_dy);
}

// triangle.rs:104-133 (Moller-Trumbore); e = {v0, e0, e1}
template <class Emit, class Soft>
RT_HD void isect_triangle(V3 o, V3 d, V3 v0, V3 e0, V3 e1, Emit&& emit, Soft&& soft) {
    V3 dir_cross_e1 = cross(d, e1);
    float det = dot(e0, dir_cross_e1);
    if (fabsf(det) <= kEps) return;
    float f = rcp_f(det);
    V3 p0 = o - v0;
    float u = f * dot(p0, dir_cross_e1);
    if (!(u >= -kGraze && u <= 1.0f + kGraze)) return;
    V3 oxe0 = cross(p0, e0);
    float v = f * dot(d, oxe0);
    if (v < -kGraze || (u + v) > 1.0f + kGraze) return;
    const float t = f * dot(e1, oxe0);
    const bool hit = (u >= 0.0f && u <= 1.0f) && !(v < 0.0f || (u + v) > 1.0f);  // triangle.rs:117-125
    if (!hit) {  // within the margin of an edge: the neighbour across the edge may or may not take the ray
        soft(t);
        return;
    }
    const bool edge = u < kGraze || v < kGraze || (u + v) > 1.0f - kGraze;
    emit(edge ? kUnsure : 0, t, u, v);
}

// Dispatch: world ray -> object space (shape.rs:79-81, ray.rs:17-22) -> primitive test.
// EXOTIC = false compiles the cylinder / triangle code out: a kernel instantiated that way (for
// scenes without such primitives — rt_scene_create knows) is 6-20 % faster on the other shapes
// because the register allocation and the instruction cache are no longer sized by the rare paths.
template <bool EXOTIC = true, class Emit, class Soft>
RT_HD void intersect_prim(const DPrim& p, const DExtra* extras, const Ray& ray, bool self, Emit&& emit, Soft&& soft) {
    switch (p.type) {
        case PRIM_PLANE: {  // plane.rs:39-51 — only the y row of the inverse is needed
            float oy = self ? ray.dq.y : fmaf(p.inv[4], ray.o.x, fmaf(p.inv[5], ray.o.y, fmaf(p.inv[6], ray.o.z, p.inv[7])));
            float dy = fmaf(p.inv[4], ray.d.x, fmaf(p.inv[5], ray.d.y, p.inv[6] * ray.d.z));
            if (!(fabsf(dy) <= kEps)) emit(0, div_f(-oy, dy), 0.0f, 0.0f);
            break;
        }
        case PRIM_SPHERE: {
            V3 o = xform_point(p.inv, ray.o), d = xform_vector(p.inv, ray.d);
            isect_sphere(o, d, self, ray.q, ray.dq, emit, soft);
            break;
        }
        case PRIM_CUBE: {
            V3 o = xform_point(p.inv, ray.o), d = xform_vector(p.inv, ray.d);
            isect_cube(o, d, self, ray.q, ray.dq, emit, soft);
            break;
        }
        case PRIM_CYLINDER: {
            if (!EXOTIC) break;
            V3 o = xform_point(p.inv, ray.o), d = xform_vector(p.inv, ray.d);
            const float* e = extras[p.extra].f;
            isect_cylinder(o, d, self, ray.q, ray.dq, e[0], e[1], e[2] != 0.0f, emit, soft);
            break;
        }
        default: {  // PRIM_TRIANGLE, PRIM_SMOOTH
            if (!EXOTIC) break;
            V3 o = xform_point(p.inv, ray.o), d = xform_vector(p.inv, ray.d);
            if (self) o = ray.q + ray.dq;
            const float* e = extras[p.extra].f;
            isect_triangle(o, d, v3(e[0], e[1], e[2]), v3(e[3], e[4], e[5]), v3(e[6], e[7], e[8]), emit, soft);
            break;
        }
    }
}
// Without the near-miss reports: `emit` still receives kUnsure in the slot and must mask it (slot & 3).
template <bool EXOTIC = true, class Emit>
RT_HD void intersect_prim(const DPrim& p, const DExtra* extras, const Ray& ray, bool self, Emit&& emit) {
    intersect_prim<EXOTIC>(p, extras, ray, self, emit, NoSoft{});
}

// Shadow rays only ask "is there an intersection with 0 < t < dist" (world.rs:73-92: the nearest
// positive hit is closer than the light iff ANY positive hit is).  For a ray that does NOT start on
// the primitive the three common shapes answer without a branch; the others go through the emitter.
RT_HD bool in_range(float t, float dist) { return (t > 0.0f) & (t < dist); }
RT_HD bool in_win(float t, float lo, float hi) { return (t > lo) & (t < hi); }
template <bool EXOTIC = true>
RT_HD bool occludes_prim(const DPrim& p, const DExtra* extras, V3 ro, V3 rd, float dist) {
    switch (p.type) {
        case PRIM_PLANE: {
            float oy = fmaf(p.inv[4], ro.x, fmaf(p.inv[5], ro.y, fmaf(p.inv[6], ro.z, p.inv[7])));
            float dy = fmaf(p.inv[4], rd.x, fmaf(p.inv[5], rd.y, p.inv[6] * rd.z));
            return !(fabsf(dy) <= kEps) & in_range(-oy * rcp_f(dy), dist);
        }
        case PRIM_SPHERE: {
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            float tc, half;
            V3 l;
            bool ok = sphere_foot(o, d, dot(d, d), tc, l, half);
            return ok & (in_range(tc - half, dist) | in_range(tc + half, dist));
        }
        case PRIM_CUBE: {
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            float tmin, tmax;
            isect_box(v3(-1.0f, -1.0f, -1.0f) - o, v3(1.0f, 1.0f, 1.0f) - o, d, tmin, tmax);
            return !(tmin > tmax) & (in_range(tmin, dist) | in_range(tmax, dist));
        }
        default: {
            if (!EXOTIC) return false;
            bool occ = false;
            Ray r;
            r.o = ro, r.d = rd, r.self = kNoPrim, r.q = v3(0.0f, 0.0f, 0.0f), r.dq = v3(0.0f, 0.0f, 0.0f);
            intersect_prim<EXOTIC>(p, extras, r, false, [&](int, float t, float, float) { occ |= in_range(t, dist); });
            return occ;
        }
    }
}

// The hot form of the question below, for the FIRST pass of the kernels with the f64 chain: the plain answer, plus
// `near` = the ray grazes this sphere, the edge of this cube, the wall or a rim of this cylinder, an edge of this triangle
// (the near things a point in the open meets; a point within 1e-4 of another surface is known from the nearest-hit
// search: its second candidate).  One compare a sphere, four instructions a cube; cylinders and triangles go through
// their emitters (kUnsure / soft), in the kernels compiled for them only.
#ifndef RT_TRI_NEAR
#define RT_TRI_NEAR 1  // triangle edges are near things for the shadow rays too: 7e-4 of obj_mesh's shadow rays leave a record,
                       // +6 % of its frame since the second count takes 16 records a pass (3x before: profiles/r03_group_boxes.txt)
#endif
constexpr float kNearSphere = 2e-5f;
template <bool EXOTIC = true>
RT_HD bool occludes_prim_near(const DPrim& p, const DExtra* extras, V3 ro, V3 rd, float dist, bool& near) {
    switch (p.type) {
        case PRIM_SPHERE: {
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            float tc, half;
            V3 l;
            const bool ok = sphere_foot(o, d, dot(d, d), tc, l, half);
            // (FP32 knows 1 - |l|^2 to about 2.5e-7 (1 + |o|): the object-space origin's rounding, twice; the band is 6x that)
            near |= fabsf(1.0f - dot(l, l)) < fmaxf(kNearSphere, 1.5e-6f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)));
            return ok & (in_range(tc - half, dist) | in_range(tc + half, dist));
        }
        case PRIM_CUBE: {
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            float tmin, tmax;
            isect_box(v3(-1.0f, -1.0f, -1.0f) - o, v3(1.0f, 1.0f, 1.0f) - o, d, tmin, tmax);
            near |= fabsf(tmax - tmin) <= kGraze * fmaxf(1.0f, fabsf(tmax));
            return !(tmin > tmax) & (in_range(tmin, dist) | in_range(tmax, dist));
        }
        case PRIM_CYLINDER: {  // a grazed wall, a crossing at a rim: isect_cylinder's near things, if they lie before the light
            if (!EXOTIC) return false;
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            const float* e = extras[p.extra].f;
            bool occ = false;
            const float w0 = amb_window(0.0f), w1 = amb_window(dist);
            isect_cylinder(
                o, d, false, v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 0.0f), e[0], e[1], e[2] != 0.0f,
                [&](int slot, float t, float, float) { occ |= in_range(t, dist), near |= (slot & kUnsure) != 0 && in_win(t, -w0, dist + w1); },
                [&](float t) { near |= in_win(t, -w0, dist + w1); });
            return occ;
        }
        case PRIM_PLANE: return occludes_prim<EXOTIC>(p, extras, ro, rd, dist);  // planes have no near things of their own
#if RT_TRI_NEAR
        default: {  // triangles: a crossing within kGraze (barycentric) of an edge, inside or outside
            if (!EXOTIC) return false;
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            const float* e = extras[p.extra].f;
            bool occ = false;
            const float w0 = amb_window(0.0f), w1 = amb_window(dist);
            isect_triangle(
                o, d, v3(e[0], e[1], e[2]), v3(e[3], e[4], e[5]), v3(e[6], e[7], e[8]),
                [&](int slot, float t, float, float) { occ |= in_range(t, dist), near |= (slot & kUnsure) != 0 && in_win(t, -w0, dist + w1); },
                [&](float t) { near |= in_win(t, -w0, dist + w1); });
            return occ;
        }
#else
        default: return occludes_prim<EXOTIC>(p, extras, ro, rd, dist);
#endif
    }
}
// ... and for the primitive the ray starts on: its own sphere grazed at the terminator (kUnsure roots, near misses)
template <bool EXOTIC = true>
RT_HD bool occludes_self_near(const DPrim& p, const DExtra* extras, const Ray& ray, float dist, bool& near) {
    bool hit = false;
    intersect_prim<EXOTIC>(
        p, extras, ray, true,
        [&](int slot, float t, float, float) {
            const bool in = in_range(t, dist);
            hit |= in, near |= in && (slot & kUnsure) != 0;
        },
        [&](float) { near = true; });
    return hit;
}

// A path ray that leaves a point where the reference's own rounding noise decides whether it meets the primitive it stands
// on AGAIN, at t ~ 1e-15 (rt_exact.h XHit::qedge: a cube's point on an edge line or with the neighbouring face's normal, a
// cylinder's wall point with an end's normal), says so to its own first search through the lowest mantissa bit of
// dq.x — 1e-12 of an offset of 1e-5, and no register, stack word or record load for it (deriving it again from q and dq
// took sheared cubes for noise: shapes_zoo +5 %).  The f64 chain then asks the primitive itself (rt_kernel.h "self_noise").
RT_HD float with_noise_flag(float dq_x, bool noise) {
    uint32_t b;
    __builtin_memcpy(&b, &dq_x, 4);
    b = (b & ~1u) | (noise ? 1u : 0u);
    __builtin_memcpy(&dq_x, &b, 4);
    return dq_x;
}
RT_HD bool noise_flag(float dq_x) {
    uint32_t b;
    __builtin_memcpy(&b, &dq_x, 4);
    return (b & 1u) != 0u;
}

// What FP32 cannot settle about a shadow ray (the exact chain decides it in f64: rt_exact.h xoccludes, rt_kernel.h
// "unsure shadow rays").  The any-hit question has three kinds of near things: a ray GRAZING a sphere or the edge of
// a cube (kGraze, object units), a crossing within w0 of the ray's ORIGIN (the point stands within 1e-5 .. 1e-4 of
// another surface: two cubes that touch, a body standing on the floor), a crossing within w1 of the LIGHT.  MAYBE =
// "occludes" with every threshold moved outwards by its margin, SURE = with every threshold moved inwards: a primitive
// that is MAYBE but not SURE is left to f64.  (The primitive a ray starts on is not asked: its self form is exact by
// construction, rt_math.h "self".)
struct ShadowWin {
    float lo_m, hi_m;  // maybe: lo_m < t < hi_m
    float lo_s, hi_s;  // sure:  lo_s < t < hi_s
};
RT_HD ShadowWin shadow_windows(float dist) {
    const float w0 = amb_window(0.0f), w1 = amb_window(dist);
    return ShadowWin{-w0, dist + w1, w0, dist - w1};
}
struct ShadowTest {
    bool maybe, sure;
};
// The primitive the shadow ray STARTS on: its self form (rt_math.h "self") knows the crossing it stands on exactly, so
// the plain range (0, dist) applies; what it cannot settle is a ray that grazes its own sphere (kUnsure roots, or a
// near miss reported to `soft`).
template <bool EXOTIC = true>
RT_HD ShadowTest occludes_self_soft(const DPrim& p, const DExtra* extras, const Ray& ray, float dist, const ShadowWin& w) {
    bool any = false, sure = false, near = false;
    intersect_prim<EXOTIC>(
        p, extras, ray, true,
        [&](int slot, float t, float, float) {
            const bool in = in_range(t, dist);
            any |= in, sure |= in && !(slot & kUnsure);
        },
        [&](float t) { near |= in_win(t, w.lo_m, w.hi_m); });
    return ShadowTest{any || near, sure};
}
template <bool EXOTIC = true>
RT_HD ShadowTest occludes_prim_soft(const DPrim& p, const DExtra* extras, V3 ro, V3 rd, const ShadowWin& w) {
    switch (p.type) {
        case PRIM_PLANE: {
            float oy = fmaf(p.inv[4], ro.x, fmaf(p.inv[5], ro.y, fmaf(p.inv[6], ro.z, p.inv[7])));
            float dy = fmaf(p.inv[4], rd.x, fmaf(p.inv[5], rd.y, p.inv[6] * rd.z));
            const float t = -oy * rcp_f(dy), ady = fabsf(dy);
            const bool maybe = !(ady <= 0.9f * kEps) & in_win(t, w.lo_m, w.hi_m), sure = !(ady <= 1.1f * kEps) & in_win(t, w.lo_s, w.hi_s);
            return ShadowTest{maybe, sure};
        }
        case PRIM_SPHERE: {
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            float tc, half;
            V3 l;
            sphere_foot(o, d, dot(d, d), tc, l, half);  // (a negative discriminant leaves half = 0: both roots at the foot)
            const float m = 1.0f - dot(l, l);
            const float t0 = tc - half, t1 = tc + half;
            const bool maybe = (m > -kGraze) & (in_win(t0, w.lo_m, w.hi_m) | in_win(t1, w.lo_m, w.hi_m));
            const bool sure = (m > kGraze) & (in_win(t0, w.lo_s, w.hi_s) | in_win(t1, w.lo_s, w.hi_s));
            return ShadowTest{maybe, sure};
        }
        case PRIM_CUBE: {
            V3 o = xform_point(p.inv, ro), d = xform_vector(p.inv, rd);
            float tmin, tmax;
            isect_box(v3(-1.0f, -1.0f, -1.0f) - o, v3(1.0f, 1.0f, 1.0f) - o, d, tmin, tmax);
            const float room = kGraze * fmaxf(1.0f, fabsf(tmax));
            const bool maybe = !(tmin > tmax + room) & (in_win(tmin, w.lo_m, w.hi_m) | in_win(tmax, w.lo_m, w.hi_m));
            const bool sure = (tmax - tmin > room) & (in_win(tmin, w.lo_s, w.hi_s) | in_win(tmax, w.lo_s, w.hi_s));
            return ShadowTest{maybe, sure};
        }
        default: {
            if (!EXOTIC) return ShadowTest{false, false};
            ShadowTest st{false, false};
            Ray r;
            r.o = ro, r.d = rd, r.self = kNoPrim, r.q = v3(0.0f, 0.0f, 0.0f), r.dq = v3(0.0f, 0.0f, 0.0f);
            intersect_prim<EXOTIC>(
                p, extras, r, false,
                [&](int slot, float t, float, float) {
                    st.maybe |= in_win(t, w.lo_m, w.hi_m);
                    // (a cylinder's wall and caps report no near things of their own: its hits are never SURE)
                    st.sure |= p.type != PRIM_CYLINDER && !(slot & kUnsure) && in_win(t, w.lo_s, w.hi_s);
                },
                [&](float t) { st.maybe |= in_win(t, w.lo_m, w.hi_m); });
            return st;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Groups.  The reference enters a Group's children only if the ray's LINE hits the group's box
// (group.rs:107, bounding_box.rs:81-84: tmin < tmax, strict, approximate max_by/min_by) — so a
// primitive inside groups G1 > G2 > ... is visible to a ray iff every one of those tests passes.
// The device does not walk the caller's tree (its leaves may hold hundreds of children): it walks
// its own bounding-volume hierarchy over the primitives (rt_pack.h) and keeps the caller's boxes as
// GATES, checked only when a primitive's own test found something.  Same visible set, same image.
struct DGate {  // 32 B: one Group of the caller's tree
    float mn[3];
    uint32_t parent;  // enclosing Group's gate, or kNoGate
    float mx[3];
    uint32_t pad;
};
constexpr uint32_t kNoGate = 0xFFFFFFFFu;

RT_HD bool gates_pass(const DGate* gates, uint32_t g, V3 o, V3 d) {
    while (g != kNoGate) {
        const DGate G = gates[g];
        float tmin, tmax;
        isect_box(v3(G.mn[0], G.mn[1], G.mn[2]) - o, v3(G.mx[0], G.mx[1], G.mx[2]) - o, d, tmin, tmax);
        if (!(tmin < tmax)) return false;
        g = G.parent;
    }
    return true;
}

// The same for the kernels with the f64 chain, which also want to know whether some box on the way was a near thing.
// The boxes here are the caller's, rounded to FP32 and padded outwards (rt_pack.h widen), so what passes here is a
// superset of what passes in the reference; a ray whose way through a box is shorter than kGraze (the rule of the cube's
// edges, isect_cube) clips it so close to an edge that only f64 with the caller's own box (rt_exact.h xgates_pass) can say
// whether the reference lets it in.
RT_HD bool gates_pass_near(const DGate* gates, uint32_t g, V3 o, V3 d, bool& near) {
    bool n = false;  // (a ray that FP32 keeps out of some box stays out in f64 too: nothing near about it)
    while (g != kNoGate) {
        const DGate G = gates[g];
        float tmin, tmax;
        isect_box(v3(G.mn[0], G.mn[1], G.mn[2]) - o, v3(G.mx[0], G.mx[1], G.mx[2]) - o, d, tmin, tmax);
        if (!(tmin < tmax)) return false;
        n |= tmax - tmin <= kGraze * fmaxf(1.0f, fabsf(tmax));
        g = G.parent;
    }
    near |= n;
    return true;
}

// A node of the device's own hierarchy: box min in inv[0..2], max in inv[3..5] (rounded outwards),
// skip link in `material`.  Plain slab test on the parameter interval [t_lo, t_hi]; a NaN from
// 0 * inf (origin on a slab plane, direction parallel to it) drops out of min/max, which keeps the
// test conservative.  inv_d = 1 / direction, per component.
template <class Rec>
RT_HD bool bvh_node_hit(const Rec& n, V3 o, V3 inv_d, float t_lo, float t_hi) {
    float x0 = (n.inv[0] - o.x) * inv_d.x, x1 = (n.inv[3] - o.x) * inv_d.x;
    float y0 = (n.inv[1] - o.y) * inv_d.y, y1 = (n.inv[4] - o.y) * inv_d.y;
    float z0 = (n.inv[2] - o.z) * inv_d.z, z1 = (n.inv[5] - o.z) * inv_d.z;
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_lo));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_hi));
    return tn <= tf;
}
RT_HD V3 inv_dir(V3 d) { return v3(rcp_f(d.x), rcp_f(d.y), rcp_f(d.z)); }
// Node test of the container pass (intersection.rs:71-103).  Only a primitive with an ODD number of
// crossings before the hit is in the container list at the hit; a sphere or a cube is convex and
// crossed twice or never, so that means the hit point lies between its two crossings — inside it, or
// within the tie window of its surface — and therefore inside every hierarchy box around it, padded
// by that window.  (Open cylinders and triangles can be crossed once; scenes holding them keep the
// ray-interval test.)  m = the padding, in world units.
template <class Rec>
RT_HD bool node_holds_point(const Rec& n, V3 p, float m) {
    return (p.x >= n.inv[0] - m) & (p.x <= n.inv[3] + m) & (p.y >= n.inv[1] - m) & (p.y <= n.inv[4] + m) & (p.z >= n.inv[2] - m) &
           (p.z <= n.inv[5] + m);
}
RT_HD float container_margin(float t_hit, V3 p) {
    return 8.0f * tie_window(t_hit) + 1e-4f + 1e-5f * (fabsf(p.x) + fabsf(p.y) + fabsf(p.z));
}

// ---------------------------------------------------------------------------------------------
// Shafts.  The shadow rays of one shading point towards one area light (light.rs:196-208) all run
// from the point to somewhere on the light's parallelogram, i.e. inside the union of the balls
// B(p + s D, s r), s in [0, 1], with D = centre - p and (centre, r) a ball around the light.  A
// record that provably stays outside that set cannot occlude any of those rays, whatever the
// jitter: the answer "not occluded" is known without casting them.  These tests only ever say
// "cannot block" with room to spare (>= 1e-3 relative, far above FP32 rounding of the exact
// tests); everything else is left to the exact intersection code, so the image does not change.
struct DBound {  // 32 B per record: world-space bounding ball (r = +inf: unbounded) and, for spheres
    float c[3];  //  and cubes, the ball inscribed around the same centre (r_in = 0: none)
    float r;
    float r_in;
    float margin;  // object-space margin of the pyramid tests (spheres and cubes)
    float pad[2];
};
struct Shaft {
    V3 p, lc, D;  // apex (the over point), light ball centre, lc - p
    float lr, dlen, inv_d2, cosk;  // ball radius, |D|, 1/|D|^2, sqrt(1 - (lr/|D|)^2)
    bool wide;    // apex inside or close to the light's ball: the cone bound says nothing
};
RT_HD void light_ball(const DLight& l, V3& c, float& r) {
    V3 a = v3(l.uvec[0], l.uvec[1], l.uvec[2]) * (0.5f * static_cast<float>(l.usteps));
    V3 b = v3(l.vvec[0], l.vvec[1], l.vvec[2]) * (0.5f * static_cast<float>(l.vsteps));
    c = v3(l.pos[0], l.pos[1], l.pos[2]) + a + b;
    r = sqrt_f(fmaxf(dot(a + b, a + b), dot(a - b, a - b))) * 1.0001f;
}
RT_HD Shaft make_shaft(V3 p, V3 lc, float lr) {
    Shaft s;
    s.p = p, s.lc = lc, s.D = lc - p, s.lr = lr;
    float d2 = dot(s.D, s.D);
    float k2 = lr * lr * rcp_f(d2);
    s.wide = !(k2 < 0.9f);
    s.inv_d2 = rcp_f(d2);
    s.dlen = d2 * rsq_f(d2);
    s.cosk = sqrt_f(fmaxf(1.0f - k2, 0.0f));
    return s;
}
// min over s of |c - p - s D| - s lr  (a lower bound of the distance between the ball's centre and
// the shaft) against the ball's radius.  NaN / inf radius -> true.
RT_HD bool shaft_meets_ball(const Shaft& s, V3 c, float r) {
    V3 e = c - s.p;
    float t = dot(e, s.D) * s.inv_d2;
    V3 w = madd(s.D, -t, e);
    float perp = sqrt_f(dot(w, w));
    float gap = perp * s.cosk - t * s.lr - 1e-5f * fmaf(fabsf(t), s.dlen, perp);
    return s.wide | !(gap > fmaf(r, 1.001f, 1e-4f));
}
// Plane (plane.rs:39-51): the exact test finds t in (0, dist) only if the ray's end lies on the
// other side of y = 0 than its origin.  oy is the origin's object-space height exactly as the
// exact test computes it; the heights of the light's points are bounded with a margin.
RT_HD bool shaft_meets_plane(const Shaft& s, const float* inv, float oy) {
    float yc = fmaf(inv[4], s.lc.x, fmaf(inv[5], s.lc.y, fmaf(inv[6], s.lc.z, inv[7])));
    float rn = sqrt_f(fmaf(inv[4], inv[4], fmaf(inv[5], inv[5], inv[6] * inv[6])));
    float l1 = fabsf(s.lc.x) + fabsf(s.lc.y) + fabsf(s.lc.z) + fabsf(s.p.x) + fabsf(s.p.y) + fabsf(s.p.z) + s.lr;
    float reach = fmaf(rn, s.lr, fmaf(1e-5f, fmaf(rn, l1, fabsf(yc)), 1e-6f));
    bool above = (oy > 0.0f) & (yc - reach > 0.0f);
    bool below = (oy < 0.0f) & (yc + reach < 0.0f);
    return !(above | below);
}
// Cube (cube.rs:39-122), after the ball test: the cube is the intersection of three slabs of its
// object space.  If the apex and the whole light lie beyond the SAME face plane, every segment
// between them does too (a half-space is convex) and misses the cube.  Same margins as the plane rule.
RT_HD bool shaft_meets_cube(const Shaft& s, const float* inv) {
    float l1 = fabsf(s.lc.x) + fabsf(s.lc.y) + fabsf(s.lc.z) + fabsf(s.p.x) + fabsf(s.p.y) + fabsf(s.p.z) + s.lr;
    bool outside = false;
    for (int k = 0; k < 3; ++k) {
        const float* r = inv + 4 * k;
        float pk = fmaf(r[0], s.p.x, fmaf(r[1], s.p.y, fmaf(r[2], s.p.z, r[3])));
        float lk = fmaf(r[0], s.lc.x, fmaf(r[1], s.lc.y, fmaf(r[2], s.lc.z, r[3])));
        float rn = sqrt_f(fmaf(r[0], r[0], fmaf(r[1], r[1], r[2] * r[2])));
        float slack = fmaf(2e-5f, rn, 1e-4f);  // the slab distances are compared approximately (cube.rs:63-73)
        float reach = fmaf(rn, s.lr, fmaf(1e-5f, fmaf(rn, l1, fabsf(lk)), slack));
        float face = 1.0f + slack;
        outside |= ((pk > face) & (lk - reach > 1.0f)) | ((pk < -face) & (lk + reach < -1.0f));
    }
    return !outside;
}
// The opposite question: does a primitive occlude EVERY ray of the shaft?  Then the point is in its
// umbra and the answer "all cells occluded" is known without casting.  Sufficient: the apex lies
// outside a ball inscribed in the primitive, the whole primitive (bounding radius r_out) is nearer
// than any point of the light, and the cone of directions towards the light's ball lies inside the
// cone of directions towards the inscribed ball: theta + beta < alpha with theta = angle(e, D),
// sin beta = lr / |D|, sin alpha = r_in / |e|.  Every such ray enters the inscribed ball ahead of the
// apex, so it crosses the primitive's surface at some 0 < t < dist, with room to spare.
RT_HD bool shaft_blocked_by_ball(const Shaft& s, V3 c, float r_in, float r_out) {
    V3 e = c - s.p;
    float de2 = dot(e, e);
    float inv_de = rsq_f(de2);
    float de = de2 * inv_de;
    float sa = r_in * inv_de;
    float inv_dl = s.dlen * s.inv_d2;
    float ct = dot(e, s.D) * inv_de * inv_dl;
    float st = sqrt_f(fmaxf(1.0f - ct * ct, 0.0f));
    float cab = ct * s.cosk - st * (s.lr * inv_dl);
    float ca = sqrt_f(fmaxf(1.0f - sa * sa, 0.0f));
    bool ok = (r_in > 0.0f) & (sa < 0.999f) & !s.wide & (ct > 0.0f);
    ok &= (de + r_out) < (s.dlen - s.lr) * 0.999f;
    return ok & (cab > ca + 1e-3f);
}
// A plane between the apex and the whole light (plane.rs:39-51): every segment crosses it.
RT_HD bool shaft_blocked_by_plane(const Shaft& s, const float* inv, float oy) {
    float yc = fmaf(inv[4], s.lc.x, fmaf(inv[5], s.lc.y, fmaf(inv[6], s.lc.z, inv[7])));
    float rn = sqrt_f(fmaf(inv[4], inv[4], fmaf(inv[5], inv[5], inv[6] * inv[6])));
    float l1 = fabsf(s.lc.x) + fabsf(s.lc.y) + fabsf(s.lc.z) + fabsf(s.p.x) + fabsf(s.p.y) + fabsf(s.p.z) + s.lr;
    float reach = fmaf(rn, s.lr, fmaf(1e-5f, fmaf(rn, l1, fabsf(yc)), 1e-6f));
    float far = 1e-4f * rn * (s.dlen + s.lr);  // the crossing is well inside (0, dist) and the ray not parallel
    bool down = (oy > far) & (yc + reach < -far);
    bool up = (oy < -far) & (yc - reach > far);
    return down | up;
}
// The apex lies INSIDE a closed convex primitive (a sphere or cube whose surface was hit from within,
// so the over point sits 1e-5 below it) and the whole light outside the primitive's bounding ball:
// every ray of the shaft crosses the surface on its way out, before it reaches the light.
RT_HD bool shaft_leaves_solid(const Shaft& s, V3 c, float r_out) {
    V3 e = s.lc - c;
    float reach = s.lr + fmaf(r_out, 1.001f, 1e-4f);
    return dot(e, e) > reach * reach;
}
// The exact shape of a shaft is the pyramid from the apex to the light's parallelogram (the jittered
// points of light.rs:211-216 all lie in it).  Seen from a sphere's or cube's OBJECT space (the map is
// affine: pyramids stay pyramids, the shape becomes the unit sphere / the cube [-1, 1]^3) two sharper
// answers than the ball tests give are cheap:
//   pyramid_misses  : one of the four side planes through the apex separates the shape (grown by the
//                     margin m) from the pyramid — no ray of the shaft can touch it;
//   pyramid_blocked : the apex lies outside the grown shape and the four rays to the light's corners
//                     all enter the shape SHRUNK by m at 1e-3 < t < 1 - 1e-3 (t = 1 at the corner).
//                     Directions that hit a convex body form a convex cone, and the convex hull of the
//                     four entry points is met by every ray of the pyramid at a parameter between the
//                     smallest and largest of the four: every ray enters the shape before the light.
// m (DBound.margin) covers the 1e-5 slack of the approximate comparisons in cube.rs:63-73 seen from
// object space, and FP32 rounding of the maps: 1e-3 + 1e-4 |inverse|_F.
struct ShaftObj {
    V3 a, l0, U, V;  // apex; the light's corner and its two edges (corners l0, l0 + U, l0 + U + V, l0 + V)
};
RT_HD ShaftObj shaft_in_object_space(const float* inv, V3 apex, const DLight& l) {
    ShaftObj s;
    s.a = xform_point(inv, apex);
    s.l0 = xform_point(inv, v3(l.pos[0], l.pos[1], l.pos[2]));
    s.U = xform_vector(inv, v3(l.uvec[0], l.uvec[1], l.uvec[2]) * static_cast<float>(l.usteps));
    s.V = xform_vector(inv, v3(l.vvec[0], l.vvec[1], l.vvec[2]) * static_cast<float>(l.vsteps));
    return s;
}
// Written so that a far apex (|a| of 1e4 object units happens: a small shape, a point near the
// horizon) costs no accuracy: the side plane through the apex and the light's edge E that starts at
// corner p has the normal (p - a) x E and the offset n . a = (p x E) . a — no difference of nearly equal
// vectors anywhere; the pyramid's orientation is the side of the light's plane the apex is on.
RT_HD bool pyramid_misses(const ShaftObj& s, bool cube, float m) {
    const V3 nl = cross(s.U, s.V);
    const V3 e0 = s.l0 - s.a;
    const float tr = dot(e0, nl);  // = (l0 - a) . ((l1 - a) x (l2 - a))
    const float sg = tr >= 0.0f ? 1.0f : -1.0f;
    const float grow = 1.0f + m;
    bool out = false;
    V3 p = s.l0, E = s.U;
    for (int k = 0; k < 4; ++k) {
        const V3 e = p - s.a;
        const V3 n = cross(e, E);
        const float ka = sg * dot(cross(p, E), s.a);  // inner side: sg n . (x - a) >= 0; the shape's centre is ka below it
        const float nn = dot(n, n);
        bool sep;
        if (cube) sep = ka > (fabsf(n.x) + fabsf(n.y) + fabsf(n.z)) * grow;
        else sep = (ka > 0.0f) & (ka * ka > nn * (grow * grow));
        // an apex (nearly) on the edge's line leaves this plane to rounding: no answer from it
        out |= sep & (nn > 1e-6f * dot(e, e) * dot(E, E));
        p = p + E;
        E = k == 0 ? s.V : (k == 1 ? neg(s.U) : neg(s.V));
    }
    // likewise an apex (nearly) in the light's plane, for the orientation
    return out & (tr * tr > 1e-6f * dot(nl, nl) * dot(e0, e0));
}
// The four rays are followed from the light's corners towards the apex (x = l + s (a - l), s = 1 - t):
// the corner's coordinates are of the shape's own size, so the quadratic / the slabs lose nothing.
RT_HD bool pyramid_blocked(const ShaftObj& s, bool cube, float m) {
    const float grow = 1.0f + m, core = 1.0f - m;
    bool ok = core > 0.5f;
    if (cube) ok &= fmaxf(fabsf(s.a.x), fmaxf(fabsf(s.a.y), fabsf(s.a.z))) > grow;
    else ok &= dot(s.a, s.a) > grow * grow;
    V3 l = s.l0;
    for (int k = 0; k < 4; ++k) {
        const V3 w = s.a - l;
        float s_in, s_out;  // where the reversed ray enters and leaves the shrunk shape
        bool hit;
        if (cube) {
            const V3 iw = v3(rcp_f(w.x), rcp_f(w.y), rcp_f(w.z));
            const float x0 = (-core - l.x) * iw.x, x1 = (core - l.x) * iw.x;
            const float y0 = (-core - l.y) * iw.y, y1 = (core - l.y) * iw.y;
            const float z0 = (-core - l.z) * iw.z, z1 = (core - l.z) * iw.z;
            s_in = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
            s_out = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            hit = s_in < s_out;  // NaN (0 * inf) compares false
        } else {
            const float A = dot(w, w), B = dot(l, w), C = dot(l, l) - core * core;
            const float disc = fmaf(B, B, -(A * C));
            hit = (disc > 0.0f) & (A > 0.0f);
            const float root = sqrt_f(fmaxf(disc, 0.0f)), ia = rcp_f(A);
            s_in = (-B - root) * ia, s_out = (-B + root) * ia;
        }
        // the light's corner outside the shape (s_in > 0), the shape entered by the forward ray at
        // t = 1 - s_out in (1e-3, 1 - 1e-3)
        ok &= hit & (s_in > 1e-3f) & (s_out < 1.0f - 1e-3f);
        l = l + (k == 0 ? s.U : (k == 1 ? s.V : neg(s.U)));
    }
    return ok;
}
// Every ray of the shaft leaves the tangent plane at the apex on the side of the normal n: a ray
// that starts (1e-5 outside) on a CONVEX primitive with outward normal n cannot meet it again.
// (the light's lowest and highest points over that plane are corners of the parallelogram.)  Returns
// +1 for that; -1 when the WHOLE light lies below the tangent plane: then every cell fails the gate of
// material.rs:145, both Phong sums are exactly zero and no shadow ray of the pair can change the pixel;
// 0 when the plane cuts the light.
RT_HD int shaft_side_of_surface(const Shaft& s, V3 n, const DLight& l) {
    const float base = dot(v3(l.pos[0], l.pos[1], l.pos[2]) - s.p, n);
    const float un = dot(v3(l.uvec[0], l.uvec[1], l.uvec[2]), n) * static_cast<float>(l.usteps);
    const float vn = dot(v3(l.vvec[0], l.vvec[1], l.vvec[2]), n) * static_cast<float>(l.vsteps);
    const float room = 1e-3f * (s.dlen + s.lr);
    if (base + fminf(un, 0.0f) + fminf(vn, 0.0f) > room) return 1;
    if (base + fmaxf(un, 0.0f) + fmaxf(vn, 0.0f) < -room) return -1;
    return 0;
}

// intersection.rs:71-103 without the sorted list.  The reference walks the sorted intersections up
// to the hit, toggling each object in a container list; n1 is the material of the list's last entry
// before the hit toggles, n2 after.  Equivalent, per primitive: it is in the list at the hit iff an
// ODD number of its intersections precede the hit, and the list's last entry is the contained
// primitive whose latest preceding crossing has the largest t.
// EXACT: a crossing of ANOTHER primitive this close to the hit is more than FP32 can order against the reference's
// 1e-5 rule: a ray that grazes the face two stacked cubes share gets t = (1 - o_k) / d_k to 1e-7 / |d_k| only
// (1.2e-5 at t = 0.58 in random_stress:148).  The kernel then places that primitive's crossings in f64 (rt_exact.h).
// (... and 2e-4 at t = 0.3 in random_stress:41, where the ray meets the shared face at 1e-3 rad: the window grows with
// 1 / |n . d| of the hit, `cos_hit`.)
#ifndef RT_AMB_GRAZE
#define RT_AMB_GRAZE 1  // experiments: 0 = the fixed window of round 2
#endif
RT_HD float container_amb_window(float t, float cos_hit) {
    if (!RT_AMB_GRAZE) return 16.0f * tie_window(t);
    return 16.0f * tie_window(t) * fminf(fmaxf(0.02f * rcp_f(fmaxf(cos_hit, 1e-6f)), 1.0f), 256.0f);
}
RT_HD bool crossing_near_hit(const Hits& h, float t_hit, float w) {
    return fabsf(h.t[0] - t_hit) <= w || fabsf(h.t[1] - t_hit) <= w || fabsf(h.t[2] - t_hit) <= w || fabsf(h.t[3] - t_hit) <= w;  // NaN: false
}
#ifndef RT_CONTAINER_TIES
#define RT_CONTAINER_TIES 1  // experiments: 0 = crossings of different primitives ordered by FP32 t alone (round 1),
                             // 2 = the best container's order key fetched in the tie case instead of carried (slower)
#endif
struct ContainerScan {
    float best_t = -INFINITY, best_ior = 1.0f, hit_last = -INFINITY;
    bool have_best = false, hit_in = false;
    uint32_t best_prim = kNoPrim;     // the record best_ior came from (its order key; the exact chain reads its f64 index)
    const uint32_t* order = nullptr;  // record -> world.objects position (null: the records ARE in that order)
#if RT_CONTAINER_TIES == 1
    uint32_t best_ord = 0u;
#endif

    // Is crossing (t, ord) later in the reference's sorted list than the best one so far?  Crossings within the
    // tie window of each other keep world.objects order (a stable sort under the approximate comparison): the
    // bottom of a glass cube standing ON the floor plane and the plane itself, seen from a point on that floor.
    RT_HD bool later_than_best(float t, uint32_t ord) const {
#if RT_CONTAINER_TIES == 0
        return t > best_t;
#elif RT_CONTAINER_TIES == 1
        return (fabsf(t - best_t) <= tie_window(best_t)) ? (ord > best_ord) : (t > best_t);
#else
        if (fabsf(t - best_t) <= tie_window(best_t)) return ord > (order ? order[best_prim] : best_prim);  // (rare: the key is fetched here)
        return t > best_t;
#endif
    }
    // ord / ord_hit: world.objects positions of this primitive and of the hit one
    RT_HD void add(const Hits& h, bool is_hit_prim, uint32_t ord, float ior, uint32_t ord_hit, float t_hit, uint32_t prim) {
        int skipk = -1;  // the hit itself = the hit primitive's intersection closest to t_hit
        float bd = INFINITY;
        for (int k = 0; k < 4; ++k)
            if (is_hit_prim && fabsf(h.t[k] - t_hit) < bd) bd = fabsf(h.t[k] - t_hit), skipk = k;
        int cnt = 0;
        float last = -INFINITY;
        for (int k = 0; k < 4; ++k)
            if (k != skipk && h.t[k] == h.t[k] && precedes_hit(h.t[k], t_hit, ord, ord_hit)) ++cnt, last = fmaxf(last, h.t[k]);
        if (cnt & 1) {
            if (is_hit_prim) hit_in = true, hit_last = last;
            else if (!have_best || later_than_best(last, ord)) {
                have_best = true, best_t = last, best_ior = ior, best_prim = prim;
#if RT_CONTAINER_TIES == 1
                best_ord = ord;
#endif
            }
        }
    }
    RT_HD bool hit_is_last(uint32_t ord_hit) const { return !have_best || later_than_best(hit_last, ord_hit); }
    RT_HD void finish(float hit_ior, uint32_t ord_hit, float& n1, float& n2) const {
        const float excl = have_best ? best_ior : 1.0f;
        if (hit_in) {
            n1 = hit_is_last(ord_hit) ? hit_ior : best_ior;
            n2 = excl;
        } else {
            n1 = excl;
            n2 = hit_ior;
        }
    }
    // The same answer as record indices: kNoPrim = outside everything (index 1.0), `hit_prim` = the hit one.
    RT_HD void finish_prims(uint32_t hit_prim, uint32_t ord_hit, uint32_t& p1, uint32_t& p2) const {
        const uint32_t excl = have_best ? best_prim : kNoPrim;
        if (hit_in) {
            p1 = hit_is_last(ord_hit) ? hit_prim : best_prim;
            p2 = excl;
        } else {
            p1 = excl;
            p2 = hit_prim;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Hit preparation: intersection.rs:40-69 + shape.rs:83-96,118-137 + per-shape normal_at.
struct HitInfo {
    V3 p;        // world hit point  (ray.position(t))
    V3 n;        // world normal, flipped towards the eye
    V3 eye;      // -direction
    V3 q;        // snapped object-space hit point
    V3 dq;       // object-space offset of over_point: eps * inv_linear * n   (under_point: -dq)
    bool inside;
    V3 eq;       // bound on the FP32 error of the object-space hit point, per axis (pattern_unsure)
    bool edge;   // the local normal was a near decision: the edge of a cube, the rim of a capped cylinder
};

#ifndef RT_DIR_ERR
#define RT_DIR_ERR 1.5e-7f  // bound on the absolute FP32 error of a component of a unit ray direction (prepare_hit)
#endif
template <bool EXOTIC = true>
RT_HD HitInfo prepare_hit(const DPrim& p, const DExtra* extras, const Ray& ray, bool self, float t, float hu, float hv,
                          bool want_eq = true) {  // want_eq = false: h.eq is not needed (solid pattern, FP32-only path)
    HitInfo h;
    V3 o = xform_point(p.inv, ray.o), d = xform_vector(p.inv, ray.d);
    if (self) o = ray.q + ray.dq;
    V3 q = madd(d, t, o);
    V3 nobj = v3(0.0f, 1.0f, 0.0f);
    h.eq = v3(0.0f, 0.0f, 0.0f);
    if (want_eq) {  // (a coordinate the snap below makes exact carries no error: the height over a plane, a cube's face)
        const float e = 3e-7f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + fabsf(t) * (fabsf(d.x) + fabsf(d.y) + fabsf(d.z)) + 1.0f);
        h.eq = v3(e, e, e);
    }
    h.edge = false;
    switch (p.type) {
        case PRIM_SPHERE: {  // sphere.rs:74-76 — hit point from the perpendicular foot, snapped onto the unit sphere
            float tc, half;
            V3 l;
            sphere_foot(o, d, dot(d, d), tc, l, half);
            q = normalize(madd(d, (t < tc) ? -half : half, l));
            nobj = q;
            break;
        }
        case PRIM_PLANE:  // plane.rs:53-55
            q.y = 0.0f;
            if (want_eq) {  // t = -o.y / d.y is ill-conditioned for a grazing ray: its error moves the point along the ray
                const float a = fabsf(p.inv[4] * ray.o.x) + fabsf(p.inv[5] * ray.o.y) + fabsf(p.inv[6] * ray.o.z) + fabsf(p.inv[7]);
                const float b = fabsf(p.inv[4] * ray.d.x) + fabsf(p.inv[5] * ray.d.y) + fabsf(p.inv[6] * ray.d.z);
                // (the FP32 direction itself is a unit vector known to an ABSOLUTE 1e-7 or so per component, whatever
                // the size of d.y: towards the horizon that, not the rounding of the sum, is what moves t)
                const float c = fabsf(p.inv[4]) + fabsf(p.inv[5]) + fabsf(p.inv[6]);
                const float dt = fmaf(2e-7f, fmaf(fabsf(t), b, a), RT_DIR_ERR * fabsf(t) * c) * rcp_f(fabsf(d.y));
                h.eq.x = fmaf(fabsf(d.x), dt, h.eq.x), h.eq.z = fmaf(fabsf(d.z), dt, h.eq.z);
            }
            h.eq.y = 0.0f;
            nobj = v3(0.0f, 1.0f, 0.0f);
            break;
        case PRIM_CUBE: {  // cube.rs:125-141: exact max, then approx tie order x, y, z
            float ax = fabsf(q.x), ay = fabsf(q.y), az = fabsf(q.z);
            float mc = fmaxf(ax, fmaxf(ay, az));
            h.edge = (ax > mc - kGraze ? 1 : 0) + (ay > mc - kGraze ? 1 : 0) + (az > mc - kGraze ? 1 : 0) >= 2;
            if (approx(mc, ax)) {
                q.x = copysignf(1.0f, q.x);
                h.eq.x = 0.0f;
                nobj = v3(q.x, 0.0f, 0.0f);
            } else if (approx(mc, ay)) {
                q.y = copysignf(1.0f, q.y);
                h.eq.y = 0.0f;
                nobj = v3(0.0f, q.y, 0.0f);
            } else {
                q.z = copysignf(1.0f, q.z);
                h.eq.z = 0.0f;
                nobj = v3(0.0f, 0.0f, q.z);
            }
            break;
        }
        case PRIM_CYLINDER: {  // cylinder.rs:172-184
            if (!EXOTIC) break;
            const float* e = extras[p.extra].f;
            float dist = fmaf(q.x, q.x, q.z * q.z);
            float top = e[1] - kEps, bot = e[0] + kEps;
            h.edge = dist > 0.99f && (fabsf(q.y - e[1]) < kGraze || fabsf(q.y - e[0]) < kGraze);
            // a cap is a plane: t = (cap - o.y) / d.y is ill-conditioned for a ray that grazes it (as for PRIM_PLANE)
            const float cap_c = fabsf(p.inv[4]) + fabsf(p.inv[5]) + fabsf(p.inv[6]);
            const float cap_dt = fmaf(2e-7f, fabsf(e[1]) + fabsf(e[0]) + fabsf(o.y) + fabsf(t * d.y), RT_DIR_ERR * fabsf(t) * cap_c) * rcp_f(fabsf(d.y));
            if (dist < 1.0f && (approx(q.y, top) || q.y > top)) {
                q.y = e[1];
                h.eq.x = fmaf(fabsf(d.x), cap_dt, h.eq.x), h.eq.z = fmaf(fabsf(d.z), cap_dt, h.eq.z), h.eq.y = 0.0f;
                nobj = v3(0.0f, 1.0f, 0.0f);
            } else if (dist < 1.0f && (approx(q.y, bot) || q.y < bot)) {
                q.y = e[0];
                h.eq.x = fmaf(fabsf(d.x), cap_dt, h.eq.x), h.eq.z = fmaf(fabsf(d.z), cap_dt, h.eq.z), h.eq.y = 0.0f;
                nobj = v3(0.0f, -1.0f, 0.0f);
            } else {
                float tc, half;
                V3 l;
                V3 d2 = v3(d.x, 0.0f, d.z);
                sphere_foot(v3(o.x, 0.0f, o.z), d2, dot(d2, d2), tc, l, half);
                V3 w = madd(d2, (t < tc) ? -half : half, l);
                float s = rsq_f(fmaf(w.x, w.x, w.z * w.z));
                q.x = w.x * s, q.z = w.z * s;
                nobj = v3(q.x, 0.0f, q.z);
            }
            break;
        }
        case PRIM_TRIANGLE: {  // triangle.rs:135-137
            if (!EXOTIC) break;
            const float* e = extras[p.extra].f;
            nobj = v3(e[9], e[10], e[11]);
            break;
        }
        default: {  // smooth_triangle.rs:26-33
            if (!EXOTIC) break;
            const float* e = extras[p.extra].f;
            V3 n0 = v3(e[12], e[13], e[14]), n1 = v3(e[15], e[16], e[17]), n2 = v3(e[18], e[19], e[20]);
            nobj = n1 * hu + n2 * hv + n0 * (1.0f - hu - hv);
            break;
        }
    }
    V3 n = normalize(xform_normal(p.inv, nobj));
    h.eye = neg(ray.d);
    h.inside = dot(n, h.eye) < 0.0f;  // intersection.rs:48
    if (h.inside) n = neg(n);
    h.n = n;
    h.p = madd(ray.d, t, ray.o);
    h.q = q;
    h.dq = xform_vector(p.inv, n) * kEps;
    return h;
}

// ---------------------------------------------------------------------------------------------
// pattern.rs:58-96, evaluated at the object-space over_point (q + dq)
// `|f % 2| <= 1e-5` (pattern.rs:74,82,89) for an integer-valued f: f/2 has no fractional part.  Exact
// (halving and truncation do not round); every float >= 2^24 is even; inf/NaN give false like fmod.
RT_HD bool is_even(float f) {
    float h = 0.5f * f;
    return h - truncf(h) == 0.0f;
}
RT_HD V3 pattern_color(const DPattern& pat, V3 obj_point) {
    V3 a = v3(pat.a[0], pat.a[1], pat.a[2]);
    if (pat.type == PAT_SOLID) return a;
    V3 b = v3(pat.b[0], pat.b[1], pat.b[2]);
    V3 pp = xform_point(pat.inv, obj_point);
    switch (pat.type) {
        case PAT_STRIPE: return is_even(floorf(pp.x)) ? a : b;
        case PAT_GRADIENT: return a + (b - a) * (pp.x - floorf(pp.x));
        case PAT_RING: return is_even(floorf(sqrtf(fmaf(pp.x, pp.x, pp.z * pp.z)))) ? a : b;
        default: return is_even(floorf(pp.x) + floorf(pp.y) + floorf(pp.z)) ? a : b;
    }
}

// Does the pattern decision at obj_point lie within FP32 reach of a boundary (then the exact chain
// decides it in f64)?  e_q bounds the error of obj_point itself; the pattern-space coordinates add the
// rounding of their own sums, relative to the size of the terms (so the height 1e-5 of an over point
// above its own plane, a sum of one small term, counts as decided).
RT_HD bool pattern_unsure(const DPattern& pat, V3 p, V3 e_q) {
    if (pat.type == PAT_SOLID) return false;
    float c[3], room[3];
    for (int k = 0; k < 3; ++k) {
        const float* m = pat.inv + 4 * k;
        c[k] = fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[3])));
        const float lin = fabsf(m[0]) * e_q.x + fabsf(m[1]) * e_q.y + fabsf(m[2]) * e_q.z;
        const float mag = fabsf(m[0] * p.x) + fabsf(m[1] * p.y) + fabsf(m[2] * p.z) + fabsf(m[3]);
        room[k] = fmaf(4e-7f, mag, lin);
    }
    auto near = [](float x, float r) {
        const float f = x - floorf(x);
        return !(fminf(f, 1.0f - f) > r);  // NaN counts as near
    };
    switch (pat.type) {
        case PAT_STRIPE: return near(c[0], room[0]);
        // (a gradient is continuous between its jumps, but its blend factor is the fractional part of a
        // coordinate that FP32 only knows to room[0]: far from the pattern's origin that alone exceeds 1e-5)
        case PAT_GRADIENT: return near(c[0], room[0]) || room[0] > 1e-5f;
        case PAT_RING: return near(sqrtf(fmaf(c[0], c[0], c[2] * c[2])), room[0] + room[2]);
        default: return near(c[0], room[0]) || near(c[1], room[1]) || near(c[2], room[2]);
    }
}

// material.rs:138-159 for ONE light cell: returns (light . normal) and the specular factor, both
// already gated (0 when the reference skips them).
// One shading point's side of material.rs:138-159; the part that does not depend on the light cell.
struct PhongPoint {
    V3 p, n, eye;
    float shininess;
    float n_eye;  // n . eye
    float gate;   // (reflect . eye) below this: the specular factor is < 2^-36 and dropped
};
RT_HD PhongPoint phong_point(V3 p, V3 n, V3 eye, float shininess) {
    PhongPoint q;
    q.p = p, q.n = n, q.eye = eye, q.shininess = shininess;
    q.n_eye = dot(n, eye);
    q.gate = shininess > 0.0f ? fast_exp2(-36.0f * rcp_f(shininess)) : 0.0f;  // a threshold: a few ulps are immaterial
    return q;
}
// (reflect . eye)^shininess inside a highlight; lv = cell - point, inv = 1/|lv|.
// x^200 turns the 1e-7 absolute rounding error of an FP32 dot product of two unit vectors into 2e-5
// RELATIVE error on highlights whose value exceeds 1, so cos(theta) is taken from the cross product
// instead: sin^2 = |r x e|^2 / (|r|^2 |e|^2) keeps full relative accuracy in (1 - cos), and the
// power is exp(shininess/2 * log1p(-sin^2)).
RT_HD float phong_highlight(const PhongPoint& q, V3 lv, float inv) {
    V3 ml = neg(lv * inv);
    V3 rv = ml - q.n * 2.0f * dot(ml, q.n);
    V3 cx = cross_exact(rv, q.eye);
    float sin2 = fminf(div_f(dot(cx, cx), dot(rv, rv) * dot(q.eye, q.eye)), 1.0f);
    if (sin2 < 0.25f) return expf(0.5f * q.shininess * log1pf(-sin2));
    return powf(dot(rv, q.eye), q.shininess);
}
RT_HD void phong_cell(const PhongPoint& q, V3 cell, float& ldn_out, float& spec_out) {
    V3 lv = cell - q.p;
    float len2 = dot(lv, lv);
    float inv = (len2 <= kEps * kEps) ? 0.0f : rsq_f(len2);  // normalize().unwrap_or(0)
    float ldn = dot(lv, q.n) * inv;
    ldn_out = 0.0f;
    spec_out = 0.0f;
    if (ldn >= -kEps) {  // float::ge(light_dot_normal, 0.0)
        ldn_out = ldn;
        // (-lightv).reflect(normalv) . eye = 2 (lightv.n)(n.eye) - lightv.eye: five instructions a
        // cell; the reflected vector itself is only formed inside a highlight.
        // factor = (reflect . eye)^shininess (material.rs:150-153).  Outside the highlight the factor
        // is below 2^-36 and is dropped (1e-11 of a colour unit) — most cells of most shading points,
        // so the wave usually skips the power altogether.
        float rde_fast = fmaf(2.0f * ldn, q.n_eye, -(dot(lv, q.eye) * inv));
        if (rde_fast > q.gate) spec_out = phong_highlight(q, lv, inv);
    }
}

// The same sums over ALL cell centres of an area light (material.rs:133-161 with light.rs:192-216's
// grid): centre(cu, cv) - p = a + cu u + cv v, so the three quantities a cell needs — |lv|^2, lv.n
// and lv.R with R = 2 (n.eye) n - eye, since (-lightv).reflect(n) . eye = lightv . R — are a quadratic
// and two linear forms in (cu, cv) whose coefficients depend on the shading point only.  A cell then
// costs five FMAs instead of forming the vector and three dot products; the vector itself is only
// formed inside a highlight (phong_highlight, unchanged).
struct PhongGrid {
    V3 a, u, v;                    // lv at cell (0, 0); the per-cell steps
    float an, un, vn;              // lv . n = an + cu un + cv vn
    float ar, ur, vr;              // lv . R = ar + cu ur + cv vr
    float aa, au2, av2, uu, vv, uv2;  // |lv|^2 = aa + cu au2 + cv av2 + cu^2 uu + cv^2 vv + cu cv uv2
};
RT_HD PhongGrid phong_grid(const PhongPoint& q, const DLight& l) {
    PhongGrid g;
    g.u = v3(l.uvec[0], l.uvec[1], l.uvec[2]), g.v = v3(l.vvec[0], l.vvec[1], l.vvec[2]);
    g.a = (v3(l.pos[0], l.pos[1], l.pos[2]) + (g.u + g.v) * 0.5f) - q.p;
    const V3 R = madd(q.n, 2.0f * q.n_eye, neg(q.eye));
    g.an = dot(g.a, q.n), g.un = dot(g.u, q.n), g.vn = dot(g.v, q.n);
    g.ar = dot(g.a, R), g.ur = dot(g.u, R), g.vr = dot(g.v, R);
    g.aa = dot(g.a, g.a), g.au2 = 2.0f * dot(g.a, g.u), g.av2 = 2.0f * dot(g.a, g.v);
    g.uu = dot(g.u, g.u), g.vv = dot(g.v, g.v), g.uv2 = 2.0f * dot(g.u, g.v);
    return g;
}
// One row of cells (fixed cv) from column `base` on: the forms in k = cu - base.
struct PhongRow {
    float c0, c1, dn0, dr0;
};
RT_HD PhongRow phong_row(const PhongGrid& g, float fcv) {
    PhongRow r;
    r.c0 = fmaf(fcv, fmaf(fcv, g.vv, g.av2), g.aa);
    r.c1 = fmaf(fcv, g.uv2, g.au2);
    r.dn0 = fmaf(fcv, g.vn, g.an);
    r.dr0 = fmaf(fcv, g.vr, g.ar);
    return r;
}
RT_HD PhongRow phong_row_from(const PhongGrid& g, const PhongRow& r, float base) {
    PhongRow s;
    s.c0 = fmaf(base, fmaf(base, g.uu, r.c1), r.c0);
    s.c1 = fmaf(2.0f * base, g.uu, r.c1);
    s.dn0 = fmaf(base, g.un, r.dn0);
    s.dr0 = fmaf(base, g.ur, r.dr0);
    return s;
}
// What is known about ALL cells of one light seen from one shading point (bounds over the ball
// around the light, with room to spare); lets the sums run without the per-cell gates.
//   far_front : no cell centre is within 1e-2 of the point, and every cell is in front of the surface
//               (material.rs:145's gate passes everywhere, normalize() never fails)
//   no_spec   : (reflect . eye) stays below the highlight gate for every cell
struct PhongBounds {
    bool far_front, no_spec;
};
RT_HD PhongBounds phong_bounds(const PhongPoint& q, V3 D, float dlen, float lr, float cosk, bool wide) {
    PhongBounds b;
    b.far_front = !wide & (dlen - lr > 1e-2f) & (dot(D, q.n) - lr > 1e-3f * (dlen + lr));
    const V3 R = madd(q.n, 2.0f * q.n_eye, neg(q.eye));
    const float r2 = dot(R, R);
    const float rlen = r2 * rsq_f(fmaxf(r2, 1e-30f));
    const float inv_dl = rcp_f(dlen);
    const float ct = fminf(fmaxf(dot(D, R) * inv_dl * rsq_f(fmaxf(r2, 1e-30f)), -1.0f), 1.0f);  // cos angle(D, R)
    const float st = sqrt_f(fmaxf(1.0f - ct * ct, 0.0f));
    const float sb = lr * inv_dl;                                  // sin of the light ball's half angle
    const float best = rlen * fmaf(ct, cosk, st * sb);             // |R| cos(theta - beta) when theta > beta
    b.no_spec = !wide & (ct < cosk - 1e-3f) & (best < q.gate - 2e-3f * fmaxf(rlen, 1.0f));
    return b;
}
// GATED: the per-cell gates of material.rs:145 and of normalize() are evaluated (else known to pass);
// SPEC: a highlight is possible.  fk = cu - base of the row.
template <bool GATED, bool SPEC>
RT_HD void phong_grid_cell(const PhongPoint& q, const PhongGrid& g, const PhongRow& r, float fk, float base, float fcv, float& sum_d,
                           float& sum_s) {
    float len2 = fmaf(fk, fmaf(fk, g.uu, r.c1), r.c0);
    float inv = (GATED && len2 <= kEps * kEps) ? 0.0f : rsq_f(len2);
    float ldn = fmaf(fk, g.un, r.dn0) * inv;
    if (!GATED || ldn >= -kEps) {
        sum_d += ldn;
        if (SPEC) {
            float rde_fast = fmaf(fk, g.ur, r.dr0) * inv;
            if (rde_fast > q.gate) sum_s += phong_highlight(q, madd(g.u, base + fk, madd(g.v, fcv, g.a)), inv);
        }
    }
}
// All cells, in the reference's order (material.rs:133-136: rows of v, cells of u).  Without
// highlights the row is straight-line code, eight cells at a time with constant indices.
template <bool GATED, bool SPEC>
RT_HD void phong_grid_sums(const PhongPoint& q, const PhongGrid& g, uint32_t usteps, uint32_t vsteps, float& sum_d, float& sum_s) {
    for (uint32_t cv = 0; cv < vsteps; ++cv) {
        const float fcv = static_cast<float>(cv);
        const PhongRow row = phong_row(g, fcv);
        if (SPEC) {
            for (uint32_t cu = 0; cu < usteps; ++cu)
                phong_grid_cell<GATED, true>(q, g, row, static_cast<float>(cu), 0.0f, fcv, sum_d, sum_s);
        } else {
            for (uint32_t cu = 0; cu < usteps; cu += 8u) {
                const float base = static_cast<float>(cu);
                const PhongRow r = cu ? phong_row_from(g, row, base) : row;
                const uint32_t n = usteps - cu;
                if (n >= 8u) {
                    phong_grid_cell<GATED, false>(q, g, r, 0.0f, base, fcv, sum_d, sum_s);
                    phong_grid_cell<GATED, false>(q, g, r, 1.0f, base, fcv, sum_d, sum_s);
                    phong_grid_cell<GATED, false>(q, g, r, 2.0f, base, fcv, sum_d, sum_s);
                    phong_grid_cell<GATED, false>(q, g, r, 3.0f, base, fcv, sum_d, sum_s);
                    phong_grid_cell<GATED, false>(q, g, r, 4.0f, base, fcv, sum_d, sum_s);
                    phong_grid_cell<GATED, false>(q, g, r, 5.0f, base, fcv, sum_d, sum_s);
                    phong_grid_cell<GATED, false>(q, g, r, 6.0f, base, fcv, sum_d, sum_s);
                    phong_grid_cell<GATED, false>(q, g, r, 7.0f, base, fcv, sum_d, sum_s);
                } else {
                    uint32_t k = 0;
                    if (n >= 4u) {
                        phong_grid_cell<GATED, false>(q, g, r, 0.0f, base, fcv, sum_d, sum_s);
                        phong_grid_cell<GATED, false>(q, g, r, 1.0f, base, fcv, sum_d, sum_s);
                        phong_grid_cell<GATED, false>(q, g, r, 2.0f, base, fcv, sum_d, sum_s);
                        phong_grid_cell<GATED, false>(q, g, r, 3.0f, base, fcv, sum_d, sum_s);
                        k = 4u;
                    }
                    for (; k < n; ++k) phong_grid_cell<GATED, false>(q, g, r, static_cast<float>(k), base, fcv, sum_d, sum_s);
                }
            }
        }
    }
}

// intersection.rs:124-141
RT_HD float schlick(V3 eye, V3 n, float n1, float n2) {
    float c = dot(eye, n);
    if (n1 > n2) {
        float r = div_f(n1, n2);
        float sin2_t = r * r * (1.0f - c * c);
        if (sin2_t > 1.0f) return 1.0f;
        c = sqrt_f(1.0f - sin2_t);
    }
    float r0 = div_f(n1 - n2, n1 + n2);
    r0 *= r0;
    float k = 1.0f - c;
    float k2 = k * k;
    return r0 + (1.0f - r0) * (k2 * k2 * k);
}

// world.rs:113-125: returns false under total internal reflection
RT_HD bool refract_dir(V3 eye, V3 n, float n1, float n2, V3& dir) {
    float r = div_f(n1, n2);
    float cos_i = dot(eye, n);
    float sin2_t = r * r * (1.0f - cos_i * cos_i);
    if (sin2_t > 1.0f) return false;
    float cos_t = sqrt_f(1.0f - sin2_t);
    dir = n * (r * cos_i - cos_t) - eye * r;
    return true;
}

// camera.rs:210-228.  direction = normalize(inv_linear * (wx, wy, -1)) — identical to
// normalize(pixel - origin) in exact arithmetic, without the cancellation for far cameras.
RT_HD void camera_ray(const DCamera& c, uint32_t x, uint32_t y, V3& origin, V3& dir) {
    float wx = c.half_width - (static_cast<float>(x) + 0.5f) * c.pixel_size;
    float wy = c.half_height - (static_cast<float>(y) + 0.5f) * c.pixel_size;
    origin = v3(c.inv[3], c.inv[7], c.inv[11]);
    dir = normalize(xform_vector(c.inv, v3(wx, wy, -1.0f)));
}

// Jitter hash shared bit-for-bit with oracle/rt_oracle.hpp (counter_jitter).
RT_HD uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
RT_HD uint32_t jitter_pixel_key(uint32_t seed, uint32_t pixel) { return fmix32(pixel ^ (seed * 0x9E3779B9u)); }
// One hash per (pixel, path node, light, cell); the u draw is its top byte, the v draw the next one.
RT_HD void counter_jitter(uint32_t pixel_key, uint32_t node, uint32_t light, uint32_t cell, float& ju, float& jv) {
    // (node < 256 up to depth 7; the bits the shift drops at depth 8 are folded in separately)
    uint32_t k = (node << 24) | (light << 16) | cell;
    uint32_t h = fmix32(pixel_key + k * 0x85EBCA77u + (node >> 8) * 0x27D4EB2Fu);
    ju = static_cast<float>(h >> 24) * (1.0f / 255.0f);
    jv = static_cast<float>((h >> 16) & 0xFFu) * (1.0f / 255.0f);
}

// light.rs:211-216
RT_HD V3 point_on_light(const DLight& l, uint32_t u, uint32_t v, float ju, float jv) {
    V3 c = v3(l.pos[0], l.pos[1], l.pos[2]);
    V3 uv = v3(l.uvec[0], l.uvec[1], l.uvec[2]), vv = v3(l.vvec[0], l.vvec[1], l.vvec[2]);
    return c + uv * (static_cast<float>(u) + ju) + vv * (static_cast<float>(v) + jv);
}

// canvas.rs:37-39: `(c * 255.0) as u8`
RT_HD uint8_t quantize_u8(float c) {
    // the product in f64, as the reference forms it: an f32 product can round UP onto an integer the
    // f64 product stays below (the kernel is HBM-bound, the f64 multiply is free)
    double v = static_cast<double>(c) * 255.0;
    if (!(v == v) || v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return static_cast<uint8_t>(v);
}

}  // namespace rtk
