// rt_exact.h — the f64 geometry chain of the render kernels ("exact chain").
//
// FP32 meets the f64 reference at 1e-4 almost everywhere (rt_math.h), but not where colour is
// discontinuous in position (pattern boundaries, silhouettes, the corner line of two walls) and not
// along deep glass paths, where every curved interface magnifies the 1e-6 position error of an FP32
// world coordinate (the round-2 parity census: 5e-4 of C4's pixels at 4K, profiles/
// r02_parity_census_before.txt).  The kernels therefore carry every PATH ray (camera, reflection,
// refraction) a second time as f64 origin + direction, and at a path node that needs it — it spawns
// children, its pattern lookup lands within FP32 error of a boundary, or the FP32 nearest-hit
// search saw a second candidate it could not tell apart — the hit is worked out again in f64 from
// the f64 ray and the f64 tables of the caller's scene, with the REFERENCE'S OWN formulas in the
// reference's operation order (no FMA contraction: the library is built with -ffp-contract=off):
// ray.rs:17-22 + matrix.rs:184-198 (fold from 0.0), sphere.rs:38-72, plane.rs:39-51, cube.rs:39-122,
// cylinder.rs:126-225, triangle.rs:104-133, intersection.rs:40-69,105-120, shape.rs:83-96,
// pattern.rs:58-96, world.rs:94-133.  The FP32 machinery keeps doing the bulk work — hierarchy walks,
// candidate search, shadow rays, Phong sums — and the chain only replaces the handful of quantities the
// decisions hang on, so a refined node reproduces the reference's t, hit point, normal, pattern
// cell and child rays to the last bit (given the same candidates), including the exact ties the
// reference itself decides by its rounding noise (a camera in the symmetry plane of two walls).
//
// Everything is RT_HD: the scalar executable spec (tests/cpp/hostsim.cpp) compiles the same text.
#pragma once

#include "rt_math.h"

namespace rtk {

struct D3 {
    double x, y, z;
};
RT_HD D3 d3(double x, double y, double z) { return D3{x, y, z}; }
RT_HD D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }  // tuple.rs:208-218
RT_HD D3 xneg(D3 a) { return D3{0.0 - a.x, 0.0 - a.y, 0.0 - a.z}; }          // tuple.rs:200-206: 0 - v
RT_HD double xdot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // tuple.rs:107-109
RT_HD D3 xcross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RT_HD V3 to_v3(D3 a) { return V3{static_cast<float>(a.x), static_cast<float>(a.y), static_cast<float>(a.z)}; }
RT_HD D3 to_d3(V3 a) { return D3{a.x, a.y, a.z}; }

constexpr double kEpsD = 1e-5;  // float.rs:3
RT_HD bool xapprox(double a, double b) {  // float.rs:5-11
    if (isinf(a) && isinf(b)) return a == b;
    return fabs(a - b) <= kEpsD;
}
RT_HD bool xge(double a, double b) { return xapprox(a, b) || a > b; }
RT_HD bool xle(double a, double b) { return xapprox(a, b) || a < b; }
// partial_cmp == Greater (float.rs:29-37: NaN falls through to Greater)
RT_HD bool xgreater(double a, double b) { return !xapprox(a, b) && !(a < b); }

// matrix.rs:184-198 on the first three rows of an affine matrix (the fourth is 0 0 0 1): each
// component is the fold 0.0 + m0 x + m1 y + m2 z + m3 w, left to right.
RT_HD D3 xpoint(const double* m, D3 p) {
    return D3{(((0.0 + m[0] * p.x) + m[1] * p.y) + m[2] * p.z) + m[3] * 1.0, (((0.0 + m[4] * p.x) + m[5] * p.y) + m[6] * p.z) + m[7] * 1.0,
              (((0.0 + m[8] * p.x) + m[9] * p.y) + m[10] * p.z) + m[11] * 1.0};
}
RT_HD D3 xvector(const double* m, D3 v) {
    return D3{(((0.0 + m[0] * v.x) + m[1] * v.y) + m[2] * v.z) + m[3] * 0.0, (((0.0 + m[4] * v.x) + m[5] * v.y) + m[6] * v.z) + m[7] * 0.0,
              (((0.0 + m[8] * v.x) + m[9] * v.y) + m[10] * v.z) + m[11] * 0.0};
}
// transform_inverse.transpose() * normal (shape.rs:89-90); the transposed matrix's fourth column is the
// inverse's fourth row (0 0 0 1), times the vector's w = 0
RT_HD D3 xnormal(const double* m, D3 n) {
    return D3{(((0.0 + m[0] * n.x) + m[4] * n.y) + m[8] * n.z) + 0.0 * 0.0, (((0.0 + m[1] * n.x) + m[5] * n.y) + m[9] * n.z) + 0.0 * 0.0,
              (((0.0 + m[2] * n.x) + m[6] * n.y) + m[10] * n.z) + 0.0 * 0.0};
}
RT_HD double xmagnitude(D3 v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }  // tuple.rs:92-94
// tuple.rs:102-104,228-236: v * (1 / |v|); false for |v| <= eps
RT_HD bool xnormalize(D3 v, D3& out) {
    const double m = xmagnitude(v);
    if (xapprox(m, 0.0)) return false;
    out = v * (1.0 / m);
    return true;
}

// The caller's scene in f64 (rt_scene_desc.*64), in stream order (rt_pack.h).
struct XPrim {  // per stream record (hierarchy nodes: unused)
    double inv[12];
};
struct XExtra {  // cylinder {min, max, closed}; triangle {v0, e0, e1, normal, n0, n1, n2}
    double f[24];
};
struct XPattern {
    double inv[12];
};
struct XCamera {
    double pixel_size, half_width, half_height;
    double inv[12];
};

// camera.rs:210-228
RT_HD void xcamera_ray(const XCamera& c, uint32_t x, uint32_t y, D3& origin, D3& dir) {
    const double xoffset = (static_cast<double>(x) + 0.5) * c.pixel_size;
    const double yoffset = (static_cast<double>(y) + 0.5) * c.pixel_size;
    const double world_x = c.half_width - xoffset, world_y = c.half_height - yoffset;
    const D3 pixel = xpoint(c.inv, d3(world_x, world_y, -1.0));
    origin = xpoint(c.inv, d3(0.0, 0.0, 0.0));
    D3 n = d3(0.0, 0.0, 0.0);
    xnormalize(pixel - origin, n);
    dir = n;
}

// The intersections of ONE primitive with the f64 ray in fixed slots, in the order the reference pushes
// them (a cylinder: wall roots 0 and 1, caps 2 and 3; the others: 0 and 1); an absent one is +inf.  Every
// index is a compile-time constant, so the slots are registers.
struct XHits {
    double t[4], u, v;
};
constexpr double kAbsent = static_cast<double>(INFINITY);
// cube.rs:79-97
RT_HD void xcheck_axis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
    const double tmin_num = mn - origin, tmax_num = mx - origin;
    if (xge(fabs(direction), kEpsD)) {  // approximate ge: true for every finite direction
        tmin = tmin_num / direction;
        tmax = tmax_num / direction;
    } else {
        tmin = tmin_num * static_cast<double>(INFINITY);
        tmax = tmax_num * static_cast<double>(INFINITY);
    }
    if (tmin > tmax) {
        const double s = tmin;
        tmin = tmax, tmax = s;
    }
}
RT_HD double xmax_by3(double a, double b, double c) {  // Iterator::max_by: the LAST of approximately equal maxima
    double m = xgreater(a, b) ? a : b;
    return xgreater(m, c) ? m : c;
}
RT_HD double xmin_by3(double a, double b, double c) {  // Iterator::min_by: the FIRST of approximately equal minima
    double m = xgreater(a, b) ? b : a;
    return xgreater(m, c) ? c : m;
}
RT_HD bool xcheck_cap(D3 o, D3 d, double t) {  // cylinder.rs:220-225
    const double x = o.x + t * d.x, z = o.z + t * d.z;
    return xle(x * x + z * z, 1.0);
}
RT_HD void xcaps(D3 o, D3 d, double ymin, double ymax, bool closed, XHits& h) {  // cylinder.rs:186-217
    if (!closed || xapprox(d.y, 0.0)) return;
    double t = (ymin - o.y) / d.y;
    if (xcheck_cap(o, d, t)) h.t[2] = t;
    t = (ymax - o.y) / d.y;
    if (xcheck_cap(o, d, t)) h.t[3] = t;
}
RT_HD XHits xabsent() {
    XHits h;
    h.u = 0.0, h.v = 0.0;
    h.t[0] = h.t[1] = h.t[2] = h.t[3] = kAbsent;
    return h;
}
// EXOTIC = false compiles the cylinder / triangle code out (rt_math.h intersect_prim does the same).
template <bool EXOTIC = true>
RT_HD XHits xintersect(uint32_t type, const double* inv, const double* e, D3 ro, D3 rd) {
    XHits h;
    h.u = 0.0, h.v = 0.0;
    h.t[0] = h.t[1] = h.t[2] = h.t[3] = kAbsent;
    const D3 o = xpoint(inv, ro), d = xvector(inv, rd);  // shape.rs:79-81, ray.rs:17-22
    switch (type) {
        case PRIM_SPHERE: {  // sphere.rs:38-72
            const double a = xdot(d, d);
            const double b = 2.0 * xdot(d, o);
            const double c = xdot(o, o) - 1.0;
            const double disc = b * b - 4.0 * a * c;
            if (disc < 0.0) break;
            h.t[0] = (-b - sqrt(disc)) / (2.0 * a);
            h.t[1] = (-b + sqrt(disc)) / (2.0 * a);
            break;
        }
        case PRIM_PLANE:  // plane.rs:39-51
            if (!xapprox(d.y, 0.0)) h.t[0] = -o.y / d.y;
            break;
        case PRIM_CUBE: {  // cube.rs:39-76,101-122
            double x0, x1, y0, y1, z0, z1;
            xcheck_axis(o.x, d.x, -1.0, 1.0, x0, x1);
            xcheck_axis(o.y, d.y, -1.0, 1.0, y0, y1);
            xcheck_axis(o.z, d.z, -1.0, 1.0, z0, z1);
            const double tmin = xmax_by3(x0, y0, z0), tmax = xmin_by3(x1, y1, z1);
            if (tmin > tmax) break;
            h.t[0] = tmin, h.t[1] = tmax;
            break;
        }
        case PRIM_CYLINDER: {  // cylinder.rs:126-170
            if (!EXOTIC) break;
            const double ymin = e[0], ymax = e[1];
            const bool closed = e[2] != 0.0;
            const double a = d.x * d.x + d.z * d.z;
            if (xapprox(a, 0.0)) {
                xcaps(o, d, ymin, ymax, closed, h);
                break;
            }
            const double b = 2.0 * o.x * d.x + 2.0 * o.z * d.z;
            const double c = o.x * o.x + o.z * o.z - 1.0;
            const double disc = b * b - 4.0 * a * c;
            if (disc < 0.0) break;
            double t0 = (-b - sqrt(disc)) / (2.0 * a), t1 = (-b + sqrt(disc)) / (2.0 * a);
            if (t0 > t1) {
                const double s = t0;
                t0 = t1, t1 = s;
            }
            const double y0 = o.y + t0 * d.y;
            if (ymin < y0 && y0 < ymax) h.t[0] = t0;
            const double y1 = o.y + t1 * d.y;
            if (ymin < y1 && y1 < ymax) h.t[1] = t1;
            xcaps(o, d, ymin, ymax, closed, h);
            break;
        }
        default: {  // triangle.rs:104-133 (smooth triangles intersect alike, smooth_triangle.rs:22-24)
            if (!EXOTIC) break;
            const D3 v0 = d3(e[0], e[1], e[2]), e0 = d3(e[3], e[4], e[5]), e1 = d3(e[6], e[7], e[8]);
            const D3 dir_cross_e1 = xcross(d, e1);
            const double det = xdot(e0, dir_cross_e1);
            if (xapprox(fabs(det), 0.0)) break;
            const double f = 1.0 / det;
            const D3 p0 = o - v0;
            const double u = f * xdot(p0, dir_cross_e1);
            if (!(u >= 0.0 && u <= 1.0)) break;
            const D3 oxe0 = xcross(p0, e0);
            const double v = f * xdot(d, oxe0);
            if (v < 0.0 || (u + v) > 1.0) break;
            h.t[0] = f * xdot(e1, oxe0), h.u = u, h.v = v;
            break;
        }
    }
    return h;
}

// The f64 crossings of one primitive handed to the FP32 container scan (rt_math.h ContainerScan): each is placed
// relative to the f64 hit by the reference's rule (|t - t_hit| <= 1e-5 is a tie) and written so that the scan's own
// FP32 rule repeats that decision: a tie becomes t_hit itself, the others keep their value but stay outside its window.
RT_HD Hits xplace(const XHits& x, double t64, float t_hit) {
    Hits h;
    const float w = 2.0f * tie_window(t_hit);
    for (int k = 0; k < 4; ++k) {
        const double t = x.t[k];
        if (t == kAbsent) h.t[k] = RT_NOHIT;
        else if (fabs(t - t64) <= kEpsD) h.t[k] = t_hit;
        else if (t < t64) h.t[k] = fminf(static_cast<float>(t), t_hit - w);
        else h.t[k] = fmaxf(static_cast<float>(t), t_hit + w);
    }
    return h;
}

// Intersection::sort + hit (intersection.rs:105-120) over the intersections of up to two candidate
// primitives, entered in world.objects order: approximate comparator, insertion sort (stable), the
// first t > 0.0 wins.  K slots per primitive (absent = +inf, which sorts last and never wins); every index
// is a compile-time constant so that the lists stay in registers.
RT_HD bool xless(double a, double b) { return !(fabs(a - b) <= kEpsD) && a < b; }  // (inf - inf = NaN: not less, like approx)
template <int K>
struct XCandidates {
    static constexpr int N = 2 * K;
    double t[N], u[2], v[2];
    uint32_t prim[2];
    int who[N];  // which of the two primitives an entry belongs to
    RT_HD XCandidates() {
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = kAbsent, who[i] = i / K;
        prim[0] = prim[1] = kNoPrim, u[0] = u[1] = v[0] = v[1] = 0.0;
    }
    RT_HD void set(int which, const XHits& h, uint32_t prim_id) {  // which = 0 first, then 1
        prim[which] = prim_id, u[which] = h.u, v[which] = h.v;
#pragma unroll
        for (int k = 0; k < K; ++k) t[which * K + k] = h.t[k];
    }
    RT_HD bool hit(double& t_out, uint32_t& prim_out, double& u_out, double& v_out) {
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const double kt = t[i];
            const int kw = who[i];
            bool moving = true;
#pragma unroll
            for (int j = i; j > 0; --j) {
                const bool shift = moving && xless(kt, t[j - 1]);
                if (shift) t[j] = t[j - 1], who[j] = who[j - 1];
                else if (moving) t[j] = kt, who[j] = kw, moving = false;
            }
            if (moving) t[0] = kt, who[0] = kw;
        }
        bool found = false;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (!found && t[i] > 0.0 && t[i] < kAbsent) {
                found = true;
                t_out = t[i], prim_out = who[i] ? prim[1] : prim[0], u_out = who[i] ? u[1] : u[0], v_out = who[i] ? v[1] : v[0];
            }
        return found;
    }
};
// The same for ONE primitive without cylinder caps (at most two intersections, pushed t[0] then t[1]).
RT_HD bool xhit_of_one(const XHits& h, double& t_out) {
    const bool swap = xless(h.t[1], h.t[0]);
    const double a = swap ? h.t[1] : h.t[0], b = swap ? h.t[0] : h.t[1];
    if (a > 0.0 && a < kAbsent) {
        t_out = a;
        return true;
    }
    if (b > 0.0 && b < kAbsent) {
        t_out = b;
        return true;
    }
    return false;
}
// Nearest hit among one or two candidate primitives (second_type = REC_NODE: none), picking the smallest
// network that holds their intersections.
template <bool EXOTIC>
RT_HD bool xselect(const XHits& ha, uint32_t prim_a, uint32_t type_a, const XHits& hb, uint32_t prim_b, uint32_t type_b, bool two,
                   double& t_out, uint32_t& prim_out, double& u_out, double& v_out) {
    const bool caps = EXOTIC && (type_a == PRIM_CYLINDER || (two && type_b == PRIM_CYLINDER));
    if (!two && !caps) {
        prim_out = prim_a, u_out = ha.u, v_out = ha.v;
        return xhit_of_one(ha, t_out);
    }
    if (!caps) {
        XCandidates<2> xc;
        xc.set(0, ha, prim_a);
        xc.set(1, hb, prim_b);
        return xc.hit(t_out, prim_out, u_out, v_out);
    }
    XCandidates<4> xc;
    xc.set(0, ha, prim_a);
    if (two) xc.set(1, hb, prim_b);
    return xc.hit(t_out, prim_out, u_out, v_out);
}

// intersection.rs:40-69 + shape.rs:83-96,118-137 + the shapes' local normals
struct XHit {
    D3 point, normal, eye, reflectv, over, under;
    D3 qobj;    // the hit point in object space; the coordinate a face / cap / plane fixes is set exactly
    int qround;  // what is left for FP32 (xsnap_q): 1 = scale onto the unit sphere, 2 = x, z onto the unit circle
    bool qedge;  // a cube's point within 1e-5 of an edge that got the NEIGHBOURING face's normal, a cylinder's wall point that got an end's (below): not snapped
    bool inside;
};
template <bool EXOTIC = true>
RT_HD XHit xprepare(uint32_t type, const double* inv, const double* e, D3 ro, D3 rd, double t, double hu, double hv) {
    XHit h;
    h.point = ro + rd * t;  // ray.rs:13-15
    h.eye = xneg(rd);
    const D3 p = xpoint(inv, h.point);
    h.qobj = p, h.qround = 0, h.qedge = false;
    D3 nobj = d3(0.0, 1.0, 0.0);
    switch (type) {
        case PRIM_SPHERE:  // sphere.rs:74-76
            nobj = p - d3(0.0, 0.0, 0.0);
            h.qround = 1;
            break;
        case PRIM_PLANE:  // plane.rs:53-55
            nobj = d3(0.0, 1.0, 0.0);
            h.qobj = d3(p.x, 0.0, p.z);
            break;
        case PRIM_CUBE: {                                       // cube.rs:125-141
            const double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
            double mc = (ax > ay) ? ax : ay;  // max_by with the exact comparison: the largest value
            mc = (mc > az) ? mc : az;
            // The coordinate of the face is set exactly — if the point IS on that face.  Within 1e-5 of an edge the
            // approximate comparison gives the normal of the first axis in x, y, z order, which may belong to the
            // NEIGHBOURING face (cover_image, pixel (1504, 406): a point on the front face, 9e-6 under the top edge, gets
            // the top face's normal); the over point then lies inside the cube and the rays that leave it meet that
            // face from within at t ~ 1e-5.  Such a point keeps its own coordinates.
            // Such a point keeps its own coordinates, and whether the rays leaving it (shadow rays first of all) meet
            // the cube again is the reference's own rounding noise — pixel (2471, 1404): the over point lies 1.8e-15
            // inside the face it stands on, so every shadow ray "leaves the cube" at t = 1e-15 > 0 and is shadowed:
            // qedge tells the shadow walks to ask f64 about the primitive itself (rt_kernel.h "unsure shadow rays").
            constexpr double kOnFace = 1e-9;
            if (xapprox(mc, ax)) {
                nobj = d3(p.x, 0.0, 0.0), h.qedge = !(fabs(ax - 1.0) <= kOnFace);
                if (!h.qedge) h.qobj.x = p.x < 0.0 ? -1.0 : 1.0;
            } else if (xapprox(mc, ay)) {
                nobj = d3(0.0, p.y, 0.0), h.qedge = !(fabs(ay - 1.0) <= kOnFace);
                if (!h.qedge) h.qobj.y = p.y < 0.0 ? -1.0 : 1.0;
            } else {
                nobj = d3(0.0, 0.0, p.z), h.qedge = !(fabs(az - 1.0) <= kOnFace);
                if (!h.qedge) h.qobj.z = p.z < 0.0 ? -1.0 : 1.0;
            }
            // ... and a point ON an edge line (two coordinates at +-1 to rounding: the far end of the room's floor, where it
            // meets the wall) stands 1e-15 inside or outside the OTHER face: the same noise decides what leaves it.
            if ((fabs(ax - 1.0) <= kOnFace) + (fabs(ay - 1.0) <= kOnFace) + (fabs(az - 1.0) <= kOnFace) >= 2) h.qedge = true;
            break;
        }
        case PRIM_CYLINDER: {  // cylinder.rs:172-184
            if (!EXOTIC) break;
            const double dist = p.x * p.x + p.z * p.z;
            // A WALL point within 1e-5 of an end gets that end's normal whenever its x^2 + z^2 happens to round below 1
            // (shapes_zoo, pixel (1424, 727): the open cylinder's wall 8e-6 above its lower end).  The over point then
            // still stands on the wall, and whether the rays that leave it meet the wall again at t ~ 1e-16 is the
            // reference's own rounding noise: qedge, as for the cube; the point keeps its height and is a wall point.
            const bool on_wall = fabs(dist - 1.0) <= 1e-9;
            if (dist < 1.0 && xge(p.y, e[1] - kEpsD)) nobj = d3(0.0, 1.0, 0.0);
            else if (dist < 1.0 && xle(p.y, e[0] + kEpsD)) nobj = d3(0.0, -1.0, 0.0);
            else nobj = d3(p.x, 0.0, p.z), h.qround = 2;
            if (h.qround != 2) {
                if (on_wall) h.qedge = true, h.qround = 2;
                else h.qobj.y = nobj.y > 0.0 ? e[1] : e[0];
            } else if (fabs(p.y - e[1]) <= 1e-9 || fabs(p.y - e[0]) <= 1e-9) {
                // ... and a wall point ON a rim line (random_field:218: seen from inside, y = ymax to the last bit) stands
                // 1e-16 under or over the cap's plane with its over point: the same noise
                h.qedge = true;
            }
            break;
        }
        case PRIM_TRIANGLE:  // triangle.rs:135-137
            if (EXOTIC) nobj = d3(e[9], e[10], e[11]);
            break;
        default: {  // smooth_triangle.rs:26-33
            if (!EXOTIC) break;
            const D3 n0 = d3(e[12], e[13], e[14]), n1 = d3(e[15], e[16], e[17]), n2 = d3(e[18], e[19], e[20]);
            nobj = n1 * hu + n2 * hv + n0 * (1.0 - hu - hv);
            break;
        }
    }
    D3 n = d3(0.0, 1.0, 0.0);
    xnormalize(xnormal(inv, nobj), n);
    h.inside = xdot(n, h.eye) < 0.0;  // intersection.rs:48
    if (h.inside) n = xneg(n);
    h.normal = n;
    h.reflectv = rd - n * 2.0 * xdot(rd, n);  // tuple.rs:121-123
    h.over = h.point + n * kEpsD;
    h.under = h.point - n * kEpsD;
    return h;
}

// The object-space point the rays leaving a refined hit stand on (Ray::q): the f64 point rounded to FP32 and put
// onto the surface the way rt_math.h prepare_hit does it (a unit vector to FP32 rounding; exact face coordinates).
RT_HD V3 xsnap_q(const XHit& h) {
    V3 q = to_v3(h.qobj);
    if (h.qround == 1) q = normalize(q);
    else if (h.qround == 2) {
        const float s = rsq_f(fmaf(q.x, q.x, q.z * q.z));
        q.x *= s, q.z *= s;
    }
    return q;
}

// `|f % 2| <= 1e-5` (pattern.rs:74,82,89) for an integer-valued f (a floor, or a sum of floors): |f % 2| is
// 0 or 1, so the test says "f is even" — halving and flooring are exact; inf / NaN give false like fmod.
RT_HD bool xeven(double f) {
    const double h = 0.5 * f;
    return h - floor(h) == 0.0;
}
// pattern.rs:58-96 at the over point (material.rs:127): object space, then pattern space.  Returns
// whether colour a applies (Solid: always; Gradient: `frac` receives the blend factor instead).
RT_HD bool xpattern_is_a(uint32_t pat_type, const double* prim_inv, const double* pat_inv, D3 over, double& frac) {
    frac = 0.0;
    if (pat_type == PAT_SOLID) return true;
    const D3 pp = xpoint(pat_inv, xpoint(prim_inv, over));
    switch (pat_type) {
        case PAT_STRIPE: return xeven(floor(pp.x));
        case PAT_GRADIENT: frac = pp.x - floor(pp.x); return true;
        case PAT_RING: return xeven(floor(hypot(pp.x, pp.z)));
        default: return xeven(floor(pp.x) + floor(pp.y) + floor(pp.z));
    }
}

// world.rs:109-130: false under total internal reflection
RT_HD bool xrefract(const XHit& h, double n1, double n2, D3& dir) {
    const double n_ratio = n1 / n2;
    const double cos_i = xdot(h.eye, h.normal);
    const double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);  // powi(x, 2) = x * x
    if (sin2_t > 1.0) return false;
    const double cos_t = sqrt(1.0 - sin2_t);
    dir = h.normal * (n_ratio * cos_i - cos_t) - h.eye * n_ratio;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Shadow rays FP32 cannot settle (rt_math.h occludes_prim_soft): World::is_shadowed (world.rs:73-92) against ONE
// primitive, in f64 from the f64 over point to the f64 light position.  hit = the first t > 0 of the (sorted) list,
// shadowed iff hit.t < distance: for one question "is there an intersection with 0 < t < distance".
struct XLight {  // light.rs:42-48,81-89 in f64: point: position; area: corner, uvec, vvec
    double pos[3], uvec[3], vvec[3];
};
// light.rs:211-216; the draws are bytes / 255 (light.rs:149-152), 0.5 in the deterministic mode
RT_HD D3 xpoint_on_light(const XLight& l, uint32_t u, uint32_t v, double ju, double jv) {
    const D3 c = d3(l.pos[0], l.pos[1], l.pos[2]), uv = d3(l.uvec[0], l.uvec[1], l.uvec[2]), vv = d3(l.vvec[0], l.vvec[1], l.vvec[2]);
    return c + uv * (static_cast<double>(u) + ju) + vv * (static_cast<double>(v) + jv);
}
template <bool EXOTIC = true>
RT_HD bool xoccludes(uint32_t type, const double* inv, const double* e, D3 point, D3 light_position) {
    const D3 point_to_light = light_position - point;
    const double distance = xmagnitude(point_to_light);
    D3 dir = d3(0.0, 0.0, 0.0);
    if (!xnormalize(point_to_light, dir)) return false;
    const XHits h = xintersect<EXOTIC>(type, inv, e, point, dir);
    bool occ = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) occ |= (h.t[k] > 0.0) & (h.t[k] < distance);  // (absent = +inf)
    return occ;
}
// group.rs:107 + bounding_box.rs:81-84 in f64: the ray sees a primitive only if it enters the box of every Group around
// it (tmin < tmax, strict).  For the rays whose FP32 answer was a near thing (rt_math.h gates_pass_near): XGate is the
// caller's own f64 box of a Group (rt_scene_desc.node_box64), parallel to DGate, whose `parent` links are followed.
struct XGate {
    double mn[3], mx[3];
};
RT_HD bool xgates_pass(const DGate* gates, const XGate* xgates, uint32_t g, D3 o, D3 d) {
    while (g != kNoGate) {
        const XGate& G = xgates[g];
        double x0, x1, y0, y1, z0, z1;
        xcheck_axis(o.x, d.x, G.mn[0], G.mx[0], x0, x1);
        xcheck_axis(o.y, d.y, G.mn[1], G.mx[1], y0, y1);
        xcheck_axis(o.z, d.z, G.mn[2], G.mx[2], z0, z1);
        if (!(xmax_by3(x0, y0, z0) < xmin_by3(x1, y1, z1))) return false;
        g = gates[g].parent;
    }
    return true;
}
// ... for a shadow ray: the ray World::is_shadowed forms (world.rs:82-84), as in xoccludes
RT_HD bool xgates_pass_shadow(const DGate* gates, const XGate* xgates, uint32_t g, D3 point, D3 light_position) {
    D3 dir = d3(0.0, 0.0, 0.0);
    if (!xnormalize(light_position - point, dir)) return false;
    return xgates_pass(gates, xgates, g, point, dir);
}
// The jitter draws of rt_math.h counter_jitter as the reference forms them: byte as f64 / 255.0
RT_HD void xcounter_jitter(uint32_t pixel_key, uint32_t node, uint32_t light, uint32_t cell, double& ju, double& jv) {
    uint32_t k = (node << 24) | (light << 16) | cell;
    uint32_t h = fmix32(pixel_key + k * 0x85EBCA77u + (node >> 8) * 0x27D4EB2Fu);
    ju = static_cast<double>(h >> 24) / 255.0;
    jv = static_cast<double>((h >> 16) & 0xFFu) / 255.0;
}

}  // namespace rtk
