// hostsim.cpp — TEST-ONLY scalar executor of the kernel's FP32 per-lane arithmetic (rt_math.h).
//
// This container has no GPU, so the FP32-vs-f64 parity of the kernel's formulation (snapped
// self-intersection, stable sphere discriminant, order-free n1/n2, iterative ray stack) is
// debugged here: the same rt_math.h functions the HIP kernel calls are compiled with g++ and
// driven by a plain scalar loop.  Wave-level orchestration (ballots, cooperative area-light
// sampling, the threaded traversal's uniform jumps) is NOT simulated — that is covered by the
// `-m gpu` tests.  Never linked into the product library; the product has no CPU path.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define RT_HOST_ULP_NOISE 1  // rt_math.h: rcp / sqrt / rsq of this build can be moved by an ulp (hostsim_set_ulp_noise)
namespace rtk {
unsigned rt_host_ulp_noise = 0;
}
#include "../../raytracer_amd/csrc/rt_pack.h"

using namespace rtk;

namespace {

struct Sim {
    PackedScene ps;
    DCamera cam;
    XCamera cam64;
    bool exact = false;  // the f64 chain of rt_exact.h is on (scene and camera carry f64 tables)
    uint32_t max_depth, jitter_mode, seed;
};

struct Closest {
    uint32_t prim;
    float t, u, v;
    // what the exact chain needs to know about the search (rt_kernel.h keeps the same three values):
    bool unsure = false;         // the winning test was a near thing
    uint32_t alt = kNoPrim;      // another primitive FP32 cannot tell from the winner (or from a hit at all)
    float alt_t = INFINITY;
};

// visit(i, record) -> keep going?   Records are walked in stream order; a node of the device's
// hierarchy whose box the ray cannot meet within [t_lo, t_hi()] is jumped over (its `next` link is
// kept in `material`).  The caller's Groups are gates checked on a primitive's own hits.
template <class THi, class Visit>
void traverse(const Sim& s, const Ray& ray, float t_lo, THi&& t_hi, Visit&& visit) {
    const V3 inv_d = inv_dir(ray.d);
    uint32_t i = 0;
    while (i < s.ps.n_items) {
        const DPrim& r = s.ps.recs[i];
        if (r.type == REC_NODE) {
            i = bvh_node_hit(r, ray.o, inv_d, t_lo, t_hi()) ? i + 1 : r.material;
            continue;
        }
        if (!visit(i, r)) return;
        ++i;
    }
}
bool gated_in(const Sim& s, uint32_t prim, const Ray& ray) {
    return !s.ps.has_gates || gates_pass(s.ps.gates.data(), s.ps.gate[prim], ray.o, ray.d);
}
// ... and whether some Group's box on the way was a near thing (rt_math.h gates_pass_near)
bool gated_in_near(const Sim& s, uint32_t prim, V3 o, V3 d, bool& near) {
    return !s.ps.has_gates || gates_pass_near(s.ps.gates.data(), s.ps.gate[prim], o, d, near);
}
// (the render kernels ask only in their EXOTIC instantiations, rt_kernel.h RT_GATE_NEAR; the second count always)
bool gated_in_near_first_pass(const Sim& s, uint32_t prim, V3 o, V3 d, bool& near) {
    if (!s.ps.exotic) return !s.ps.has_gates || gates_pass(s.ps.gates.data(), s.ps.gate[prim], o, d);
    return gated_in_near(s, prim, o, d, near);
}

Closest closest_hit(const Sim& s, const Ray& ray, uint32_t excl0 = kNoPrim, uint32_t excl1 = kNoPrim) {
    Closest c{kNoPrim, INFINITY, 0.0f, 0.0f};
    auto note_alt = [&](uint32_t prim, float t) {  // a candidate that is not the winner: keep the nearest one
        // (the primitive the ray starts on: its near roots are exact by construction, rt_math.h "self")
        if (t <= c.t + amb_window(c.t) && t < c.alt_t && t > -amb_window(t) && prim != c.prim && prim != ray.self && gated_in(s, prim, ray))
            c.alt = prim, c.alt_t = t;
    };
    traverse(s, ray, 0.0f, [&] { return c.t + (s.exact ? amb_window(c.t) : tie_window(c.t)); }, [&](uint32_t prim, const DPrim& p) {
        if (prim == excl0 || prim == excl1) return true;
        intersect_prim(p, s.ps.extras.data(), ray, prim == ray.self, [&](int slot, float t, float u, float v) {
            bool better = closer_hit(t, c.t);
            if (!better && c.prim != kNoPrim && ties_hit(t, c.t)) better = s.ps.order[prim] < s.ps.order[c.prim];
            if (better) {
                if (!gated_in(s, prim, ray)) return;  // group.rs:107: the ray does not see this primitive
                const Closest old = c;
                c.prim = prim, c.t = t, c.u = u, c.v = v, c.unsure = (slot & kUnsure) != 0;
                // the old winner is the nearest loser so far, if it is within the new winner's window
                if (old.prim != kNoPrim && old.prim != prim && old.t <= t + amb_window(t)) c.alt = old.prim, c.alt_t = old.t;
                else if (c.alt != kNoPrim && !(c.alt_t <= t + amb_window(t))) c.alt = kNoPrim, c.alt_t = INFINITY;
            } else if (s.exact) {
                note_alt(prim, t);
            }
        }, [&](float t) { if (s.exact) note_alt(prim, t); });
        return true;
    });
    if (c.alt != kNoPrim && c.prim != kNoPrim && !(c.alt_t <= c.t + amb_window(c.t))) c.alt = kNoPrim;  // too far behind to matter
    return c;
}

// world.rs:73-92: shadowed iff the nearest positive hit is closer than the light
bool occluded(const Sim& s, const Ray& ray, float dist) {
    bool occ = false;
    traverse(s, ray, 0.0f, [&] { return dist; }, [&](uint32_t prim, const DPrim& p) {
        bool hit = false;
        if (prim == ray.self)
            intersect_prim(p, s.ps.extras.data(), ray, true, [&](int, float t, float, float) { hit |= in_range(t, dist); });
        else
            hit = occludes_prim(p, s.ps.extras.data(), ray.o, ray.d, dist);
        occ = hit && gated_in(s, prim, ray);
        return !occ;
    });
    return occ;
}
// The same with the exact chain (rt_kernel.h "unsure shadow rays").  The render kernel's walk: the plain answer, and
// `near` = some test on the way was a near thing (a grazed sphere, cube edge, cylinder wall or rim, triangle edge, a Group's
// box clipped at a corner, the own sphere at the terminator).
std::atomic<uint64_t> g_sh_rays{0}, g_sh_unsure{0}, g_sh_f64{0}, g_sh_redo{0}, g_sh_why_qedge{0}, g_sh_why_near{0};
bool occluded_near(const Sim& s, const Ray& ray, float dist, bool& near) {
    bool occ = false;
    g_sh_rays += 1;
    traverse(s, ray, 0.0f, [&] { return dist; }, [&](uint32_t prim, const DPrim& p) {
        bool hit;
        if (prim == ray.self) hit = occludes_self_near(p, s.ps.extras.data(), ray, dist, near);
        else hit = occludes_prim_near(p, s.ps.extras.data(), ray.o, ray.d, dist, near);
        occ = hit && gated_in_near_first_pass(s, prim, ray.o, ray.d, near);
        return !occ;
    });
    return occ;
}
// The fix kernel's walk (rt_aux_kernels.h fix_occluded): FP32 soft tests where they are sure, f64 from the f64 over point
// to the f64 light position where they are not, and always for the primitive the point stands on.
bool occluded_fix(const Sim& s, uint32_t self_prim, V3 o, V3 d, float dist, D3 over64, D3 light64) {
    const ShadowWin w = shadow_windows(dist);
    Ray ray;
    ray.o = o, ray.d = d, ray.self = kNoPrim, ray.q = v3(0, 0, 0), ray.dq = v3(0, 0, 0);
    bool occ = false;
    traverse(s, ray, w.lo_m, [&] { return w.hi_m; }, [&](uint32_t prim, const DPrim& p) {
        ShadowTest st{true, false};
        if (prim != self_prim) st = occludes_prim_soft(p, s.ps.extras.data(), o, d, w);
        bool gate_near = false;
        bool hit = st.maybe && gated_in_near(s, prim, o, d, gate_near);
        if (hit && !st.sure) {
            g_sh_f64 += 1;
            hit = xoccludes(p.type, s.ps.xprims[prim].inv, s.ps.xextras[p.extra].f, over64, light64);
        }
        if (hit && gate_near) hit = xgates_pass_shadow(s.ps.gates.data(), s.ps.xgates.data(), s.ps.gate[prim], over64, light64);
        occ = hit;
        return !occ;
    });
    return occ;
}

// intersection.rs:71-103 without the sorted list: a primitive is a "container" at the hit iff an
// odd number of its intersections precede the hit; `visited.last()` is the container whose last
// crossing before the hit has the largest t.
void containers(const Sim& s, const Ray& ray, uint32_t hit_prim, float t_hit, float hit_ior, float& n1, float& n2, uint32_t& n1_prim,
                uint32_t& n2_prim, bool refined = false, double t64 = 0.0, D3 o64 = D3{0, 0, 0}, D3 d64 = D3{0, 0, 0}, float cos_hit = 1.0f) {
    uint32_t amb_prim = kNoPrim;
    const float amb_w = container_amb_window(t_hit, cos_hit);
    ContainerScan cs;
    cs.order = s.ps.order.data();
    const V3 hp = madd(ray.d, t_hit, ray.o);
    const float hm = container_margin(t_hit, hp);
    const V3 inv_d = inv_dir(ray.d);
    for (uint32_t i = 0; i < s.ps.n_items;) {  // the kernel's container walk: its own node test
        const DPrim& p = s.ps.recs[i];
        if (p.type == REC_NODE) {
            const bool in = s.ps.exotic ? bvh_node_hit(p, ray.o, inv_d, -INFINITY, t_hit + tie_window(t_hit)) : node_holds_point(p, hp, hm);
            i = in ? i + 1 : p.material;
            continue;
        }
        const uint32_t prim = i++;
        Hits h;
        h.t[0] = h.t[1] = h.t[2] = h.t[3] = RT_NOHIT;
        bool any = false;
        auto crossing = [&](int slot, float t, float, float) { h.t[slot & 3] = t, any = true; };
        if (s.ps.exotic && ray.self == prim)  // (rt_kernel.h: the triangle the ray starts on, a hair past its edge in FP32)
            intersect_prim(p, s.ps.extras.data(), ray, true, crossing, [&](float t) { if (p.type >= PRIM_TRIANGLE) h.t[0] = t, any = true; });
        else
            intersect_prim(p, s.ps.extras.data(), ray, ray.self == prim, crossing);
        if (any && gated_in(s, prim, ray)) {
            // the kernel's rule: the first primitive with a crossing FP32 cannot order against the hit waits for f64
            if (s.exact && refined && prim != hit_prim && amb_prim == kNoPrim && crossing_near_hit(h, t_hit, amb_w)) amb_prim = prim;
            else cs.add(h, prim == hit_prim, s.ps.order[prim], p.ior, s.ps.order[hit_prim], t_hit, prim);
        }
    }
    if (amb_prim != kNoPrim) {
        const DPrim& pa = s.ps.recs[amb_prim];
        const XHits hx = xintersect<true>(pa.type, s.ps.xprims[amb_prim].inv, s.ps.xextras[pa.extra].f, o64, d64);
        cs.add(xplace(hx, t64, t_hit), false, s.ps.order[amb_prim], pa.ior, s.ps.order[hit_prim], t_hit, amb_prim);
    }
    cs.finish(hit_ior, s.ps.order[hit_prim], n1, n2);
    cs.finish_prims(hit_prim, s.ps.order[hit_prim], n1_prim, n2_prim);
}

// ---------------------------------------------------------------------------------------------
// Audit of the kernel's shaft tests (rt_math.h "Shafts", rt_kernel.h area-light pass).  The kernel
// skips the shadow rays of a (shading point, area light) pair against every record its shaft test
// rules out.  Here the same tests are evaluated, then EVERY primitive is tested exactly against
// rays to the jittered point and to the four corners of every light cell: a primitive that
// occludes one of them although it was ruled out is a violation (there must be none).
std::atomic<bool> g_audit{false};
std::atomic<uint64_t> g_audit_pairs{0}, g_audit_ruled_out{0}, g_audit_open{0}, g_audit_violations{0}, g_audit_umbra{0},
    g_audit_umbra_violations{0}, g_audit_front{0}, g_audit_dull{0}, g_audit_phong_violations{0}, g_audit_behind{0};

void audit_shaft(const Sim& s, const DLight& l, uint32_t li, uint32_t pixel_key, uint32_t node, V3 over, const HitInfo& h,
                 uint32_t self, float shininess) {
    const uint32_t n = s.ps.n_items;
    std::vector<uint8_t> may(n, 0);
    V3 lcen;
    float lrad;
    light_ball(l, lcen, lrad);
    const Shaft sh = make_shaft(over, lcen, lrad);
    {  // the bounds that let the kernel drop the per-cell gates of the Phong sums (rt_math.h PhongBounds):
       // checked against the per-cell evaluation of material.rs:138-159 (phong_cell) at every cell centre
        const PhongPoint pp = phong_point(over, h.n, h.eye, shininess);
        const PhongBounds pb = phong_bounds(pp, sh.D, sh.dlen, sh.lr, sh.cosk, sh.wide);
        g_audit_front += pb.far_front ? 1u : 0u, g_audit_dull += pb.no_spec ? 1u : 0u;
        if (pb.far_front || pb.no_spec)
            for (uint32_t cv = 0; cv < l.vsteps; ++cv)
                for (uint32_t cu = 0; cu < l.usteps; ++cu) {
                    const V3 centre = point_on_light(l, cu, cv, 0.5f, 0.5f);
                    const V3 lv = centre - over;
                    float ldn, spec;
                    phong_cell(pp, centre, ldn, spec);
                    if (pb.far_front && !(ldn > 0.0f && dot(lv, lv) > 1e-6f)) g_audit_phong_violations += 1;
                    if (pb.no_spec && spec != 0.0f) g_audit_phong_violations += 1;
                }
    }
    const int side = shaft_side_of_surface(sh, h.n, l);
    const bool self_clear = !h.inside && side > 0;
    if (side < 0) {  // declared "whole light behind the surface": every cell's Phong terms must be gated off
        const PhongPoint pp = phong_point(over, h.n, h.eye, shininess);
        g_audit_behind += 1;
        for (uint32_t cv = 0; cv < l.vsteps; ++cv)
            for (uint32_t cu = 0; cu < l.usteps; ++cu) {
                float ldn, spec;
                phong_cell(pp, point_on_light(l, cu, cv, 0.5f, 0.5f), ldn, spec);
                if (ldn != 0.0f || spec != 0.0f) g_audit_phong_violations += 1;
            }
        return;  // the kernel asks nothing else about this pair
    }
    uint32_t prims = 0, ruled_out = 0;
    bool umbra = false;  // some primitive occludes EVERY ray of the shaft (rt_math.h shaft_blocked_by_*)
    for (uint32_t i = 0; i < n;) {  // the walk of rt_kernel.h, one "lane"
        const DPrim& r = s.ps.recs[i];
        const DBound& b = s.ps.bound[i];
        if (r.type == REC_NODE) {
            i = shaft_meets_ball(sh, v3(b.c[0], b.c[1], b.c[2]), b.r) ? i + 1 : r.material;
            continue;
        }
        bool mb;
        if (r.type == PRIM_PLANE)
            mb = shaft_meets_plane(sh, r.inv, i == self ? h.dq.y : fmaf(r.inv[4], over.x, fmaf(r.inv[5], over.y, fmaf(r.inv[6], over.z, r.inv[7]))));
        else if (i == self) {
            mb = r.type >= PRIM_CYLINDER || !self_clear;
            if (h.inside && r.type < PRIM_CYLINDER && (!s.ps.has_gates || s.ps.gate[i] == kNoGate))
                umbra |= shaft_leaves_solid(sh, v3(b.c[0], b.c[1], b.c[2]), b.r);
        }
        else mb = shaft_meets_ball(sh, v3(b.c[0], b.c[1], b.c[2]), b.r) && (r.type != PRIM_CUBE || shaft_meets_cube(sh, r.inv));
        const bool ungated = !s.ps.has_gates || s.ps.gate[i] == kNoGate;
        if (mb && i != self && ungated) {
            if (r.type == PRIM_PLANE)
                umbra |= shaft_blocked_by_plane(sh, r.inv, fmaf(r.inv[4], over.x, fmaf(r.inv[5], over.y, fmaf(r.inv[6], over.z, r.inv[7]))));
            else if (b.r_in > 0.0f)
                umbra |= shaft_blocked_by_ball(sh, v3(b.c[0], b.c[1], b.c[2]), b.r_in, b.r);
        }
        if (mb && i != self && r.type != PRIM_PLANE && b.margin > 0.0f) {  // the pyramid tests (rt_math.h ShaftObj)
            const ShaftObj so = shaft_in_object_space(r.inv, over, l);
            if (pyramid_misses(so, r.type == PRIM_CUBE, b.margin)) mb = false;
            else if (ungated) umbra |= pyramid_blocked(so, r.type == PRIM_CUBE, b.margin);
        }
        may[i] = mb;
        ++i;
    }
    if (umbra) {  // then every ray to every cell (jittered point and corners) must be occluded
        g_audit_umbra += 1;
        const uint32_t ncell = l.usteps * l.vsteps;
        for (uint32_t cell = 0; cell < ncell; ++cell) {
            const uint32_t cu = cell % l.usteps, cv = cell / l.usteps;
            float ju = 0.5f, jv = 0.5f;
            if (s.jitter_mode == RT_JITTER_COUNTER) counter_jitter(pixel_key, node, li, cell, ju, jv);
            const float us[5] = {ju, 0.0f, 1.0f, 0.0f, 1.0f}, vs[5] = {jv, 0.0f, 0.0f, 1.0f, 1.0f};
            for (int k = 0; k < 5; ++k) {
                const V3 tl = point_on_light(l, cu, cv, us[k], vs[k]) - over;
                const float dist2 = dot(tl, tl);
                const float inv_dist = rsq_f(dist2), dist = dist2 * inv_dist;
                Ray sr;
                sr.o = over, sr.d = tl * inv_dist, sr.self = self, sr.q = h.q, sr.dq = h.dq;
                if (!(dist2 > kEps * kEps) || !occluded(s, sr, dist)) g_audit_umbra_violations += 1, g_audit_violations += 1;
            }
        }
    }
    for (uint32_t i = 0; i < n; ++i)
        if (s.ps.recs[i].type != REC_NODE) ++prims, ruled_out += may[i] ? 0u : 1u;
    g_audit_pairs += 1, g_audit_ruled_out += ruled_out, g_audit_open += (ruled_out == prims) ? 1u : 0u;
    if (ruled_out == 0) return;
    const uint32_t cells = l.usteps * l.vsteps;
    for (uint32_t cell = 0; cell < cells; ++cell) {
        const uint32_t cu = cell % l.usteps, cv = cell / l.usteps;
        float ju = 0.5f, jv = 0.5f;
        if (s.jitter_mode == RT_JITTER_COUNTER) counter_jitter(pixel_key, node, li, cell, ju, jv);
        const float us[5] = {ju, 0.0f, 1.0f, 0.0f, 1.0f}, vs[5] = {jv, 0.0f, 0.0f, 1.0f, 1.0f};
        for (int k = 0; k < 5; ++k) {
            const V3 tl = point_on_light(l, cu, cv, us[k], vs[k]) - over;
            const float dist2 = dot(tl, tl);
            if (!(dist2 > kEps * kEps)) continue;
            const float inv_dist = rsq_f(dist2), dist = dist2 * inv_dist;
            Ray sr;
            sr.o = over, sr.d = tl * inv_dist, sr.self = self, sr.q = h.q, sr.dq = h.dq;
            for (uint32_t i = 0; i < n; ++i) {
                const DPrim& p = s.ps.recs[i];
                if (p.type == REC_NODE || may[i]) continue;
                bool hit = false;
                if (i == self) intersect_prim(p, s.ps.extras.data(), sr, true, [&](int, float t, float, float) { hit |= in_range(t, dist); });
                else hit = occludes_prim(p, s.ps.extras.data(), sr.o, sr.d, dist);
                if (hit && gated_in(s, i, sr)) g_audit_violations += 1;
            }
        }
    }
}

struct PathNode {
    Ray ray;
    D3 o64, d64;  // the same ray in f64 (the exact chain, rt_exact.h)
    V3 w;
    uint32_t depth, node;
};

// tie-window audit (hostsim_tie_audit): nearest-hit decisions that the widened tie window
// (rt_math.h tie_window: 1e-5 max(1, t/4)) takes differently from the reference's fixed 1e-5 window
std::atomic<bool> g_tie_audit{false};
std::atomic<uint64_t> g_tie_hits{0}, g_tie_differs{0}, g_tie_wide_used{0};
Closest closest_hit_fixed_window(const Sim& s, const Ray& ray) {
    Closest c{kNoPrim, INFINITY, 0.0f, 0.0f};
    traverse(s, ray, 0.0f, [&] { return c.t + tie_window(c.t); }, [&](uint32_t prim, const DPrim& p) {
        intersect_prim(p, s.ps.extras.data(), ray, prim == ray.self, [&](int, float t, float u, float v) {
            bool better = t > 0.0f && t < c.t - kEps;
            if (!better && c.prim != kNoPrim && t > 0.0f && fabsf(t - c.t) <= kEps) better = s.ps.order[prim] < s.ps.order[c.prim];
            if (better && gated_in(s, prim, ray)) c = Closest{prim, t, u, v};
        });
        return true;
    });
    return c;
}

// Statistics of the exact chain (hostsim_exact_stats): path nodes, nodes refined in f64, by reason.
std::atomic<uint64_t> g_x_nodes{0}, g_x_refined{0}, g_x_children{0}, g_x_pattern{0}, g_x_amb{0}, g_x_edge{0}, g_x_changed{0},
    g_x_retry{0};

V3 trace_pixel(const Sim& s, uint32_t x, uint32_t y, bool debug = false, std::vector<double>* trace = nullptr) {
    V3 acc = v3(0, 0, 0);
    if (s.ps.n_lights == 0) return acc;  // world.rs:44: fold over no lights == BLACK
    uint32_t pixel_key = jitter_pixel_key(s.seed, y * s.cam.hsize + x);
    PathNode stack[16];
    int sp = 0;
    PathNode cur;
    camera_ray(s.cam, x, y, cur.ray.o, cur.ray.d);
    cur.o64 = d3(0, 0, 0), cur.d64 = d3(0, 0, 0);
    if (s.exact) xcamera_ray(s.cam64, x, y, cur.o64, cur.d64);
    cur.ray.self = kNoPrim, cur.ray.q = v3(0, 0, 0), cur.ray.dq = v3(0, 0, 0);
    cur.w = v3(1, 1, 1), cur.depth = s.max_depth, cur.node = 0;
    const float nl = static_cast<float>(s.ps.n_lights);
    const DExtra* ex = s.ps.extras.data();
    for (;;) {
        bool alive = false;
        // ---- nearest hit: FP32 search; the exact chain settles what FP32 cannot and refines what matters
        Closest c = closest_hit(s, cur.ray);
        // A ray that leaves a point where rounding noise decides whether it meets its own primitive again (rt_math.h
        // with_noise_flag, set by the parent node): f64 is asked about the primitive itself first; a crossing within 1e-9 of the start is the hit
        // (rt_kernel.h "self_noise"), anything else and the search runs again.
        const bool self_noise = s.exact && cur.ray.self != kNoPrim && noise_flag(cur.ray.dq.x);

        if (g_tie_audit && c.prim != kNoPrim) {
            const Closest f = closest_hit_fixed_window(s, cur.ray);
            g_tie_hits += 1, g_tie_differs += (f.prim != c.prim) ? 1u : 0u, g_tie_wide_used += (tie_window(c.t) > kEps) ? 1u : 0u;
        }
        bool refined = false;
        XHit xh{};
        double t64 = 0.0, u64 = 0.0, v64 = 0.0;
        uint32_t hit_prim = c.prim;
        float hit_t = c.t, hit_u = c.u, hit_v = c.v;
        if (s.exact) {
            g_x_nodes += 1;
            // (the same two steps as rt_kernel.h: selection in f64 among what FP32 could not order and for every
            // node that spawns children; then, for a hit that stands, the pattern / edge decisions)
            for (int attempt = 0; attempt < 2; ++attempt) {
                // a winner inside a Group whose box the ray clips within FP32 reach of an edge (rt_kernel.h "gate_out"): the
                // reference's own test in f64; if the ray does not see it, the search runs again without it
                if (attempt == 0 && s.ps.has_gates && s.ps.exotic && c.prim != kNoPrim) {
                    bool gate_near = false;
                    gated_in_near(s, c.prim, cur.ray.o, cur.ray.d, gate_near);
                    if (gate_near && !xgates_pass(s.ps.gates.data(), s.ps.xgates.data(), s.ps.gate[c.prim], cur.o64, cur.d64)) {
                        g_x_retry += 1;
                        c = closest_hit(s, cur.ray, c.prim, kNoPrim);
                        hit_prim = c.prim, hit_t = c.t, hit_u = c.u, hit_v = c.v;
                        attempt = 1;
                    }
                }
                if (attempt == 0 && self_noise) c = Closest{cur.ray.self, 0.0f, 0.0f, 0.0f}, c.unsure = true;
                bool amb = c.unsure || c.alt != kNoPrim;
                bool why_children = false;
                const bool why_pattern = false, why_edge = false;
                if (c.prim != kNoPrim) {
                    const DMaterial& m = s.ps.mats[s.ps.recs[c.prim].material];
                    why_children = cur.depth > 0 && (!(fabsf(m.reflectivity) <= kEps) || !(fabsf(m.transparency) <= kEps));
                }
                if (!(amb || why_children)) break;
                const uint32_t a = c.prim, b = c.alt;  // entered in world.objects order
                const bool a_first = b == kNoPrim || (a != kNoPrim && s.ps.order[a] < s.ps.order[b]);
                const uint32_t first = a_first ? a : b, second = a_first ? b : a;
                const DPrim& pa = s.ps.recs[first];
                const XHits ha = xintersect(pa.type, s.ps.xprims[first].inv, s.ps.xextras[pa.extra].f, cur.o64, cur.d64);
                XHits hb = ha;
                uint32_t type_b = REC_NODE;
                if (second != kNoPrim) {
                    const DPrim& pb = s.ps.recs[second];
                    type_b = pb.type;
                    hb = xintersect(pb.type, s.ps.xprims[second].inv, s.ps.xextras[pb.extra].f, cur.o64, cur.d64);
                }
                uint32_t pr = kNoPrim;
                // (rt_kernel.h "hidden": a near miss in front of the winner that f64 clears may hide another one)
                const bool alt_front = s.ps.exotic && attempt == 0 && a != kNoPrim && b != kNoPrim && a != b && c.alt_t < c.t - amb_window(c.t);
                const bool found = xselect<true>(ha, first, pa.type, hb, second, type_b, second != kNoPrim, t64, pr, u64, v64) &&
                                   !(self_noise && attempt == 0 && !(t64 < 1e-9));
                const bool hidden = found && alt_front && pr == a;
                if (found && !hidden) {
                    g_x_refined += 1, g_x_children += why_children, g_x_pattern += why_pattern, g_x_amb += amb, g_x_edge += why_edge;
                    g_x_changed += (pr != c.prim) ? 1u : 0u;
                    refined = true;
                    hit_prim = pr, hit_t = static_cast<float>(t64), hit_u = static_cast<float>(u64), hit_v = static_cast<float>(v64);
                    break;
                }
                // f64 finds none of the candidates in front of the ray: search again without them
                g_x_retry += 1;
                hit_prim = kNoPrim;
                if (attempt == 1) break;
                if (hidden) c = closest_hit(s, cur.ray, b, kNoPrim);
                else if (self_noise) c = closest_hit(s, cur.ray);  // (nothing is left out: the primitive's far side may be the hit)
                else c = closest_hit(s, cur.ray, a, b);
                hit_prim = c.prim, hit_t = c.t, hit_u = c.u, hit_v = c.v;
            }
        }
        if (hit_prim != kNoPrim) {
            const DPrim& p = s.ps.recs[hit_prim];
            const DMaterial& m = s.ps.mats[p.material];
            HitInfo h = prepare_hit(p, ex, cur.ray, cur.ray.self == hit_prim, hit_t, hit_u, hit_v);
            V3 over = h.p + h.n * kEps;
            V3 pc = pattern_color(s.ps.pats[m.pattern], h.q + h.dq);
            if (s.exact && !refined) {
                const bool why_pattern = pattern_unsure(s.ps.pats[m.pattern], h.q + h.dq, h.eq);
                if (why_pattern || h.edge) {
                    const XHits ha = xintersect(p.type, s.ps.xprims[hit_prim].inv, s.ps.xextras[p.extra].f, cur.o64, cur.d64);
                    uint32_t pr = kNoPrim;
                    refined = xselect<true>(ha, hit_prim, p.type, ha, kNoPrim, REC_NODE, false, t64, pr, u64, v64);
                    if (refined) g_x_refined += 1, g_x_pattern += why_pattern, g_x_edge += h.edge;
                }
            }
            if (refined) {  // the reference's own hit geometry (rt_exact.h)
                xh = xprepare(p.type, s.ps.xprims[hit_prim].inv, s.ps.xextras[p.extra].f, cur.o64, cur.d64, t64, u64, v64);
                over = to_v3(xh.over), h.n = to_v3(xh.normal), h.inside = xh.inside, h.p = to_v3(xh.point);
                h.q = xsnap_q(xh), h.dq = xform_vector(p.inv, h.n) * kEps;  // (the kernel's rule: children start on the f64 point)
                const DPattern& pat = s.ps.pats[m.pattern];
                double frac;
                const bool is_a = xpattern_is_a(pat.type, s.ps.xprims[hit_prim].inv, s.ps.xpats[m.pattern].inv, xh.over, frac);
                const V3 ca = v3(pat.a[0], pat.a[1], pat.a[2]), cb = v3(pat.b[0], pat.b[1], pat.b[2]);
                pc = pat.type == PAT_GRADIENT ? ca + (cb - ca) * static_cast<float>(frac) : (is_a ? ca : cb);
            }
            if (s.exact) h.dq.x = with_noise_flag(h.dq.x, refined && xh.qedge);  // (for the rays that leave this point)
            V3 surface = v3(0, 0, 0);
            double lit_by_light[4] = {-1.0, -1.0, -1.0, -1.0};
            for (uint32_t li = 0; li < s.ps.n_lights; ++li) {
                const DLight& l = s.ps.lights[li];
                V3 lc = v3(l.color[0], l.color[1], l.color[2]);
                V3 ec = pc * lc;
                uint32_t cells = l.usteps * l.vsteps;
                if (g_audit && l.type == RT_LIGHT_AREA) audit_shaft(s, l, li, pixel_key, cur.node, over, h, hit_prim, m.shininess);
                uint32_t lit = 0;
                float sum_d = 0.0f, sum_s = 0.0f;
                bool near_any = false;
                for (uint32_t cell = 0; cell < cells; ++cell) {
                    uint32_t cu = cell % l.usteps, cv = cell / l.usteps;
                    float ju = 0.5f, jv = 0.5f;
                    if (l.type == RT_LIGHT_AREA && s.jitter_mode == RT_JITTER_COUNTER)
                        counter_jitter(pixel_key, cur.node, li, cell, ju, jv);
                    V3 lp = l.type == RT_LIGHT_AREA ? point_on_light(l, cu, cv, ju, jv) : v3(l.pos[0], l.pos[1], l.pos[2]);
                    V3 tl = lp - over;
                    float dist = sqrtf(dot(tl, tl));
                    bool shadowed = false;
                    if (!(fabsf(dist) <= kEps)) {  // null vector => not shadowed (world.rs:77-81)
                        Ray sr;
                        sr.o = over, sr.d = tl * (1.0f / dist), sr.self = hit_prim, sr.q = h.q, sr.dq = h.dq;
                        if (s.exact) shadowed = occluded_near(s, sr, dist, near_any);
                        else shadowed = occluded(s, sr, dist);
                    }
                    if (!shadowed) ++lit;
                    V3 centre = l.type == RT_LIGHT_AREA ? point_on_light(l, cu, cv, 0.5f, 0.5f) : lp;
                    float ldn, spec;
                    phong_cell(phong_point(over, h.n, h.eye, m.shininess), centre, ldn, spec);
                    sum_d += ldn, sum_s += spec;
                }
                if (s.exact && (near_any || (refined && xh.qedge))) {
                    // rt_kernel.h "unsure shadow rays": a FixRecord, and the fix kernel counts this light's cells again
                    g_sh_redo += 1, g_sh_why_near += near_any, g_sh_why_qedge += refined && xh.qedge;
                    D3 over64;
                    bool ok = true;
                    if (refined) {  // (the f64 over point travels whole: FixRecord::ray[0..2])
                        over64 = xh.over;
                    } else {  // the node's f64 hit, worked out again from its f64 ray
                        const XHits ha = xintersect(p.type, s.ps.xprims[hit_prim].inv, s.ps.xextras[p.extra].f, cur.o64, cur.d64);
                        double ft = 0.0, fu = 0.0, fv = 0.0;
                        uint32_t pr = kNoPrim;
                        ok = xselect<true>(ha, hit_prim, p.type, ha, kNoPrim, REC_NODE, false, ft, pr, fu, fv);
                        if (ok) over64 = xprepare(p.type, s.ps.xprims[hit_prim].inv, s.ps.xextras[p.extra].f, cur.o64, cur.d64, ft, fu, fv).over;
                    }
                    if (ok) {
                        const XLight& xl = s.ps.xlights[li];
                        const V3 o = to_v3(over64);
                        lit = 0;
                        for (uint32_t cell = 0; cell < cells; ++cell) {
                            const uint32_t cu = cell % l.usteps, cv = cell / l.usteps;
                            float ju = 0.5f, jv = 0.5f;
                            double ju64 = 0.5, jv64 = 0.5;
                            if (l.type == RT_LIGHT_AREA && s.jitter_mode == RT_JITTER_COUNTER) {
                                counter_jitter(pixel_key, cur.node, li, cell, ju, jv);
                                xcounter_jitter(pixel_key, cur.node, li, cell, ju64, jv64);
                            }
                            const V3 lp = l.type == RT_LIGHT_AREA ? point_on_light(l, cu, cv, ju, jv) : v3(l.pos[0], l.pos[1], l.pos[2]);
                            const D3 lp64 = l.type == RT_LIGHT_AREA ? xpoint_on_light(xl, cu, cv, ju64, jv64) : d3(xl.pos[0], xl.pos[1], xl.pos[2]);
                            const V3 tl = lp - o;
                            const float dist2 = dot(tl, tl);
                            bool unocc = true;
                            if (dist2 > 0.0f) {
                                const float inv_dist = rsq_f(dist2);
                                unocc = !occluded_fix(s, hit_prim, o, tl * inv_dist, dist2 * inv_dist, over64, lp64);
                            }
                            lit += unocc ? 1u : 0u;
                        }
                    }
                }
                float intensity = static_cast<float>(lit) * l.inv_samples;
                if (li < 4) lit_by_light[li] = intensity;
                V3 shade = ec * (m.diffuse * sum_d) + lc * (m.specular * sum_s);
                surface = surface + ec * m.ambient + shade * (l.inv_samples * intensity);
            }
            acc = acc + cur.w * surface;
            if (debug)
                std::printf("  [sim] node=%u depth=%u t=%.9g prim=%u type=%u p=(%.9g %.9g %.9g) n=(%.9g %.9g %.9g) inside=%d surf=(%.9g %.9g %.9g) w=(%.6g %.6g %.6g)\n",
                            cur.node, cur.depth, hit_t, hit_prim, p.type, h.p.x, h.p.y, h.p.z, h.n.x, h.n.y, h.n.z, int(h.inside), surface.x,
                            surface.y, surface.z, cur.w.x, cur.w.y, cur.w.z);

            // children (world.rs:59-69,94-133); the per-light duplication becomes a factor n_lights
            bool want_refl = !(fabsf(m.reflectivity) <= kEps) && cur.depth > 0;
            bool want_refr = !(fabsf(m.transparency) <= kEps) && cur.depth > 0;
            float n1 = 1.0f, n2 = 1.0f;
            uint32_t n1_prim = kNoPrim, n2_prim = kNoPrim;
            bool use_schlick = (m.reflectivity * m.transparency) > 0.0f;
            if (cur.depth > 0 && m.transparency > 0.0f) containers(s, cur.ray, hit_prim, hit_t, m.ior, n1, n2, n1_prim, n2_prim, refined, t64, cur.o64, cur.d64, fabsf(dot(h.n, cur.ray.d)));
            if (debug) std::printf("        n1=%.6g n2=%.6g\n", n1, n2);
            V3 rdir = v3(0, 0, 0);
            D3 rdir64 = d3(0, 0, 0);
            if (refined) {  // world.rs:109-130 in f64, with the f64 indices of the records the container pass named
                const double x1 = n1_prim == kNoPrim ? 1.0 : s.ps.xior[s.ps.recs[n1_prim].material];
                const double x2 = n2_prim == kNoPrim ? 1.0 : s.ps.xior[s.ps.recs[n2_prim].material];
                if (want_refr && !xrefract(xh, x1, x2, rdir64)) want_refr = false;
                rdir = to_v3(rdir64);
            } else if (want_refr && !refract_dir(h.eye, h.n, n1, n2, rdir)) {
                want_refr = false;
            }
            if (trace) {  // census: the same 24-double event the oracle records (rt_oracle.hpp kTraceEventDoubles)
                const double ev[24] = {double(cur.node), hit_t, h.p.x, h.p.y, h.p.z, h.n.x, h.n.y, h.n.z, h.inside ? 1.0 : 0.0, n1, n2,
                                       pc.x, pc.y, pc.z, lit_by_light[0], lit_by_light[1], lit_by_light[2], lit_by_light[3],
                                       want_refl ? 1.0 : 0.0, want_refr ? 1.0 : 0.0, double(p.type), refined ? 1.0 : 0.0, 0, 0};
                trace->insert(trace->end(), ev, ev + 24);
            }
            float fr = use_schlick ? schlick(h.eye, h.n, n1, n2) : 0.0f;
            float k_refl = nl * m.reflectivity * (use_schlick ? fr : 1.0f);
            float k_refr = nl * m.transparency * (use_schlick ? (1.0f - fr) : 1.0f);
            PathNode refl, refr;
            if (want_refl) {
                refl.ray.o = over, refl.ray.d = cur.ray.d - h.n * 2.0f * dot(cur.ray.d, h.n);
                refl.o64 = to_d3(refl.ray.o), refl.d64 = to_d3(refl.ray.d);
                if (refined) refl.o64 = xh.over, refl.d64 = xh.reflectv, refl.ray.d = to_v3(xh.reflectv);
                refl.ray.self = hit_prim, refl.ray.q = h.q, refl.ray.dq = h.dq;
                refl.w = cur.w * k_refl, refl.depth = cur.depth - 1, refl.node = 2 * cur.node + 1;
            }
            if (want_refr) {
                refr.ray.o = h.p - h.n * kEps, refr.ray.d = rdir;
                refr.o64 = to_d3(refr.ray.o), refr.d64 = to_d3(refr.ray.d);
                if (refined) refr.o64 = xh.under, refr.d64 = rdir64, refr.ray.o = to_v3(xh.under);
                refr.ray.self = hit_prim, refr.ray.q = h.q, refr.ray.dq = neg(h.dq);
                refr.w = cur.w * k_refr, refr.depth = cur.depth - 1, refr.node = 2 * cur.node + 2;
            }
            if (want_refl && want_refr) {
                stack[sp++] = refr;
                cur = refl;
                alive = true;
            } else if (want_refl) {
                cur = refl;
                alive = true;
            } else if (want_refr) {
                cur = refr;
                alive = true;
            }
        }
        if (!alive) {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return acc;
}
}  // namespace

extern "C" int hostsim_render(const rt_scene_desc* desc, const rt_camera_desc* camera, const rt_options* options,
                              float* rgb_out, char* err_out, int err_cap) {
    Sim s;
    std::string err;
    int rc = pack_scene(desc, s.ps, err);
    if (rc != RT_OK) {
        if (err_out && err_cap > 0) std::snprintf(err_out, static_cast<size_t>(err_cap), "%s", err.c_str());
        return rc;
    }
    s.cam = pack_camera(*camera);
    s.cam64 = pack_camera64(*camera);
    s.exact = s.ps.has_f64 && camera->has_f64 != 0 && !(options && (options->flags & RT_FLAG_F32_ONLY));
    options_max_depth(options, s.max_depth);
    s.jitter_mode = options ? options->jitter_mode : 0u;
    s.seed = options ? options->seed : 0u;
    const uint32_t W = camera->hsize, H = camera->vsize;
    if (const char* dbg = std::getenv("HOSTSIM_DEBUG_PIXEL")) {
        unsigned dx = 0, dy = 0;
        if (std::sscanf(dbg, "%u,%u", &dx, &dy) == 2 && dx < W && dy < H) trace_pixel(s, dx, dy, true);
    }
    // Row bands exactly as rt_options documents them (tile_rows == 0: the whole image).
    std::vector<uint32_t> rows;  // global row of each packed output row
    {
        const uint32_t tr = (options && options->tile_rows) ? options->tile_rows : H;
        const uint32_t first = (options && options->tile_rows) ? options->tile_first : 0u;
        const uint32_t stride = (options && options->tile_rows) ? (options->tile_stride ? options->tile_stride : 1u) : 1u;
        for (uint64_t b = first; b * tr < H; b += stride)
            for (uint64_t y = b * tr; y < (b + 1) * tr && y < H; ++y) rows.push_back(static_cast<uint32_t>(y));
    }
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 4;
    if (nthreads > 8) nthreads = 8;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t] {
            for (size_t r = t; r < rows.size(); r += nthreads)
                for (uint32_t x = 0; x < W; ++x) {
                    V3 c = trace_pixel(s, x, rows[r]);
                    float* o = rgb_out + (r * W + x) * 3;
                    o[0] = c.x, o[1] = c.y, o[2] = c.z;
                }
        });
    for (auto& th : pool) th.join();
    return RT_OK;
}

// Path traces of n pixels for the parity census (same event layout as oracle_trace_pixels).
extern "C" int hostsim_trace_pixels(const rt_scene_desc* desc, const rt_camera_desc* camera, const rt_options* options, const int* xy,
                                    int n, double* events, int cap, int* counts, float* rgb) {
    Sim s;
    std::string err;
    int rc = pack_scene(desc, s.ps, err);
    if (rc != RT_OK) return rc;
    s.cam = pack_camera(*camera);
    s.cam64 = pack_camera64(*camera);
    s.exact = s.ps.has_f64 && camera->has_f64 != 0 && !(options && (options->flags & RT_FLAG_F32_ONLY));
    options_max_depth(options, s.max_depth);
    s.jitter_mode = options ? options->jitter_mode : 0u;
    s.seed = options ? options->seed : 0u;
    std::vector<double> tr;
    for (int i = 0; i < n; ++i) {
        tr.clear();
        const V3 c = trace_pixel(s, static_cast<uint32_t>(xy[2 * i]), static_cast<uint32_t>(xy[2 * i + 1]), false, &tr);
        if (rgb) rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
        int stored = static_cast<int>(tr.size() / 24);
        if (stored > cap) stored = cap;
        std::memcpy(events + static_cast<size_t>(i) * cap * 24, tr.data(), static_cast<size_t>(stored) * 24 * sizeof(double));
        counts[i] = stored;
    }
    return RT_OK;
}

// Renders like hostsim_render and counts how often the widened tie window of rt_math.h decides a nearest
// hit differently from the reference's fixed 1e-5 window: out[0] = hits, out[1] = decided differently,
// out[2] = hits whose window is wider than 1e-5 (t > 4).
extern "C" int hostsim_tie_audit(const rt_scene_desc* desc, const rt_camera_desc* camera, const rt_options* options, float* rgb_out,
                                 uint64_t* out, char* err_out, int err_cap) {
    g_tie_hits = 0, g_tie_differs = 0, g_tie_wide_used = 0;
    g_tie_audit = true;
    const int rc = hostsim_render(desc, camera, options, rgb_out, err_out, err_cap);
    g_tie_audit = false;
    out[0] = g_tie_hits, out[1] = g_tie_differs, out[2] = g_tie_wide_used;
    return rc;
}

// Shadow rays of the last renders with the exact chain: out[0] = cast, out[2] = primitives f64 was asked about,
// out[3] = (node, light) pairs counted again, out[5] / out[6] = of those: because of a quirk-edge hit / a near thing
// Experiments and tests: move every rcp / sqrt / rsq of the spec by -1, 0 or +1 ulp (a hash of the argument and `seed`
// chooses; 0 = exact again) — the freedom the hardware's v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 have.  What the near-thing bands
// of rt_math.h are for: no decision may hang on it.
extern "C" void hostsim_set_ulp_noise(unsigned seed) { rtk::rt_host_ulp_noise = seed; }

extern "C" void hostsim_shadow_stats(uint64_t* out, int reset) {
    out[0] = g_sh_rays, out[1] = g_sh_unsure, out[2] = g_sh_f64, out[3] = g_sh_redo;
    out[4] = 0, out[5] = g_sh_why_qedge, out[6] = g_sh_why_near;
    if (reset) g_sh_rays = 0, g_sh_unsure = 0, g_sh_f64 = 0, g_sh_redo = 0, g_sh_why_qedge = 0, g_sh_why_near = 0;
}

// Renders like hostsim_render and reports what the exact chain did: out[0] = path nodes, out[1] = nodes refined
// in f64, out[2..5] = of those: because of children / a pattern boundary / an ambiguous search / an edge
// (reasons overlap), out[6] = nodes whose primitive changed, out[7] = searches repeated without the candidates.
extern "C" int hostsim_exact_stats(const rt_scene_desc* desc, const rt_camera_desc* camera, const rt_options* options, float* rgb_out,
                                   uint64_t* out, char* err_out, int err_cap) {
    g_x_nodes = 0, g_x_refined = 0, g_x_children = 0, g_x_pattern = 0, g_x_amb = 0, g_x_edge = 0, g_x_changed = 0, g_x_retry = 0;
    const int rc = hostsim_render(desc, camera, options, rgb_out, err_out, err_cap);
    out[0] = g_x_nodes, out[1] = g_x_refined, out[2] = g_x_children, out[3] = g_x_pattern, out[4] = g_x_amb, out[5] = g_x_edge;
    out[6] = g_x_changed, out[7] = g_x_retry;
    return rc;
}

// Renders like hostsim_render with the shaft audit switched on.  out[0] = (point, area light) pairs
// audited, out[1] = primitives ruled out in total, out[2] = pairs with every primitive ruled out,
// out[3] = violations (both kinds), out[4] = pairs declared fully occluded, out[5] = rays of those found unoccluded,
// out[6] = pairs whose cells are all declared far and in front, out[7] = pairs declared free of highlights,
// out[8] = cells contradicting either declaration (or the one of out[9]), out[9] = pairs whose light is
// declared wholly behind the surface.
extern "C" int hostsim_shaft_audit(const rt_scene_desc* desc, const rt_camera_desc* camera, const rt_options* options,
                                   float* rgb_out, uint64_t* out, char* err_out, int err_cap) {
    g_audit_pairs = 0, g_audit_ruled_out = 0, g_audit_open = 0, g_audit_violations = 0, g_audit_umbra = 0, g_audit_umbra_violations = 0;
    g_audit_front = 0, g_audit_dull = 0, g_audit_phong_violations = 0, g_audit_behind = 0;
    g_audit = true;
    const int rc = hostsim_render(desc, camera, options, rgb_out, err_out, err_cap);
    g_audit = false;
    out[0] = g_audit_pairs, out[1] = g_audit_ruled_out, out[2] = g_audit_open, out[3] = g_audit_violations;
    out[4] = g_audit_umbra, out[5] = g_audit_umbra_violations;
    out[6] = g_audit_front, out[7] = g_audit_dull, out[8] = g_audit_phong_violations, out[9] = g_audit_behind;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// pack_scene's walk streams as rt_scene_create keeps them on the device, laid out on the HOST (what the GPU makes must
// equal them byte for byte): a scene with a hierarchy: the eight direction orders, the canonical stream, the eight top
// regions (kTopK records each); any other scene: the canonical stream.  out may be null to ask for the size.
extern "C" int hostsim_pack_streams(const rt_scene_desc* desc, int staged_top, void* out, size_t cap, size_t* bytes_out) {
    PackedScene ps;
    std::string err;
    const int rc = pack_scene(desc, ps, err, /*host_streams=*/true, staged_top != 0);
    if (rc != RT_OK) return rc;
    const size_t bytes = (ps.lstreams.size() + ps.lnodes.size() + (ps.has_bvh ? ps.top.size() : 0)) * sizeof(LNode);
    if (bytes_out) *bytes_out = bytes;
    if (out) {
        if (cap < bytes) return RT_ERR_CAPACITY;
        char* o = static_cast<char*>(out);
        std::memcpy(o, ps.lstreams.data(), ps.lstreams.size() * sizeof(LNode));
        o += ps.lstreams.size() * sizeof(LNode);
        std::memcpy(o, ps.lnodes.data(), ps.lnodes.size() * sizeof(LNode));
        o += ps.lnodes.size() * sizeof(LNode);
        if (ps.has_bvh) std::memcpy(o, ps.top.data(), ps.top.size() * sizeof(LNode));
    }
    return RT_OK;
}

// The direction-ordered compact streams of rt_pack.h and their staged tops (lane-mode walks of rt_kernel.h): structure of
// all eight, and a closest-hit query the way traverse_lane walks — from the top region of the ray's own order, through
// the portals into that order's stream — against the same query through the canonical stream, on random rays.
// out[0] = structural errors, out[1] = rays whose two answers differ, out[2] = records looked at in canonical order,
// out[3] = records looked at in the ray's own order (both summed over the rays), out[4] = n_items, out[5] = rays that
// hit something, out[6] = records of out[3] that were read from a top region, out[7] = records in use per top region.
extern "C" int hostsim_lane_streams_check(const rt_scene_desc* desc, int staged_top, int n_rays, uint32_t seed, const float* box6, uint64_t* out,
                                          char* err_out, int err_cap) {
    PackedScene ps;
    std::string err;
    const int rc = pack_scene(desc, ps, err, /*host_streams=*/true, staged_top != 0);
    if (rc != RT_OK) {
        if (err_out && err_cap > 0) std::snprintf(err_out, static_cast<size_t>(err_cap), "%s", err.c_str());
        return rc;
    }
    const uint32_t n = ps.n_items;
    const uint32_t streams = ps.has_bvh ? 8u : 1u;
    uint64_t bad = 0;
    if (ps.lnodes.size() != std::max<size_t>(ps.recs.size(), 1)) bad++;
    if (ps.has_bvh && ps.lstreams.size() != 8 * ps.lnodes.size()) bad++;
    if (ps.top.size() != 8 * kTopK || ps.top_k > kTopK) bad++;
    auto stream_of = [&](uint32_t k) { return ps.has_bvh ? ps.lstreams.data() + static_cast<size_t>(k) * n : ps.lnodes.data(); };
    auto fetch = [&](uint32_t k, uint32_t i, bool& ok) {  // index space of rt_math.h: < n memory, n end, > n top region
        ok = i < n || (i > n && i - n - 1u < ps.top_k);
        if (!ok) return LNode{};
        return i < n ? stream_of(k)[i] : ps.top[k * kTopK + (i - n - 1u)];
    };
    for (uint32_t k = 0; k < streams && !bad; ++k) {
        // entering everything: every primitive record exactly once, and the walk ends; entering nothing: it ends too
        for (int enter = 1; enter >= 0 && !bad; --enter) {
            std::vector<uint8_t> seen(n, 0);
            uint32_t i = ps.top_k ? n + 1u : 0u;
            uint64_t steps = 0;
            while (i != n && !bad) {
                bool ok;
                const LNode nd = fetch(k, i, ok);
                if (!ok || ++steps > 4ull * n + 64) {
                    bad++;
                    break;
                }
                const uint32_t type = nd.ref >> kLRefShift, ref = nd.ref & kLRefMask;
                if (type == REC_NODE) i = enter ? i + 1u : nd.next;
                else if (type == REC_PORTAL) i = enter ? ref : nd.next;
                else {
                    if (ref >= n || seen[ref] || ps.recs[ref].type != type) bad++;
                    else seen[ref] = 1;
                    i = nd.next;
                }
            }
            if (enter)
                for (uint32_t r = 0; r < n; ++r)
                    if (ps.recs[r].type != REC_NODE && !seen[r]) bad++;
        }
    }
    out[0] = bad, out[1] = out[2] = out[3] = out[5] = 0, out[4] = n;
    if (bad) return RT_OK;
    uint32_t state = seed * 2654435761u + 12345u;
    auto rnd = [&] {
        state = state * 1664525u + 1013904223u;
        return static_cast<float>(state >> 8) * (1.0f / 16777216.0f);
    };
    out[6] = 0, out[7] = ps.top_k;
    // k < 0: the canonical stream from its first record; else order k from its top region (traverse_lane, rt_kernel.h)
    auto query = [&](int k, const Ray& ray, uint64_t& looked, uint32_t& prim_out, float& t_out) {
        const V3 inv_d = inv_dir(ray.d);
        prim_out = kNoPrim, t_out = INFINITY;
        uint32_t i = (k >= 0 && ps.top_k) ? n + 1u : 0u;
        while (i != n) {
            bool ok;
            LNode nd;
            if (k < 0) nd = ps.lnodes[i];
            else nd = fetch(static_cast<uint32_t>(k), i, ok);
            ++looked;
            if (k >= 0 && i > n) out[6]++;
            const bool in = bvh_node_hit(nd, ray.o, inv_d, 0.0f, t_out + tie_window(t_out));
            const uint32_t type = nd.ref >> kLRefShift;
            if (type == REC_NODE) {
                i = in ? i + 1u : nd.next;
                continue;
            }
            if (type == REC_PORTAL) {
                i = in ? (nd.ref & kLRefMask) : nd.next;
                continue;
            }
            i = nd.next;
            if (!in) continue;
            const uint32_t prim = nd.ref & kLRefMask;
            intersect_prim(ps.recs[prim], ps.extras.data(), ray, false, [&](int, float t, float, float) {
                bool better = closer_hit(t, t_out);
                if (!better && prim_out != kNoPrim && ties_hit(t, t_out)) better = ps.order[prim] < ps.order[prim_out];
                if (better) prim_out = prim, t_out = t;
            });
        }
    };
    for (int r = 0; r < n_rays; ++r) {
        Ray ray;
        ray.o = v3(box6[0] + rnd() * (box6[3] - box6[0]), box6[1] + rnd() * (box6[4] - box6[1]), box6[2] + rnd() * (box6[5] - box6[2]));
        V3 d = v3(rnd() * 2.0f - 1.0f, rnd() * 2.0f - 1.0f, rnd() * 2.0f - 1.0f);
        const float len = std::sqrt(dot(d, d));
        if (!(len > 1e-3f)) continue;
        ray.d = d * (1.0f / len);
        ray.self = kNoPrim, ray.q = v3(0, 0, 0), ray.dq = v3(0, 0, 0);
        uint32_t pa, pb;
        float ta, tb;
        query(-1, ray, out[2], pa, ta);
        const uint32_t oct = (ray.d.x < 0.0f ? 1u : 0u) | (ray.d.y < 0.0f ? 2u : 0u) | (ray.d.z < 0.0f ? 4u : 0u);
        query(static_cast<int>(streams == 8 ? oct : 0), ray, out[3], pb, tb);
        if (pa != kNoPrim) out[5]++;
        // the two orders may settle a tie inside the tie window differently; anything else is an error
        if (pa != pb && !(pa != kNoPrim && pb != kNoPrim && ties_hit(ta, tb))) out[1]++;
        else if (pa == pb && pa != kNoPrim && ta != tb) out[1]++;
    }
    return RT_OK;
}
