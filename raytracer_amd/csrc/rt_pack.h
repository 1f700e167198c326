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
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_exact.h"
#include "rt_math.h"

namespace rtk {

struct PackedScene {
    std::vector<DPrim> recs;      // the traversal stream
    std::vector<uint32_t> order;  // per record: the primitive's position in world.objects order (nodes: 0)
    std::vector<uint32_t> gate;   // per record: innermost enclosing Group (index into gates) or kNoGate
    std::vector<DBound> bound;    // per record: world-space bounding ball (shaft tests)
    std::vector<LNode> lnodes;    // per record: the compact twin of the stream (lane-mode walks)
    std::vector<DGate> gates;     // the caller's Groups
    std::vector<DExtra> extras;
    std::vector<DMaterial> mats;
    std::vector<DPattern> pats;
    std::vector<DLight> lights;
    // the caller's f64 tables for the exact chain (rt_exact.h), laid out like their FP32 twins: per stream
    // record, per DExtra slot, per material, per pattern.  Empty when the caller did not provide them.
    std::vector<XPrim> xprims;
    std::vector<XExtra> xextras;
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

// The device's own hierarchy over the bounded primitives: a binary tree built top-down with the surface-area
// heuristic (16 bins per axis over the centroids; expected cost of a split = area(left) * n(left) + area(right) *
// n(right); the median along the widest axis when the bins cannot separate anything), at most two primitives
// per leaf.  It is emitted once in canonical order (low side of every split first) into the record stream, and
// eight times — one per sign pattern of a ray direction — into the compact LNode streams (rt_math.h).
struct BvhBuild {
    struct TNode {
        HostBox box;
        int left = -1, right = -1;  // subtrees (indices into `tree`), or -1
        int axis = 0;               // split axis (inner nodes)
        uint32_t lo = 0, hi = 0;    // leaf: perm[lo, hi) (1 or 2 primitives); a single primitive has no box record
        uint32_t rec = 0;           // position of the node's record in the canonical stream (boxed nodes)
    };
    std::vector<HostBox> boxes;  // per bounded primitive (index into `ids`)
    std::vector<uint32_t> ids;   // item index of each bounded primitive
    std::vector<uint32_t> perm;  // working permutation
    std::vector<TNode> tree;
    std::vector<uint32_t> prim_rec;  // per bounded primitive k: its position in the canonical stream
    PackedScene* out = nullptr;
    const std::vector<DPrim>* item_rec = nullptr;  // record of each item (primitives only)
    const std::vector<uint32_t>* item_gate = nullptr;

    static double area(const HostBox& b) {
        const double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    int build(uint32_t lo, uint32_t hi, int depth = 0) {  // perm[lo, hi) -> index into `tree`
        const uint32_t n = hi - lo;
        TNode node;
        node.box = HostBox::empty();
        HostBox cen = HostBox::empty();
        for (uint32_t k = lo; k < hi; ++k) {
            const HostBox& b = boxes[perm[k]];
            node.box.merge(b);
            double c[3] = {0.5 * (b.mn[0] + b.mx[0]), 0.5 * (b.mn[1] + b.mx[1]), 0.5 * (b.mn[2] + b.mx[2])};
            cen.add(c);
        }
        node.lo = lo, node.hi = hi;
        if (n <= 2) {
            tree.push_back(node);
            return static_cast<int>(tree.size()) - 1;
        }
        constexpr int kBins = 16;
        int best_axis = -1, best_bin = 0;
        double best_cost = INFINITY;
        for (int axis = 0; axis < 3; ++axis) {
            const double c0 = cen.mn[axis], ext = cen.mx[axis] - cen.mn[axis];
            if (!(ext > 0.0) || !std::isfinite(ext)) continue;
            HostBox bin_box[kBins];
            uint32_t bin_n[kBins] = {0};
            for (auto& b : bin_box) b = HostBox::empty();
            for (uint32_t k = lo; k < hi; ++k) {
                const HostBox& b = boxes[perm[k]];
                int bi = static_cast<int>((0.5 * (b.mn[axis] + b.mx[axis]) - c0) / ext * kBins);
                bi = bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
                bin_box[bi].merge(b), bin_n[bi]++;
            }
            double right_area[kBins];
            uint32_t right_n[kBins];
            HostBox acc = HostBox::empty();
            uint32_t cnt = 0;
            for (int i = kBins - 1; i > 0; --i) {
                if (bin_n[i]) acc.merge(bin_box[i]);
                cnt += bin_n[i];
                right_area[i] = area(acc), right_n[i] = cnt;
            }
            acc = HostBox::empty(), cnt = 0;
            for (int i = 0; i + 1 < kBins; ++i) {  // split between bin i and bin i + 1
                if (bin_n[i]) acc.merge(bin_box[i]);
                cnt += bin_n[i];
                if (cnt == 0 || right_n[i + 1] == 0) continue;
                const double cost = area(acc) * cnt + right_area[i + 1] * right_n[i + 1];
                if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = i;
            }
        }
        uint32_t mid = lo + n / 2;
        if (best_axis >= 0 && depth < 40) {  // (deeper than that the splits are lopsided: halve instead)
            const int axis = best_axis;
            const double c0 = cen.mn[axis], ext = cen.mx[axis] - cen.mn[axis];
            auto bin_of = [&](uint32_t a) {
                int bi = static_cast<int>((0.5 * (boxes[a].mn[axis] + boxes[a].mx[axis]) - c0) / ext * kBins);
                return bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
            };
            auto it = std::stable_partition(perm.begin() + lo, perm.begin() + hi, [&](uint32_t a) { return bin_of(a) <= best_bin; });
            mid = static_cast<uint32_t>(it - perm.begin());
            node.axis = axis;
        }
        if (best_axis < 0 || depth >= 40 || mid == lo || mid == hi) {  // nothing to separate by: the median along the widest axis
            int axis = 0;
            double ext = -1.0;
            for (int k = 0; k < 3; ++k)
                if (cen.mx[k] - cen.mn[k] > ext) ext = cen.mx[k] - cen.mn[k], axis = k;
            mid = lo + n / 2;
            std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi, [&](uint32_t a, uint32_t b) {
                const double ca = boxes[a].mn[axis] + boxes[a].mx[axis], cb = boxes[b].mn[axis] + boxes[b].mx[axis];
                return ca < cb || (ca == cb && a < b);
            });
            node.axis = axis;
        }
        const int l = build(lo, mid, depth + 1);
        const int r = build(mid, hi, depth + 1);
        node.left = l, node.right = r;
        tree.push_back(node);
        return static_cast<int>(tree.size()) - 1;
    }
    void emit_prim(uint32_t k) {
        const uint32_t item = ids[k];
        prim_rec[k] = static_cast<uint32_t>(out->recs.size());
        out->recs.push_back((*item_rec)[item]);
        out->order.push_back(item);
        out->gate.push_back((*item_gate)[item]);
    }
    // canonical stream: pre-order, low side first
    void emit(int t) {
        TNode& nd = tree[t];
        if (nd.left < 0 && nd.hi - nd.lo == 1) {
            emit_prim(perm[nd.lo]);
            return;
        }
        const size_t at = out->recs.size();
        nd.rec = static_cast<uint32_t>(at);
        DPrim node{};
        for (int k = 0; k < 3; ++k) {
            node.inv[k] = widen(std::nextafter(static_cast<float>(nd.box.mn[k]), -INFINITY), false);
            node.inv[3 + k] = widen(std::nextafter(static_cast<float>(nd.box.mx[k]), INFINITY), true);
        }
        node.type = REC_NODE;
        out->recs.push_back(node);
        out->order.push_back(0);
        out->gate.push_back(kNoGate);
        if (nd.left < 0) {
            for (uint32_t k = nd.lo; k < nd.hi; ++k) emit_prim(perm[k]);
        } else {
            emit(nd.left);
            emit(nd.right);
        }
        out->recs[at].material = static_cast<uint32_t>(out->recs.size());  // skip link
    }
    // the compact stream of one direction octant (bit k of `oct` set: the ray's direction is negative along axis k)
    void emit_lane(int t, uint32_t oct, std::vector<LNode>& dst, const std::vector<LNode>& prim_node) const {
        const TNode& nd = tree[t];
        if (nd.left < 0 && nd.hi - nd.lo == 1) {
            LNode ln = prim_node[prim_rec[perm[nd.lo]]];
            ln.next = static_cast<uint32_t>(dst.size()) + 1u;
            dst.push_back(ln);
            return;
        }
        const size_t at = dst.size();
        LNode ln{};
        for (int k = 0; k < 6; ++k) ln.inv[k] = out->recs[nd.rec].inv[k];
        ln.ref = REC_NODE << kLRefShift;
        dst.push_back(ln);
        if (nd.left < 0) {
            for (uint32_t k = nd.lo; k < nd.hi; ++k) {
                LNode pl = prim_node[prim_rec[perm[k]]];
                pl.next = static_cast<uint32_t>(dst.size()) + 1u;
                dst.push_back(pl);
            }
        } else {
            const bool flip = ((oct >> nd.axis) & 1u) != 0;
            emit_lane(flip ? nd.right : nd.left, oct, dst, prim_node);
            emit_lane(flip ? nd.left : nd.right, oct, dst, prim_node);
        }
        dst[at].next = static_cast<uint32_t>(dst.size());
    }
};

// Returns RT_OK or a negative rt_status; `err` receives the message.
inline int pack_scene(const rt_scene_desc* d, PackedScene& out, std::string& err) {
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
    std::vector<DPrim> prims;  // by primitive index
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        if (d->prim_type[i] > RT_PRIM_SMOOTH_TRIANGLE) {
            err = "unknown primitive tag";
            return RT_ERR_UNSUPPORTED;
        }
        if (d->prim_material[i] >= d->n_materials) return bad("primitive material index out of range");
        DPrim p{};
        std::memcpy(p.inv, d->prim_inv + 12 * i, sizeof(p.inv));
        p.type = d->prim_type[i];
        p.material = d->prim_material[i];
        p.ior = out.mats[p.material].ior;
        p.extra = 0;
        if (p.type >= RT_PRIM_CYLINDER) {
            if (!d->prim_param) return bad("prim_param is null but a cylinder/triangle needs it");
            DExtra e{};
            std::memcpy(e.f, d->prim_param + 24 * i, 24 * sizeof(float));
            p.extra = static_cast<uint32_t>(out.extras.size());
            out.extras.push_back(e);
            if (out.has_f64) {
                XExtra xe{};
                std::memcpy(xe.f, d->prim_param64 + 24 * i, sizeof(xe.f));
                out.xextras.push_back(xe);
            }
        }
        prims.push_back(p);
    }
    if (out.extras.empty()) out.extras.push_back(DExtra{});  // never hand a null table to the kernel
    if (out.xextras.empty()) out.xextras.push_back(XExtra{});

    // The caller's item tree: pre-order with forward skip links.  i < next <= n_items and proper
    // nesting are checked; Groups become gates, primitives remember their innermost Group.
    std::vector<DPrim> item_rec(d->n_items);
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
            } else {
                if (ref >= d->n_prims) return bad("item primitive index out of range");
                if (next != i + 1) return bad("a primitive item must link to the next item");
                if (prim_seen[ref]) return bad("primitive referenced by more than one item");
                prim_seen[ref] = 1;
                item_rec[i] = prims[ref];
                item_gate[i] = enclosing;
                prim_items.push_back(i);
                out.exotic |= prims[ref].type >= PRIM_CYLINDER;
                out.has_gates |= enclosing != kNoGate;
            }
        }
    }
    if (out.gates.empty()) out.gates.push_back(DGate{});

    // Lay the stream out.
    auto emit_flat = [&](uint32_t item) {
        out.recs.push_back(item_rec[item]);
        out.order.push_back(item);
        out.gate.push_back(item_gate[item]);
    };
    BvhBuild b;
    int root = -1;
    uint32_t n_top = 0;
    if (prim_items.size() <= kFlatLimit && !out.has_gates) {
        for (uint32_t item : prim_items) emit_flat(item);
    } else {
        b.out = &out, b.item_rec = &item_rec, b.item_gate = &item_gate;
        for (uint32_t item : prim_items) {
            const uint32_t ref = d->item_ref[item];
            const HostBox box = prim_world_box(d->prim_type[ref], d->prim_inv + 12 * ref,
                                               d->prim_param ? d->prim_param + 24 * ref : nullptr);
            if (box.bounded) b.boxes.push_back(box), b.ids.push_back(item);
            else emit_flat(item);  // unbounded primitives stay at the top, in order
        }
        if (b.ids.size() <= 2) {
            for (uint32_t item : b.ids) emit_flat(item);
        } else {
            b.perm.resize(b.ids.size());
            b.prim_rec.assign(b.ids.size(), 0u);
            for (uint32_t k = 0; k < b.perm.size(); ++k) b.perm[k] = k;
            root = b.build(0, static_cast<uint32_t>(b.perm.size()));
            n_top = static_cast<uint32_t>(out.recs.size());  // the unbounded primitives in front of the tree
            b.emit(root);
            out.has_bvh = true;
        }
    }
    out.n_items = static_cast<uint32_t>(out.recs.size());
    if (out.recs.empty()) out.recs.push_back(DPrim{}), out.order.push_back(0), out.gate.push_back(kNoGate);
    for (const DPrim& r : out.recs)
        out.bound.push_back(r.type == REC_NODE ? node_ball(r) : prim_world_ball(r.type, r.inv, out.extras[r.extra].f));
    if (out.recs.size() >= (1u << kLRefShift)) {
        err = "too many records for the compact streams";
        return RT_ERR_CAPACITY;
    }
    {  // the compact twin of every record, canonical order (stream 0 of the lane-mode walks)
        std::vector<LNode> canon;
        for (size_t i = 0; i < out.recs.size(); ++i) {
            const DPrim& r = out.recs[i];
            LNode ln{};
            if (r.type == REC_NODE) {
                for (int k = 0; k < 6; ++k) ln.inv[k] = r.inv[k];
                ln.next = r.material;
                ln.ref = REC_NODE << kLRefShift;
            } else {
                const HostBox box = prim_world_box(r.type, r.inv, out.extras[r.extra].f);
                for (int k = 0; k < 3; ++k) {
                    ln.inv[k] = box.bounded ? widen(std::nextafter(static_cast<float>(box.mn[k]), -INFINITY), false) : -INFINITY;
                    ln.inv[3 + k] = box.bounded ? widen(std::nextafter(static_cast<float>(box.mx[k]), INFINITY), true) : INFINITY;
                }
                ln.next = static_cast<uint32_t>(i) + 1u;
                ln.ref = (r.type << kLRefShift) | static_cast<uint32_t>(i);
            }
            canon.push_back(ln);
        }
        out.lnodes = canon;
        if (out.has_bvh) {  // + the seven other direction orders of the same tree
            for (uint32_t oct = 1; oct < 8; ++oct) {
                std::vector<LNode> dst(canon.begin(), canon.begin() + n_top);
                b.emit_lane(root, oct, dst, canon);
                if (dst.size() != canon.size()) return bad("internal: direction-ordered stream of the wrong length");
                out.lnodes.insert(out.lnodes.end(), dst.begin(), dst.end());
            }
        }
    }
    if (out.has_f64) {  // record i of the stream is the primitive of item order[i]
        out.xprims.resize(out.recs.size());
        for (size_t i = 0; i < out.n_items; ++i) {
            if (out.recs[i].type == REC_NODE) continue;
            const uint32_t ref = d->item_ref[out.order[i]];
            std::memcpy(out.xprims[i].inv, d->prim_inv64 + 12 * static_cast<size_t>(ref), sizeof(out.xprims[i].inv));
        }
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
