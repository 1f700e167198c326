// rt_pack.h — host side of rt_scene_create: validates an rt_scene_desc and repacks its SoA
// arrays into the device records of rt_math.h (array-of-64-byte-records: one scalar load per
// wave-uniform primitive, one cache line per divergent gather).  Pure C++ (no HIP), so the
// test-only host simulator can reuse exactly the same tables.
#pragma once

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_math.h"

namespace rtk {

struct PackedScene {
    std::vector<DPrim> recs;  // the traversal stream: one record per item, in item order
    std::vector<DExtra> extras;
    std::vector<DMaterial> mats;
    std::vector<DPattern> pats;
    std::vector<DLight> lights;
    uint32_t n_items = 0, n_lights = 0;  // real counts (the vectors are padded to be non-empty)
    uint32_t max_cells = 1;             // largest usteps*vsteps over the lights
};

// Widen a box bound outwards by a relative 4e-6 (+ a tiny absolute term): the slab test on the
// device runs in FP32, and a child whose intersection exists in f64 must never be culled because
// the box test rounded the other way.  Culling is purely an optimisation — the reference's own
// result is "every child whose line-box test passes", and boxes bound their children.
inline float widen(float v, bool up) {
    if (std::isinf(v) || std::isnan(v)) return v;
    float pad = std::fabs(v) * 4e-6f + 1e-6f;
    return up ? v + pad : v - pad;
}

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
    if (d->n_prims >= 0x7FFFFFFFu || d->n_items >= 0x7FFFFFFFu) {
        err = "too many primitives";
        return RT_ERR_CAPACITY;
    }

    out = PackedScene{};
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
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->mat_pattern[i] >= d->n_patterns) return bad("material pattern index out of range");
        const float* m = d->mat_param + 8 * i;
        out.mats.push_back(DMaterial{m[0], m[1], m[2], m[3], m[4], m[5], m[6], d->mat_pattern[i]});
    }
    std::vector<DPrim> prims;  // by primitive index; moved into stream order below
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
        }
        prims.push_back(p);
    }
    if (out.extras.empty()) out.extras.push_back(DExtra{});  // never hand a null table to the kernel
    // Item tree: pre-order with forward skip links.  i < next <= n_items and proper nesting
    // guarantee that every traversal (which only ever moves to i+1 or next[i]) terminates.
    {
        std::vector<uint32_t> open;  // ends of the enclosing subtrees
        std::vector<uint8_t> prim_seen(d->n_prims, 0);
        for (uint32_t i = 0; i < d->n_items; ++i) {
            while (!open.empty() && open.back() == i) open.pop_back();
            uint32_t ref = d->item_ref[i], next = d->item_next[i];
            if (next <= i || next > d->n_items) return bad("item_next out of range");
            if (!open.empty() && next > open.back()) return bad("item subtree crosses its parent's end");
            if (ref & RT_ITEM_NODE) {
                if ((ref & ~RT_ITEM_NODE) >= d->n_nodes) return bad("item node index out of range");
                if (next > i + 1) open.push_back(next);
            } else {
                if (ref >= d->n_prims) return bad("item primitive index out of range");
                if (next != i + 1) return bad("a primitive item must link to the next item");
                if (prim_seen[ref]) return bad("primitive referenced by more than one item");
                prim_seen[ref] = 1;
            }
            if (ref & RT_ITEM_NODE) {
                const uint32_t n = ref & ~RT_ITEM_NODE;
                DPrim r{};
                for (int k = 0; k < 3; ++k) {
                    r.inv[k] = widen(d->node_min[3 * n + k], false);
                    r.inv[3 + k] = widen(d->node_max[3 * n + k], true);
                }
                r.type = REC_NODE;
                r.material = next;
                out.recs.push_back(r);
            } else {
                out.recs.push_back(prims[ref]);
            }
        }
    }
    out.n_items = d->n_items;
    if (out.recs.empty()) out.recs.push_back(DPrim{});  // placeholder, n_items stays 0
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
    }
    out.n_lights = d->n_lights;
    if (out.lights.empty()) out.lights.push_back(DLight{});
    return RT_OK;
}

inline DCamera pack_camera(const rt_camera_desc& c) {
    DCamera d{};
    d.pixel_size = c.pixel_size, d.half_width = c.half_width, d.half_height = c.half_height;
    d.hsize = c.hsize, d.vsize = c.vsize;
    std::memcpy(d.inv, c.inv, sizeof(d.inv));
    return d;
}

}  // namespace rtk
