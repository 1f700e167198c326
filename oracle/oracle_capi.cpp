// oracle_capi.cpp — C entry points over oracle/rt_oracle.hpp for tests/, smoke() and the
// cpu_baseline leg of bench.py (ctypes).  TEST INFRASTRUCTURE ONLY — never linked into or
// called by the product library (raytracer_amd/_lib/librt_amd.so).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "rt_oracle.hpp"

#define RT_API_NS oracle
#include "../scenes/examples.inc"

using namespace oracle;

namespace {
thread_local std::string g_error;
int fail(const std::string& msg) {
    g_error = msg;
    return -1;
}
void stats_to_array(const RayStats& s, uint64_t* out) {
    out[0] = s.primary, out[1] = s.reflect, out[2] = s.refract, out[3] = s.shadow;
    out[4] = s.shaded_hits, out[5] = s.light_cells, out[6] = s.pattern_lookups;
    for (int i = 0; i < 7; ++i) out[7 + i] = s.prim_tests[i];
}
}  // namespace

extern "C" {

const char* oracle_last_error() { return g_error.c_str(); }

// Renders rows [row_begin, row_end) (row_end == 0 => all rows) of the named example scene at
// width x height into rgb[height*width*3] (f64, row-major, row 0 = top; rows outside the range
// are left untouched).  jitter_mode: 0 centre (0.5), 1 counter RNG (shared with the HIP kernel),
// 2 independent thread RNG (reference-like u8/255 stream).  stats_out: 14 u64 or null:
// primary, reflect, refract, shadow rays (reference-equivalent counts, i.e. including the
// per-light duplicated recursion), shaded hits, light cells, pattern lookups, then primitive
// tests by Shape kind {cube, cylinder, group(box), plane, smooth triangle, sphere, triangle}.
// seconds_out: wall time of Camera::render only (scene construction excluded).
// depth: the `remaining` argument World::color_at starts with (world.rs:24); the reference's Camera::render passes
// RECURSION_DEPTH = 5 (camera.rs:193, world.rs:11); negative = that default.
int oracle_render_example_depth(const char* name, int width, int height, int jitter_mode, uint32_t seed, int threads,
                                int row_begin, int row_end, double* rgb, uint64_t* stats_out, double* seconds_out, int depth) {
    try {
        if (depth > 255) return fail("depth out of range");
        if (!name || !rgb || width <= 0 || height <= 0) return fail("bad argument");
        Scene scene;
        if (!examples::build(name, static_cast<size_t>(width), static_cast<size_t>(height), scene))
            return fail(std::string("unknown scene: ") + name);
        RayStats stats;
        RenderOptions opt;
        opt.jitter = static_cast<JitterMode>(jitter_mode);
        opt.seed = seed;
        opt.threads = threads > 0 ? static_cast<unsigned>(threads) : 0u;
        opt.row_begin = static_cast<size_t>(row_begin);
        opt.row_end = static_cast<size_t>(row_end);
        opt.stats = stats_out ? &stats : nullptr;
        if (depth >= 0) opt.depth = static_cast<uint8_t>(depth);
        auto t0 = std::chrono::steady_clock::now();
        Canvas img = scene.camera.render(scene.world, opt);
        auto t1 = std::chrono::steady_clock::now();
        if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
        size_t y0 = opt.row_begin, y1 = opt.row_end ? opt.row_end : static_cast<size_t>(height);
        for (size_t y = y0; y < y1; ++y)
            for (size_t x = 0; x < static_cast<size_t>(width); ++x) {
                const Color& c = img.pixel_at(x, y);
                double* p = rgb + (y * static_cast<size_t>(width) + x) * 3;
                p[0] = c.red, p[1] = c.green, p[2] = c.blue;
            }
        if (stats_out) stats_to_array(stats, stats_out);
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

int oracle_render_example(const char* name, int width, int height, int jitter_mode, uint32_t seed, int threads,
                          int row_begin, int row_end, double* rgb, uint64_t* stats_out, double* seconds_out) {
    return oracle_render_example_depth(name, width, height, jitter_mode, seed, threads, row_begin, row_end, rgb, stats_out, seconds_out, -1);
}

// Colours of n individual pixels (xy = n pairs) of the scene at full resolution width x height.
int oracle_probe_pixels(const char* name, int width, int height, int jitter_mode, uint32_t seed, const int* xy, int n,
                        double* rgb) {
    try {
        Scene scene;
        if (!examples::build(name, static_cast<size_t>(width), static_cast<size_t>(height), scene))
            return fail(std::string("unknown scene: ") + name);
        RenderOptions opt;
        opt.jitter = static_cast<JitterMode>(jitter_mode);
        opt.seed = seed;
        std::mt19937 rng(seed + 1u);
        TraceCtx ctx;
        ctx.mode = opt.jitter;
        ctx.rng = &rng;
        tls_debug = std::getenv("ORACLE_DEBUG") != nullptr;
        for (int i = 0; i < n; ++i) {
            Color c = scene.camera.color_for_pixel(scene.world, static_cast<size_t>(xy[2 * i]),
                                                   static_cast<size_t>(xy[2 * i + 1]), opt, ctx);
            rgb[3 * i] = c.red, rgb[3 * i + 1] = c.green, rgb[3 * i + 2] = c.blue;
        }
        tls_debug = false;
        return 0;
    } catch (const std::exception& e) {
        return fail(e.what());
    }
}

// Path traces of n pixels (parity census): events[i * cap * 24 ...] receives up to `cap` events of pixel i
// (rt_oracle.hpp kTraceEventDoubles; repeated visits of a node by the per-light re-tracing are dropped),
// counts[i] the number stored.
int oracle_trace_pixels(const char* name, int width, int height, int jitter_mode, uint32_t seed, const int* xy, int n,
                        double* events, int cap, int* counts) {
    try {
        Scene scene;
        if (!examples::build(name, static_cast<size_t>(width), static_cast<size_t>(height), scene))
            return fail(std::string("unknown scene: ") + name);
        RenderOptions opt;
        opt.jitter = static_cast<JitterMode>(jitter_mode);
        opt.seed = seed;
        std::mt19937 rng(seed + 1u);
        TraceCtx ctx;
        ctx.mode = opt.jitter;
        ctx.rng = &rng;
        std::vector<f64> tr;
        for (int i = 0; i < n; ++i) {
            tr.clear();
            tls_trace = &tr;
            scene.camera.color_for_pixel(scene.world, static_cast<size_t>(xy[2 * i]), static_cast<size_t>(xy[2 * i + 1]), opt, ctx);
            tls_trace = nullptr;
            int stored = 0;
            std::vector<uint32_t> seen;
            for (size_t e = 0; e + kTraceEventDoubles <= tr.size() && stored < cap; e += kTraceEventDoubles) {
                const uint32_t node = static_cast<uint32_t>(tr[e]);
                bool dup = false;
                for (uint32_t s : seen) dup |= s == node;
                if (dup) continue;
                seen.push_back(node);
                std::memcpy(events + (static_cast<size_t>(i) * cap + stored) * kTraceEventDoubles, &tr[e], kTraceEventDoubles * sizeof(double));
                ++stored;
            }
            counts[i] = stored;
        }
        return 0;
    } catch (const std::exception& e) {
        tls_trace = nullptr;
        return fail(e.what());
    }
}

// canvas.rs:31-45 quantiser on a flat f64 buffer.
void oracle_quantize_u8(const double* rgb, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) out[i] = quantize_u8(rgb[i]);
}

// The jitter hash, exposed so tests can pin host/kernel/oracle agreement on it.
double oracle_counter_jitter(uint32_t seed, uint32_t pixel, uint32_t node, uint32_t light, uint32_t cell, uint32_t axis) {
    return counter_jitter(jitter_pixel_key(seed, pixel), node, light, cell, axis);
}

int oracle_scene_count() { return static_cast<int>(examples::names().size()); }
const char* oracle_scene_name(int i) {
    const auto& n = examples::names();
    return (i >= 0 && i < static_cast<int>(n.size())) ? n[static_cast<size_t>(i)].c_str() : nullptr;
}

}  // extern "C"
