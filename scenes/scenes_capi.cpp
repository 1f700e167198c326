// scenes_capi.cpp — the example scenes built through the PRODUCT host API (include/raytracer_amd/
// raytracer.hpp, namespace rt) and exposed to Python (ctypes) for bench.py and the parity tests:
//   * rtx_scene_build / rtx_scene_desc / rtx_camera_desc hand out the flattened rt_scene_desc so the
//     tests can drive the C ABI of librt_amd.so directly;
//   * rtx_render_example is `camera.render(&world)` of examples/*.rs:  World -> flatten -> C ABI.
#include <cstdint>
#include <cstdio>
#include <string>

#include "../include/raytracer_amd/png.hpp"
#include "../include/raytracer_amd/raytracer.hpp"

#define RT_API_NS rt
#include "examples.inc"

namespace {
struct Built {
    rt::FlatScene flat;
    rt_scene_desc scene_desc;
    rt_camera_desc camera_desc;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

const char* rtx_last_error() { return g_err.c_str(); }

void* rtx_scene_build(const char* name, int width, int height) {
    try {
        rt::Scene scene;
        if (!name || !rt::examples::build(name, static_cast<size_t>(width), static_cast<size_t>(height), scene)) {
            g_err = std::string("unknown scene: ") + (name ? name : "(null)");
            return nullptr;
        }
        Built* b = new Built;
        b->flat = rt::flatten(scene.world);
        b->scene_desc = b->flat.desc();
        b->camera_desc = scene.camera.desc();
        return b;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
const rt_scene_desc* rtx_scene_desc(void* h) { return h ? &static_cast<Built*>(h)->scene_desc : nullptr; }
const rt_camera_desc* rtx_camera_desc(void* h) { return h ? &static_cast<Built*>(h)->camera_desc : nullptr; }
void rtx_scene_free(void* h) { delete static_cast<Built*>(h); }

// camera.render(&world) through the host API.  rgb: rows*width*3 floats (rows from rt_output_rows).
int rtx_render_example(const char* name, int width, int height, const rt_options* options, int device, float* rgb,
                       rt_stats* stats) {
    try {
        rt::Scene scene;
        if (!name || !rt::examples::build(name, static_cast<size_t>(width), static_cast<size_t>(height), scene)) {
            g_err = std::string("unknown scene: ") + (name ? name : "(null)");
            return RT_ERR_BAD_ARGUMENT;
        }
        rt::Canvas canvas = scene.camera.render(scene.world, options, device, stats);
        std::copy(canvas.rgb.begin(), canvas.rgb.end(), rgb);
        return RT_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RT_ERR_BAD_ARGUMENT;
    }
}

// `camera.render(&world).to_image()` in one call (Camera::render_image): rgb8 = rows*width*3 bytes.
int rtx_render_example_u8(const char* name, int width, int height, const rt_options* options, int device, uint8_t* rgb8,
                          rt_stats* stats) {
    try {
        rt::Scene scene;
        if (!name || !rgb8 || !rt::examples::build(name, static_cast<size_t>(width), static_cast<size_t>(height), scene)) {
            g_err = std::string("unknown scene: ") + (name ? name : "(null)");
            return RT_ERR_BAD_ARGUMENT;
        }
        const std::vector<uint8_t> bytes = scene.camera.render_image(scene.world, options, device, stats);
        std::copy(bytes.begin(), bytes.end(), rgb8);
        return RT_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RT_ERR_BAD_ARGUMENT;
    }
}

// `canvas.to_image().save(path)` of examples/*.rs: render, quantise on the device, write a PNG.
int rtx_render_example_png(const char* name, int width, int height, const rt_options* options, int device, const char* path) {
    try {
        rt::Scene scene;
        if (!name || !path || !rt::examples::build(name, static_cast<size_t>(width), static_cast<size_t>(height), scene)) {
            g_err = std::string("unknown scene: ") + (name ? name : "(null)");
            return RT_ERR_BAD_ARGUMENT;
        }
        const std::vector<uint8_t> bytes = scene.camera.render_image(scene.world, options, device, nullptr);
        rt::save_png(path, bytes.data(), static_cast<size_t>(width), bytes.size() / (static_cast<size_t>(width) * 3));
        return RT_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RT_ERR_BAD_ARGUMENT;
    }
}

// PNG encoder alone (host only): out may be null to query the size.
int rtx_png_encode(const uint8_t* rgb, int width, int height, uint8_t* out, size_t cap, size_t* len) {
    try {
        const std::vector<uint8_t> png = rt::encode_png(rgb, static_cast<size_t>(width), static_cast<size_t>(height));
        if (len) *len = png.size();
        if (out) {
            if (cap < png.size()) {
                g_err = "output buffer too small";
                return RT_ERR_CAPACITY;
            }
            std::copy(png.begin(), png.end(), out);
        }
        return RT_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RT_ERR_BAD_ARGUMENT;
    }
}

int rtx_scene_count() { return static_cast<int>(rt::examples::names().size()); }
const char* rtx_scene_name(int i) {
    const auto& n = rt::examples::names();
    return (i >= 0 && i < static_cast<int>(n.size())) ? n[static_cast<size_t>(i)].c_str() : nullptr;
}

}  // extern "C"
