// Times pack_scene (rt_pack.h) on a triangle soup like that of tests/helpers.py (own RNG); phases with -DRT_PACK_TIMING.
//   g++ -std=c++17 -O2 -pthread -ffp-contract=off -DRT_PACK_TIMING scripts/pack_time.cpp -o /tmp/pack_time && /tmp/pack_time 1000000 [host_streams 0|1]
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "../raytracer_amd/csrc/rt_pack.h"
int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 1000000;
    std::mt19937 rng(42);
    std::uniform_real_distribution<float> U(-6, 6);
    std::normal_distribution<float> N(0, 0.12f);
    std::vector<float> params((n + 1) * 24, 0.f), inv((n + 1) * 12, 0.f);
    std::vector<uint8_t> types(n + 1, 4);
    types[0] = 1;
    std::vector<uint32_t> mats(n + 1), ref(n + 1), nxt(n + 1);
    for (uint32_t i = 0; i <= n; ++i) {
        inv[12 * i] = inv[12 * i + 5] = inv[12 * i + 10] = 1.f;
        mats[i] = i % 3, ref[i] = i, nxt[i] = i + 1;
        if (i) {
            float* p = &params[24 * i];
            p[0] = U(rng), p[1] = U(rng) * 0.35f + 2.2f, p[2] = U(rng);
            for (int k = 3; k < 9; ++k) p[k] = N(rng);
            p[9] = 0, p[10] = 1, p[11] = 0;
        }
    }
    float mat_param[24] = {0.1f, 0.9f, 0.3f, 50, 1, 0, 0, 0, 0.1f, 0.7f, 0.2f, 10, 1, 0.2f, 0, 0, 0.2f, 0.8f, 0.9f, 300, 1, 0, 0, 0};
    uint32_t mat_pat[3] = {0, 1, 2};
    uint8_t pat_type[3] = {0, 0, 0};
    float pat_a[9] = {.9f, .9f, .9f, .8f, .3f, .2f, .2f, .5f, .9f}, pat_inv[36] = {0};
    for (int i = 0; i < 3; ++i) pat_inv[12 * i] = pat_inv[12 * i + 5] = pat_inv[12 * i + 10] = 1;
    uint8_t lt[1] = {0};
    float lp[3] = {-8, 12, 6}, lv[3] = {0, 0, 0}, lc[3] = {1, 1, 1};
    uint32_t ls[1] = {1};
    rt_scene_desc d{};
    d.abi_version = RT_ABI_VERSION;
    d.n_prims = n + 1, d.prim_type = types.data(), d.prim_inv = inv.data(), d.prim_material = mats.data(), d.prim_param = params.data();
    d.n_items = n + 1, d.item_ref = ref.data(), d.item_next = nxt.data();
    d.n_materials = 3, d.mat_param = mat_param, d.mat_pattern = mat_pat;
    d.n_patterns = 3, d.pat_type = pat_type, d.pat_color_a = pat_a, d.pat_color_b = pat_a, d.pat_inv = pat_inv;
    d.n_lights = 1, d.light_type = lt, d.light_pos = lp, d.light_uvec = lv, d.light_vvec = lv, d.light_usteps = ls, d.light_vsteps = ls, d.light_color = lc;
    for (int rep = 0; rep < 2; ++rep) {
        rtk::PackedScene ps;
        std::string err;
        auto t0 = std::chrono::steady_clock::now();
        int rc = rtk::pack_scene(&d, ps, err, argc > 2 ? atoi(argv[2]) != 0 : true);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t bytes) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull; };
        mix(ps.recs.data(), ps.recs.size() * sizeof(ps.recs[0]));
        mix(ps.lnodes.data(), ps.lnodes.size() * sizeof(ps.lnodes[0]));
        mix(ps.bound.data(), ps.bound.size() * sizeof(ps.bound[0]));
        mix(ps.order.data(), ps.order.size() * 4);
        printf("rc %d  %u triangles: pack_scene %.1f ms, %zu records, %zu lnodes, hash %016llx  [%s]\n", rc, n, ms, ps.recs.size(), ps.lnodes.size(), (unsigned long long)h, err.c_str());
#ifdef RT_PACK_TIMING
        for (int i = 0; i < 8; ++i) printf("   phase %d: %.1f ms\n", i, rtk::g_pack_marks[i + 1] - rtk::g_pack_marks[i]);
#endif
    }
}
