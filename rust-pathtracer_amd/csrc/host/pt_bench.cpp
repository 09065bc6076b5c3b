// pt_bench — host driver with the reference's `--benchmark` entry point (src/main.rs:59-153) over the C ABI.
//
// It does what a Rust host linking libmi355_pt.so would do: build the scene on the host
// (libpt_host.so stands in for load_gltf + Scene::new — no glTF assets or Rust toolchain exist here, so the "model"
// is one of the seeded procedural scenes), upload it once, run the `for sample in 0..16` loop on the GPU and print
// the same lines as main.rs:77-152.  Defaults are the benchmark's: 640x360, 16 samples, PathTracer<8>, solid sky
// (0.5, 0.6, 0.8), no sun (main.rs:94-120).
//
//   pt_bench --benchmark [scene] [--seed N] [--detail F] [--app-lighting] [--width W --height H --samples S --bounces B]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_host.h"
#include "ptx_core.h"

static double now_s() {
    using namespace std::chrono;
    return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        const int32_t st_ = (call);                                                                  \
        if (st_ != PTX_OK) {                                                                         \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, st_, ptx_last_error(ctx));           \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv) {
    std::string scene_name = "bistro_like";   // main.rs:75 defaults to assets/models/bistro.glb
    uint64_t seed = 1;
    float detail = 1.0f;
    uint32_t flags = 0, width = 640, height = 360, samples = 16, bounces = 8;
    bool benchmark = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(2); }
            return argv[++i];
        };
        if (a == "--benchmark") benchmark = true;
        else if (a == "--seed") seed = std::strtoull(next("--seed"), nullptr, 10);
        else if (a == "--detail") detail = std::strtof(next("--detail"), nullptr);
        else if (a == "--width") width = (uint32_t)std::strtoul(next("--width"), nullptr, 10);
        else if (a == "--height") height = (uint32_t)std::strtoul(next("--height"), nullptr, 10);
        else if (a == "--samples") samples = (uint32_t)std::strtoul(next("--samples"), nullptr, 10);
        else if (a == "--bounces") bounces = (uint32_t)std::strtoul(next("--bounces"), nullptr, 10);
        else if (a == "--app-lighting") flags |= 1u;   // HDR sky + sun of raytracer/app.rs:129-143
        else if (a[0] != '-') scene_name = a;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (!benchmark) {
        std::fprintf(stderr, "usage: pt_bench --benchmark [scene] [--seed N] [--detail F] [--app-lighting]\n"
                             "       scenes: cornell helmet_like sponza_like bistro_like instanced_stress sss_test glass_test\n");
        return 2;
    }

    std::printf("Benchmark: %s\n", scene_name.c_str());
    std::printf("Loading model...\n");
    const double load_start = now_s();
    PthScene* scene = pth_scene_procedural(scene_name.c_str(), seed, detail, flags);
    if (!scene) { std::fprintf(stderr, "unknown procedural scene %s\n", scene_name.c_str()); return 1; }
    const PtxSceneDesc* desc = pth_scene_desc(scene);
    // the host library builds geometry and acceleration structure in one go; report it under both headings
    const double load_time = now_s() - load_start;
    std::printf("  Loaded in %.3fs: %u instances, %u materials, %u textures\n", load_time, desc->n_instances, desc->n_materials,
                desc->n_textures);

    std::printf("Building acceleration structure...\n");
    const double build_start = now_s();
    PtxContext* ctx = nullptr;
    if (ptx_create(0, &ctx) != PTX_OK) { std::fprintf(stderr, "ptx_create failed: %s\n", ptx_last_error(nullptr)); return 1; }
    PtxCamera cam;
    const float aspect = (float)width / (float)height;
    if (pth_scene_default_camera(scene_name.c_str(), aspect, &cam) != 0) {
        const float eye[3] = {0.0f, 1.0f, 3.0f}, at[3] = {0.0f, 0.5f, 0.0f};   // main.rs:98-100
        pth_camera_new(eye, at, 60.0f, aspect, &cam);
    }
    CHECK(ptx_upload_scene(ctx, desc));
    CHECK(ptx_set_camera(ctx, &cam));
    const double build_time = now_s() - build_start + load_time;
    std::printf("  Built in %.3fs\n", build_time);

    std::printf("Rendering %u samples at %ux%u...\n", samples, width, height);
    CHECK(ptx_fb_reset(ctx, width, height, 0));
    CHECK(ptx_sync(ctx));
    const double render_start = now_s();
    for (uint32_t done = 0; done < samples;) {
        const uint32_t chunk = samples - done < 4 ? samples - done : 4;   // main.rs:135 reports every 4 samples
        CHECK(ptx_fb_render_passes(ctx, done, chunk, bounces));
        CHECK(ptx_sync(ctx));
        done += chunk;
        if (done % 4 == 0) std::printf("  Sample %u/%u: %.2f spp/sec\n", done, samples, (double)done / (now_s() - render_start));
    }
    std::vector<float> mean((size_t)width * height * 3);
    CHECK(ptx_fb_read_mean(ctx, mean.data()));
    const double render_time = now_s() - render_start;

    const double total_primary_rays = (double)width * height * samples;
    std::printf("\nResults:\n");
    std::printf("  Build time:   %.3fs\n", build_time);
    std::printf("  Render time:  %.3fs\n", render_time);
    std::printf("  Primary rays: %.2fM\n", total_primary_rays / 1e6);
    std::printf("  Performance:  %.2f Mprimary/s\n", total_primary_rays / render_time / 1e6);
    std::printf("  Samples/sec:  %.2f\n", samples / render_time);
    double sum = 0.0;
    for (float v : mean) sum += v;
    std::printf("  Mean radiance: %.6f\n", sum / (double)mean.size());   // not in the reference: a checksum for the log
    ptx_destroy(ctx);
    pth_scene_free(scene);
    return 0;
}
