// spt — command-line driver with the reference's flags (reference src/main.rs:26-66):
//   spt -s scene.json -r renderer.json [-w 512] [-h 512] -o out.png [-c camera]
// plus --seed, --device, --gpus-shard i/n for rendering one shard.  It loads the scene with
// libspt_host, renders with libspt_hip (HIP kernels only) and writes the PNG; like the
// reference it reports the time spent inside `render`.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/spt_host.h"

static void usage() {
    std::fprintf(stderr,
                 "usage: spt -s <scene.json> -r <renderer.json> -o <out.png> [-w 512] [-h 512] [-c camera]\n"
                 "           [--seed N] [--device D] [--spp N]\n");
}

int main(int argc, char** argv) {
    std::string scene_path, renderer_path, out_path, camera;
    uint32_t width = 512, height = 512, spp_override = 0;
    uint64_t seed = 1;
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) { usage(); std::exit(2); }
            return argv[++i];
        };
        if (a == "-s" || a == "--scene") scene_path = next();
        else if (a == "-r" || a == "--renderer") renderer_path = next();
        else if (a == "-o" || a == "--output") out_path = next();
        else if (a == "-w" || a == "--width") width = (uint32_t)std::atoi(next());
        else if (a == "-h" || a == "--height") height = (uint32_t)std::atoi(next());
        else if (a == "-c" || a == "--camera") camera = next();
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--spp") spp_override = (uint32_t)std::atoi(next());
        else { usage(); return 2; }
    }
    if (scene_path.empty() || renderer_path.empty() || out_path.empty()) { usage(); return 2; }

    // this binary and the library it found at run time must agree on the struct layouts of include/spt_abi.h
    if (spt_abi_version() != SPT_ABI_VERSION) {
        std::fprintf(stderr, "Error: libspt_hip.so exports ABI version %u, this program was built against %u\n", spt_abi_version(), (unsigned)SPT_ABI_VERSION);
        return 1;
    }
    std::fprintf(stderr, "Loading from JSON and building aggregate...\n");
    spt_host_scene* hs = nullptr;
    if (spt_host_load_scene(scene_path.c_str(), &hs) != SPT_OK) {
        std::fprintf(stderr, "Error: %s\n", spt_host_last_error());
        return 1;
    }
    spt_render_params params;
    std::memset(&params, 0, sizeof params);
    float radius = 0.5f;
    if (spt_host_load_renderer(renderer_path.c_str(), &params, &radius) != SPT_OK) {
        std::fprintf(stderr, "Error: %s\n", spt_host_last_error());
        return 1;
    }
    if (spp_override && params.sampler != SPT_SAMPLER_JITTERED) params.spp = spp_override;
    spt_camera cam;
    if (spt_host_scene_camera(hs, camera.empty() ? nullptr : camera.c_str(), &cam) != SPT_OK) {
        std::fprintf(stderr, "Error: %s\n", spt_host_last_error());
        return 1;
    }
    params.width = width;
    params.height = height;
    params.seed = seed;
    params.shard_index = 0;
    params.shard_count = 1;
    params.strip_rows = 16;
    spt_scene* ds = nullptr;
    if (spt_scene_create(spt_host_scene_desc(hs), device, &ds) != SPT_OK) {
        std::fprintf(stderr, "Error: %s\n", spt_last_error());
        return 1;
    }
    std::fprintf(stderr, "Scene JSON is loaded successfully. Rendering...\n");
    std::vector<float> film((size_t)width * height * 3);
    spt_render_stats st;
    params.stats_size = (uint32_t)sizeof st;   // the library writes no more than this (ABI v9)
    auto t0 = std::chrono::steady_clock::now();
    if (spt_render(ds, &cam, &params, film.data(), &st) != SPT_OK) {
        std::fprintf(stderr, "Error: %s\n", spt_last_error());
        return 1;
    }
    std::vector<uint8_t> rgb8(film.size());
    spt_host_film_to_rgb8(film.data(), (uint64_t)width * height, rgb8.data());
    if (spt_host_write_image(out_path.c_str(), rgb8.data(), width, height) != SPT_OK)
        std::printf("Failed to save image, err: %s\n", spt_host_last_error());  // printed and ignored, like pt.rs:292-294
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "Finished, time used: %.3fs (%.1f Msamples/s on the GPU, %.3f ms)\n", sec,
                 (double)st.samples / (st.gpu_ms * 1e3), st.gpu_ms);
    spt_scene_destroy(ds);
    spt_host_scene_free(hs);
    return 0;
}
