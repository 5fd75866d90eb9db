// spt — command-line driver with the reference's flags (reference src/main.rs:26-66):
//   spt -s scene.json -r renderer.json [-w 512] [-h 512] -o out.png [-c camera]
// plus --seed, --spp, --device D | --gpus N (one image over N devices: one worker thread and one scene replica per device,
// interleaved row strips, one film - spt_host_multi_*, the counterpart of the thread fan-out of pt.rs:243-287),
// --strip-rows R, --debug-normal and --bezier-ni (the reference's two cargo features, Cargo.toml:34-36: pt.rs:113-118, bezier.rs:58-103).  It loads the scene
// with libspt_host, renders with libspt_hip (HIP kernels only) and writes the image; like the reference it reports the
// time spent inside `render`.
#include <algorithm>
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
                 "           [--seed N] [--spp N] [--device D | --gpus N | --devices a,b,..] [--strip-rows R] [--debug-normal] [--bezier-ni]\n");
}

int main(int argc, char** argv) {
    std::string scene_path, renderer_path, out_path, camera;
    uint32_t width = 512, height = 512, spp_override = 0;
    uint64_t seed = 1;
    int device = 0, gpus = 0;
    uint32_t strip_rows = 0;
    bool debug_normal = false, bezier_ni = false;
    std::vector<int32_t> device_list;
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
        else if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--devices") {      // explicit device list "0,1,2" (an index may repeat: e.g. 0,0 rehearses two workers on one GPU)
            std::string list = next();
            for (size_t pos = 0; pos <= list.size();) {
                const size_t comma = std::min(list.find(',', pos), list.size());
                if (comma > pos) device_list.push_back(std::atoi(list.substr(pos, comma - pos).c_str()));
                pos = comma + 1;
            }
            gpus = (int)device_list.size();
        }
        else if (a == "--strip-rows") strip_rows = (uint32_t)std::atoi(next());
        else if (a == "--debug-normal") debug_normal = true;
        else if (a == "--bezier-ni") bezier_ni = true;
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
    if (bezier_ni) spt_host_scene_set_bezier_newton(hs, 1);   // `cargo build --features bezier_ni`
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
    if (debug_normal) params.flags |= SPT_RENDER_DEBUG_NORMAL;
    std::vector<float> film((size_t)width * height * 3);
    std::vector<spt_render_stats> st((size_t)std::max(gpus, 1));
    std::memset(st.data(), 0, st.size() * sizeof(spt_render_stats));
    params.stats_size = (uint32_t)sizeof(spt_render_stats);   // the library writes no more than this (ABI v9)
    std::chrono::steady_clock::time_point t0;
    spt_scene* ds = nullptr;
    spt_host_multi* multi = nullptr;
    if (gpus > 0) {
        // --gpus N: devices 0 .. N-1, one replica and one worker thread each, one film
        int32_t have = 0;
        if (device_list.empty() && (spt_device_count(&have) != SPT_OK || have < gpus)) {
            std::fprintf(stderr, "Error: --gpus %d but %d usable gfx950 device(s) are visible\n", gpus, have);
            return 1;
        }
        std::vector<int32_t> devs = device_list;
        if (devs.empty())
            for (int k = 0; k < gpus; ++k) devs.push_back(k);
        const spt_device_api api = {spt_scene_create, spt_scene_destroy, spt_render, spt_last_error, spt_pin_host, spt_unpin_host};
        if (spt_host_multi_create(spt_host_scene_desc(hs), &api, (uint32_t)gpus, devs.data(), &multi) != SPT_OK) {
            std::fprintf(stderr, "Error: %s\n", spt_host_last_error());
            return 1;
        }
        std::fprintf(stderr, "Scene JSON is loaded successfully. Rendering on %d device(s)...\n", gpus);
        t0 = std::chrono::steady_clock::now();
        if (spt_host_multi_render(multi, &cam, &params, strip_rows, film.data(), st.data()) != SPT_OK) {
            std::fprintf(stderr, "Error: %s\n", spt_host_last_error());
            return 1;
        }
    } else {
        if (spt_scene_create(spt_host_scene_desc(hs), device, &ds) != SPT_OK) {
            std::fprintf(stderr, "Error: %s\n", spt_last_error());
            return 1;
        }
        std::fprintf(stderr, "Scene JSON is loaded successfully. Rendering...\n");
        t0 = std::chrono::steady_clock::now();
        if (spt_render(ds, &cam, &params, film.data(), st.data()) != SPT_OK) {
            std::fprintf(stderr, "Error: %s\n", spt_last_error());
            return 1;
        }
    }
    std::vector<uint8_t> rgb8(film.size());
    spt_host_film_to_rgb8(film.data(), (uint64_t)width * height, rgb8.data());
    if (spt_host_write_image(out_path.c_str(), rgb8.data(), width, height) != SPT_OK)
        std::printf("Failed to save image, err: %s\n", spt_host_last_error());  // printed and ignored, like pt.rs:292-294
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t samples = 0;
    double gpu_ms = 0.0;      // the slowest device's time on its stream
    for (const spt_render_stats& d : st) { samples += d.samples; gpu_ms = std::max(gpu_ms, d.gpu_ms); }
    std::fprintf(stderr, "Finished, time used: %.3fs (%.1f Msamples/s on the GPU%s, %.3f ms)\n", sec, (double)samples / (gpu_ms * 1e3),
                 gpus > 1 ? "s" : "", gpu_ms);
    if (multi) spt_host_multi_destroy(multi);
    if (ds) spt_scene_destroy(ds);
    spt_host_scene_free(hs);
    return 0;
}
