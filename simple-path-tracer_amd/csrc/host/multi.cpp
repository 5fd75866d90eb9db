// One call, N devices, one film (include/spt_host.h): the counterpart of the thread fan-out of PathTracer::render
// (reference src/renderer/pt.rs:243-287) with a GPU behind every worker.  The reference cuts the image into contiguous
// row bands, one per thread (src/renderer/util.rs:6-19), and lets every thread write its pixels into the one film through
// UnsafeFilm (src/core/film.rs:101-116).  Here the bands are interleaved strips - a contiguous band would put the whole
// object of a typical scene on two or three of eight devices - and "write into the one film" is a strided device-to-host
// copy per worker; nothing else changes hands, so there is no collective and no ordering between workers to get wrong.
//
// Workers are persistent threads (created with the replicas, parked on a condition variable between frames): a frame of
// the headline workload is ~0.5 ms per device at n = 8, thread creation would be a tenth of that.
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/spt_host.h"

namespace spt_host {
void set_error(const std::string& m);
}

struct spt_host_multi {
    spt_device_api api{};
    std::vector<int32_t> devices;
    std::vector<spt_scene*> scenes;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t job_id = 0;          // incremented per frame; a worker runs job k once
    uint32_t pending = 0;         // workers still busy with the current job
    bool quit = false;
    // the current job
    const spt_camera* cam = nullptr;
    spt_render_params params{};
    float* film = nullptr;
    spt_render_stats* stats = nullptr;
    std::vector<spt_status> status;
    std::vector<std::string> errors;
    void* pinned = nullptr;       // film currently page-locked through api.pin_host
    uint64_t pinned_bytes = 0;
};

namespace {

void worker_main(spt_host_multi* m, uint32_t k) {
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lock(m->mu);
            m->cv_job.wait(lock, [&] { return m->quit || m->job_id != seen; });
            if (m->quit) return;
            seen = m->job_id;
        }
        const uint32_t n = (uint32_t)m->devices.size();
        spt_render_params p = m->params;
        p.shard_index = k;
        p.shard_count = n;
        const uint64_t row_bytes = (uint64_t)p.width * 3u * sizeof(float);
        p.out_strip_stride = (uint64_t)n * p.strip_rows * row_bytes;      // this worker's strips, in place in the full film
        float* first = m->film + (size_t)k * p.strip_rows * p.width * 3u;
        // a worker whose first strip lies below the image has no rows at all (more devices than strips)
        if ((uint64_t)k * p.strip_rows >= p.height) first = m->film;
        spt_render_stats* st = m->stats ? reinterpret_cast<spt_render_stats*>(reinterpret_cast<char*>(m->stats) + (size_t)k * p.stats_size) : nullptr;
        const spt_status rc = m->api.render(m->scenes[k], m->cam, &p, first, st);
        std::string err;
        if (rc != SPT_OK) {
            const char* e = m->api.last_error ? m->api.last_error() : nullptr;   // thread-local in libspt_hip.so: read it on THIS thread
            err = "device " + std::to_string(m->devices[k]) + " (shard " + std::to_string(k) + " of " + std::to_string(n) + "): " + (e ? e : "render failed");
        }
        {
            std::lock_guard<std::mutex> lock(m->mu);
            m->status[k] = rc;
            m->errors[k] = err;
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

void stop_workers(spt_host_multi* m) {
    {
        std::lock_guard<std::mutex> lock(m->mu);
        m->quit = true;
    }
    m->cv_job.notify_all();
    for (auto& t : m->workers)
        if (t.joinable()) t.join();
    m->workers.clear();
}

}  // namespace

extern "C" {

spt_status spt_host_multi_create(const spt_scene_desc* desc, const spt_device_api* api, uint32_t n_devices, const int32_t* devices,
                                 spt_host_multi** out) {
    if (!desc || !api || !devices || !out || n_devices == 0) { spt_host::set_error("multi_create: null argument or no devices"); return SPT_ERR_INVALID_ARG; }
    if (!api->scene_create || !api->scene_destroy || !api->render) { spt_host::set_error("multi_create: the device table lacks scene_create / scene_destroy / render"); return SPT_ERR_INVALID_ARG; }
    if (n_devices > 1024u) { spt_host::set_error("multi_create: more than 1024 devices"); return SPT_ERR_INVALID_ARG; }
    *out = nullptr;
    spt_host_multi* m = new spt_host_multi();
    m->api = *api;
    m->devices.assign(devices, devices + n_devices);
    m->scenes.assign(n_devices, nullptr);
    m->status.assign(n_devices, SPT_OK);
    m->errors.assign(n_devices, std::string());
    // the replicas are uploaded side by side (a scene create is a BVH build + an upload: hundreds of ms for a large scene)
    {
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_devices; ++k)
            th.emplace_back([m, desc, k] {
                const spt_status rc = m->api.scene_create(desc, m->devices[k], &m->scenes[k]);
                m->status[k] = rc;
                if (rc != SPT_OK) {
                    const char* e = m->api.last_error ? m->api.last_error() : nullptr;
                    m->errors[k] = "device " + std::to_string(m->devices[k]) + ": " + (e ? e : "scene_create failed");
                }
            });
        for (auto& t : th) t.join();
    }
    for (uint32_t k = 0; k < n_devices; ++k) {
        if (m->status[k] != SPT_OK) {
            const spt_status rc = m->status[k];
            spt_host::set_error("multi_create: " + m->errors[k]);
            for (spt_scene* s : m->scenes)
                if (s) m->api.scene_destroy(s);
            delete m;
            return rc;
        }
    }
    for (uint32_t k = 0; k < n_devices; ++k) m->workers.emplace_back(worker_main, m, k);
    *out = m;
    return SPT_OK;
}

uint32_t spt_host_multi_device_count(const spt_host_multi* m) { return m ? (uint32_t)m->devices.size() : 0u; }

spt_status spt_host_multi_render(spt_host_multi* m, const spt_camera* cam, const spt_render_params* params, uint32_t strip_rows,
                                 float* film, spt_render_stats* stats) {
    if (!m || !cam || !params || !film) { spt_host::set_error("multi_render: null argument"); return SPT_ERR_INVALID_ARG; }
    if (params->width == 0 || params->height == 0) { spt_host::set_error("multi_render: width and height must be > 0"); return SPT_ERR_INVALID_ARG; }
    if (stats && params->stats_size < 8u) { spt_host::set_error("multi_render: stats given but params.stats_size is not set"); return SPT_ERR_INVALID_ARG; }
    if (params->flags & SPT_RENDER_ASYNC) { spt_host::set_error("multi_render: SPT_RENDER_ASYNC is not supported (the call returns a complete film)"); return SPT_ERR_INVALID_ARG; }
    const uint32_t n = (uint32_t)m->devices.size();
    if (strip_rows == 0) {
        // Even shares: every device should own several strips spread over the whole image.  16 rows (one tile row of the
        // primary kernel) while that still leaves >= 8 strips per device, finer below (measured on the headline image at n = 8:
        // slowest / mean device 1.15 with 16-row strips, 1.05 with 4-row strips, tools/strip_rows_sweep.py)
        strip_rows = 16;
        while (strip_rows > 1 && (uint64_t)params->height < (uint64_t)strip_rows * n * 8u) strip_rows /= 2;
    }
    const uint64_t bytes = (uint64_t)params->width * params->height * 3u * sizeof(float);
    if (m->api.pin_host && m->api.unpin_host && (m->pinned != film || m->pinned_bytes != bytes)) {
        if (m->pinned) m->api.unpin_host(m->pinned);
        m->pinned = nullptr;
        if (m->api.pin_host(film, bytes) == SPT_OK) { m->pinned = film; m->pinned_bytes = bytes; }   // not fatal: the copy-out is slower, not wrong
    }
    {
        std::unique_lock<std::mutex> lock(m->mu);
        m->cam = cam;
        m->params = *params;
        m->params.strip_rows = strip_rows;
        m->film = film;
        m->stats = stats;
        m->pending = n;
        ++m->job_id;
        m->cv_job.notify_all();
        m->cv_done.wait(lock, [&] { return m->pending == 0; });
    }
    for (uint32_t k = 0; k < n; ++k)
        if (m->status[k] != SPT_OK) {
            spt_host::set_error("multi_render: " + m->errors[k]);
            return m->status[k];
        }
    return SPT_OK;
}

void spt_host_multi_destroy(spt_host_multi* m) {
    if (!m) return;
    stop_workers(m);
    if (m->pinned && m->api.unpin_host) m->api.unpin_host(m->pinned);
    for (spt_scene* s : m->scenes)
        if (s) m->api.scene_destroy(s);
    delete m;
}

}  // extern "C"
