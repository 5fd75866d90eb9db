// C entry points of libspt_host.so (see include/spt_host.h for the reference
// counterparts of each function).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "host_scene.hpp"
#include "json.hpp"

namespace spt_host {

static thread_local std::string g_error;
void set_error(const std::string& m) { g_error = m; }

}  // namespace spt_host

using namespace spt_host;

struct spt_host_scene {
    HostScene* hs;
};

extern "C" {

const char* spt_host_last_error(void) { return g_error.c_str(); }

spt_status spt_host_catmull_clark(const char* ply_path, uint32_t fas_times, uint32_t* n_patches, float** control_points_out) {
    if (!ply_path || !n_patches || !control_points_out) { set_error("catmull_clark: null argument"); return SPT_ERR_INVALID_ARG; }
    *n_patches = 0;
    *control_points_out = nullptr;
    try {
        const std::vector<float> cps = catmull_clark_patches(ply_path, fas_times);
        float* out = static_cast<float*>(std::malloc(std::max<size_t>(cps.size(), 1) * sizeof(float)));
        if (!out) { set_error("catmull_clark: out of memory"); return SPT_ERR_OUT_OF_MEMORY; }
        std::memcpy(out, cps.data(), cps.size() * sizeof(float));
        *n_patches = (uint32_t)(cps.size() / 48);
        *control_points_out = out;
        return SPT_OK;
    } catch (const HostError& e) {
        set_error(e.msg);
        return e.code;
    } catch (const std::exception& e) {
        set_error(std::string("catmull_clark: ") + e.what());
        return SPT_HOST_ERR_PARSE;
    }
}

spt_status spt_host_load_scene(const char* scene_json_path, spt_host_scene** out) {
    if (!scene_json_path || !out) { set_error("load_scene: null argument"); return SPT_ERR_INVALID_ARG; }
    *out = nullptr;
    try {
        HostScene* hs = load_scene_file(scene_json_path);
        *out = new spt_host_scene{hs};
        return SPT_OK;
    } catch (const HostError& e) {
        set_error(e.msg);
        return e.code;
    } catch (const std::exception& e) {
        set_error(std::string("load_scene: ") + e.what());
        return SPT_HOST_ERR_PARSE;
    }
}

const spt_scene_desc* spt_host_scene_desc(const spt_host_scene* scene) { return scene ? &scene->hs->desc : nullptr; }

spt_status spt_host_scene_set_bezier_newton(spt_host_scene* scene, int32_t newton) {
    if (!scene) { set_error("scene_set_bezier_newton: null argument"); return SPT_ERR_INVALID_ARG; }
    for (spt_bezier_patch& bp : scene->hs->bezier_patches) bp.cp[0][0][3] = newton ? SPT_BEZIER_NEWTON : 0.0f;   // the desc points at this array
    return SPT_OK;
}

// Scene::get_camera (src/core/scene.rs:29-41): by name, or the only one
spt_status spt_host_scene_camera(const spt_host_scene* scene, const char* name, spt_camera* out) {
    if (!scene || !out) { set_error("scene_camera: null argument"); return SPT_ERR_INVALID_ARG; }
    const HostScene& hs = *scene->hs;
    if (name) {
        auto it = hs.camera_index.find(name);
        if (it == hs.camera_index.end()) { set_error(std::string("There is no camera names ") + name); return SPT_HOST_ERR_SCHEMA; }
        *out = hs.cameras[it->second];
        return SPT_OK;
    }
    if (hs.cameras.size() == 1) { *out = hs.cameras[0]; return SPT_OK; }
    set_error("There are multiple cameras so a name must be given");
    return SPT_HOST_ERR_SCHEMA;
}

void spt_host_scene_free(spt_host_scene* scene) {
    if (!scene) return;
    delete scene->hs;
    delete scene;
}

// loader::load_renderer (src/loader/json.rs:19-51) + create_sampler_from_params
// (src/pixel_sampler/mod.rs:28-43) + create_filter_from_params (src/filter/mod.rs:19-32)
spt_status spt_host_load_renderer(const char* path, spt_render_params* params, float* filter_radius) {
    if (!path || !params) { set_error("load_renderer: null argument"); return SPT_ERR_INVALID_ARG; }
    try {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw HostError(SPT_HOST_ERR_IO, std::string("cannot open '") + path + "'");
        std::stringstream ss;
        ss << f.rdbuf();
        std::string text = ss.str();
        JsonValue root;
        try { root = JsonParser(text).parse(); } catch (const std::runtime_error& e) { throw HostError(SPT_HOST_ERR_PARSE, e.what()); }
        const JsonValue* md = root.get("max_depth");
        if (!md) throw HostError(SPT_HOST_ERR_SCHEMA, "renderer - There is no 'max_depth' field");
        if (md->kind != JsonValue::Int) throw HostError(SPT_HOST_ERR_SCHEMA, "renderer - 'max_depth' shoule be integer");
        const JsonValue* sv = root.get("sampler");
        if (!sv || sv->kind != JsonValue::Object) throw HostError(SPT_HOST_ERR_SCHEMA, "renderer - There is no 'sampler' field");
        auto need = [&](const JsonValue* o, const char* key, JsonValue::Kind kind, const char* owner, const char* hint) {
            const JsonValue* v = o->get(key);
            if (!v) throw HostError(SPT_HOST_ERR_SCHEMA, std::string(owner) + " - there is no '" + key + "' field");
            if (v->kind != kind) throw HostError(SPT_HOST_ERR_SCHEMA, std::string(owner) + " - '" + key + "' should be " + hint);
            return v;
        };
        std::string sty = need(sv, "type", JsonValue::String, "sampler", "string")->s;
        uint32_t sampler, spp, dx = 0, dy = 0;
        if (sty == "random" || sty == "recurrence") {
            sampler = sty == "random" ? SPT_SAMPLER_RANDOM : SPT_SAMPLER_RECURRENCE;
            spp = (uint32_t)need(sv, "spp", JsonValue::Int, ("sampler-" + sty).c_str(), "integer")->i;
        } else if (sty == "jittered") {
            sampler = SPT_SAMPLER_JITTERED;
            dx = (uint32_t)need(sv, "division_x", JsonValue::Int, "sampler-jittered", "integer")->i;
            dy = (uint32_t)need(sv, "division_y", JsonValue::Int, "sampler-jittered", "integer")->i;
            spp = dx * dy;
        } else {
            throw HostError(SPT_HOST_ERR_SCHEMA, "sampler: unknown type '" + sty + "'");
        }
        const JsonValue* fv = root.get("filter");
        if (!fv || fv->kind != JsonValue::Object) throw HostError(SPT_HOST_ERR_SCHEMA, "renderer - There is no 'filter' field");
        std::string fty = need(fv, "type", JsonValue::String, "filter", "string")->s;
        if (fty != "box") throw HostError(SPT_HOST_ERR_SCHEMA, "filter: unknown type '" + fty + "'");
        float radius = (float)need(fv, "radius", JsonValue::Float, "filter-box", "float")->f;
        const JsonValue* ty = root.get("type");
        if (!ty) throw HostError(SPT_HOST_ERR_SCHEMA, "renderer - There is no 'type' field");
        if (ty->kind != JsonValue::String) throw HostError(SPT_HOST_ERR_SCHEMA, "renderer - 'type' shoule be string");
        if (ty->s != "pt") throw HostError(SPT_HOST_ERR_SCHEMA, "renderer - unknown type '" + ty->s + "'");
        params->max_depth = (uint32_t)md->i;
        params->sampler = sampler;
        params->spp = spp;
        params->division_x = dx;
        params->division_y = dy;
        // BoxFilter::new (src/filter/boxf.rs:11-14): any radius; 0.5 is the plain per-pixel mean
        params->filter_radius = radius;
        if (radius != 0.5f) params->flags |= SPT_RENDER_BOX_RADIUS;
        if (filter_radius) *filter_radius = radius;
        return SPT_OK;
    } catch (const HostError& e) {
        set_error(e.msg);
        return e.code;
    }
}

}  // extern "C"
