// Scene JSON -> flattened spt_scene_desc.
//
// Host-side counterpart of the reference's loader + scene assembly:
//   src/loader/json.rs:53-242            section order, object | array | external-file values
//   src/core/loader.rs:286-305,404-438   typed getters (Int != Float), unused-key warnings
//   src/core/scene_resources.rs:85-168   aggregate, light list, ShapeLight per emissive instance
//   src/primitive/instance.rs:35-85      T * Rz * Rx * Ry * S * M transform composition
//   src/primitive/triangle.rs:57-108,339-388  OBJ -> MeshVertex, tangents
// The object graph is not kept: everything ends up in the POD arrays of spt_abi.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

#include "../../../include/spt_host.h"
#include "bvh.hpp"
#include "hmath.hpp"
#include "host_scene.hpp"
#include "json.hpp"

namespace spt_host {

namespace {

bool g_verbose() {
    static int v = -1;
    if (v < 0) v = std::getenv("SPT_LOG") ? 1 : 0;
    return v == 1;
}
void warn(const std::string& m) {
    if (g_verbose()) std::fprintf(stderr, "[spt warn] %s\n", m.c_str());
}

std::string with_file_name(const std::string& base, const std::string& file) {
    size_t p = base.find_last_of('/');
    if (p == std::string::npos) return file;
    return base.substr(0, p + 1) + file;
}

std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw HostError(SPT_HOST_ERR_IO, "cannot open '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

JsonValue parse_json_file(const std::string& path) {
    std::string text = read_file(path);
    try {
        return JsonParser(text).parse();
    } catch (const std::runtime_error& e) {
        throw HostError(SPT_HOST_ERR_PARSE, path + ": " + e.what());
    }
}

// InputParams (src/core/loader.rs:11-16): typed access + visited-key tracking.
class Params {
   public:
    Params(const JsonValue& v, std::string base_path) : base_(std::move(base_path)) {
        if (v.kind != JsonValue::Object) throw HostError(SPT_HOST_ERR_SCHEMA, "expected a JSON object");
        for (auto& kv : v.obj) {
            if (kv.second.kind == JsonValue::Null)
                throw HostError(SPT_HOST_ERR_SCHEMA, "can't convert to InputParamsValue from null json");
            if (kv.second.kind == JsonValue::Object)
                throw HostError(SPT_HOST_ERR_SCHEMA, "can't convert to InputParamsValue from object json");
            vals_[kv.first] = &kv.second;
        }
    }
    void set_name(const std::string& n) { name_ = n; }
    const std::string& name() const { return name_; }
    bool contains(const std::string& k) const { return vals_.count(k) != 0; }

    std::string get_str(const std::string& k) {
        const JsonValue* v = find(k);
        if (v->kind != JsonValue::String) bad(k, "string");
        visited_.insert(k);
        return v->s;
    }
    float get_float(const std::string& k) {
        const JsonValue* v = find(k);
        if (v->kind != JsonValue::Float) bad(k, "float");  // Int is NOT accepted (loader.rs:286-299)
        visited_.insert(k);
        return (float)v->f;
    }
    float get_float_or(const std::string& k, float fb) {
        auto it = vals_.find(k);
        if (it == vals_.end() || it->second->kind != JsonValue::Float) return fb;
        visited_.insert(k);
        return (float)it->second->f;
    }
    int32_t get_int(const std::string& k) {
        const JsonValue* v = find(k);
        if (v->kind != JsonValue::Int) bad(k, "integer");
        visited_.insert(k);
        return (int32_t)v->i;
    }
    bool get_bool_or(const std::string& k, bool fb) {
        auto it = vals_.find(k);
        if (it == vals_.end() || it->second->kind != JsonValue::Bool) return fb;
        visited_.insert(k);
        return it->second->b;
    }
    V3 get_float3(const std::string& k) {
        const JsonValue* v = find(k);
        V3 out;
        if (!float_vec(v, 3, &out.x)) bad(k, "array with 3 floats");
        visited_.insert(k);
        return out;
    }
    V3 get_float3_or(const std::string& k, V3 fb) {
        auto it = vals_.find(k);
        V3 out;
        if (it == vals_.end() || !float_vec(it->second, 3, &out.x)) return fb;
        visited_.insert(k);
        return out;
    }
    std::array<float, 2> get_float2_or(const std::string& k, std::array<float, 2> fb) {
        auto it = vals_.find(k);
        std::array<float, 2> out;
        if (it == vals_.end() || !float_vec(it->second, 2, out.data())) return fb;
        visited_.insert(k);
        return out;
    }
    // get_float_3darray (loader.rs:201-262) with all three lengths given; `stride` floats per innermost row in `out`
    void get_float_3darray(const std::string& k, size_t n1, size_t n2, size_t n3, float* out, size_t stride) {
        const JsonValue* v = find(k);
        const std::string what = "3D array with " + std::to_string(n1) + "x" + std::to_string(n2) + "x" + std::to_string(n3) + " floats";
        if (v->kind != JsonValue::Array || v->arr.size() != n1) bad(k, what.c_str());
        for (size_t a = 0; a < n1; ++a) {
            const JsonValue& v2 = v->arr[a];
            if (v2.kind != JsonValue::Array || v2.arr.size() != n2) bad(k, what.c_str());
            for (size_t b = 0; b < n2; ++b) {
                const JsonValue& v3 = v2.arr[b];
                if (v3.kind != JsonValue::Array || v3.arr.size() != n3) bad(k, what.c_str());
                for (size_t c = 0; c < n3; ++c) {
                    if (v3.arr[c].kind != JsonValue::Float) bad(k, what.c_str());
                    out[(a * n2 + b) * stride + c] = (float)v3.arr[c].f;
                }
            }
        }
        visited_.insert(k);
    }
    // get_matrix (loader.rs:307-335): 16 values column-major; non-Float entries keep identity.
    Affine get_matrix(const std::string& k) {
        const JsonValue* v = find(k);
        if (v->kind != JsonValue::Array) bad(k, "an array");
        if (v->arr.size() != 16) bad(k, "an array of 16 floats");
        float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int i = 0; i < 16; ++i)
            if (v->arr[i].kind == JsonValue::Float) m[i] = (float)v->arr[i].f;
        visited_.insert(k);
        Affine a;  // Affine3A::from_mat4: upper 3x3 + translation column
        a.m.c0 = {m[0], m[1], m[2]};
        a.m.c1 = {m[4], m[5], m[6]};
        a.m.c2 = {m[8], m[9], m[10]};
        a.t = {m[12], m[13], m[14]};
        return a;
    }
    std::string get_file_path(const std::string& k) { return with_file_name(base_, get_str(k)); }
    size_t num_unused() const {
        size_t n = 0;
        for (auto& kv : vals_)
            if (kv.first.rfind("#", 0) != 0 && !visited_.count(kv.first)) ++n;
        return n;
    }
    void check_unused() const {
        for (auto& kv : vals_)
            if (kv.first.rfind("#", 0) != 0 && !visited_.count(kv.first)) warn(name_ + " - unused key '" + kv.first + "'");
    }
    void mark(const std::string& k) { visited_.insert(k); }

   private:
    std::map<std::string, const JsonValue*> vals_;
    std::set<std::string> visited_;
    std::string name_ = "?";
    std::string base_;

    const JsonValue* find(const std::string& k) {
        auto it = vals_.find(k);
        if (it == vals_.end()) throw HostError(SPT_HOST_ERR_SCHEMA, name_ + " - there is no '" + k + "' field");
        return it->second;
    }
    [[noreturn]] void bad(const std::string& k, const char* hint) {
        throw HostError(SPT_HOST_ERR_SCHEMA, name_ + " - '" + k + "' should be " + hint);
    }
    static bool float_vec(const JsonValue* v, size_t n, float* out) {
        if (v->kind != JsonValue::Array || v->arr.size() != n) return false;
        for (size_t i = 0; i < n; ++i) {
            if (v->arr[i].kind != JsonValue::Float) return false;
            out[i] = (float)v->arr[i].f;
        }
        return true;
    }
};

float luminance(V3 c) { return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z; }

float fresnel_moment1(float eta) {  // src/bxdf/util.rs:123-134
    float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1.0f)
        return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945f * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}

float srgb_to_linear(float s) {  // src/texture/srgb_tex.rs
    if (s <= 0.04045f) return s / 12.92f;
    return std::pow((s + 0.055f) / 1.055f, 2.4f);
}

struct ObjMesh {
    std::vector<V3> pos, nrm, tan, bit;
    std::vector<float> uv;  // 2 per vertex
    std::vector<uint32_t> idx;
};

// tobj::load_obj with triangulate + single_index (src/primitive/triangle.rs:57-63):
// each distinct (v, vt, vn) tuple becomes one vertex in first-seen order; polygons
// are fanned.  All models of the file are merged into one vertex/index list.
ObjMesh load_obj(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw HostError(SPT_HOST_ERR_IO, "cannot open OBJ '" + path + "'");
    std::vector<V3> v, vn;
    std::vector<std::pair<float, float>> vt;
    ObjMesh m;
    std::map<std::array<int, 3>, uint32_t> remap;
    bool any_vt = false, any_vn = false;
    std::vector<std::array<int, 3>> corners;
    std::string line;
    struct Face { std::vector<std::array<int, 3>> c; };
    std::vector<Face> faces;
    while (std::getline(f, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v") {
            V3 p; ls >> p.x >> p.y >> p.z; v.push_back(p);
        } else if (tag == "vn") {
            V3 p; ls >> p.x >> p.y >> p.z; vn.push_back(p);
        } else if (tag == "vt") {
            float a = 0, b = 0; ls >> a >> b; vt.emplace_back(a, b);
        } else if (tag == "f") {
            Face face;
            std::string tok;
            while (ls >> tok) {
                std::array<int, 3> c = {0, 0, 0};  // 1-based, 0 = absent
                int field = 0;
                std::string cur;
                auto flush = [&]() {
                    if (!cur.empty() && field < 3) c[field] = std::atoi(cur.c_str());
                    cur.clear();
                };
                for (char ch : tok) {
                    if (ch == '/') { flush(); ++field; } else cur += ch;
                }
                flush();
                if (c[0] < 0) c[0] = (int)v.size() + c[0] + 1;
                if (c[1] < 0) c[1] = (int)vt.size() + c[1] + 1;
                if (c[2] < 0) c[2] = (int)vn.size() + c[2] + 1;
                if (c[0] <= 0 || c[0] > (int)v.size()) throw HostError(SPT_HOST_ERR_PARSE, path + ": face index out of range");
                if (c[1] > (int)vt.size() || c[2] > (int)vn.size()) throw HostError(SPT_HOST_ERR_PARSE, path + ": face index out of range");
                if (c[1] > 0) any_vt = true;
                if (c[2] > 0) any_vn = true;
                face.c.push_back(c);
            }
            if (face.c.size() >= 3) faces.push_back(std::move(face));
        }
    }
    auto vertex_of = [&](const std::array<int, 3>& c) -> uint32_t {
        auto it = remap.find(c);
        if (it != remap.end()) return it->second;
        uint32_t id = (uint32_t)m.pos.size();
        remap[c] = id;
        m.pos.push_back(v[c[0] - 1]);
        // MeshVertex::default: normal Z, texcoords 0, tangent X, bitangent Y (triangle.rs:29-38)
        m.nrm.push_back((any_vn && c[2] > 0) ? vn[c[2] - 1] : V3{0, 0, 1});
        if (any_vt && c[1] > 0) { m.uv.push_back(vt[c[1] - 1].first); m.uv.push_back(vt[c[1] - 1].second); }
        else { m.uv.push_back(0); m.uv.push_back(0); }
        m.tan.push_back({1, 0, 0});
        m.bit.push_back({0, 1, 0});
        return id;
    };
    for (auto& face : faces)
        for (size_t k = 1; k + 1 < face.c.size(); ++k) {
            m.idx.push_back(vertex_of(face.c[0]));
            m.idx.push_back(vertex_of(face.c[k]));
            m.idx.push_back(vertex_of(face.c[k + 1]));
        }
    return m;
}

// TriMesh::calc_tangents (src/primitive/triangle.rs:339-388)
void calc_tangents(ObjMesh& m) {
    size_t nv = m.pos.size();
    std::vector<V3> ts(nv), bs(nv);
    std::vector<int> deg(nv, 0);
    for (size_t t = 0; t + 2 < m.idx.size(); t += 3) {
        uint32_t i0 = m.idx[t], i1 = m.idx[t + 1], i2 = m.idx[t + 2];
        V3 e1 = m.pos[i1] - m.pos[i0], e2 = m.pos[i2] - m.pos[i0];
        float u1x = m.uv[2 * i1] - m.uv[2 * i0], u1y = m.uv[2 * i1 + 1] - m.uv[2 * i0 + 1];
        float u2x = m.uv[2 * i2] - m.uv[2 * i0], u2y = m.uv[2 * i2 + 1] - m.uv[2 * i0 + 1];
        float det = u1x * u2y - u1y * u2x;
        if (det != 0.0f) {
            det = 1.0f / det;
            V3 tg = normalize((e1 * u2y - e2 * u1y) * det);
            V3 bt = normalize((e2 * u1x - e1 * u2x) * det);
            for (uint32_t i : {i0, i1, i2}) { ts[i] = ts[i] + tg; bs[i] = bs[i] + bt; deg[i]++; }
        }
    }
    for (size_t i = 0; i < nv; ++i)
        if (deg[i] != 0) {
            float inv = 1.0f / (float)deg[i];
            m.tan[i] = normalize(ts[i] * inv);
            m.bit[i] = normalize(bs[i] * inv);
        }
}


// ---- glTF 2.0 import helpers (role of the `gltf` crate's import(), reference src/loader/gltf.rs:19-20) ----
std::vector<uint8_t> base64_decode(const std::string& in, const std::string& what) {
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (char ch : in) {
        int v;
        if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
        else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
        else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
        else if (ch == '+' || ch == '-') v = 62;
        else if (ch == '/' || ch == '_') v = 63;
        else if (ch == '=' || ch == '\n' || ch == '\r') continue;
        else throw HostError(SPT_HOST_ERR_PARSE, what + ": bad base64 data");
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xffu)); }
    }
    return out;
}
std::vector<uint8_t> read_bytes(const std::string& path) {
    std::string s = read_file(path);
    return std::vector<uint8_t>(s.begin(), s.end());
}
// a uri of a buffer or image: "data:<mime>;base64,<payload>" or a path relative to the glTF file
std::vector<uint8_t> load_uri(const std::string& uri, const std::string& gltf_path) {
    if (uri.rfind("data:", 0) == 0) {
        size_t c = uri.find(',');
        if (c == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > c)
            throw HostError(SPT_HOST_ERR_UNSUPPORTED, "gltf: only base64 data URIs are supported");
        return base64_decode(uri.substr(c + 1), "gltf data URI");
    }
    return read_bytes(with_file_name(gltf_path, uri));
}
// TriMesh::calc_normals (src/primitive/triangle.rs:305-337)
void calc_normals(ObjMesh& m) {
    size_t nv = m.pos.size();
    std::vector<V3> sum(nv);
    std::vector<int> deg(nv, 0);
    for (size_t t = 0; t + 2 < m.idx.size(); t += 3) {
        uint32_t i0 = m.idx[t], i1 = m.idx[t + 1], i2 = m.idx[t + 2];
        V3 n = normalize(cross(m.pos[i1] - m.pos[i0], m.pos[i2] - m.pos[i0]));
        for (uint32_t i : {i0, i1, i2}) { sum[i] = sum[i] + n; deg[i]++; }
    }
    for (size_t i = 0; i < nv; ++i)
        if (deg[i] != 0) m.nrm[i] = normalize(sum[i] / (float)deg[i]);
}
struct M4 {   // column-major, glam Mat4
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};
inline M4 operator*(const M4& a, const M4& b) {   // glam Mat4 * Mat4: each column of b transformed by a
    M4 r;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k)
            r.m[4 * c + k] = ((a.m[k] * b.m[4 * c] + a.m[4 + k] * b.m[4 * c + 1]) + a.m[8 + k] * b.m[4 * c + 2]) + a.m[12 + k] * b.m[4 * c + 3];
    return r;
}
double jnum(const JsonValue* v, double fb) { return (v && v->is_number()) ? v->as_double() : fb; }
bool jvec(const JsonValue* v, size_t n, float* out) {
    if (!v || v->kind != JsonValue::Array || v->arr.size() != n) return false;
    for (size_t i = 0; i < n; ++i) {
        if (!v->arr[i].is_number()) return false;
        out[i] = (float)v->arr[i].as_double();
    }
    return true;
}

struct PrimRec { uint32_t type; uint32_t id; Box box; };
constexpr uint32_t kPrimCatmullClark = 100u;   // host-only kind: PrimRec::id indexes Loader::catmulls; its patches become Bezier instances
struct SurfaceRec { uint32_t index; bool emissive; };

}  // namespace

// ---------------------------------------------------------------------------------------

struct SceneBuilder {
    HostScene& hs;
    std::string path;
    // a named texture = a node of hs.textures; `constant` ones (no image below them) are folded to `value`
    struct TexInfo { uint32_t node; bool constant; V3 value; };
    std::map<std::string, TexInfo> textures;
    std::map<std::string, uint32_t> image_ids;        // by resolved file name: one mip pyramid per file
    bool any_textured = false;                        // a material / surface samples an image texture per hit
    std::map<std::string, uint32_t> materials, surfaces, mediums;
    // materials / primitives of kinds that are not built: the scene files of the reference pull whole libraries of
    // them in (common_materials.json, common_primitives.json), so the error is raised where one is actually USED
    std::map<std::string, std::string> unsupported_materials, unsupported_prims;
    std::map<std::string, PrimRec> prims;
    std::vector<V3> avg_emissive;                     // per surface
    struct InstRec { spt_instance inst; Affine trans; std::string name; };
    struct CatmullRec { std::string ply_path, label; uint32_t fas_times = 4; bool loaded = false; uint32_t first_patch = 0, n_patches = 0; };
    std::vector<CatmullRec> catmulls;
    std::set<std::string> catmull_instances;
    std::map<std::string, InstRec> instances;         // name-sorted (Q7: reference order is HashMap order)
    std::map<std::string, spt_light> lights;
    bool has_env = false;
    std::vector<ObjMesh> mesh_src;                    // per mesh, leaf order (for area / power)

    explicit SceneBuilder(HostScene& h, std::string p) : hs(h), path(std::move(p)) {}

    using LoadFn = void (SceneBuilder::*)(Params&);

    // load_from_value_or_external (src/loader/json.rs:212-242)
    void load_section(const JsonValue& v, const char* env, LoadFn fn, bool allow_array) {
        if (v.kind == JsonValue::String) {
            std::string ext = with_file_name(path, v.s);
            JsonValue sub;
            try {
                sub = parse_json_file(ext);
            } catch (const HostError& e) {
                if (e.code == SPT_HOST_ERR_IO) throw HostError(SPT_HOST_ERR_IO, std::string(env) + " - External json file not found: " + ext);
                throw;
            }
            load_section(sub, env, fn, allow_array);
        } else if (v.kind == JsonValue::Array) {
            if (!allow_array) throw HostError(SPT_HOST_ERR_SCHEMA, std::string(env) + " - Field should not be an array");
            for (auto& e : v.arr) load_section(e, env, fn, true);
        } else {
            Params p(v, path);
            (this->*fn)(p);
        }
    }

    const TexInfo& tex_info(const std::string& name) {
        auto it = textures.find(name);
        if (it == textures.end()) throw HostError(SPT_HOST_ERR_SCHEMA, "There is no texture named '" + name + "'");
        return it->second;
    }
    // value of a constant texture; for an image-backed one its TextureT::average_color
    V3 texture(const std::string& name) {
        const TexInfo& t = tex_info(name);
        return t.constant ? t.value : tex_average(t.node);
    }

    // TextureT::average_color (scalar.rs:37-39, image_tex.rs:41-44, binary_op.rs:58-60, srgb_tex.rs:30-32,
    // input_modifier.rs:88-90)
    V3 tex_average(uint32_t node) {
        const spt_texture& t = hs.textures[node];
        switch (t.type) {
        case SPT_TEX_SCALAR: return {t.value[0], t.value[1], t.value[2]};
        case SPT_TEX_IMAGE: {
            const spt_image& im = hs.images[t.image];
            uint32_t px = hs.texels[hs.image_levels[im.first_level + im.n_levels - 1].first_texel];
            return {(float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f};
        }
        case SPT_TEX_ADD: return tex_average(t.a) + tex_average(t.b);
        case SPT_TEX_SUB: return tex_average(t.a) - tex_average(t.b);
        case SPT_TEX_MUL: return tex_average(t.a) * tex_average(t.b);
        case SPT_TEX_DIV: { V3 a = tex_average(t.a), b = tex_average(t.b); return {a.x / b.x, a.y / b.y, a.z / b.z}; }
        case SPT_TEX_SRGB: { V3 a = tex_average(t.a); return {srgb_to_linear(a.x), srgb_to_linear(a.y), srgb_to_linear(a.z)}; }
        default: return tex_average(t.a);
        }
    }

    // ImageTex::new + generate_mipmap (src/texture/image_tex.rs:12-15,66-100): box-filtered u8 pyramid,
    // odd sizes round up and clamp the second tap, the mean is truncated (`as u8`)
    uint32_t add_image(const std::string& file) {
        auto it = image_ids.find(file);
        if (it != image_ids.end()) return it->second;
        uint32_t w = 0, h = 0;
        std::vector<uint32_t> lvl;
        read_png_rgba8(file, &w, &h, &lvl);
        return add_image_pixels(file, w, h, lvl);
    }
    uint32_t add_image_pixels(const std::string& key, uint32_t w, uint32_t h, std::vector<uint32_t> lvl) {
        spt_image im;
        im.first_level = (uint32_t)hs.image_levels.size();
        im.n_levels = 0;
        while (true) {
            if (hs.texels.size() + lvl.size() > 0xffffffffull) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "image textures exceed 2^32 texels");
            spt_image_level L;
            L.width = w; L.height = h; L.first_texel = (uint32_t)hs.texels.size(); L.pad = 0;
            hs.image_levels.push_back(L);
            hs.texels.insert(hs.texels.end(), lvl.begin(), lvl.end());
            ++im.n_levels;
            if (w <= 1 && h <= 1) break;
            uint32_t nw = (w + 1) >> 1, nh = (h + 1) >> 1;
            std::vector<uint32_t> next((size_t)nw * nh);
            for (uint32_t i = 0; i < nw; ++i)
                for (uint32_t j = 0; j < nh; ++j) {
                    uint32_t x0 = 2 * i, x1 = std::min(2 * i + 1, w - 1), y0 = 2 * j, y1 = std::min(2 * j + 1, h - 1);
                    uint32_t p0 = lvl[(size_t)y0 * w + x0], p1 = lvl[(size_t)y1 * w + x0], p2 = lvl[(size_t)y0 * w + x1], p3 = lvl[(size_t)y1 * w + x1];
                    uint32_t out = 0;
                    for (int c = 0; c < 4; ++c) {
                        float sum = (((float)((p0 >> (8 * c)) & 255u) + (float)((p1 >> (8 * c)) & 255u)) + (float)((p2 >> (8 * c)) & 255u)) + (float)((p3 >> (8 * c)) & 255u);
                        out |= ((uint32_t)(sum * 0.25f) & 255u) << (8 * c);
                    }
                    next[(size_t)j * nw + i] = out;
                }
            lvl.swap(next);
            w = nw; h = nh;
        }
        hs.images.push_back(im);
        image_ids[key] = (uint32_t)hs.images.size() - 1;
        return (uint32_t)hs.images.size() - 1;
    }

    uint32_t add_tex_node(uint32_t type, uint32_t a = 0, uint32_t b = 0) {
        spt_texture t;
        std::memset(&t, 0, sizeof t);
        t.type = type; t.a = a; t.b = b;
        t.mode = -1; t.wrap = -1;
        t.tiling[0] = t.tiling[1] = t.tiling[2] = 1.0f;
        hs.textures.push_back(t);
        return (uint32_t)hs.textures.size() - 1;
    }

    // camera::create_camera_from_params + PerspectiveCamera::new (src/camera/perspective.rs:15-37)
    void load_camera(Params& p) {
        p.set_name("camera");
        std::string ty = p.get_str("type"), name = p.get_str("name");
        p.set_name("camera-" + ty + "-" + name);
        if (ty != "perspective") throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + ": unknown type '" + ty + "'");
        V3 eye = p.get_float3("eye"), fwd = p.get_float3("forward"), up = p.get_float3("up");
        float fov = p.get_float("fov") * 3.14159265358979323846f / 180.0f;
        if (hs.camera_index.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated camera name '" + name + "'");
        add_camera(name, eye, fwd, up, fov);
        p.check_unused();
    }
    // PerspectiveCamera::new (src/camera/perspective.rs:15-27), fov in radians
    void add_camera(const std::string& name, V3 eye, V3 fwd, V3 up, float fov) {
        spt_camera c;
        V3 f = normalize(fwd);
        V3 r = normalize(cross(f, up));
        V3 u = cross(r, f);
        c.eye[0] = eye.x; c.eye[1] = eye.y; c.eye[2] = eye.z;
        c.forward[0] = f.x; c.forward[1] = f.y; c.forward[2] = f.z;
        c.up[0] = u.x; c.up[1] = u.y; c.up[2] = u.z;
        c.right[0] = r.x; c.right[1] = r.y; c.right[2] = r.z;
        c.half_cot_half_fov = 0.5f / std::tan(fov * 0.5f);
        hs.camera_index[name] = hs.cameras.size();
        hs.cameras.push_back(c);
    }

    // texture::create_texture_from_params (src/texture/mod.rs:210-243)
    void load_texture(Params& p) {
        p.set_name("texture");
        std::string ty = p.get_str("type"), name = p.get_str("name");
        p.set_name("texture-" + ty + "-" + name);
        TexInfo info;
        info.constant = true;
        info.value = {0, 0, 0};
        if (ty == "scalar") {
            info.value = p.get_float3("value");
            info.node = add_tex_node(SPT_TEX_SCALAR);
            hs.textures[info.node].value[0] = info.value.x;
            hs.textures[info.node].value[1] = info.value.y;
            hs.textures[info.node].value[2] = info.value.z;
        } else if (ty == "add" || ty == "sub" || ty == "mul" || ty == "div") {
            const TexInfo ta = tex_info(p.get_str("t1")), tb = tex_info(p.get_str("t2"));
            V3 a = ta.value, b = tb.value;
            uint32_t op;
            if (ty == "add") { info.value = a + b; op = SPT_TEX_ADD; }
            else if (ty == "sub") { info.value = a - b; op = SPT_TEX_SUB; }
            else if (ty == "mul") { info.value = a * b; op = SPT_TEX_MUL; }
            else { info.value = {a.x / b.x, a.y / b.y, a.z / b.z}; op = SPT_TEX_DIV; }
            info.constant = ta.constant && tb.constant;
            info.node = add_tex_node(op, ta.node, tb.node);
        } else if (ty == "image") {
            // InputParams::get_image (src/core/loader.rs:366-371): relative to the scene file
            std::string file = with_file_name(path, p.get_str("image_file"));
            uint32_t im;
            try {
                im = add_image(file);
            } catch (const HostError& e) {
                throw HostError(e.code, p.name() + " - " + e.msg);
            }
            info.constant = false;
            info.node = add_tex_node(SPT_TEX_IMAGE);
            hs.textures[info.node].image = im;
        } else {
            throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + ": unknown type '" + ty + "'");
        }
        if (p.get_bool_or("is_srgb", false)) {
            info.value = {srgb_to_linear(info.value.x), srgb_to_linear(info.value.y), srgb_to_linear(info.value.z)};
            info.node = add_tex_node(SPT_TEX_SRGB, info.node);
        }
        // any key left (tiling / offset / mode / wrap, or a typo) wraps the texture in a TexInputModifier
        // (mod.rs:235-237, input_modifier.rs:52-71); it is the identity on a constant texture
        if (p.num_unused() > 0) {
            int32_t mode = -1, wrap = -1;
            if (p.contains("mode")) {
                std::string m = p.get_str("mode");
                if (m == "texcoords") mode = SPT_TEXMODE_TEXCOORDS;
                else if (m == "position") mode = SPT_TEXMODE_POSITION;
                else if (m == "normal") mode = SPT_TEXMODE_NORMAL;
                else if (m == "tangent") mode = SPT_TEXMODE_TANGENT;
                else if (m == "bitangent") mode = SPT_TEXMODE_BITANGENT;
                else throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + " - Unknown texture input mode '" + m + "'");
            }
            if (p.contains("wrap")) {
                std::string m = p.get_str("wrap");
                if (m == "repeat") wrap = SPT_TEXWRAP_REPEAT;
                else if (m == "mirror_repeat") wrap = SPT_TEXWRAP_MIRROR_REPEAT;
                else if (m == "clamp") wrap = SPT_TEXWRAP_CLAMP;
                else if (m == "mirror_clamp") wrap = SPT_TEXWRAP_MIRROR_CLAMP;
                else throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + " - Unknown texture input wrap mode '" + m + "'");
            }
            V3 tiling, offset;
            if (mode < 0 || mode == SPT_TEXMODE_TEXCOORDS) {
                std::array<float, 2> t2 = p.get_float2_or("tiling", {1.0f, 1.0f}), o2 = p.get_float2_or("offset", {0.0f, 0.0f});
                tiling = {t2[0], t2[1], 1.0f};
                offset = {o2[0], o2[1], 0.0f};
            } else {
                tiling = p.get_float3_or("tiling", {1, 1, 1});
                offset = p.get_float3_or("offset", {0, 0, 0});
            }
            info.node = add_tex_node(SPT_TEX_MODIFIER, info.node);
            spt_texture& t = hs.textures[info.node];
            t.mode = mode; t.wrap = wrap;
            t.tiling[0] = tiling.x; t.tiling[1] = tiling.y; t.tiling[2] = tiling.z;
            t.offset[0] = offset.x; t.offset[1] = offset.y; t.offset[2] = offset.z;
        }
        if (textures.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated texture name '" + name + "'");
        textures[name] = info;
        p.check_unused();
    }

    // material::create_material_from_params + MaterialT::bxdf_context for scalar textures
    void load_material(Params& p) {
        p.set_name("material");
        std::string ty = p.get_str("type"), name = p.get_str("name");
        p.set_name("material-" + ty + "-" + name);
        spt_material m;
        std::memset(&m, 0, sizeof m);
        auto roughness = [&](float& ax, float& ay) {
            float rx, ry;
            if (p.contains("roughness")) {
                rx = ry = texture(p.get_str("roughness")).x;
            } else {
                rx = texture(p.get_str("roughness_x")).x;
                ry = texture(p.get_str("roughness_y")).x;
            }
            ax = rx * rx;  // powi(2)
            ay = ry * ry;
        };
        if (ty == "lambert") {
            V3 a = texture(p.get_str("albedo"));
            m.bxdf = SPT_BXDF_LAMBERT;
            m.c0[0] = a.x; m.c0[1] = a.y; m.c0[2] = a.z;
        } else if (ty == "conductor") {
            V3 ior = texture(p.get_str("ior")), k = texture(p.get_str("ior_k"));
            roughness(m.ax, m.ay);
            m.bxdf = (m.ax < 0.0001f || m.ay < 0.0001f) ? SPT_BXDF_SPECULAR_CONDUCTOR : SPT_BXDF_MICROFACET_CONDUCTOR;
            m.c0[0] = ior.x; m.c0[1] = ior.y; m.c0[2] = ior.z;
            m.c1[0] = k.x; m.c1[1] = k.y; m.c1[2] = k.z;
        } else if (ty == "dielectric") {
            float int_ior = p.get_float("int_ior");
            float ext_ior = p.get_float_or("ext_ior", 1.0f);
            (void)texture(p.get_str("reflectance"));    // loaded, never used (dielectric.rs:67-68)
            (void)texture(p.get_str("transmittance"));
            roughness(m.ax, m.ay);
            m.ior = int_ior / ext_ior;
            m.bxdf = (m.ax < 0.0001f || m.ay < 0.0001f) ? SPT_BXDF_SPECULAR_DIELECTRIC : SPT_BXDF_MICROFACET_DIELECTRIC;
        } else if (ty == "pseudo") {
            m.bxdf = SPT_BXDF_PSEUDO;
        } else if (ty == "plastic") {
            // src/material/plastic.rs:49-85: roughness is NOT squared here; Diffuse substrate + DielectricFresnel
            float int_ior = p.get_float("int_ior");
            float ext_ior = p.get_float_or("ext_ior", 1.0f);
            V3 albedo = texture(p.get_str("albedo"));
            float rx, ry;
            if (p.contains("roughness")) {
                rx = ry = texture(p.get_str("roughness")).x;
            } else {
                rx = texture(p.get_str("roughness_x")).x;
                ry = texture(p.get_str("roughness_y")).x;
            }
            m.ior = int_ior / ext_ior;
            m.ax = rx; m.ay = ry;
            m.bxdf = (rx < 0.0001f || ry < 0.0001f) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
            m.fresnel = SPT_FRESNEL_DIELECTRIC;
            m.substrate = SPT_SUBSTRATE_DIFFUSE;
            m.c0[0] = albedo.x; m.c0[1] = albedo.y; m.c0[2] = albedo.z;
            // Diffuse::new (src/bxdf/substrate.rs:127-137)
            float fdr = 2.0f * fresnel_moment1(1.0f / m.ior);
            V3 num = albedo * 0.318309886183790671538f;
            V3 den = ((V3{1, 1, 1} - albedo * fdr) * m.ior) * m.ior;
            m.c2[0] = num.x / den.x; m.c2[1] = num.y / den.y; m.c2[2] = num.z / den.z;
        } else if (ty == "pbr_metallic" || ty == "pbr_specular") {
            // src/material/pbr_metallic.rs:76-104, pbr_specular.rs:66-92: Lambert substrate + SchlickFresnel
            V3 diffuse, specular;
            if (ty == "pbr_metallic") {
                V3 base = texture(p.get_str("base_color"));
                roughness(m.ax, m.ay);
                float metallic = texture(p.get_str("metallic")).x;
                specular = base * metallic + V3{0.04f, 0.04f, 0.04f} * (1.0f - metallic);
                diffuse = base * (1.0f - metallic);
            } else {
                diffuse = texture(p.get_str("diffuse"));
                specular = texture(p.get_str("specular"));
                roughness(m.ax, m.ay);
            }
            m.bxdf = (m.ax < 0.0001f || m.ay < 0.0001f) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
            m.fresnel = SPT_FRESNEL_SCHLICK;
            m.substrate = SPT_SUBSTRATE_LAMBERT;
            m.c0[0] = diffuse.x; m.c0[1] = diffuse.y; m.c0[2] = diffuse.z;
            m.c1[0] = specular.x; m.c1[1] = specular.y; m.c1[2] = specular.z;
        } else if (ty == "subsurface") {
            // src/material/subsurface.rs:66-93: roughness squared; bxdf::Subsurface::new (substrate.rs:199-211) over a
            // dielectric-Fresnel coat
            float int_ior = p.get_float("int_ior");
            float ext_ior = p.get_float_or("ext_ior", 1.0f);
            V3 albedo = texture(p.get_str("albedo"));
            float ld = texture(p.get_str("ld")).x;
            roughness(m.ax, m.ay);
            m.ior = int_ior / ext_ior;
            m.bxdf = (m.ax < 0.0001f || m.ay < 0.0001f) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
            m.fresnel = SPT_FRESNEL_DIELECTRIC;
            m.substrate = SPT_SUBSTRATE_SUBSURFACE;
            m.c0[0] = albedo.x; m.c0[1] = albedo.y; m.c0[2] = albedo.z;
            float fdr = 2.0f * fresnel_moment1(1.0f / m.ior);
            V3 num = albedo * 0.318309886183790671538f;
            V3 den = ((V3{1, 1, 1} - albedo * fdr) * m.ior) * m.ior;
            m.c2[0] = num.x / den.x; m.c2[1] = num.y / den.y; m.c2[2] = num.z / den.z;
            const float a[3] = {albedo.x, albedo.y, albedo.z};
            for (int k = 0; k < 3; ++k) {
                float q = a[k] - 0.33f, q2 = q * q;   // powi(4)
                m.c1[k] = ld / (3.5f + 100.0f * (q2 * q2));
            }
        } else if (ty == "pndf_conductor" || ty == "pndf_plastic") {
            // PndfConductor::load / ::new (src/material/pndf_conductor.rs:31-154), PndfPlastic (pndf_plastic.rs:30-161).
            // The constants of `m` are the fallback the material takes when a hit has no pixel footprint
            // (pndf_conductor.rs:181-194, pndf_plastic.rs:188-209) at the textures' averages; the recipe below is what
            // every hit evaluates
            const bool plastic = ty == "pndf_plastic";
            float ior = 0.0f;
            if (plastic) {
                const float int_ior = p.get_float("int_ior");
                ior = int_ior / p.get_float_or("ext_ior", 1.0f);
            }
            V3 albedo = texture(p.get_str("albedo"));
            const float fr = texture(p.get_str("fallback_roughness")).x;
            m.ax = m.ay = fr * fr;
            m.c0[0] = albedo.x; m.c0[1] = albedo.y; m.c0[2] = albedo.z;
            if (plastic) {
                m.ior = ior;
                m.bxdf = (m.ax < 0.0001f) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
                m.fresnel = SPT_FRESNEL_DIELECTRIC;
                m.substrate = SPT_SUBSTRATE_DIFFUSE;
                const float fdr = 2.0f * fresnel_moment1(1.0f / m.ior);   // Diffuse::new (src/bxdf/substrate.rs:127-137)
                const V3 num = albedo * 0.318309886183790671538f;
                const V3 den = ((V3{1, 1, 1} - albedo * fdr) * m.ior) * m.ior;
                m.c2[0] = num.x / den.x; m.c2[1] = num.y / den.y; m.c2[2] = num.z / den.z;
            } else {
                m.bxdf = (m.ax < 0.0001f) ? SPT_BXDF_SPECULAR_CONDUCTOR : SPT_BXDF_MICROFACET_CONDUCTOR;
                m.fresnel = SPT_FRESNEL_SCHLICK;
            }
            spt_material_recipe r;
            std::memset(&r, 0, sizeof r);
            r.type = plastic ? SPT_MAT_PNDF_PLASTIC : SPT_MAT_PNDF_CONDUCTOR;
            r.ior = ior;
            r.tex[0] = tex_info(p.get_str("albedo")).node;
            r.tex[1] = build_pndf(hs, tex_info(p.get_str("base_normal")).node, p.get_float("sigma_r"), p.get_float("h"), p.name());
            r.tex[2] = tex_info(p.get_str("fallback_roughness")).node;
            r.tex[3] = r.tex[2];
            r.rough_chan = SPT_CHAN_R;
            hs.material_recipes.push_back(r);
            m.recipe = (uint32_t)hs.material_recipes.size();
            any_textured = true;
            if (materials.count(name) || unsupported_materials.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated material name '" + name + "'");
            materials[name] = (uint32_t)hs.materials.size();
            hs.materials.push_back(m);
            p.check_unused();
            return;
        } else {
            throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + ": unknown type '" + ty + "'");
        }
        // a parameter backed by an image: keep the recipe, the device evaluates bxdf_context at every hit
        {
            spt_material_recipe r;
            std::memset(&r, 0, sizeof r);
            bool textured = false;
            auto slot = [&](int i, const char* key) {
                const TexInfo& t = tex_info(p.get_str(key));
                r.tex[i] = t.node;
                textured = textured || !t.constant;
            };
            auto rough_slots = [&]() {
                if (p.contains("roughness")) { slot(2, "roughness"); r.tex[3] = r.tex[2]; }
                else { slot(2, "roughness_x"); slot(3, "roughness_y"); }
            };
            r.rough_chan = SPT_CHAN_R;
            r.metal_chan = SPT_CHAN_R;
            r.ior = m.ior;
            if (ty == "lambert") { r.type = SPT_MAT_LAMBERT; slot(0, "albedo"); }
            else if (ty == "conductor") { r.type = SPT_MAT_CONDUCTOR; slot(0, "ior"); slot(1, "ior_k"); rough_slots(); }
            else if (ty == "dielectric") { r.type = SPT_MAT_DIELECTRIC; rough_slots(); }
            else if (ty == "plastic") { r.type = SPT_MAT_PLASTIC; slot(0, "albedo"); rough_slots(); }
            else if (ty == "pbr_metallic") { r.type = SPT_MAT_PBR_METALLIC; slot(0, "base_color"); slot(1, "metallic"); rough_slots(); }
            else if (ty == "pbr_specular") { r.type = SPT_MAT_PBR_SPECULAR; slot(0, "diffuse"); slot(1, "specular"); rough_slots(); }
            else if (ty == "subsurface") { r.type = SPT_MAT_SUBSURFACE; slot(0, "albedo"); slot(1, "ld"); rough_slots(); }
            if (textured) {
                hs.material_recipes.push_back(r);
                m.recipe = (uint32_t)hs.material_recipes.size();
                any_textured = true;
            }
        }
        if (materials.count(name) || unsupported_materials.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated material name '" + name + "'");
        materials[name] = (uint32_t)hs.materials.size();
        hs.materials.push_back(m);
        p.check_unused();
    }

    // medium::create_medium_from_params + Homogeneous::load (src/medium/homogeneous.rs:21-27):
    // sigma_s is read from the key "sigma_a" (reference quirk Q4, replicated).
    void load_medium(Params& p) {
        p.set_name("medium");
        std::string ty = p.get_str("type"), name = p.get_str("name");
        p.set_name("medium-" + ty + "-" + name);
        if (ty != "homogeneous") throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + ": unknown type '" + ty + "'");
        V3 sa = p.get_float3("sigma_a");
        V3 ss = p.get_float3("sigma_a");
        float g = p.get_float("asymmetric");
        spt_medium m;
        std::memset(&m, 0, sizeof m);
        V3 st = sa + ss;
        m.sigma_t[0] = st.x; m.sigma_t[1] = st.y; m.sigma_t[2] = st.z;
        m.sigma_s[0] = ss.x; m.sigma_s[1] = ss.y; m.sigma_s[2] = ss.z;
        m.g = g;
        if (mediums.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated medium name '" + name + "'");
        mediums[name] = (uint32_t)hs.mediums.size();
        hs.mediums.push_back(m);
        p.check_unused();
    }

    // primitive::create_primitive_from_params (src/primitive/mod.rs:55-77)
    void load_primitive(Params& p) {
        p.set_name("primitive");
        std::string ty = p.get_str("type"), name = p.get_str("name");
        p.set_name("primitive-" + ty + "-" + name);
        PrimRec rec;
        if (ty == "sphere") {
            V3 c = p.get_float3_or("center", {0, 0, 0});
            float r = p.get_float("radius");
            spt_sphere s = {{c.x, c.y, c.z}, r};
            rec.type = SPT_PRIM_SPHERE;
            rec.id = (uint32_t)hs.spheres.size();
            rec.box.lo = c - V3{r, r, r};
            rec.box.hi = c + V3{r, r, r};
            hs.spheres.push_back(s);
        } else if (ty == "trimesh") {
            ObjMesh m = load_obj(p.get_file_path("obj_file"));
            calc_tangents(m);
            rec.type = SPT_PRIM_MESH;
            rec.id = add_mesh(m, rec.box);
        } else if (ty == "cubic_bezier") {
            // CubicBezier::load / ::new (src/primitive/bezier.rs:24-38, 136-148): 4 x 4 control points, the box of the hull
            spt_bezier_patch bp;
            std::memset(&bp, 0, sizeof bp);
            p.get_float_3darray("control_points", 4, 4, 3, &bp.cp[0][0][0], 4);
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) rec.box.grow(V3{bp.cp[i][j][0], bp.cp[i][j][1], bp.cp[i][j][2]});
            rec.type = SPT_PRIM_BEZIER;
            rec.id = (uint32_t)hs.bezier_patches.size();
            hs.bezier_patches.push_back(bp);
        } else if (ty == "catmull_clark") {
            // CatmullClark::load (src/primitive/catmull.rs:93-101).  The control mesh is read and subdivided when the first
            // instance uses the primitive (catmull.cpp): the reference's scene files pull whole primitive libraries in
            CatmullRec cr;
            cr.ply_path = p.get_file_path("ply_file");
            cr.fas_times = p.contains("fas_times") ? (uint32_t)std::max(0, p.get_int("fas_times")) : 4u;
            cr.label = p.name();
            rec.type = kPrimCatmullClark;
            rec.id = (uint32_t)catmulls.size();
            catmulls.push_back(cr);
        } else {
            throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + ": unknown type '" + ty + "'");
        }
        if (prims.count(name) || unsupported_prims.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated primitive name '" + name + "'");
        prims[name] = rec;
        p.check_unused();
    }

    // TriMesh::new (src/primitive/triangle.rs:42-55): triangles -> BLAS(4, 16), leaf order
    uint32_t add_mesh(const ObjMesh& m, Box& root_box) {
        uint32_t ntri = (uint32_t)(m.idx.size() / 3);
        if (ntri == 0) throw HostError(SPT_HOST_ERR_SCHEMA, "trimesh without triangles");
        std::vector<Box> boxes(ntri);
        for (uint32_t t = 0; t < ntri; ++t)
            for (int k = 0; k < 3; ++k) boxes[t].grow(m.pos[m.idx[3 * t + k]]);
        BvhBuildResult bvh = BvhBuilder(boxes, 4, 16).build();
        spt_mesh mesh;
        mesh.root = (uint32_t)hs.blas_nodes.size();
        mesh.node_count = (uint32_t)bvh.nodes.size();
        mesh.tri_first = (uint32_t)hs.tri_pos.size();
        mesh.tri_count = ntri;
        for (auto nd : bvh.nodes) {
            if (nd.b & SPT_LEAF_FLAG) nd.a += mesh.tri_first;
            else { nd.a += mesh.root; nd.b += mesh.root; }
            hs.blas_nodes.push_back(nd);
        }
        for (uint32_t k = 0; k < ntri; ++k) {
            uint32_t t = bvh.order[k];
            spt_tri_pos tp;
            spt_tri_attr ta;
            std::memset(&tp, 0, sizeof tp);
            std::memset(&ta, 0, sizeof ta);
            float* pp[3] = {tp.p0, tp.p1, tp.p2};
            for (int c = 0; c < 3; ++c) {
                uint32_t vi = m.idx[3 * t + c];
                pp[c][0] = m.pos[vi].x; pp[c][1] = m.pos[vi].y; pp[c][2] = m.pos[vi].z;
                ta.n[c][0] = m.nrm[vi].x; ta.n[c][1] = m.nrm[vi].y; ta.n[c][2] = m.nrm[vi].z;
                ta.t[c][0] = m.tan[vi].x; ta.t[c][1] = m.tan[vi].y; ta.t[c][2] = m.tan[vi].z;
                ta.b[c][0] = m.bit[vi].x; ta.b[c][1] = m.bit[vi].y; ta.b[c][2] = m.bit[vi].z;
                ta.uv[c][0] = m.uv[2 * vi]; ta.uv[c][1] = m.uv[2 * vi + 1];
            }
            hs.tri_pos.push_back(tp);
            hs.tri_attr.push_back(ta);
        }
        const spt_bvh_node& r = hs.blas_nodes[mesh.root];
        root_box.lo = {r.bmin[0], r.bmin[1], r.bmin[2]};
        root_box.hi = {r.bmax[0], r.bmax[1], r.bmax[2]};
        hs.meshes.push_back(mesh);
        return (uint32_t)hs.meshes.size() - 1;
    }

    uint32_t add_surface(uint32_t material, V3 emissive, bool double_sided, int32_t medium) {
        spt_surface s;
        std::memset(&s, 0, sizeof s);
        s.material = material;
        s.flags = double_sided ? (uint32_t)SPT_SURF_DOUBLE_SIDED : 0u;
        s.inside_medium = medium;
        s.emissive[0] = emissive.x; s.emissive[1] = emissive.y; s.emissive[2] = emissive.z;
        hs.surfaces.push_back(s);
        avg_emissive.push_back(emissive);
        return (uint32_t)hs.surfaces.size() - 1;
    }

    // Surface::load (src/core/surface.rs:117-164)
    void load_surface(Params& p) {
        p.set_name("surface");
        std::string name = p.get_str("name");
        p.set_name("surface-" + name);
        std::string mat = p.get_str("material");
        if (unsupported_materials.count(mat)) throw HostError(SPT_HOST_ERR_UNSUPPORTED, unsupported_materials[mat]);
        auto mi = materials.find(mat);
        if (mi == materials.end()) throw HostError(SPT_HOST_ERR_SCHEMA, "There is no material named '" + mat + "'");
        uint32_t normal_map = 0, emissive_map = 0;   // texture node + 1
        if (p.contains("normal_map")) normal_map = tex_info(p.get_str("normal_map")).node + 1;
        if (p.contains("displacement_map")) (void)texture(p.get_str("displacement_map"));  // loaded, unused (surface.rs:17)
        V3 em = p.get_float3_or("emissive", {0, 0, 0});
        V3 em_avg = em;                              // Surface::average_emissive (surface.rs:57-63)
        if (p.contains("emissive_map")) {
            std::string en = p.get_str("emissive_map");
            emissive_map = tex_info(en).node + 1;
            em_avg = em * texture(en);
        }
        bool ds = p.get_bool_or("double_sided", false);
        int32_t med = -1;
        if (p.contains("inside_medium")) {
            std::string mn = p.get_str("inside_medium");
            auto it = mediums.find(mn);
            if (it == mediums.end()) throw HostError(SPT_HOST_ERR_SCHEMA, "There is no medium named '" + mn + "'");
            med = (int32_t)it->second;
        }
        if (surfaces.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated surface name '" + name + "'");
        uint32_t si = add_surface(mi->second, em, ds, med);
        hs.surfaces[si].normal_map = normal_map;
        hs.surfaces[si].emissive_map = emissive_map;
        avg_emissive[si] = em_avg;
        if (normal_map || emissive_map) any_textured = true;
        surfaces[name] = si;
        p.check_unused();
    }

    // Instance::load + Instance::new (src/primitive/instance.rs:19-85)
    void load_instance(Params& p) {
        p.set_name("instance");
        std::string name = p.get_str("name");
        p.set_name("instance-" + name);
        const float deg = 3.14159265358979323846f / 180.0f;
        Affine trans;
        if (p.contains("matrix")) trans = p.get_matrix("matrix");
        if (p.contains("scale")) trans = from_scale(p.get_float3("scale")) * trans;
        if (p.contains("rotate")) {
            V3 r = p.get_float3("rotate");
            trans = from_rotation_z(r.z * deg) * from_rotation_x(r.x * deg) * from_rotation_y(r.y * deg) * trans;
        }
        if (p.contains("translate")) trans = from_translation(p.get_float3("translate")) * trans;
        if (determinant(trans.m) == 0.0f) warn(p.name() + ": transform matrix is singular");
        uint32_t surf;
        if (p.contains("surface")) {
            std::string sn = p.get_str("surface");
            auto it = surfaces.find(sn);
            if (it == surfaces.end()) throw HostError(SPT_HOST_ERR_SCHEMA, "There is no surface named '" + sn + "'");
            surf = it->second;
        } else {
            std::string mn = p.get_str("material");
            if (unsupported_materials.count(mn)) throw HostError(SPT_HOST_ERR_UNSUPPORTED, unsupported_materials[mn]);
            auto it = materials.find(mn);
            if (it == materials.end()) throw HostError(SPT_HOST_ERR_SCHEMA, "There is no material named '" + mn + "'");
            surf = add_surface(it->second, {0, 0, 0}, false, -1);
        }
        std::string pn = p.get_str("primitive");
        if (unsupported_prims.count(pn)) throw HostError(SPT_HOST_ERR_UNSUPPORTED, unsupported_prims[pn]);
        auto pi = prims.find(pn);
        if (pi == prims.end()) throw HostError(SPT_HOST_ERR_SCHEMA, "There is no primitive named '" + pn + "'");
        if (instances.count(name) || catmull_instances.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated instance name '" + name + "'");
        if (pi->second.type == kPrimCatmullClark) {
            // CatmullClark = BvhAccel<CubicBezier> (catmull.rs:82-85, 445): every patch becomes an instance of its own with this
            // instance's transform and surface, and the scene's TLAS is that BVH.  (Instance order is name order: the
            // patches of one primitive stay together, zero-padded.)
            CatmullRec& cr = catmulls[pi->second.id];
            if (!cr.loaded) {
                const std::vector<float> cps = catmull_clark_patches(cr.ply_path, cr.fas_times);
                cr.first_patch = (uint32_t)hs.bezier_patches.size();
                cr.n_patches = (uint32_t)(cps.size() / 48);
                if (cr.n_patches == 0) throw HostError(SPT_HOST_ERR_SCHEMA, cr.label + ": the control mesh yields no patch");
                for (uint32_t k = 0; k < cr.n_patches; ++k) {
                    spt_bezier_patch bp;
                    std::memset(&bp, 0, sizeof bp);
                    for (int i = 0; i < 4; ++i)
                        for (int j = 0; j < 4; ++j)
                            for (int c = 0; c < 3; ++c) bp.cp[i][j][c] = cps[(size_t)k * 48 + (size_t)(i * 4 + j) * 3 + (size_t)c];
                    hs.bezier_patches.push_back(bp);
                }
                cr.loaded = true;
            }
            if (luminance(V3{hs.surfaces[surf].emissive[0], hs.surfaces[surf].emissive[1], hs.surfaces[surf].emissive[2]}) > 0.0f)
                throw HostError(SPT_HOST_ERR_UNSUPPORTED, p.name() + ": an emissive surface on a catmull_clark primitive (CatmullClark::sample / surface_area are unimplemented in the reference, catmull.rs:120-130)");
            char suffix[32];
            for (uint32_t k = 0; k < cr.n_patches; ++k) {
                const spt_bezier_patch& bp = hs.bezier_patches[cr.first_patch + k];
                PrimRec pr;
                pr.type = SPT_PRIM_BEZIER;
                pr.id = cr.first_patch + k;
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) pr.box.grow(V3{bp.cp[i][j][0], bp.cp[i][j][1], bp.cp[i][j][2]});
                std::snprintf(suffix, sizeof suffix, "\x01patch%07u", k);
                const std::string pname = name + suffix;
                instances[pname] = make_instance(pname, trans, pr, surf);
            }
            catmull_instances.insert(name);
        } else {
            instances[name] = make_instance(name, trans, pi->second, surf);
        }
        p.check_unused();
    }

    // Instance::new (src/primitive/instance.rs:53-85)
    InstRec make_instance(const std::string& name, const Affine& trans, const PrimRec& prim, uint32_t surf) {
        InstRec rec;
        rec.name = name;
        rec.trans = trans;
        spt_instance& in = rec.inst;
        std::memset(&in, 0, sizeof in);
        Affine inv = inverse(trans);
        M3 it3 = transpose(inv.m);  // Transform::new: trans_it = inverse.matrix3.transpose()
        auto put = [](float* dst, const Affine& a) {
            const V3 cols[4] = {a.m.c0, a.m.c1, a.m.c2, a.t};
            for (int c = 0; c < 4; ++c) { dst[3 * c] = cols[c].x; dst[3 * c + 1] = cols[c].y; dst[3 * c + 2] = cols[c].z; }
        };
        put(in.inv, inv);
        put(in.fwd, trans);
        const V3 nc[3] = {it3.c0, it3.c1, it3.c2};
        for (int c = 0; c < 3; ++c) { in.nrm[3 * c] = nc[c].x; in.nrm[3 * c + 1] = nc[c].y; in.nrm[3 * c + 2] = nc[c].z; }
        in.prim_type = prim.type;
        in.prim_id = prim.id;
        in.surface = surf;
        in.light = -1;
        // Bbox::transformed_by (src/core/bbox.rs:40-61): the 8 corners
        const Box& pb = prim.box;
        Box wb;
        for (int c = 0; c < 8; ++c)
            wb.grow(trans.point({(c & 4) ? pb.hi.x : pb.lo.x, (c & 2) ? pb.hi.y : pb.lo.y, (c & 1) ? pb.hi.z : pb.lo.z}));
        in.bmin[0] = wb.lo.x; in.bmin[1] = wb.lo.y; in.bmin[2] = wb.lo.z;
        in.bmax[0] = wb.hi.x; in.bmax[1] = wb.hi.y; in.bmax[2] = wb.hi.z;
        return rec;
    }

    // light::create_light_from_params (src/light/mod.rs:37-59)
    void load_light(Params& p) {
        p.set_name("light");
        std::string ty = p.get_str("type"), name = p.get_str("name");
        p.set_name("light-" + ty + "-" + name);
        spt_light l;
        std::memset(&l, 0, sizeof l);
        auto set3 = [](float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
        if (ty == "directional") {
            l.type = SPT_LIGHT_DIRECTIONAL;
            V3 d = p.get_float3("direction"), s = p.get_float3("strength");
            set3(l.dir, normalize(d));  // DirLight::new normalises
            set3(l.strength, s);
        } else if (ty == "point") {
            l.type = SPT_LIGHT_POINT;
            set3(l.pos, p.get_float3("position"));
            set3(l.strength, p.get_float3("strength"));
        } else if (ty == "spot") {
            l.type = SPT_LIGHT_SPOT;
            const float deg = 3.14159265358979323846f / 180.0f;
            set3(l.pos, p.get_float3("position"));
            set3(l.dir, p.get_float3("direction"));  // NOT normalised (spot.rs:18-31)
            float inner = p.get_float_or("inner_angle", 0.0f) * deg;
            float outer = p.get_float_or("outer_angle", 90.0f) * deg;
            l.cos_inner = std::cos(inner);
            l.cos_outer = std::cos(outer);
            set3(l.strength, p.get_float3("strength"));
        } else {
            throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + ": unknown type '" + ty + "'");
        }
        l.power = luminance({l.strength[0], l.strength[1], l.strength[2]});
        if (lights.count(name)) throw HostError(SPT_HOST_ERR_SCHEMA, "Duplicated light name '" + name + "'");
        lights[name] = l;
        p.check_unused();
    }

    // EnvLight::load / ::new (src/light/environment.rs:19-49, 86-107)
    void load_env(Params& p) {
        p.set_name("environment");
        std::string ty = p.get_str("type");
        V3 scale = p.get_float3_or("scale", {1, 1, 1});
        uint32_t w = 0, h = 0;
        std::vector<float> tex;
        if (ty == "color") {
            V3 c = p.get_float3("color");
            w = h = 1;
            tex = {c.x, c.y, c.z};
        } else if (ty == "exr") {
            std::string file = p.get_file_path("exr_file");
            float* data = nullptr;
            spt_status st = spt_host_read_exr(file.c_str(), &w, &h, &data);
            if (st != SPT_OK) throw HostError(st, spt_host_last_error());
            tex.assign(data, data + (size_t)w * h * 3);
            spt_host_free(data);
        } else {
            throw HostError(SPT_HOST_ERR_SCHEMA, p.name() + " - unknown type");
        }
        if (has_env) throw HostError(SPT_HOST_ERR_SCHEMA, "Environment has been set before");
        has_env = true;
        size_t n = (size_t)w * h;
        hs.env_texels = tex;
        hs.env_props.resize(n);
        float sum = 0.0f;
        float height_inv = 1.0f / (float)h;
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const float* px = &tex[3 * ((size_t)y * w + x)];
                float theta = ((float)y + 0.5f) * height_inv;  // NOT scaled by pi (quirk Q5)
                float prop = luminance({px[0], px[1], px[2]}) * std::sin(theta);
                sum += prop;
                hs.env_props[(size_t)y * w + x] = prop;
            }
        float sum_inv = 1.0f / sum;
        for (auto& pr : hs.env_props) pr *= sum_inv;
        float avg_power = sum / (float)n;
        build_alias(hs.env_props, hs.env_u, hs.env_k);
        hs.env_w = w; hs.env_h = h;
        hs.env_scale[0] = scale.x; hs.env_scale[1] = scale.y; hs.env_scale[2] = scale.z;
        spt_light l;
        std::memset(&l, 0, sizeof l);
        l.type = SPT_LIGHT_ENV;
        l.power = avg_power * 4.0f * 3.14159265358979323846f;
        lights["$env"] = l;  // add_environment registers it as light "$env" (scene_resources.rs:158-168)
        p.check_unused();
    }

    // AliasTable::new (src/core/alias_table.rs:7-58), same pairing order
    static void build_alias(const std::vector<float>& props, std::vector<float>& u, std::vector<uint32_t>& k) {
        size_t n = props.size();
        u.resize(n);
        k.resize(n);
        for (size_t i = 0; i < n; ++i) { u[i] = props[i] * (float)n; k[i] = (uint32_t)i; }
        const size_t NONE = (size_t)-1;
        size_t poor = NONE, rich = NONE;
        for (size_t i = 0; i < n; ++i) if (u[i] < 1.0f) { poor = i; break; }
        size_t poor_max = poor;
        for (size_t i = 0; i < n; ++i) if (u[i] > 1.0f) { rich = i; break; }
        while (poor != NONE && rich != NONE) {
            float diff = 1.0f - u[poor];
            u[rich] -= diff;
            k[poor] = (uint32_t)rich;
            if (u[rich] < 1.0f && rich < poor_max) {
                poor = rich;
            } else {
                poor = NONE;
                for (size_t i = poor_max + 1; i < n; ++i)
                    if (u[i] < 1.0f) { poor = i; poor_max = i; break; }
            }
            size_t start = rich;
            rich = NONE;
            for (size_t i = start; i < n; ++i)
                if (u[i] > 1.0f) { rich = i; break; }
        }
    }

    // PrimitiveT::surface_area for an instance under `trans` (Instance/TriMesh/Sphere ::surface_area)
    // ---- loader::gltf::load_scene_resources (src/loader/gltf.rs:19-43) -----------------------------------
    // Everything the glTF defines is resolved inside the glTF's own namespace; cameras, instances and
    // lights are then merged by name, existing (JSON) names winning (SceneResources::merge,
    // src/core/scene_resources.rs:264-310).
    struct GltfTex { uint32_t node; bool constant; V3 value; float alpha; };
    GltfTex gl_scalar(V3 v) {
        uint32_t n = add_tex_node(SPT_TEX_SCALAR);
        hs.textures[n].value[0] = v.x; hs.textures[n].value[1] = v.y; hs.textures[n].value[2] = v.z;
        return {n, true, v, 1.0f};
    }
    GltfTex gl_binary(uint32_t op, const GltfTex& a, const GltfTex& b) {
        GltfTex r;
        r.node = add_tex_node(op, a.node, b.node);
        r.constant = a.constant && b.constant;
        if (op == SPT_TEX_MUL) { r.value = a.value * b.value; r.alpha = a.alpha * b.alpha; }
        else { r.value = a.value - b.value; r.alpha = a.alpha - b.alpha; }
        return r;
    }
    GltfTex gl_srgb(const GltfTex& a) {
        GltfTex r = a;
        r.node = add_tex_node(SPT_TEX_SRGB, a.node);
        r.value = {srgb_to_linear(a.value.x), srgb_to_linear(a.value.y), srgb_to_linear(a.value.z)};
        return r;
    }
    static float gl_chan(const GltfTex& t, uint32_t chan) {
        return chan == SPT_CHAN_R ? t.value.x : (chan == SPT_CHAN_G ? t.value.y : (chan == SPT_CHAN_B ? t.value.z : t.alpha));
    }

    void import_gltf(const std::string& file) {
        // --- container: .gltf (JSON) or .glb (JSON chunk + BIN chunk)
        std::vector<uint8_t> raw = read_bytes(file);
        std::string json_text;
        std::vector<uint8_t> glb_bin;
        if (raw.size() >= 12 && std::memcmp(raw.data(), "glTF", 4) == 0) {
            size_t p = 12;
            while (p + 8 <= raw.size()) {
                uint32_t len, type;
                std::memcpy(&len, &raw[p], 4);
                std::memcpy(&type, &raw[p + 4], 4);
                if (p + 8 + (size_t)len > raw.size()) throw HostError(SPT_HOST_ERR_PARSE, "gltf '" + file + "': truncated GLB chunk");
                if (type == 0x4E4F534Au) json_text.assign((const char*)&raw[p + 8], len);
                else if (type == 0x004E4942u && glb_bin.empty()) glb_bin.assign(raw.begin() + p + 8, raw.begin() + p + 8 + len);
                p += 8 + (size_t)len;
            }
        } else {
            json_text.assign(raw.begin(), raw.end());
        }
        JsonValue doc;
        try {
            doc = JsonParser(json_text).parse();
        } catch (const std::exception& e) {
            throw HostError(SPT_HOST_ERR_PARSE, "gltf '" + file + "': " + e.what());
        }
        if (doc.kind != JsonValue::Object) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf '" + file + "': top level must be an object");
        auto arr = [&](const JsonValue& o, const char* key) -> const std::vector<JsonValue>& {
            static const std::vector<JsonValue> empty;
            const JsonValue* v = o.get(key);
            return (v && v->kind == JsonValue::Array) ? v->arr : empty;
        };
        auto index_of = [&](const JsonValue& o, const char* key) -> int64_t {
            const JsonValue* v = o.get(key);
            return (v && v->kind == JsonValue::Int) ? v->i : -1;
        };
        // --- buffers
        std::vector<std::vector<uint8_t>> buffers;
        for (const JsonValue& b : arr(doc, "buffers")) {
            const JsonValue* uri = b.get("uri");
            if (uri && uri->kind == JsonValue::String) buffers.push_back(load_uri(uri->s, file));
            else buffers.push_back(glb_bin);
        }
        const auto& views = arr(doc, "bufferViews");
        const auto& accessors = arr(doc, "accessors");
        struct Acc { const uint8_t* data; size_t count; int64_t comp; std::string type; size_t elem; };
        // get_data_of_accessor (gltf.rs:448-460): the accessor's elements are read tightly packed
        auto accessor = [&](int64_t ai, const std::string& what) -> Acc {
            if (ai < 0 || (size_t)ai >= accessors.size()) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: " + what + ": accessor index out of range");
            const JsonValue& a = accessors[(size_t)ai];
            int64_t vi = index_of(a, "bufferView");
            if (vi < 0 || (size_t)vi >= views.size()) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: Accessor has no buffer view");
            const JsonValue& v = views[(size_t)vi];
            int64_t bi = index_of(v, "buffer");
            if (bi < 0 || (size_t)bi >= buffers.size()) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: buffer index out of range");
            Acc r;
            r.count = (size_t)std::max<int64_t>(0, index_of(a, "count"));
            r.comp = index_of(a, "componentType");
            const JsonValue* ty = a.get("type");
            r.type = (ty && ty->kind == JsonValue::String) ? ty->s : "";
            size_t comps = r.type == "SCALAR" ? 1 : r.type == "VEC2" ? 2 : r.type == "VEC3" ? 3 : r.type == "VEC4" ? 4 : 0;
            size_t csize = (r.comp == 5120 || r.comp == 5121) ? 1 : (r.comp == 5122 || r.comp == 5123) ? 2 : (r.comp == 5125 || r.comp == 5126) ? 4 : 0;
            r.elem = comps * csize;
            if (r.elem == 0) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: " + what + ": unsupported accessor type");
            int64_t stride = index_of(v, "byteStride");
            if (stride > 0 && (size_t)stride != r.elem)
                throw HostError(SPT_HOST_ERR_UNSUPPORTED, "gltf: " + what + ": interleaved buffer views (byteStride) are not supported");
            size_t voff = (size_t)std::max<int64_t>(0, index_of(v, "byteOffset")), vlen = (size_t)std::max<int64_t>(0, index_of(v, "byteLength"));
            size_t aoff = (size_t)std::max<int64_t>(0, index_of(a, "byteOffset"));
            const std::vector<uint8_t>& buf = buffers[(size_t)bi];
            if (voff + vlen > buf.size() || aoff + r.count * r.elem > vlen) throw HostError(SPT_HOST_ERR_PARSE, "gltf: " + what + ": accessor exceeds its buffer view");
            r.data = buf.data() + voff + aoff;
            return r;
        };
        // --- images -> ImageTex "image_<i>" (load_images, gltf.rs:50-103)
        std::vector<GltfTex> image_tex;
        {
            size_t i = 0;
            for (const JsonValue& im : arr(doc, "images")) {
                std::vector<uint8_t> bytes;
                const JsonValue* uri = im.get("uri");
                if (uri && uri->kind == JsonValue::String) {
                    bytes = load_uri(uri->s, file);
                } else {
                    int64_t vi = index_of(im, "bufferView");
                    if (vi < 0 || (size_t)vi >= views.size()) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: image without uri or bufferView");
                    const JsonValue& v = views[(size_t)vi];
                    int64_t bi = index_of(v, "buffer");
                    size_t off = (size_t)std::max<int64_t>(0, index_of(v, "byteOffset")), len = (size_t)std::max<int64_t>(0, index_of(v, "byteLength"));
                    if (bi < 0 || (size_t)bi >= buffers.size() || off + len > buffers[(size_t)bi].size()) throw HostError(SPT_HOST_ERR_PARSE, "gltf: image buffer view out of range");
                    bytes.assign(buffers[(size_t)bi].begin() + off, buffers[(size_t)bi].begin() + off + len);
                }
                uint32_t w = 0, h = 0;
                std::vector<uint32_t> px;
                decode_png_rgba8(bytes, file + "#image_" + std::to_string(i), &w, &h, &px);
                uint32_t id = add_image_pixels(file + "#image_" + std::to_string(i), w, h, std::move(px));
                GltfTex t;
                t.node = add_tex_node(SPT_TEX_IMAGE);
                hs.textures[t.node].image = id;
                t.constant = false;
                t.value = tex_average(t.node);   // the folded constants of an image-backed material hold the averages
                t.alpha = 1.0f;
                image_tex.push_back(t);
                ++i;
            }
        }
        const GltfTex scalar_one = gl_scalar({1, 1, 1});
        // `image_{texture.index()}`: the reference looks the IMAGE up with the TEXTURE's index (gltf.rs:127,192,...)
        auto image_of = [&](const JsonValue& mat_part, const char* key, bool* present) -> GltfTex {
            const JsonValue* t = mat_part.get(key);
            *present = t && t->kind == JsonValue::Object;
            if (!*present) return scalar_one;
            int64_t ti = index_of(*t, "index");
            if (ti < 0 || (size_t)ti >= image_tex.size()) throw HostError(SPT_HOST_ERR_SCHEMA, "There is no texture named 'image_" + std::to_string(ti) + "'");
            return image_tex[(size_t)ti];
        };
        // --- materials -> one Surface per glTF material (load_materials, gltf.rs:105-253)
        std::vector<uint32_t> mat_surface;
        for (const JsonValue& gm : arr(doc, "materials")) {
            spt_material m;
            std::memset(&m, 0, sizeof m);
            spt_material_recipe r;
            std::memset(&r, 0, sizeof r);
            GltfTex t0, t1, trough;
            const JsonValue* ext = gm.get("extensions");
            const JsonValue* sg = (ext && ext->kind == JsonValue::Object) ? ext->get("KHR_materials_pbrSpecularGlossiness") : nullptr;
            bool has;
            if (sg && sg->kind == JsonValue::Object) {
                float df[4] = {1, 1, 1, 1}, sf[3] = {1, 1, 1};
                jvec(sg->get("diffuseFactor"), 4, df);
                jvec(sg->get("specularFactor"), 3, sf);
                const float gloss = (float)jnum(sg->get("glossinessFactor"), 1.0);
                GltfTex dimg = image_of(*sg, "diffuseTexture", &has);
                t0 = gl_scalar({df[0], df[1], df[2]});
                if (has) t0 = gl_binary(SPT_TEX_MUL, t0, gl_srgb(dimg));
                GltfTex simg = image_of(*sg, "specularGlossinessTexture", &has);
                t1 = gl_scalar({sf[0], sf[1], sf[2]});
                if (has) {
                    t1 = gl_binary(SPT_TEX_MUL, t1, gl_srgb(simg));
                    trough = gl_binary(SPT_TEX_SUB, scalar_one, gl_binary(SPT_TEX_MUL, gl_scalar({gloss, gloss, gloss}), simg));
                    r.rough_chan = SPT_CHAN_A;
                } else {
                    trough = gl_scalar({1.0f - gloss, 1.0f - gloss, 1.0f - gloss});
                    r.rough_chan = SPT_CHAN_R;
                }
                r.type = SPT_MAT_PBR_SPECULAR;
                r.metal_chan = SPT_CHAN_R;
                // pbr_specular.rs:60-92 at the constant values (only used when nothing is image-backed)
                float rv = gl_chan(trough, r.rough_chan);
                m.ax = m.ay = rv * rv;
                m.c0[0] = t0.value.x; m.c0[1] = t0.value.y; m.c0[2] = t0.value.z;
                m.c1[0] = t1.value.x; m.c1[1] = t1.value.y; m.c1[2] = t1.value.z;
            } else {
                static const JsonValue empty_obj = [] { JsonValue v; v.kind = JsonValue::Object; return v; }();
                const JsonValue* pm = gm.get("pbrMetallicRoughness");
                const JsonValue& mr = (pm && pm->kind == JsonValue::Object) ? *pm : empty_obj;
                float bf[4] = {1, 1, 1, 1};
                jvec(mr.get("baseColorFactor"), 4, bf);
                const float mf = (float)jnum(mr.get("metallicFactor"), 1.0), rf = (float)jnum(mr.get("roughnessFactor"), 1.0);
                GltfTex bimg = image_of(mr, "baseColorTexture", &has);
                t0 = gl_scalar({bf[0], bf[1], bf[2]});
                if (has) t0 = gl_binary(SPT_TEX_MUL, t0, gl_srgb(bimg));
                GltfTex mimg = image_of(mr, "metallicRoughnessTexture", &has);
                t1 = gl_scalar({mf, mf, mf});
                trough = gl_scalar({rf, rf, rf});
                if (has) {
                    t1 = gl_binary(SPT_TEX_MUL, t1, mimg);
                    trough = gl_binary(SPT_TEX_MUL, trough, mimg);
                }
                r.type = SPT_MAT_PBR_METALLIC;
                r.rough_chan = SPT_CHAN_G;
                r.metal_chan = SPT_CHAN_B;
                // pbr_metallic.rs:75-104 at the constant values
                float rv = gl_chan(trough, r.rough_chan), metallic = gl_chan(t1, r.metal_chan);
                m.ax = m.ay = rv * rv;
                V3 spec = t0.value * metallic + V3{0.04f, 0.04f, 0.04f} * (1.0f - metallic), diff = t0.value * (1.0f - metallic);
                m.c0[0] = diff.x; m.c0[1] = diff.y; m.c0[2] = diff.z;
                m.c1[0] = spec.x; m.c1[1] = spec.y; m.c1[2] = spec.z;
            }
            m.bxdf = (m.ax < 0.0001f || m.ay < 0.0001f) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
            m.fresnel = SPT_FRESNEL_SCHLICK;
            m.substrate = SPT_SUBSTRATE_LAMBERT;
            r.tex[0] = t0.node; r.tex[1] = t1.node; r.tex[2] = trough.node; r.tex[3] = trough.node;
            if (!(t0.constant && t1.constant && trough.constant)) {
                hs.material_recipes.push_back(r);
                m.recipe = (uint32_t)hs.material_recipes.size();
                any_textured = true;
            }
            hs.materials.push_back(m);
            float em[3] = {0, 0, 0};
            jvec(gm.get("emissiveFactor"), 3, em);
            const JsonValue* ds = gm.get("doubleSided");
            uint32_t si = add_surface((uint32_t)hs.materials.size() - 1, {em[0], em[1], em[2]}, ds && ds->kind == JsonValue::Bool && ds->b, -1);
            GltfTex et = image_of(gm, "emissiveTexture", &has);
            if (has) {
                hs.surfaces[si].emissive_map = et.node + 1;
                avg_emissive[si] = V3{em[0], em[1], em[2]} * tex_average(et.node);
                any_textured = true;
            }
            GltfTex nt = image_of(gm, "normalTexture", &has);
            if (has) { hs.surfaces[si].normal_map = nt.node + 1; any_textured = true; }
            mat_surface.push_back(si);
        }
        // --- meshes -> one TriMesh per primitive (load_primitives, gltf.rs:255-343)
        struct GPrim { PrimRec rec; int64_t material; };
        std::vector<std::vector<GPrim>> mesh_prims;
        std::vector<std::string> mesh_names;
        {
            size_t mi = 0;
            for (const JsonValue& mesh : arr(doc, "meshes")) {
                const JsonValue* nm = mesh.get("name");
                std::string mesh_name = (nm && nm->kind == JsonValue::String) ? nm->s : "mesh_" + std::to_string(mi);
                mesh_names.push_back(mesh_name);
                std::vector<GPrim> prims_of;
                size_t pi = 0;
                for (const JsonValue& prim : arr(mesh, "primitives")) {
                    const std::string prim_name = mesh_name + "_prim_" + std::to_string(pi);
                    int64_t ia = index_of(prim, "indices");
                    if (ia < 0) throw HostError(SPT_HOST_ERR_SCHEMA, "Primitives '" + prim_name + "' doesn't have indices");
                    Acc idx = accessor(ia, prim_name + " indices");
                    ObjMesh m;
                    m.idx.assign(idx.count, 0u);
                    if (idx.comp == 5125) { for (size_t k = 0; k < idx.count; ++k) std::memcpy(&m.idx[k], idx.data + 4 * k, 4); }
                    else if (idx.comp == 5123) { for (size_t k = 0; k < idx.count; ++k) { uint16_t v; std::memcpy(&v, idx.data + 2 * k, 2); m.idx[k] = v; } }
                    else throw HostError(SPT_HOST_ERR_UNSUPPORTED, "gltf: " + prim_name + ": only u16 / u32 indices are read (the reference leaves other types at 0)");
                    const JsonValue* attrs = prim.get("attributes");
                    if (!attrs || attrs->kind != JsonValue::Object || index_of(*attrs, "POSITION") < 0)
                        throw HostError(SPT_HOST_ERR_SCHEMA, "Primitive '" + prim_name + "' doesn't have positions");
                    Acc pos = accessor(index_of(*attrs, "POSITION"), prim_name + " POSITION");
                    if (pos.comp != 5126 || pos.type != "VEC3") throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: " + prim_name + ": POSITION must be f32 VEC3");
                    const size_t nv = pos.count;
                    m.pos.resize(nv);
                    // MeshVertex::default (triangle.rs:29-38)
                    m.nrm.assign(nv, V3{0, 0, 1}); m.tan.assign(nv, V3{1, 0, 0}); m.bit.assign(nv, V3{0, 1, 0}); m.uv.assign(2 * nv, 0.0f);
                    for (size_t k = 0; k < nv; ++k) std::memcpy(&m.pos[k], pos.data + 12 * k, 12);
                    for (uint32_t v : m.idx) if (v >= nv) throw HostError(SPT_HOST_ERR_PARSE, "gltf: " + prim_name + ": vertex index out of range");
                    int64_t ta = index_of(*attrs, "TEXCOORD_0");
                    if (ta >= 0) {
                        Acc uv = accessor(ta, prim_name + " TEXCOORD_0");
                        if (uv.comp != 5126 || uv.type != "VEC2" || uv.count < nv) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "gltf: " + prim_name + ": TEXCOORD_0 must be f32 VEC2");
                        std::memcpy(m.uv.data(), uv.data, 8 * nv);
                    }
                    int64_t na = index_of(*attrs, "NORMAL");
                    if (na >= 0) {
                        Acc nr = accessor(na, prim_name + " NORMAL");
                        if (nr.comp != 5126 || nr.type != "VEC3" || nr.count < nv) throw HostError(SPT_HOST_ERR_UNSUPPORTED, "gltf: " + prim_name + ": NORMAL must be f32 VEC3");
                        for (size_t k = 0; k < nv; ++k) std::memcpy(&m.nrm[k], nr.data + 12 * k, 12);
                    } else {
                        calc_normals(m);
                    }
                    calc_tangents(m);
                    GPrim g;
                    g.rec.type = SPT_PRIM_MESH;
                    g.rec.id = add_mesh(m, g.rec.box);
                    g.material = index_of(prim, "material");
                    prims_of.push_back(g);
                    ++pi;
                }
                mesh_prims.push_back(std::move(prims_of));
                ++mi;
            }
        }
        // --- node hierarchy (parse_nodes, gltf.rs:345-446)
        const auto& nodes = arr(doc, "nodes");
        const auto& cams = arr(doc, "cameras");
        const JsonValue* dext = doc.get("extensions");
        const JsonValue* lext = (dext && dext->kind == JsonValue::Object) ? dext->get("KHR_lights_punctual") : nullptr;
        static const std::vector<JsonValue> no_lights;
        const std::vector<JsonValue>& glights = (lext && lext->kind == JsonValue::Object) ? arr(*lext, "lights") : no_lights;
        std::function<void(int64_t, const M4&, int)> visit = [&](int64_t ni, const M4& parent, int depth) {
            if (ni < 0 || (size_t)ni >= nodes.size()) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: node index out of range");
            if (depth > 256) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: node hierarchy too deep (cycle?)");
            const JsonValue& node = nodes[(size_t)ni];
            M4 local;
            if (!jvec(node.get("matrix"), 16, local.m)) {
                // Transform::Decomposed -> T * R * S (gltf crate, scene::Transform::matrix)
                float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, sc[3] = {1, 1, 1};
                jvec(node.get("translation"), 3, t);
                jvec(node.get("rotation"), 4, q);
                jvec(node.get("scale"), 3, sc);
                const float x = q[0], y = q[1], z = q[2], w = q[3];
                const float r[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),
                                    2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),
                                    2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)};   // columns
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < 3; ++k) local.m[4 * c + k] = r[3 * c + k] * sc[c];
                local.m[12] = t[0]; local.m[13] = t[1]; local.m[14] = t[2];
            }
            const M4 trans = parent * local;
            const V3 col1{trans.m[4], trans.m[5], trans.m[6]}, col2{trans.m[8], trans.m[9], trans.m[10]}, col3{trans.m[12], trans.m[13], trans.m[14]};
            const JsonValue* nname = node.get("name");
            const bool named = nname && nname->kind == JsonValue::String;
            int64_t mi = index_of(node, "mesh");
            if (mi >= 0) {
                if ((size_t)mi >= mesh_prims.size()) throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: mesh index out of range");
                Affine a;   // Affine3A::from_mat4
                a.m.c0 = {trans.m[0], trans.m[1], trans.m[2]};
                a.m.c1 = col1;
                a.m.c2 = col2;
                a.t = col3;
                for (size_t k = 0; k < mesh_prims[(size_t)mi].size(); ++k) {
                    const GPrim& g = mesh_prims[(size_t)mi][k];
                    const std::string inst_name = mesh_names[(size_t)mi] + "_prim_" + std::to_string(k) + "_node_" + std::to_string(ni);
                    if (g.material < 0 || (size_t)g.material >= mat_surface.size())
                        throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: primitive '" + inst_name + "' has no material (the reference has no default material)");
                    if (!instances.count(inst_name)) instances[inst_name] = make_instance(inst_name, a, g.rec, mat_surface[(size_t)g.material]);
                }
            }
            int64_t ci = index_of(node, "camera");
            if (ci >= 0 && (size_t)ci < cams.size()) {
                const std::string cam_name = named ? nname->s : "camera_" + std::to_string(ni);
                const JsonValue* ty = cams[(size_t)ci].get("type");
                const JsonValue* persp = cams[(size_t)ci].get("perspective");
                if (ty && ty->kind == JsonValue::String && ty->s == "perspective" && persp && persp->kind == JsonValue::Object) {
                    if (!hs.camera_index.count(cam_name)) add_camera(cam_name, col3, -col2, col1, (float)jnum(persp->get("yfov"), 1.0));
                } else {
                    warn("Camera '" + cam_name + "' is orthographic and is not supported yet");
                }
            }
            const JsonValue* next = node.get("extensions");
            const JsonValue* nl = (next && next->kind == JsonValue::Object) ? next->get("KHR_lights_punctual") : nullptr;
            int64_t li = (nl && nl->kind == JsonValue::Object) ? index_of(*nl, "light") : -1;
            if (li >= 0 && (size_t)li < glights.size()) {
                const JsonValue& gl = glights[(size_t)li];
                const std::string light_name = named ? nname->s : "light_" + std::to_string(ni);
                float color[3] = {1, 1, 1};
                jvec(gl.get("color"), 3, color);
                const float intensity = (float)jnum(gl.get("intensity"), 1.0);
                spt_light l;
                std::memset(&l, 0, sizeof l);
                auto set3 = [](float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
                set3(l.strength, V3{color[0], color[1], color[2]} * intensity);
                const JsonValue* kind = gl.get("type");
                const std::string k = (kind && kind->kind == JsonValue::String) ? kind->s : "";
                if (k == "directional") {
                    l.type = SPT_LIGHT_DIRECTIONAL;
                    set3(l.dir, normalize(-col2));   // DirLight::new normalises
                } else if (k == "point") {
                    l.type = SPT_LIGHT_POINT;
                    set3(l.pos, col3);
                } else if (k == "spot") {
                    l.type = SPT_LIGHT_SPOT;
                    set3(l.pos, col3);
                    set3(l.dir, -col2);              // SpotLight::new keeps the direction as it is
                    const JsonValue* spot = gl.get("spot");
                    l.cos_inner = std::cos((float)jnum(spot ? spot->get("innerConeAngle") : nullptr, 0.0));
                    l.cos_outer = std::cos((float)jnum(spot ? spot->get("outerConeAngle") : nullptr, 0.7853981633974483));
                } else {
                    throw HostError(SPT_HOST_ERR_SCHEMA, "gltf: light '" + light_name + "': unknown type '" + k + "'");
                }
                l.power = luminance({l.strength[0], l.strength[1], l.strength[2]});
                if (!lights.count(light_name)) lights[light_name] = l;
            }
            for (const JsonValue& ch : arr(node, "children"))
                if (ch.kind == JsonValue::Int) visit(ch.i, trans, depth + 1);
        };
        for (const JsonValue& scn : arr(doc, "scenes"))
            for (const JsonValue& root : arr(scn, "nodes"))
                if (root.kind == JsonValue::Int) visit(root.i, M4(), 0);
    }

    float instance_area(const InstRec& r) {
        const spt_instance& in = r.inst;
        if (in.prim_type == SPT_PRIM_BEZIER)   // the reference reaches `unimplemented!` (bezier.rs:188-190) when it builds the ShapeLight
            throw HostError(SPT_HOST_ERR_UNSUPPORTED, "instance '" + r.name + "': an emissive surface on a cubic_bezier primitive (CubicBezier::surface_area is unimplemented in the reference)");
        if (in.prim_type == SPT_PRIM_SPHERE) {
            float rad = hs.spheres[in.prim_id].radius * 0.5f;
            V3 v0 = r.trans.vector({-rad, -rad, -rad}), v1 = r.trans.vector({-rad, -rad, rad});
            V3 v2 = r.trans.vector({-rad, rad, -rad}), v3 = r.trans.vector({rad, -rad, -rad});
            auto d2 = [](V3 a, V3 b) { V3 d = a - b; return dot(d, d); };
            float a2 = d2(v0, v1), b2 = d2(v0, v2), c2 = d2(v0, v3);
            return 4.0f * 3.14159265358979323846f * std::sqrt((a2 * b2 + b2 * c2 + c2 * a2) / 3.0f);
        }
        const spt_mesh& m = hs.meshes[in.prim_id];
        float sum = 0.0f;
        for (uint32_t t = m.tri_first; t < m.tri_first + m.tri_count; ++t) {
            const spt_tri_pos& tp = hs.tri_pos[t];
            V3 p0 = r.trans.point({tp.p0[0], tp.p0[1], tp.p0[2]});
            V3 p1 = r.trans.point({tp.p1[0], tp.p1[1], tp.p1[2]});
            V3 p2 = r.trans.point({tp.p2[0], tp.p2[1], tp.p2[2]});
            sum += length(cross(p1 - p0, p2 - p0)) * 0.5f;
        }
        return sum;
    }

    void finish(const JsonValue& root) {
        // aggregate: "group" | "bvh" (default bvh(4,16))  (scene_resources.rs:85-103)
        uint32_t aggregate = SPT_AGGREGATE_BVH;
        if (const JsonValue* a = root.get("aggregate")) {
            if (a->kind != JsonValue::String) throw HostError(SPT_HOST_ERR_SCHEMA, "scene - 'aggregate' should be string");
            if (a->s == "group") aggregate = SPT_AGGREGATE_GROUP;
            else if (a->s == "bvh") aggregate = SPT_AGGREGATE_BVH;
            else throw HostError(SPT_HOST_ERR_SCHEMA, "Unknown aggregate type '" + a->s + "'");
        }
        uint32_t sampler = SPT_LIGHT_SAMPLER_UNIFORM;
        if (const JsonValue* a = root.get("light_sampler")) {
            if (a->kind != JsonValue::String) throw HostError(SPT_HOST_ERR_SCHEMA, "scene - 'light_sampler' should be string");
            if (a->s == "uniform") sampler = SPT_LIGHT_SAMPLER_UNIFORM;
            else if (a->s == "power_is") sampler = SPT_LIGHT_SAMPLER_POWER_IS;
            else throw HostError(SPT_HOST_ERR_SCHEMA, "Unknown light sampler type '" + a->s + "'");
        }
        if (hs.cameras.empty()) throw HostError(SPT_HOST_ERR_SCHEMA, "At least one camera is needed");

        // instances in name order, then TLAS
        std::vector<const InstRec*> recs;
        for (auto& kv : instances) recs.push_back(&kv.second);
        std::vector<Box> boxes(recs.size());
        for (size_t i = 0; i < recs.size(); ++i) {
            const spt_instance& in = recs[i]->inst;
            boxes[i].lo = {in.bmin[0], in.bmin[1], in.bmin[2]};
            boxes[i].hi = {in.bmax[0], in.bmax[1], in.bmax[2]};
        }
        std::vector<const InstRec*> ordered;
        if (aggregate == SPT_AGGREGATE_GROUP || recs.empty()) {
            spt_bvh_node root_node;
            std::memset(&root_node, 0, sizeof root_node);
            Box all;
            for (auto& b : boxes) all.grow(b);
            root_node.bmin[0] = all.lo.x; root_node.bmin[1] = all.lo.y; root_node.bmin[2] = all.lo.z;
            root_node.bmax[0] = all.hi.x; root_node.bmax[1] = all.hi.y; root_node.bmax[2] = all.hi.z;
            root_node.a = 0;
            root_node.b = SPT_LEAF_FLAG | (uint32_t)recs.size();
            hs.tlas_nodes.push_back(root_node);
            ordered = recs;
        } else {
            BvhBuildResult bvh = BvhBuilder(boxes, 4, 16).build();
            hs.tlas_nodes = bvh.nodes;
            for (uint32_t k : bvh.order) ordered.push_back(recs[k]);
        }
        std::map<std::string, uint32_t> inst_index;
        for (auto* r : ordered) {
            inst_index[r->name] = (uint32_t)hs.instances.size();
            hs.instances.push_back(r->inst);
        }

        // lights: named lights (incl. "$env") in name order, then one ShapeLight per emissive
        // instance in name order (scene_resources.rs:105-122)
        int32_t env_index = -1;
        for (auto& kv : lights) {
            if (kv.second.type == SPT_LIGHT_ENV) env_index = (int32_t)hs.lights.size();
            hs.lights.push_back(kv.second);
        }
        for (auto& kv : instances) {
            const InstRec& r = kv.second;
            // Surface::is_emissive (surface.rs:45-47) looks at the CONSTANT emissive only: a surface with emissive > 0 and an
            // all-black emissive_map is still a light (it takes a slot of the sampler and of the alias table); the map's
            // average only enters ShapeLight::power (shape_light.rs:79-82)
            const spt_surface& sfc = hs.surfaces[r.inst.surface];
            V3 em = avg_emissive[r.inst.surface];
            if (luminance(V3{sfc.emissive[0], sfc.emissive[1], sfc.emissive[2]}) > 0.0f) {
                spt_light l;
                std::memset(&l, 0, sizeof l);
                l.type = SPT_LIGHT_SHAPE;
                l.instance = inst_index[r.name];
                l.power = instance_area(r) * luminance(em);
                hs.instances[l.instance].light = (int32_t)hs.lights.size();
                hs.lights.push_back(l);
            }
        }
        hs.aggregate = aggregate;
        hs.light_sampler = sampler;
        hs.env_light_index = env_index;
        if (sampler == SPT_LIGHT_SAMPLER_POWER_IS) {
            // PowerIsLightSampler::new (src/light_sampler/power_is.rs:24-46)
            size_t n = hs.lights.size();
            hs.light_props.assign(n, 0.0f);
            float sum = 0.0f;
            for (size_t i = 0; i < n; ++i) { hs.light_props[i] = hs.lights[i].power; sum += hs.light_props[i]; }
            float sum_inv = 1.0f / sum;
            for (auto& pr : hs.light_props) pr *= sum_inv;
            build_alias(hs.light_props, hs.light_u, hs.light_k);
        }
        hs.finalize_desc();
    }
};

void HostScene::finalize_desc() {
    std::memset(&desc, 0, sizeof desc);
    desc.abi_version = SPT_ABI_VERSION;
    desc.aggregate = aggregate;
    desc.n_tlas_nodes = (uint32_t)tlas_nodes.size(); desc.tlas_nodes = tlas_nodes.data();
    desc.n_instances = (uint32_t)instances.size(); desc.instances = instances.data();
    desc.n_meshes = (uint32_t)meshes.size(); desc.meshes = meshes.data();
    desc.n_blas_nodes = (uint32_t)blas_nodes.size(); desc.blas_nodes = blas_nodes.data();
    desc.n_tris = (uint32_t)tri_pos.size(); desc.tri_pos = tri_pos.data(); desc.tri_attr = tri_attr.data();
    desc.n_spheres = (uint32_t)spheres.size(); desc.spheres = spheres.data();
    // SPT_BEZIER_NI=1: the reference built with `--features bezier_ni` (Cargo.toml:34-36) - every patch is intersected by
    // Newton's iteration instead of Bezier clipping (spt_abi.h: cp[0][0][3])
    if (const char* ni = std::getenv("SPT_BEZIER_NI"))
        for (spt_bezier_patch& bp : bezier_patches) bp.cp[0][0][3] = (ni[0] != '\0' && ni[0] != '0') ? SPT_BEZIER_NEWTON : 0.0f;
    desc.n_bezier_patches = (uint32_t)bezier_patches.size(); desc.bezier_patches = bezier_patches.data();
    desc.n_pndfs = (uint32_t)pndfs.size(); desc.pndfs = pndfs.data();
    desc.n_pndf_terms = (uint32_t)pndf_terms.size(); desc.pndf_terms = pndf_terms.data();
    desc.n_pndf_nodes = (uint32_t)pndf_nodes.size(); desc.pndf_nodes = pndf_nodes.data();
    desc.n_pndf_refs = (uint32_t)pndf_refs.size(); desc.pndf_refs = pndf_refs.data();
    desc.n_pndf_roots = (uint32_t)pndf_roots.size(); desc.pndf_roots = pndf_roots.data();
    desc.n_surfaces = (uint32_t)surfaces.size(); desc.surfaces = surfaces.data();
    desc.n_materials = (uint32_t)materials.size(); desc.materials = materials.data();
    desc.n_mediums = (uint32_t)mediums.size(); desc.mediums = mediums.data();
    desc.n_lights = (uint32_t)lights.size(); desc.lights = lights.data();
    desc.light_sampler = light_sampler;
    desc.env_light_index = env_light_index;
    desc.light_alias.n = (uint32_t)light_props.size();
    desc.light_alias.props = light_props.data();
    desc.light_alias.u = light_u.data();
    desc.light_alias.k = light_k.data();
    desc.env.width = env_w; desc.env.height = env_h;
    desc.env.texels = env_texels.data();
    for (int i = 0; i < 3; ++i) desc.env.scale[i] = env_scale[i];
    desc.env.alias.n = (uint32_t)env_props.size();
    desc.env.alias.props = env_props.data();
    desc.env.alias.u = env_u.data();
    desc.env.alias.k = env_k.data();
    desc.n_textures = (uint32_t)textures.size(); desc.textures = textures.data();
    desc.n_images = (uint32_t)images.size(); desc.images = images.data();
    desc.n_image_levels = (uint32_t)image_levels.size(); desc.image_levels = image_levels.data();
    desc.n_texels = (uint32_t)texels.size(); desc.texels = texels.data();
    desc.n_material_recipes = (uint32_t)material_recipes.size(); desc.material_recipes = material_recipes.data();
}

// loader::load_scene (src/loader/json.rs:53-199): fixed section order
HostScene* load_scene_file(const std::string& path) {
    // loader::load_scene (src/loader/mod.rs:20-31): by file extension
    const bool is_gltf = (path.size() >= 5 && path.substr(path.size() - 5) == ".gltf") || (path.size() >= 4 && path.substr(path.size() - 4) == ".glb");
    if (is_gltf) {   // gltf::load_scene (gltf.rs:13-17): to_scene(None, None) = bvh aggregate, uniform light sampler
        std::unique_ptr<HostScene> hs(new HostScene());
        SceneBuilder b(*hs, path);
        b.import_gltf(path);
        JsonValue none;
        none.kind = JsonValue::Object;
        b.finish(none);
        return hs.release();
    }
    if (path.size() < 5 || path.substr(path.size() - 5) != ".json")
        throw HostError(SPT_HOST_ERR_SCHEMA, "File extension is not recognized");
    JsonValue root = parse_json_file(path);
    if (root.kind != JsonValue::Object) throw HostError(SPT_HOST_ERR_SCHEMA, "scene - top level must be an object");
    std::unique_ptr<HostScene> hs(new HostScene());
    SceneBuilder b(*hs, path);
    struct Sec { const char* key; const char* env; SceneBuilder::LoadFn fn; };
    const Sec secs[] = {
        {"cameras", "json-cameras", &SceneBuilder::load_camera},
        {"textures", "json-textures", &SceneBuilder::load_texture},
        {"materials", "json-materials", &SceneBuilder::load_material},
        {"mediums", "json-mediums", &SceneBuilder::load_medium},
        {"primitives", "json-primitives", &SceneBuilder::load_primitive},
        {"surfaces", "json-surfaces", &SceneBuilder::load_surface},
        {"instances", "json-instances", &SceneBuilder::load_instance},
        {"lights", "json-lights", &SceneBuilder::load_light},
    };
    for (auto& s : secs) {
        const JsonValue* v = root.get(s.key);
        if (!v) throw HostError(SPT_HOST_ERR_SCHEMA, std::string("scene - There is no '") + s.key + "' field");
        b.load_section(*v, s.env, s.fn, true);
    }
    if (const JsonValue* v = root.get("environment")) b.load_section(*v, "json-environment", &SceneBuilder::load_env, false);
    if (const JsonValue* g = root.get("gltf")) {   // json.rs:168-175: merged after the JSON sections
        if (g->kind != JsonValue::String) throw HostError(SPT_HOST_ERR_SCHEMA, "json - 'gltf' should be string");
        b.import_gltf(with_file_name(path, g->s));
    }
    b.finish(root);
    return hs.release();
}

}  // namespace spt_host
