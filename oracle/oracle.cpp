// oracle.cpp — CPU restatement of the reference's per-pixel radiance loop.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (libspt_hip.so, libspt_host.so,
// the CLI, the python package) links, loads or calls this file; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
//
// What it restates (reference = /root/reference, PepcyCh/simple-path-tracer, Rust):
//   src/renderer/pt.rs            trace_ray, shadow_ray_from_medium, render, power_heuristic
//   src/primitive/{bvh,group,instance,triangle,sphere,bezier}.rs   intersect / intersect_test / sample / pdf
//   src/core/{ray,bbox,intersection,coord,transform,color,film}.rs
//   src/core/surface.rs, src/material/*.rs (resolved on the host for scalar textures)
//   src/bxdf/{lambert,util,fresnel,microfacet,microfacet_conductor,microfacet_dielectric,
//             specular_conductor,specular_dielectric,pseudo,microfacet_plastic,specular_plastic,substrate}.rs
//   src/texture/*.rs, src/core/intersection.rs (ray differentials), src/filter/boxf.rs
//   src/light_sampler/{uniform,power_is}.rs, src/core/alias_table.rs, src/light/*.rs
//   src/medium/{homogeneous,util}.rs, src/camera/perspective.rs, src/pixel_sampler/*.rs
// Each function cites the lines it follows.  Scalar f32, one expression per
// reference expression, same operation order (glam 0.20 conventions: dot =
// (x*x'+y*y')+z*z', Mat3*v = (c0*x+c1*y)+c2*z, normalize = v/sqrt(dot)).
//
// PARITY STATUS: UNPINNED against the Rust reference itself.  The reference cannot
// be built here (no cargo/rustc, un-vendored crates) and ships no tests, golden
// images or fixtures for this path.  The oracle is pinned instead by
// closed-form known answers for the two shipped scenes and by hand-derived unit
// KATs (tests/test_oracle_*.py).
//
// Deliberate, documented differences from a literal transcription:
//   D1  RNG: per-sample PCG32 keyed by (seed, pixel, sample) instead of a per-thread
//       entropy-seeded Xoshiro (src/core/rng.rs:8-12); recurrence sampler in closed
//       form (include/spt_detmath.h).
//   D2  sin/cos/ln/exp/acos/atan2 come from include/spt_detmath.h so that the GPU can
//       match bit-for-bit (ORACLE_LIBM switches to libm to measure the difference).
//   D3  Film: for the box filter of radius 0.5 a running f32 sum per pixel instead of a Vec of
//       samples (src/core/film.rs:47-51), identical result; any other radius keeps the samples and
//       runs filter_pixel as written (film.rs:71-92).
//   D4  BVH: the flattened arrays of include/spt_abi.h built by the host loader, traversed
//       in the reference's order (push left, push right, pop; src/primitive/bvh.rs:262-283).
//       ORACLE_SLAB_RECIPROCAL evaluates the slab test with a precomputed 1/d (what the
//       kernels do) instead of the reference's six divisions (src/core/bbox.rs:68-79);
//       ORACLE_BRUTE_FORCE ignores every BVH and tests all primitives linearly.
//   D5  Where the reference would panic (no light at all: src/light_sampler/uniform.rs:34-36;
//       in-medium ray that leaves the scene: src/renderer/pt.rs:77 `unwrap`), the oracle
//       skips the light sample / uses the light distance, as the kernels do.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../include/spt_abi.h"
#include "../include/spt_detmath.h"
#include "../include/spt_pndf.h"
#include "oracle.h"

namespace {

// ---------------------------------------------------------------- math (glam order)
struct Vec3 {
    float x, y, z;
};
inline Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 v3(const float* p) { return Vec3{p[0], p[1], p[2]}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length_squared(Vec3 a) { return dot(a, a); }
inline float length(Vec3 a) { return spt_sqrt(dot(a, a)); }
inline Vec3 normalize(Vec3 a) { return a / length(a); }

// src/core/color.rs
struct Color {
    float r, g, b;
};
inline Color col(float r, float g, float b) { return Color{r, g, b}; }
inline Color col(const float* p) { return Color{p[0], p[1], p[2]}; }
inline Color gray(float v) { return Color{v, v, v}; }
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Color operator-(Color a) { return {-a.r, -a.g, -a.b}; }
inline Color operator*(Color a, float s) { return {a.r * s, a.g * s, a.b * s}; }
inline Color operator*(float s, Color a) { return a * s; }                                 // color.rs:123-129
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Color operator/(Color a, float s) { return a * (1.0f / s); }                         // color.rs:145-151
inline Color operator/(Color a, Color b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }
inline float luminance(Color c) { return 0.299f * c.r + 0.587f * c.g + 0.114f * c.b; }      // color.rs:30-32
inline float avg(Color c) { return (c.r + c.g + c.b) / 3.0f; }
inline bool is_finite(Color c) { return spt_is_finite(c.r) && spt_is_finite(c.g) && spt_is_finite(c.b); }

struct Flags {
    bool slab_recip, brute, libm, tie_min_id;
};

struct Math {  // D2: deterministic kernels by default, libm on request
    bool libm;
    void sincos(float x, float* s, float* c) const {
        if (libm) { *s = std::sin(x); *c = std::cos(x); } else spt_sincos(x, s, c);
    }
    float sin(float x) const { return libm ? std::sin(x) : spt_sin(x); }
    float cos(float x) const { return libm ? std::cos(x) : spt_cos(x); }
    float ln(float x) const { return libm ? std::log(x) : spt_log(x); }
    float exp(float x) const { return libm ? std::exp(x) : spt_exp(x); }
    float acos(float x) const { return libm ? std::acos(x) : spt_acos(x); }
    float atan2(float y, float x) const { return libm ? std::atan2(y, x) : spt_atan2(y, x); }
    float powf(float x, float y) const { return libm ? std::pow(x, y) : spt_pow(x, y); }
    float log2(float x) const { return libm ? std::log2(x) : spt_log2(x); }
    Color exp(Color c) const { return col(exp(c.r), exp(c.g), exp(c.b)); }
};

// src/core/rng.rs:14-20
struct Rng {
    spt_rng s;
    float uniform_1d() { return spt_rng_f32(&s); }
    void uniform_2d(float* a, float* b) { *a = uniform_1d(); *b = uniform_1d(); }
    // Rng::gaussian_2d (src/core/rng.rs:28-42): Box-Muller, redrawing while the first number is <= 1e-6
    void gaussian_2d(float mu, float sigma, float* x, float* y) {
        float rx, ry;
        do { uniform_2d(&rx, &ry); } while (!(rx > 1e-6f));
        const float mag = sigma * spt_sqrt(-2.0f * spt_log(rx));
        const float temp = 2.0f * SPT_PI * ry;
        *x = mag * spt_cos(temp) + mu;
        *y = mag * spt_sin(temp) + mu;
    }
};

// src/core/ray.rs:2-27 (aux rays are only consumed by image textures: not carried)
struct Ray {
    Vec3 origin, direction;
    float t_min;
    // AuxiliaryRay (ray.rs:9-15): only camera rays carry one (generate_ray_with_aux_ray, camera/mod.rs:15-21)
    bool has_aux = false;
    Vec3 x_origin{0, 0, 0}, x_direction{0, 0, 0}, y_origin{0, 0, 0}, y_direction{0, 0, 0};
};
constexpr float T_MIN_EPS = 0.0001f;
inline Ray make_ray(Vec3 o, Vec3 d) {
    Ray r;
    r.origin = o; r.direction = d; r.t_min = T_MIN_EPS;
    return r;
}
inline Vec3 point_at(const Ray& r, float t) { return r.origin + r.direction * t; }

// glam Affine3A stored as 3 columns + translation (spt_instance::inv / fwd)
inline Vec3 xf_vector(const float* m, Vec3 v) {
    return (v3(m) * v.x + v3(m + 3) * v.y) + v3(m + 6) * v.z;
}
inline Vec3 xf_point(const float* m, Vec3 p) { return xf_vector(m, p) + v3(m + 9); }
inline Vec3 mat3_mul(const float* m, Vec3 v) { return (v3(m) * v.x + v3(m + 3) * v.y) + v3(m + 6) * v.z; }

// src/core/intersection.rs:6-18 (+ the ids the flattened scene needs)
struct Inter {
    float t = SPT_F32_MAX;
    Vec3 position{0, 0, 0}, tangent{1, 0, 0}, bitangent{0, 1, 0}, normal{0, 0, 1};
    int32_t instance = -1;   // Option<&Instance>
    int32_t prim = -1;       // BasicPrimitiveRef: triangle (absolute) or sphere index
    int32_t prim_type = -1;  // SPT_PRIM_*
    float bv = 0, bw = 0;    // barycentrics of the accepted triangle hit
    float texcoords[2] = {0, 0}, duvdx[2] = {0, 0}, duvdy[2] = {0, 0};
    int32_t cand_instance = -1;  // instance being traversed (for the ORACLE_TIE_MIN_ID rule)
};

struct Counters {
    uint64_t closest = 0, shadow = 0, nodes = 0, tris = 0, spheres = 0, insts = 0;
};

struct Ctx {
    const spt_scene_desc* d;
    Flags f;
    Math m;
    Counters* c;
};

// ---------------------------------------------------------------- src/core/bbox.rs:63-93
inline bool bbox_intersect_test(const Ctx& cx, const spt_bvh_node& n, const Ray& ray, Vec3 inv_d, float t_max, bool tie_cull = false) {
    cx.c->nodes++;
    if (n.bmin[0] > n.bmax[0] || n.bmin[1] > n.bmax[1] || n.bmin[2] > n.bmax[2]) return false;  // is_empty
    float x0, x1, y0, y1, z0, z1;
    if (cx.f.slab_recip) {
        x0 = (n.bmin[0] - ray.origin.x) * inv_d.x; x1 = (n.bmax[0] - ray.origin.x) * inv_d.x;
        y0 = (n.bmin[1] - ray.origin.y) * inv_d.y; y1 = (n.bmax[1] - ray.origin.y) * inv_d.y;
        z0 = (n.bmin[2] - ray.origin.z) * inv_d.z; z1 = (n.bmax[2] - ray.origin.z) * inv_d.z;
    } else {
        x0 = (n.bmin[0] - ray.origin.x) / ray.direction.x; x1 = (n.bmax[0] - ray.origin.x) / ray.direction.x;
        y0 = (n.bmin[1] - ray.origin.y) / ray.direction.y; y1 = (n.bmax[1] - ray.origin.y) / ray.direction.y;
        z0 = (n.bmin[2] - ray.origin.z) / ray.direction.z; z1 = (n.bmax[2] - ray.origin.z) / ray.direction.z;
    }
    float xa = spt_min(x0, x1), xb = spt_max(x0, x1);
    float ya = spt_min(y0, y1), yb = spt_max(y0, y1);
    float za = spt_min(z0, z1), zb = spt_max(z0, z1);
    float t0 = spt_max(xa, spt_max(ya, za));
    float t1 = spt_min(xb, spt_min(yb, zb));
    if (!(t0 <= t1)) return false;
    return t1 > ray.t_min && (tie_cull ? t0 <= t_max : t0 < t_max);
}
inline Vec3 recip_dir(const Ray& r) { return v3(1.0f / r.direction.x, 1.0f / r.direction.y, 1.0f / r.direction.z); }

// ---------------------------------------------------------------- src/primitive/triangle.rs:124-147
inline bool triangle_intersect_ray(const Ctx& cx, const spt_tri_pos& tp, const Ray& ray, float* t, float* v_out, float* w_out) {
    cx.c->tris++;
    Vec3 p0 = v3(tp.p0), p1 = v3(tp.p1), p2 = v3(tp.p2);
    Vec3 e1 = p1 - p0;
    Vec3 e2 = p2 - p0;
    Vec3 q = cross(ray.direction, e2);
    float det = dot(e1, q);
    if (det != 0.0f) {
        det = 1.0f / det;
        Vec3 s = ray.origin - p0;
        float v = dot(s, q) * det;
        if (v >= 0.0f) {
            Vec3 r = cross(s, e1);
            float w = dot(ray.direction, r) * det;
            float u = 1.0f - v - w;
            if (w >= 0.0f && u >= 0.0f) {
                *t = dot(e2, r) * det;
                *v_out = v;
                *w_out = w;
                return true;
            }
        }
    }
    return false;
}

// acceptance of a candidate at distance t: the reference's `t < inter.t`, or the order-independent
// (t, instance, prim) minimum under ORACLE_TIE_MIN_ID
inline bool closer(const Ctx& cx, float t, int32_t prim, const Inter& inter) {
    if (t < inter.t) return true;
    if (!cx.f.tie_min_id || t != inter.t || inter.instance < 0) return false;
    if (inter.cand_instance != inter.instance) return inter.cand_instance < inter.instance;
    return prim < inter.prim;
}

// triangle.rs:176-218: accept + interpolate (object space)
inline bool triangle_intersect(const Ctx& cx, uint32_t tri, const Ray& ray, Inter& inter) {
    float t, v, w;
    if (triangle_intersect_ray(cx, cx.d->tri_pos[tri], ray, &t, &v, &w)) {
        if (t > ray.t_min && closer(cx, t, (int32_t)tri, inter)) {
            float u = 1.0f - v - w;
            const spt_tri_attr& a = cx.d->tri_attr[tri];
            inter.t = t;
            inter.normal = normalize((v3(a.n[0]) * u + v3(a.n[1]) * v) + v3(a.n[2]) * w);
            inter.tangent = (v3(a.t[0]) * u + v3(a.t[1]) * v) + v3(a.t[2]) * w;
            inter.bitangent = (v3(a.b[0]) * u + v3(a.b[1]) * v) + v3(a.b[2]) * w;
            inter.texcoords[0] = (a.uv[0][0] * u + a.uv[1][0] * v) + a.uv[2][0] * w;   // lerp_point2, triangle.rs:291-302
            inter.texcoords[1] = (a.uv[0][1] * u + a.uv[1][1] * v) + a.uv[2][1] * w;
            inter.prim = (int32_t)tri;
            inter.instance = inter.cand_instance;
            inter.prim_type = SPT_PRIM_MESH;
            inter.bv = v;
            inter.bw = w;
            return true;
        }
    }
    return false;
}
inline bool triangle_intersect_test(const Ctx& cx, uint32_t tri, const Ray& ray, float t_max) {
    float t, v, w;
    if (triangle_intersect_ray(cx, cx.d->tri_pos[tri], ray, &t, &v, &w)) return t > ray.t_min && t < t_max;
    return false;
}

// ---------------------------------------------------------------- src/primitive/sphere.rs:25-39
inline bool sphere_intersect_ray(const Ctx& cx, const spt_sphere& s, const Ray& ray, float* mn, float* mx) {
    cx.c->spheres++;
    Vec3 oc = ray.origin - v3(s.center);
    float a = length_squared(ray.direction);
    float b = dot(ray.direction, oc);
    float c = length_squared(oc) - s.radius * s.radius;
    float delta = b * b - a * c;
    if (delta >= 0.0f) {
        delta = spt_sqrt(delta);
        *mn = (-b - delta) / a;
        *mx = (-b + delta) / a;
        return true;
    }
    return false;
}
// sphere.rs:70-82 / 124-134: tangent frame from the unit normal
inline void sphere_frame(Vec3 norm, Vec3* tangent, Vec3* bitangent) {
    float sin_theta = spt_sqrt(1.0f - norm.y * norm.y);
    if (sin_theta != 0.0f) {
        Vec3 bt = norm * (-norm.y / sin_theta);
        bt.y = sin_theta;
        *bitangent = bt;
        *tangent = cross(bt, norm);
    } else if (norm.y > 0.0f) {
        *bitangent = v3(1, 0, 0);
        *tangent = v3(0, 0, 1);
    } else {
        *bitangent = v3(-1, 0, 0);
        *tangent = v3(0, 0, -1);
    }
}
// sphere.rs:138-145
inline void sphere_normal_to_texcoords(const Ctx& cx, Vec3 p, float* uv) {
    float theta = cx.m.acos(p.y);
    float phi = cx.m.atan2(p.x, p.z) + SPT_PI;
    uv[0] = phi * 0.5f * SPT_FRAC_1_PI;
    uv[1] = theta * SPT_FRAC_1_PI;
}
// sphere.rs:59-84
inline bool sphere_intersect(const Ctx& cx, uint32_t si, const Ray& ray, Inter& inter) {
    const spt_sphere& s = cx.d->spheres[si];
    float mn, mx;
    if (sphere_intersect_ray(cx, s, ray, &mn, &mx)) {
        float t = (mn < ray.t_min) ? mx : mn;
        if (ray.t_min < t && closer(cx, t, (int32_t)si, inter)) {
            inter.t = t;
            inter.instance = inter.cand_instance;
            Vec3 norm = (point_at(ray, t) - v3(s.center)) / s.radius;
            inter.normal = norm;
            sphere_frame(norm, &inter.tangent, &inter.bitangent);
            sphere_normal_to_texcoords(cx, norm, inter.texcoords);
            inter.prim = (int32_t)si;
            inter.prim_type = SPT_PRIM_SPHERE;
            inter.bv = 0.0f;
            inter.bw = 0.0f;
            return true;
        }
    }
    return false;
}
inline bool sphere_intersect_test(const Ctx& cx, uint32_t si, const Ray& ray, float t_max) {
    float mn, mx;
    if (sphere_intersect_ray(cx, cx.d->spheres[si], ray, &mn, &mx)) return mn < t_max && mx > ray.t_min;  // sphere.rs:51-56
    return false;
}

// ---------------------------------------------------------------- src/primitive/bezier.rs (Bezier clipping build)
// glam 0.20 Vec2 (scalar): dot = x*x + y*y, normalize = v * (1 / length), Vec2 / f32 divides each component
struct V2 { float x, y; };
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }
inline V2 operator/(V2 a, float s) { return {a.x / s, a.y / s}; }
inline V2 normalize2(V2 a) { return a * (1.0f / spt_sqrt(a.x * a.x + a.y * a.y)); }
struct Patch2 { V2 p[4][4]; };
struct OptF { bool some; float v; };

constexpr uint32_t CLIPPING_MAX_TIMES = 16;    // bezier.rs:14-17
constexpr float CLIPPING_EPS = 0.00001f;

inline void cubic_bezier_at(float u, float* b) {      // bezier.rs:206-209
    float iu = 1.0f - u;
    b[0] = iu * iu * iu; b[1] = 3.0f * iu * iu * u; b[2] = 3.0f * u * u * iu; b[3] = u * u * u;
}
inline void cubic_bezier_du_at(float u, float* b) {   // bezier.rs:211-219
    float iu = 1.0f - u;
    b[0] = -3.0f * iu * iu;
    b[1] = 3.0f * iu * iu - 6.0f * iu * u;
    b[2] = 6.0f * u * iu - 3.0f * u * u;
    b[3] = 3.0f * u * u;
}
inline Vec3 cubic_bezier_sum(const spt_bezier_patch& bp, const float* bu, const float* bv) {   // bezier.rs:222-236
    Vec3 result = v3(0, 0, 0);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) result = result + v3(bp.cp[i][j]) * (bu[j] * bv[i]);
    return result;
}
inline Vec3 bezier_point_at(const spt_bezier_patch& bp, float u, float v) {      // bezier.rs:40-44
    float bu[4], bv[4];
    cubic_bezier_at(u, bu); cubic_bezier_at(v, bv);
    return cubic_bezier_sum(bp, bu, bv);
}
inline Vec3 bezier_tangent_at(const spt_bezier_patch& bp, float u, float v) {    // bezier.rs:46-50
    float bu[4], bv[4];
    cubic_bezier_du_at(u, bu); cubic_bezier_at(v, bv);
    return cubic_bezier_sum(bp, bu, bv);
}
inline Vec3 bezier_bitangent_at(const spt_bezier_patch& bp, float u, float v) {  // bezier.rs:52-56
    float bu[4], bv[4];
    cubic_bezier_at(u, bu); cubic_bezier_du_at(v, bv);
    return cubic_bezier_sum(bp, bu, bv);
}
inline void clip_bezier_by(const V2* pt, float u_min, float u_max, V2* out) {     // bezier.rs:425-455
    float b[4];
    cubic_bezier_at(u_min, b);
    V2 p_min = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    cubic_bezier_du_at(u_min, b);
    V2 d_min = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    d_min = d_min * (u_max - u_min);
    cubic_bezier_at(u_max, b);
    V2 p_max = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    cubic_bezier_du_at(u_max, b);
    V2 d_max = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    d_max = d_max * (u_max - u_min);
    out[0] = p_min; out[1] = p_min + d_min / 3.0f; out[2] = p_max - d_max / 3.0f; out[3] = p_max;
}
inline void clip_bezier_at_midpoint(const V2* pt, V2* l, V2* r) {                // bezier.rs:458-485
    float b[4];
    cubic_bezier_at(0.5f, b);
    V2 p_mid = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    cubic_bezier_du_at(0.5f, b);
    V2 d_mid = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    d_mid = d_mid * 0.5f / 3.0f;
    l[0] = pt[0]; l[1] = (pt[0] + pt[1]) * 0.5f; l[2] = p_mid - d_mid; l[3] = p_mid;
    r[0] = p_mid; r[1] = p_mid + d_mid; r[2] = (pt[2] + pt[3]) * 0.5f; r[3] = pt[3];
}
inline Patch2 transposed(const V2 rows[4][4]) {   // the `swap` re-indexing of bezier.rs:307-318, 381-386
    Patch2 o;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) o.p[a][b] = rows[b][a];
    return o;
}
// bezier.rs:239-422, the recursion as written; results are appended in the reference's order
void bezier_clipping(const Patch2& patch, V2 lu, V2 lv, float au0, float au1, float av0, float av1, bool real_u, OptF calculated,
                     uint32_t times, std::vector<V2>& results) {
    if (times == CLIPPING_MAX_TIMES) {
        float u = 0.5f * au0 + au1;
        float v = calculated.some ? calculated.v : 0.5f * av0 + av1;
        results.push_back(real_u ? V2{u, v} : V2{v, u});
        return;
    }
    float upper[4] = {0, 0, 0, 0}, lower[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float dist = patch.p[i][j].x * lu.y - patch.p[i][j].y * lu.x;
            if (i == 0 || dist > upper[j]) upper[j] = dist;
            if (i == 0 || dist < lower[j]) lower[j] = dist;
        }
    static const int pairs[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    float u_min = (upper[0] >= 0.0f && lower[0] <= 0.0f) ? 0.0f : 1.0f;
    float u_max = (upper[3] >= 0.0f && lower[3] <= 0.0f) ? 1.0f : 0.0f;
    for (const auto& pr : pairs) {
        const int a = pr[0], b = pr[1];
        if (upper[a] * upper[b] <= 0.0f) {
            float diff = upper[b] - upper[a];
            if (diff == 0.0f) {
                u_min = spt_min(u_min, (float)a / 3.0f);
                u_max = spt_max(u_max, (float)b / 3.0f);
            } else {
                float k = (float)(b - a) / 3.0f / diff;
                float c = (float)a / 3.0f - k * upper[a];
                u_min = spt_min(u_min, c);
                u_max = spt_max(u_max, c);
            }
        }
        if (lower[a] * lower[b] <= 0.0f) {
            float diff = lower[b] - lower[a];
            if (diff == 0.0f) {
                u_min = spt_min(u_min, (float)a / 3.0f);
                u_max = spt_max(u_max, (float)b / 3.0f);
            } else {
                float k = (float)(b - a) / 3.0f / diff;
                float c = (float)b / 3.0f - k * lower[b];
                u_min = spt_min(u_min, c);
                u_max = spt_max(u_max, c);
            }
        }
    }
    if (u_max < u_min) return;
    const bool swap = !calculated.some;
    if (u_max - u_min > 0.8f) {
        V2 l[4][4], r[4][4];
        for (int k = 0; k < 4; ++k) clip_bezier_at_midpoint(patch.p[k], l[k], r[k]);
        if (swap) {
            bezier_clipping(transposed(l), lv, lu, av0, av1, au0 * 0.5f, au1, !real_u, OptF{false, 0.0f}, times + 1, results);
            bezier_clipping(transposed(r), lv, lu, av0, av1, au0 * 0.5f, au0 * 0.5f + au1, !real_u, OptF{false, 0.0f}, times + 1, results);
        } else {
            Patch2 pl, prr;
            std::memcpy(pl.p, l, sizeof l); std::memcpy(prr.p, r, sizeof r);
            bezier_clipping(pl, lu, lv, au0 * 0.5f, au1, av0, av1, real_u, calculated, times + 1, results);
            bezier_clipping(prr, lu, lv, au0 * 0.5f, au0 * 0.5f + au1, av0, av1, real_u, calculated, times + 1, results);
        }
        return;
    }
    float u_len = u_max - u_min;
    bool stop = u_len * au0 < CLIPPING_EPS;
    if (stop) {
        float u = 0.5f * (u_max + u_min) * au0 + au1;
        if (calculated.some) {
            results.push_back(real_u ? V2{u, calculated.v} : V2{calculated.v, u});
            return;
        }
        calculated = OptF{true, u};
    }
    V2 n[4][4];
    for (int k = 0; k < 4; ++k) clip_bezier_by(patch.p[k], u_min, u_max, n[k]);
    if (swap) {
        bezier_clipping(transposed(n), lv, lu, av0, av1, au0 * u_len, au0 * u_min + au1, !real_u, calculated, times + 1, results);
    } else {
        Patch2 pn;
        std::memcpy(pn.p, n, sizeof n);
        bezier_clipping(pn, lu, lv, au0 * u_len, au0 * u_min + au1, av0, av1, real_u, calculated, times + 1, results);
    }
}
// bezier.rs:105-134: (u, v, t) of the nearest accepted intersection
// CubicBezier::intersect_ray of the `bezier_ni` build (bezier.rs:58-103) with Bbox::intersect_ray (bbox.rs:63-85)
inline bool bezier_intersect_ray_newton(const spt_bezier_patch& bp, const Ray& ray, float* u_out, float* v_out, float* t_out) {
    Vec3 lo = v3(bp.cp[0][0]), hi = lo;      // CubicBezier::new (bezier.rs:26-38)
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const Vec3 p = v3(bp.cp[i][j]);
            lo = v3(spt_min(lo.x, p.x), spt_min(lo.y, p.y), spt_min(lo.z, p.z));
            hi = v3(spt_max(hi.x, p.x), spt_max(hi.y, p.y), spt_max(hi.z, p.z));
        }
    const float x0 = (lo.x - ray.origin.x) / ray.direction.x, x1 = (hi.x - ray.origin.x) / ray.direction.x;
    const float y0 = (lo.y - ray.origin.y) / ray.direction.y, y1 = (hi.y - ray.origin.y) / ray.direction.y;
    const float z0 = (lo.z - ray.origin.z) / ray.direction.z, z1 = (hi.z - ray.origin.z) / ray.direction.z;
    const float t0 = spt_max(spt_min(x0, x1), spt_max(spt_min(y0, y1), spt_min(z0, z1)));
    const float t1 = spt_min(spt_max(x0, x1), spt_min(spt_max(y0, y1), spt_max(z0, z1)));
    if (!(t0 <= t1)) return false;
    float t = 0.5f * (t0 + t1), u = 0.5f, v = 0.5f;
    for (int it = 0; it < 16; ++it) {            // NEWTON_ITERATION_MAX_TIMES
        const Vec3 point = bezier_point_at(bp, u, v);
        const Vec3 diff = point_at(ray, t) - point;
        if (!spt_is_finite(t) || !spt_is_finite(u) || !spt_is_finite(v)) break;
        if (dot(diff, diff) < 0.000000001f) {    // NEWTON_ITERATION_EPS
            if (u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f && t > ray.t_min) {
                *u_out = u; *v_out = v; *t_out = t;
                return true;
            }
            break;
        }
        const Vec3 dpdu = bezier_tangent_at(bp, u, v), dpdv = bezier_bitangent_at(bp, u, v);
        const Vec3 n = cross(dpdu, dpdv);
        float det = dot(ray.direction, n);
        if (det == 0.0f) break;
        det = 1.0f / det;
        const float dt = dot(diff, n) * det;
        const Vec3 q = cross(ray.direction, diff);
        const float du = -dot(dpdv, q) * det;
        const float dv = dot(dpdu, q) * det;
        t -= dt; u -= du; v -= dv;
    }
    return false;
}

inline bool bezier_intersect_ray(const Ctx& cx, uint32_t bi, const Ray& ray, float* u_out, float* v_out, float* t_out) {
    cx.c->spheres++;
    const spt_bezier_patch& bp = cx.d->bezier_patches[bi];
    if (bp.cp[0][0][3] != 0.0f) return bezier_intersect_ray_newton(bp, ray, u_out, v_out, t_out);   // SPT_BEZIER_NEWTON (ABI v12)
    Vec3 n1 = normalize(v3(-ray.direction.y, ray.direction.x, 0.0f));
    Vec3 n2 = normalize(v3(0.0f, -ray.direction.z, ray.direction.y));
    Patch2 patch;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            Vec3 diff = v3(bp.cp[i][j]) - ray.origin;
            patch.p[i][j] = V2{dot(diff, n1), dot(diff, n2)};
        }
    V2 lu = normalize2((patch.p[3][0] - patch.p[0][0]) + (patch.p[3][3] - patch.p[0][3]));
    V2 lv = normalize2((patch.p[0][3] - patch.p[0][0]) + (patch.p[3][3] - patch.p[3][0]));
    std::vector<V2> inters;
    bezier_clipping(patch, lu, lv, 1.0f, 0.0f, 1.0f, 0.0f, true, OptF{false, 0.0f}, 0, inters);
    float t_min = SPT_F32_MAX;
    bool found = false;
    for (const V2& it : inters) {
        Vec3 p = bezier_point_at(bp, it.x, it.y);
        Vec3 diff = p - ray.origin;
        Vec3 c = cross(diff, ray.direction);
        if (length_squared(c) < CLIPPING_EPS) {
            float t = dot(diff, ray.direction) / length_squared(ray.direction);
            if (t > ray.t_min && t < t_min) {
                t_min = t;
                *u_out = it.x; *v_out = it.y; *t_out = t;
                found = true;
            }
        }
    }
    return found;
}
// bezier.rs:160-174
inline bool bezier_intersect(const Ctx& cx, uint32_t bi, const Ray& ray, Inter& inter) {
    float u, v, t;
    if (bezier_intersect_ray(cx, bi, ray, &u, &v, &t)) {
        if (t > ray.t_min && closer(cx, t, (int32_t)bi, inter)) {
            const spt_bezier_patch& bp = cx.d->bezier_patches[bi];
            inter.t = t;
            inter.texcoords[0] = u; inter.texcoords[1] = v;
            inter.tangent = bezier_tangent_at(bp, u, v);
            inter.bitangent = bezier_bitangent_at(bp, u, v);
            inter.normal = normalize(cross(inter.tangent, inter.bitangent));
            inter.prim = (int32_t)bi;
            inter.instance = inter.cand_instance;
            inter.prim_type = SPT_PRIM_BEZIER;
            inter.bv = u;      // the hit record carries the patch parameters where a triangle has its barycentrics
            inter.bw = v;
            return true;
        }
    }
    return false;
}
// bezier.rs:152-158
inline bool bezier_intersect_test(const Ctx& cx, uint32_t bi, const Ray& ray, float t_max) {
    float u, v, t;
    if (bezier_intersect_ray(cx, bi, ray, &u, &v, &t)) return t > ray.t_min && t < t_max;
    return false;
}

// ---------------------------------------------------------------- src/primitive/bvh.rs:237-283 over a BLAS
constexpr int STACK_MAX = 128;
inline bool blas_intersect(const Ctx& cx, const spt_mesh& mesh, const Ray& ray, Inter& inter) {
    bool result = false;
    if (cx.f.brute) {
        for (uint32_t i = mesh.tri_first; i < mesh.tri_first + mesh.tri_count; ++i) result |= triangle_intersect(cx, i, ray, inter);
        return result;
    }
    Vec3 inv_d = recip_dir(ray);
    uint32_t stack[STACK_MAX];
    int sp = 0;
    stack[sp++] = mesh.root;
    while (sp > 0) {
        const spt_bvh_node& u = cx.d->blas_nodes[stack[--sp]];
        if (!bbox_intersect_test(cx, u, ray, inv_d, inter.t, cx.f.tie_min_id)) continue;
        if (u.b & SPT_LEAF_FLAG) {
            uint32_t n = u.b & ~SPT_LEAF_FLAG;
            for (uint32_t i = u.a; i < u.a + n; ++i) result |= triangle_intersect(cx, i, ray, inter);
        } else if (sp + 2 <= STACK_MAX) {
            stack[sp++] = u.a;  // lc
            stack[sp++] = u.b;  // rc: popped first
        }
    }
    return result;
}
inline bool blas_intersect_test(const Ctx& cx, const spt_mesh& mesh, const Ray& ray, float t_max) {
    if (cx.f.brute) {
        for (uint32_t i = mesh.tri_first; i < mesh.tri_first + mesh.tri_count; ++i)
            if (triangle_intersect_test(cx, i, ray, t_max)) return true;
        return false;
    }
    Vec3 inv_d = recip_dir(ray);
    uint32_t stack[STACK_MAX];
    int sp = 0;
    stack[sp++] = mesh.root;
    while (sp > 0) {
        const spt_bvh_node& u = cx.d->blas_nodes[stack[--sp]];
        if (!bbox_intersect_test(cx, u, ray, inv_d, t_max)) continue;
        if (u.b & SPT_LEAF_FLAG) {
            uint32_t n = u.b & ~SPT_LEAF_FLAG;
            for (uint32_t i = u.a; i < u.a + n; ++i)
                if (triangle_intersect_test(cx, i, ray, t_max)) return true;
        } else if (sp + 2 <= STACK_MAX) {
            stack[sp++] = u.a;
            stack[sp++] = u.b;
        }
    }
    return false;
}

// ---------------------------------------------------------------- src/primitive/instance.rs:88-109
inline Ray transformed_by(const Ray& r, const float* m) {  // src/core/ray.rs:33-41: direction NOT renormalised
    Ray t = r;   // `..self`: the auxiliary ray stays in world space (never read in object space)
    t.origin = xf_point(m, r.origin);
    t.direction = xf_vector(m, r.direction);
    return t;
}
inline bool instance_intersect(const Ctx& cx, uint32_t ii, const Ray& ray, Inter& inter) {
    cx.c->insts++;
    const spt_instance& in = cx.d->instances[ii];
    Ray tr = transformed_by(ray, in.inv);
    inter.cand_instance = (int32_t)ii;
    bool hit = (in.prim_type == SPT_PRIM_SPHERE)   ? sphere_intersect(cx, in.prim_id, tr, inter)
               : (in.prim_type == SPT_PRIM_BEZIER) ? bezier_intersect(cx, in.prim_id, tr, inter)
                                                   : blas_intersect(cx, cx.d->meshes[in.prim_id], tr, inter);
    if (hit) {
        inter.instance = (int32_t)ii;
        inter.position = point_at(ray, inter.t);
        inter.normal = normalize(mat3_mul(in.nrm, inter.normal));  // Transform::transform_normal3a
        inter.tangent = xf_vector(in.fwd, inter.tangent);
        inter.bitangent = xf_vector(in.fwd, inter.bitangent);
        return true;
    }
    return false;
}
inline bool instance_intersect_test(const Ctx& cx, uint32_t ii, const Ray& ray, float t_max) {
    cx.c->insts++;
    const spt_instance& in = cx.d->instances[ii];
    Ray tr = transformed_by(ray, in.inv);
    return (in.prim_type == SPT_PRIM_SPHERE)   ? sphere_intersect_test(cx, in.prim_id, tr, t_max)
           : (in.prim_type == SPT_PRIM_BEZIER) ? bezier_intersect_test(cx, in.prim_id, tr, t_max)
                                               : blas_intersect_test(cx, cx.d->meshes[in.prim_id], tr, t_max);
}

// ---------------------------------------------------------------- scene.aggregate(): Group (group.rs:24-40) or BvhAccel<Instance>
bool aggregate_intersect(const Ctx& cx, const Ray& ray, Inter& inter) {
    cx.c->closest++;
    const spt_scene_desc& d = *cx.d;
    bool result = false;
    if (d.aggregate == SPT_AGGREGATE_GROUP || cx.f.brute) {
        for (uint32_t i = 0; i < d.n_instances; ++i) result |= instance_intersect(cx, i, ray, inter);
        return result;
    }
    if (d.n_tlas_nodes == 0) return false;
    Vec3 inv_d = recip_dir(ray);
    uint32_t stack[STACK_MAX];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const spt_bvh_node& u = d.tlas_nodes[stack[--sp]];
        if (!bbox_intersect_test(cx, u, ray, inv_d, inter.t, cx.f.tie_min_id)) continue;
        if (u.b & SPT_LEAF_FLAG) {
            uint32_t n = u.b & ~SPT_LEAF_FLAG;
            for (uint32_t i = u.a; i < u.a + n; ++i) result |= instance_intersect(cx, i, ray, inter);
        } else if (sp + 2 <= STACK_MAX) {
            stack[sp++] = u.a;
            stack[sp++] = u.b;
        }
    }
    return result;
}
bool aggregate_intersect_test(const Ctx& cx, const Ray& ray, float t_max) {
    cx.c->shadow++;
    const spt_scene_desc& d = *cx.d;
    if (d.aggregate == SPT_AGGREGATE_GROUP || cx.f.brute) {
        for (uint32_t i = 0; i < d.n_instances; ++i)
            if (instance_intersect_test(cx, i, ray, t_max)) return true;
        return false;
    }
    if (d.n_tlas_nodes == 0) return false;
    Vec3 inv_d = recip_dir(ray);
    uint32_t stack[STACK_MAX];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const spt_bvh_node& u = d.tlas_nodes[stack[--sp]];
        if (!bbox_intersect_test(cx, u, ray, inv_d, t_max)) continue;
        if (u.b & SPT_LEAF_FLAG) {
            uint32_t n = u.b & ~SPT_LEAF_FLAG;
            for (uint32_t i = u.a; i < u.a + n; ++i)
                if (instance_intersect_test(cx, i, ray, t_max)) return true;
        } else if (sp + 2 <= STACK_MAX) {
            stack[sp++] = u.a;
            stack[sp++] = u.b;
        }
    }
    return false;
}

// ---------------------------------------------------------------- src/core/coord.rs:10-59
struct Coordinate {
    Vec3 x_world, y_world, z_world, hemisphere;
    Vec3 to_local(Vec3 w) const { return v3(dot(x_world, w), dot(y_world, w), dot(z_world, w)); }  // transpose * w
    Vec3 to_world(Vec3 l) const { return (x_world * l.x + y_world * l.y) + z_world * l.z; }
};
inline Coordinate coord_from_tangent_normal(Vec3 t, Vec3 n, Vec3 hemisphere) {
    Coordinate c;
    c.z_world = n;
    c.y_world = normalize(cross(c.z_world, t));
    c.x_world = cross(c.y_world, c.z_world);
    c.hemisphere = hemisphere;
    return c;
}


// ---------------------------------------------------------------- src/core/intersection.rs:28-84
inline bool solve_linear_system_2x2(const float a[2][2] /* columns */, const float b[2], float* x1, float* x2) {  // intersection.rs:104-118
    float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];   // Mat2::determinant
    if (det != 0.0f) {
        float temp = b[1] * a[0][0] - b[0] * a[0][1];
        *x2 = temp / det;
        *x1 = (spt_abs(a[0][0]) > spt_abs(a[0][1])) ? (b[0] - a[1][0] * *x2) / a[0][0] : (b[1] - a[1][1] * *x2) / a[0][1];
        return true;
    }
    return false;
}
inline void calc_differential(Inter& it, const Ray& ray) {
    if (!ray.has_aux) return;
    Vec3 p = point_at(ray, it.t);
    float d = dot(p, it.normal);
    float tx = (d - dot(ray.x_origin, it.normal)) / dot(ray.x_direction, it.normal);
    Vec3 px = ray.x_origin + ray.x_direction * tx;
    float ty = (d - dot(ray.y_origin, it.normal)) / dot(ray.y_direction, it.normal);
    Vec3 py = ray.y_origin + ray.y_direction * ty;
    Vec3 dpdx = px - p, dpdy = py - p;
    float bx[2], by[2], a[2][2];
    float ax = spt_abs(it.normal.x), ay = spt_abs(it.normal.y), az = spt_abs(it.normal.z);
    if (ax >= ay && ax >= az) {
        bx[0] = dpdx.y; bx[1] = dpdx.z; by[0] = dpdy.y; by[1] = dpdy.z;
        a[0][0] = it.tangent.y; a[0][1] = it.tangent.z; a[1][0] = it.bitangent.y; a[1][1] = it.bitangent.z;
    } else if (ay >= az) {
        bx[0] = dpdx.z; bx[1] = dpdx.x; by[0] = dpdy.z; by[1] = dpdy.x;
        a[0][0] = it.tangent.z; a[0][1] = it.tangent.x; a[1][0] = it.bitangent.z; a[1][1] = it.bitangent.x;
    } else {
        bx[0] = dpdx.x; bx[1] = dpdx.y; by[0] = dpdy.x; by[1] = dpdy.y;
        a[0][0] = it.tangent.x; a[0][1] = it.tangent.y; a[1][0] = it.bitangent.x; a[1][1] = it.bitangent.y;
    }
    float x1, x2;
    if (solve_linear_system_2x2(a, bx, &x1, &x2)) { it.duvdx[0] = x1; it.duvdx[1] = x2; }
    if (solve_linear_system_2x2(a, by, &x1, &x2)) { it.duvdy[0] = x1; it.duvdy[1] = x2; }
}

// ---------------------------------------------------------------- src/texture/*.rs
struct Vec4 {
    float x, y, z, w;
};
inline Vec4 operator+(Vec4 a, Vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Vec4 operator-(Vec4 a, Vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Vec4 operator*(Vec4 a, Vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline Vec4 operator/(Vec4 a, Vec4 b) { return {a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w}; }
inline Vec4 operator*(Vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

// TextureInput (mod.rs:50-62); `specified` is never selectable from a scene file
struct TexInput {
    Vec3 position, normal, tangent, bitangent;
    float texcoords[2], duvdx[2], duvdy[2];
    int32_t mode = SPT_TEXMODE_TEXCOORDS, wrap = SPT_TEXWRAP_REPEAT;
};
inline TexInput tex_input(const Inter& it) {  // From<&Intersection> (mod.rs:143-157)
    TexInput in;
    in.position = it.position; in.normal = it.normal; in.tangent = it.tangent; in.bitangent = it.bitangent;
    for (int k = 0; k < 2; ++k) { in.texcoords[k] = it.texcoords[k]; in.duvdx[k] = it.duvdx[k]; in.duvdy[k] = it.duvdy[k]; }
    return in;
}
inline void tex_value_vec2_wrapped(const TexInput& in, float* u, float* v) {  // mod.rs:73-141
    float val[2];
    switch (in.mode) {
    case SPT_TEXMODE_TEXCOORDS: val[0] = in.texcoords[0]; val[1] = in.texcoords[1]; break;
    case SPT_TEXMODE_POSITION: val[0] = in.position.x; val[1] = in.position.y; break;
    case SPT_TEXMODE_NORMAL: val[0] = in.normal.x; val[1] = in.normal.y; break;
    case SPT_TEXMODE_TANGENT: val[0] = in.tangent.x; val[1] = in.tangent.y; break;
    case SPT_TEXMODE_BITANGENT: val[0] = in.bitangent.x; val[1] = in.bitangent.y; break;
    default: val[0] = 0.0f; val[1] = 0.0f; break;
    }
    float out[2];
    for (int k = 0; k < 2; ++k) {
        float x = val[k];
        switch (in.wrap) {
        case SPT_TEXWRAP_REPEAT: {
            float fr = spt_fract(x);
            out[k] = (x >= 0.0f) ? fr : 1.0f + fr;
            break;
        }
        case SPT_TEXWRAP_MIRROR_REPEAT: {
            float fr = spt_fract(x);
            float xn = (x >= 0.0f) ? fr : 1.0f + fr;
            out[k] = (spt_f2i_sat(x) % 2 == 0) ? xn : 1.0f - xn;
            break;
        }
        case SPT_TEXWRAP_CLAMP: out[k] = spt_clamp(x, 0.0f, 1.0f); break;
        default: out[k] = spt_abs(spt_clamp(x, 0.0f, 1.0f)); break;
        }
    }
    *u = out[0];
    *v = out[1];
}
inline Vec4 rgba_to_vec4(uint32_t px) {  // image_tex.rs:153-160
    return {(float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f, (float)(px >> 24) / 255.0f};
}
inline Vec4 sample_bilinear(const Ctx& cx, const spt_image_level& L, float u, float v) {  // image_tex.rs:102-125
    const uint32_t* tx = cx.d->texels + L.first_texel;
    float x = u * (float)L.width;
    int32_t x1 = spt_f2i_sat(spt_round(x));
    int32_t x0 = (int32_t)((uint32_t)x1 - 1u);   // wrapping, as release-mode Rust
    float xt = x - (float)x0 - 0.5f;
    int32_t wmax = (int32_t)L.width - 1, hmax = (int32_t)L.height - 1;
    x0 = x0 < 0 ? 0 : (x0 > wmax ? wmax : x0);
    x1 = x1 < 0 ? 0 : (x1 > wmax ? wmax : x1);
    float y = v * (float)L.height;
    int32_t y1 = spt_f2i_sat(spt_round(y));
    int32_t y0 = (int32_t)((uint32_t)y1 - 1u);
    float yt = y - (float)y0 - 0.5f;
    y0 = y0 < 0 ? 0 : (y0 > hmax ? hmax : y0);
    y1 = y1 < 0 ? 0 : (y1 > hmax ? hmax : y1);
    Vec4 c00 = rgba_to_vec4(tx[(size_t)y0 * L.width + x0]), c01 = rgba_to_vec4(tx[(size_t)y1 * L.width + x0]);
    Vec4 c10 = rgba_to_vec4(tx[(size_t)y0 * L.width + x1]), c11 = rgba_to_vec4(tx[(size_t)y1 * L.width + x1]);
    Vec4 c0 = c00 * (1.0f - yt) + c01 * yt;
    Vec4 c1 = c10 * (1.0f - yt) + c11 * yt;
    return c0 * (1.0f - xt) + c1 * xt;
}
inline Vec4 sample_trilinear(const Ctx& cx, const spt_image& im, float u, float v, const float* duvdx, const float* duvdy) {  // image_tex.rs:127-151
    if (im.n_levels == 0) return {0, 0, 0, 0};
    const spt_image_level* lv = cx.d->image_levels + im.first_level;
    float sx = (float)lv[0].width, sy = (float)lv[0].height;
    float dxx = duvdx[0] * sx, dxy = duvdx[1] * sy, dyx = duvdy[0] * sx, dyy = duvdy[1] * sy;
    float lx = spt_sqrt(dxx * dxx + dxy * dxy), ly = spt_sqrt(dyx * dyx + dyy * dyy);
    float level = spt_clamp(cx.m.log2(spt_max(lx, ly) + 0.001f), 0.0f, (float)(im.n_levels - 1));
    uint32_t l0 = spt_f2u_sat(spt_floor(level));
    if (l0 + 1 == im.n_levels) return sample_bilinear(cx, lv[l0], u, v);
    float lt = level - (float)l0;
    Vec4 c0 = sample_bilinear(cx, lv[l0], u, v), c1 = sample_bilinear(cx, lv[l0 + 1], u, v);
    return c0 * (1.0f - lt) + c1 * lt;
}
inline float srgb_to_linear(const Ctx& cx, float sv) {  // srgb_tex.rs:53-59
    return (sv <= 0.04045f) ? sv / 12.92f : cx.m.powf((sv + 0.055f) / 1.055f, 2.4f);
}
// (color_at, float_at) of the closed Texture enum as one RGBA evaluation: every variant acts per channel,
// ScalarTex reports alpha 1 (scalar.rs:29-34) and SrgbTex leaves alpha alone (srgb_tex.rs:22-28)
Vec4 tex_eval(const Ctx& cx, uint32_t node, TexInput in) {
    const spt_texture& t = cx.d->textures[node];
    switch (t.type) {
    case SPT_TEX_SCALAR: return {t.value[0], t.value[1], t.value[2], 1.0f};
    case SPT_TEX_IMAGE: {
        float u, v;
        tex_value_vec2_wrapped(in, &u, &v);
        return sample_trilinear(cx, cx.d->images[t.image], u, v, in.duvdx, in.duvdy);
    }
    case SPT_TEX_ADD: return tex_eval(cx, t.a, in) + tex_eval(cx, t.b, in);
    case SPT_TEX_SUB: return tex_eval(cx, t.a, in) - tex_eval(cx, t.b, in);
    case SPT_TEX_MUL: return tex_eval(cx, t.a, in) * tex_eval(cx, t.b, in);
    case SPT_TEX_DIV: return tex_eval(cx, t.a, in) / tex_eval(cx, t.b, in);
    case SPT_TEX_SRGB: {
        Vec4 c = tex_eval(cx, t.a, in);
        return {srgb_to_linear(cx, c.x), srgb_to_linear(cx, c.y), srgb_to_linear(cx, c.z), c.w};
    }
    default: {  // TexInputModifier::apply_modifier (input_modifier.rs:35-50)
        Vec3 tl = v3(t.tiling), of = v3(t.offset);
        auto app = [&](Vec3 a) { return v3(a.x * tl.x + of.x, a.y * tl.y + of.y, a.z * tl.z + of.z); };
        TexInput m = in;
        m.position = app(in.position); m.normal = app(in.normal); m.tangent = app(in.tangent); m.bitangent = app(in.bitangent);
        m.texcoords[0] = in.texcoords[0] * tl.x + of.x; m.texcoords[1] = in.texcoords[1] * tl.y + of.y;
        m.duvdx[0] = in.duvdx[0] * tl.x; m.duvdx[1] = in.duvdx[1] * tl.y;
        m.duvdy[0] = in.duvdy[0] * tl.x; m.duvdy[1] = in.duvdy[1] * tl.y;
        if (t.mode >= 0) m.mode = t.mode;
        if (t.wrap >= 0) m.wrap = t.wrap;
        return tex_eval(cx, t.a, m);
    }
    }
}
inline Color tex_color(const Ctx& cx, uint32_t node, const TexInput& in) { Vec4 c = tex_eval(cx, node, in); return col(c.x, c.y, c.z); }
inline float tex_float(const Ctx& cx, uint32_t node, const TexInput& in, uint32_t chan) {
    Vec4 c = tex_eval(cx, node, in);
    return chan == SPT_CHAN_R ? c.x : (chan == SPT_CHAN_G ? c.y : (chan == SPT_CHAN_B ? c.z : c.w));
}

inline float fresnel_moment1(float eta) {  // src/bxdf/util.rs:123-134
    float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1.0f) return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945f * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
// MaterialT::bxdf_context (src/material/{lambert,conductor,dielectric,plastic,pbr_metallic,pbr_specular}.rs)
// for a material with image-backed parameters; constant materials were evaluated by the loader.
inline spt_material material_at(const Ctx& cx, const spt_material& constant, const Inter& inter) {
    if (constant.recipe == 0) return constant;
    const spt_material_recipe& r = cx.d->material_recipes[constant.recipe - 1];
    TexInput in = tex_input(inter);
    spt_material m;
    std::memset(&m, 0, sizeof m);
    m.recipe = constant.recipe;
    auto store = [](float* dst, Color c) { dst[0] = c.r; dst[1] = c.g; dst[2] = c.b; };
    auto roughness = [&](bool squared) {
        float rx = tex_float(cx, r.tex[2], in, r.rough_chan), ry = tex_float(cx, r.tex[3], in, r.rough_chan);
        m.ax = squared ? rx * rx : rx;
        m.ay = squared ? ry * ry : ry;
        return m.ax < 0.0001f || m.ay < 0.0001f;
    };
    switch (r.type) {
    case SPT_MAT_LAMBERT:
        m.bxdf = SPT_BXDF_LAMBERT;
        store(m.c0, tex_color(cx, r.tex[0], in));
        break;
    case SPT_MAT_CONDUCTOR:
        store(m.c0, tex_color(cx, r.tex[0], in));
        store(m.c1, tex_color(cx, r.tex[1], in));
        m.bxdf = roughness(true) ? SPT_BXDF_SPECULAR_CONDUCTOR : SPT_BXDF_MICROFACET_CONDUCTOR;
        break;
    case SPT_MAT_DIELECTRIC:
        m.ior = r.ior;
        m.bxdf = roughness(true) ? SPT_BXDF_SPECULAR_DIELECTRIC : SPT_BXDF_MICROFACET_DIELECTRIC;
        break;
    case SPT_MAT_PLASTIC: {  // plastic.rs:60-85: roughness NOT squared; Diffuse::new (substrate.rs:127-137)
        Color albedo = tex_color(cx, r.tex[0], in);
        m.ior = r.ior;
        m.bxdf = roughness(false) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        m.fresnel = SPT_FRESNEL_DIELECTRIC;
        m.substrate = SPT_SUBSTRATE_DIFFUSE;
        store(m.c0, albedo);
        float fdr = 2.0f * fresnel_moment1(1.0f / r.ior);
        store(m.c2, (albedo * SPT_FRAC_1_PI) / (((gray(1.0f) - albedo * fdr) * r.ior) * r.ior));
        break;
    }
    case SPT_MAT_PBR_METALLIC: {  // pbr_metallic.rs:75-104
        Color base = tex_color(cx, r.tex[0], in);
        bool spec = roughness(true);
        float metallic = tex_float(cx, r.tex[1], in, r.metal_chan);
        store(m.c1, metallic * base + (1.0f - metallic) * gray(0.04f));
        store(m.c0, base * (1.0f - metallic));
        m.bxdf = spec ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        m.fresnel = SPT_FRESNEL_SCHLICK;
        m.substrate = SPT_SUBSTRATE_LAMBERT;
        break;
    }
    case SPT_MAT_SUBSURFACE: {  // material/subsurface.rs:66-93, bxdf::Subsurface::new (substrate.rs:199-211)
        Color albedo = tex_color(cx, r.tex[0], in);
        float ld = tex_float(cx, r.tex[1], in, SPT_CHAN_R);
        m.ior = r.ior;
        m.bxdf = roughness(true) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        m.fresnel = SPT_FRESNEL_DIELECTRIC;
        m.substrate = SPT_SUBSTRATE_SUBSURFACE;
        store(m.c0, albedo);
        float fdr = 2.0f * fresnel_moment1(1.0f / r.ior);
        store(m.c2, (albedo * SPT_FRAC_1_PI) / (((gray(1.0f) - albedo * fdr) * r.ior) * r.ior));
        const float a[3] = {albedo.r, albedo.g, albedo.b};
        for (int k = 0; k < 3; ++k) {
            float q = a[k] - 0.33f, q2 = q * q;
            m.c1[k] = ld / (3.5f + 100.0f * (q2 * q2));
        }
        break;
    }
    case SPT_MAT_PNDF_CONDUCTOR:    // pndf_conductor.rs:156-196
    case SPT_MAT_PNDF_PLASTIC: {    // pndf_plastic.rs:163-211
        const bool plastic = r.type == SPT_MAT_PNDF_PLASTIC;
        const spt_pndf& pd = cx.d->pndfs[r.tex[1]];
        const Color albedo = tex_color(cx, r.tex[0], in);
        store(m.c0, albedo);
        if (plastic) {   // DielectricFresnel::new(ior), Diffuse::new(albedo, ior) (substrate.rs:127-137)
            m.ior = r.ior;
            m.fresnel = SPT_FRESNEL_DIELECTRIC;
            m.substrate = SPT_SUBSTRATE_DIFFUSE;
            float fdr = 2.0f * fresnel_moment1(1.0f / r.ior);
            store(m.c2, (albedo * SPT_FRAC_1_PI) / (((gray(1.0f) - albedo * fdr) * r.ior) * r.ior));
        } else {
            m.fresnel = SPT_FRESNEL_SCHLICK;
        }
        const float ux = spt_pndf_wrap(inter.texcoords[0] * pd.tiling[0] + pd.offset[0]), uy = spt_pndf_wrap(inter.texcoords[1] * pd.tiling[1] + pd.offset[1]);
        const float dxx = inter.duvdx[0] * pd.tiling[0], dxy = inter.duvdx[1] * pd.tiling[1];
        const float dyx = inter.duvdy[0] * pd.tiling[0], dyy = inter.duvdy[1] * pd.tiling[1];
        const float sigma_p = spt_max(spt_sqrt(dxx * dxx + dxy * dxy), spt_sqrt(dyx * dyx + dyy * dyy)) / 3.0f;
        if (sigma_p > 0.0f) {
            // PndfMicrofacet::new (microfacet.rs:67-95): the normalisation of the footprint's terms
            spt_pndf_view v{&pd, cx.d->pndf_terms, cx.d->pndf_nodes, cx.d->pndf_refs, cx.d->pndf_roots};
            const float sum = spt_pndf_uv_walk(&v, ux, uy, sigma_p, 0, 0.0f, 0.0f, nullptr);
            m.bxdf = plastic ? SPT_BXDF_PNDF_PLASTIC : SPT_BXDF_PNDF_CONDUCTOR;
            m.ax = ux; m.ay = uy;
            m.c1[0] = 1.0f / sum;            // per-hit record of the two P-NDF lobes: (ax, ay) = u, c1 = (1 / sum, sigma_p, table)
            m.c1[1] = sigma_p;
            m.c1[2] = spt_u2f(r.tex[1]);
        } else {
            const float fr = tex_float(cx, r.tex[2], in, r.rough_chan);
            m.ax = m.ay = fr * fr;
            if (plastic) m.bxdf = m.ax < 0.0001f ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
            else m.bxdf = m.ax < 0.0001f ? SPT_BXDF_SPECULAR_CONDUCTOR : SPT_BXDF_MICROFACET_CONDUCTOR;
        }
        break;
    }
    default: {  // pbr_specular.rs:60-92
        store(m.c0, tex_color(cx, r.tex[0], in));
        store(m.c1, tex_color(cx, r.tex[1], in));
        m.bxdf = roughness(true) ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        m.fresnel = SPT_FRESNEL_SCHLICK;
        m.substrate = SPT_SUBSTRATE_LAMBERT;
        break;
    }
    }
    return m;
}
// Surface::emissive (surface.rs:49-55)
inline Color surface_emissive(const Ctx& cx, const spt_surface& s, const Inter& inter) {
    Color e = col(s.emissive);
    if (s.emissive_map) e = e * tex_color(cx, s.emissive_map - 1, tex_input(inter));
    return e;
}

// src/core/surface.rs:65-95
inline Coordinate surface_coord(const Ctx& cx, const spt_surface& s, const Ray& ray, const Inter& inter) {
    Vec3 shade_normal = inter.normal;
    if (s.normal_map) {
        Color value = tex_color(cx, s.normal_map - 1, tex_input(inter));
        Color nc = value * 2.0f - gray(1.0f);
        Vec3 local = normalize(v3(nc.r, nc.g, nc.b));
        shade_normal = normalize((local.x * normalize(inter.tangent) + local.y * normalize(inter.bitangent)) + local.z * inter.normal);
    }
    bool hit_back = dot(ray.direction, inter.normal) > 0.0f;
    bool ds = (s.flags & SPT_SURF_DOUBLE_SIDED) != 0;
    return coord_from_tangent_normal(inter.tangent, (ds && hit_back) ? -shade_normal : shade_normal,
                                     hit_back ? -inter.normal : inter.normal);
}

// ---------------------------------------------------------------- src/bxdf/util.rs
inline float pow2(float x) { return x * x; }
inline Vec3 reflect(Vec3 i) { return v3(-i.x, -i.y, i.z); }                       // util.rs:3-5
inline Vec3 reflect_n(Vec3 i, Vec3 n) { return (2.0f * dot(i, n)) * n - i; }     // util.rs:7-9
inline bool refract(Vec3 i, float ior, Vec3* out) {                              // util.rs:11-24
    float ior_ratio = (i.z >= 0.0f) ? 1.0f / ior : ior;
    float o_z_sqr = 1.0f - (1.0f - i.z * i.z) * ior_ratio * ior_ratio;
    if (o_z_sqr >= 0.0f) {
        float o_z = (i.z >= 0.0f) ? -spt_sqrt(o_z_sqr) : spt_sqrt(o_z_sqr);
        *out = v3(-i.x * ior_ratio, -i.y * ior_ratio, o_z);
        return true;
    }
    return false;
}
inline bool refract_n(Vec3 i, Vec3 n, float ior, Vec3* out) {                    // util.rs:26-46
    float cos_i = dot(i, n);
    if (cos_i >= 0.0f) {
        float ior_ratio = 1.0f / ior;
        float o_z_sqr = 1.0f - (1.0f - cos_i * cos_i) * ior_ratio * ior_ratio;
        if (o_z_sqr >= 0.0f) {
            *out = (ior_ratio * cos_i - spt_sqrt(o_z_sqr)) * n - ior_ratio * i;
            return true;
        }
        return false;
    }
    float ior_ratio = ior;
    float o_z_sqr = 1.0f - (1.0f - cos_i * cos_i) * ior_ratio * ior_ratio;
    if (o_z_sqr >= 0.0f) {
        *out = (spt_sqrt(o_z_sqr) + ior_ratio * cos_i) * n - ior_ratio * i;
        return true;
    }
    return false;
}
inline float fresnel_n(float ior, Vec3 i, Vec3 n) {                              // util.rs:56-81
    float i_ior, o_ior;
    if (dot(i, n) >= 0.0f) { i_ior = 1.0f; o_ior = ior; } else { i_ior = ior; o_ior = 1.0f; }
    Vec3 rf;
    if (refract_n(i, n, ior, &rf)) {
        float idotn = spt_abs(dot(i, n));
        float rdotn = spt_abs(dot(rf, n));
        float denom = i_ior * idotn + o_ior * rdotn;
        float num = i_ior * idotn - o_ior * rdotn;
        float rs = num / denom;
        rs = rs * rs;
        denom = i_ior * rdotn + o_ior * idotn;
        num = i_ior * rdotn - o_ior * idotn;
        float rp = num / denom;
        rp = rp * rp;
        return 0.5f * (rs + rp);
    }
    return 1.0f;
}
inline Color csqrt(Color c) { return col(spt_sqrt(c.r), spt_sqrt(c.g), spt_sqrt(c.b)); }
inline Color fresnel_conductor_n(Color ior, Color ior_k, Vec3 i, Vec3 n) {       // util.rs:87-112
    float cosv = dot(i, n);
    Color ior_ratio, k_ratio;
    if (cosv >= 0.0f) { ior_ratio = ior; k_ratio = ior_k; } else { ior_ratio = gray(1.0f) / ior; k_ratio = gray(1.0f) / ior_k; }
    float cos2 = cosv * cosv;
    float sin2 = 1.0f - cos2;
    Color ior_ratio2 = ior_ratio * ior_ratio;
    Color k_ratio2 = k_ratio * k_ratio;
    Color t0 = ior_ratio2 - k_ratio2 - gray(sin2);
    Color a2_b2 = csqrt(t0 * t0 + 4.0f * ior_ratio2 * k_ratio2);
    Color t1 = a2_b2 + gray(cos2);
    Color a = csqrt(0.5f * (a2_b2 + t0));
    Color t2 = (2.0f * cosv) * a;
    Color rs = (t1 - t2) / (t1 + t2);
    Color t3 = cos2 * a2_b2 + gray(sin2 * sin2);
    Color t4 = t2 * sin2;
    Color rp = rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (rs + rp);
}
inline Vec3 half_from_reflect(Vec3 i, Vec3 o) {                                   // util.rs:136-142
    return (i.z >= 0.0f) ? normalize(i + o) : -normalize(i + o);
}
inline Vec3 half_from_refract(Vec3 i, Vec3 o, float ior) {                        // util.rs:144-155
    Vec3 h = (i.z >= 0.0f) ? normalize(i + ior * o) : normalize(ior * i + o);
    if (h.z < 0.0f) h = -h;
    return h;
}
inline float ggx_ndf_aniso(Vec3 h, float ax, float ay) {                          // util.rs:162-165
    return SPT_FRAC_1_PI / spt_max(ax * ay * pow2(pow2(h.x / ax) + pow2(h.y / ay) + pow2(h.z)), 0.0001f);
}
inline float smith_g1_aniso(Vec3 v, float ax, float ay) {                         // util.rs:172-174
    return 2.0f / (1.0f + spt_sqrt(1.0f + (pow2(ax * v.x) + pow2(ay * v.y)) / spt_max(pow2(v.z), 0.0001f)));
}
inline float smith_separable_visible_aniso(Vec3 v, Vec3 l, float ax, float ay) {  // util.rs:176-180
    float vv = spt_abs(v.z) + spt_sqrt(pow2(ax * v.x) + pow2(ay * v.y) + pow2(v.z));
    float ll = spt_abs(l.z) + spt_sqrt(pow2(ax * l.x) + pow2(ay * l.y) + pow2(l.z));
    return 1.0f / (vv * ll);
}
inline float ggx_smith_vndf_pdf(Vec3 h, Vec3 v, float ax, float ay) {             // util.rs:189-194
    if (!(v.z >= 0.0f)) v = -v;
    return smith_g1_aniso(v, ax, ay) * ggx_ndf_aniso(h, ax, ay) * spt_max(dot(v, h), 0.0f) / spt_max(v.z, 0.0001f);
}
inline Vec3 ggx_smith_vndf_sample(const Math& m, Vec3 ve, float ax, float ay, float r0, float r1, float* pdf) {  // util.rs:196-224
    if (!(ve.z >= 0.0f)) ve = -ve;
    Vec3 vh = normalize(v3(ax * ve.x, ay * ve.y, ve.z));
    float len_sqr = vh.x * vh.x + vh.y * vh.y;
    Vec3 t_vec1 = (len_sqr > 0.0f) ? v3(-vh.y, vh.x, 0.0f) / spt_sqrt(len_sqr) : v3(1, 0, 0);
    Vec3 t_vec2 = cross(vh, t_vec1);
    float r = spt_sqrt(r0);
    float phi = 2.0f * SPT_PI * r1;
    float sp, cp;
    m.sincos(phi, &sp, &cp);
    float t1 = r * cp;
    float t2 = r * sp;
    float s = 0.5f * (1.0f + vh.z);
    t2 = (1.0f - s) * spt_sqrt(1.0f - t1 * t1) + s * t2;
    Vec3 nh = (t1 * t_vec1 + t2 * t_vec2) + spt_sqrt(spt_max(1.0f - t1 * t1 - t2 * t2, 0.0f)) * vh;
    Vec3 ne = normalize(v3(ax * nh.x, ay * nh.y, spt_max(nh.z, 0.0f)));
    *pdf = ggx_smith_vndf_pdf(ne, ve, ax, ay);
    return ne;
}

// ---------------------------------------------------------------- Bxdf (src/bxdf/mod.rs:72-102)
enum DirType { REFLECT = 0, TRANSMIT = 1 };
struct BxdfSample {
    Vec3 wi;
    int dir;
    Color bxdf;
    float pdf;
};

// SchlickFresnel::fresnel (fresnel.rs:49-52) = schlick_fresnel_with_r0 (util.rs:119-121), cos = i.n unclamped
inline float pow5(float x) { return x * x * x * x * x; }
inline Color mat_fresnel(const spt_material& mt, Vec3 i, Vec3 n) {  // src/bxdf/fresnel.rs:29-59
    if (mt.bxdf == SPT_BXDF_MICROFACET_CONDUCTOR || mt.bxdf == SPT_BXDF_SPECULAR_CONDUCTOR || mt.bxdf == SPT_BXDF_PNDF_CONDUCTOR) {
        if (mt.fresnel == SPT_FRESNEL_SCHLICK) {   // the conductors of pndf_conductor.rs: SchlickFresnel::new(albedo)
            const Color r0 = col(mt.c0);
            return r0 + (gray(1.0f) - r0) * pow5(1.0f - dot(i, n));
        }
        return fresnel_conductor_n(col(mt.c0), col(mt.c1), i, n);
    }
    return gray(fresnel_n(mt.ior, i, n));
}
inline Color plastic_fresnel(const spt_material& mt, Vec3 i, Vec3 n) {
    if (mt.fresnel == SPT_FRESNEL_SCHLICK) {
        Color r0 = col(mt.c1);
        return r0 + (gray(1.0f) - r0) * pow5(1.0f - dot(i, n));
    }
    return gray(fresnel_n(mt.ior, i, n));
}
// SubstrateT for Lambert (substrate.rs:29-45) and Diffuse (substrate.rs:139-180)
inline Color substrate_reflectance(const spt_material& mt) { return col(mt.c0); }
inline float substrate_pdf(const spt_material&, Vec3 wo, Vec3 wi) {
    return (wo.z * wi.z >= 0.0f) ? spt_abs(wi.z) * SPT_FRAC_1_PI : 1.0f;
}
inline Color substrate_eval(const spt_material& mt, Vec3 wo, Vec3 wi) {
    if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
    if (mt.substrate != SPT_SUBSTRATE_LAMBERT) {   // Diffuse, and Subsurface through its `diffuse` member (substrate.rs:339-349)
        float fi = fresnel_n(mt.ior, wi, v3(0, 0, 1));
        return (1.0f - fi) * col(mt.c2);
    }
    return col(mt.c0) * SPT_FRAC_1_PI;
}
inline float ndf_visible(const spt_material& mt, Vec3 wo, Vec3 wi, Vec3 h) {  // src/bxdf/microfacet.rs:47-53
    float ndf = ggx_ndf_aniso(h, mt.ax, mt.ay);
    float vis = smith_separable_visible_aniso(wo, wi, mt.ax, mt.ay);
    return ndf * vis;
}

bool bxdf_is_delta(const spt_material& mt) {
    return mt.bxdf == SPT_BXDF_SPECULAR_CONDUCTOR || mt.bxdf == SPT_BXDF_SPECULAR_DIELECTRIC || mt.bxdf == SPT_BXDF_PSEUDO;
}

// ---- Subsurface substrate (src/bxdf/substrate.rs:182-350) -----------------------------------------------
struct SsCdfTable {
    float x[SPT_SS_CDF_SIZE], y[SPT_SS_CDF_SIZE];
    SsCdfTable() { for (uint32_t i = 0; i < SPT_SS_CDF_SIZE; ++i) spt_ss_cdf_entry(i, &x[i], &y[i]); }
};
inline const SsCdfTable& ss_cdf_table() { static const SsCdfTable t; return t; }
inline float ss_sample_r(float rand) {  // substrate.rs:219-229
    const SsCdfTable& t = ss_cdf_table();
    for (uint32_t i = 1; i < SPT_SS_CDF_SIZE; ++i)
        if (t.y[i] >= rand) {
            float w = (rand - t.y[i - 1]) / (t.y[i] - t.y[i - 1]);
            return t.x[i] * w + t.x[i - 1] * (1.0f - w);
        }
    return -1.0f;
}
inline Color ss_sp(const Math& m, Color d, float r) {  // substrate.rs:213-217
    Color e1 = m.exp(col(-r / d.r, -r / d.g, -r / d.b));
    Color e2 = m.exp(col(-r / d.r, -r / d.g, -r / d.b) / 3.0f);
    return ((e1 + e2) * SPT_FRAC_1_PI) / ((8.0f * d) * r);
}
// what BxdfInputs adds for the BSSRDF (bxdf/mod.rs:62-67) and what BxdfSubsurfaceSample hands back (mod.rs:69-74)
struct SubsurfaceIo {
    const Ctx* cx = nullptr;
    Vec3 po{0, 0, 0};
    Coordinate coord_po;
    bool has = false;
    Vec3 pi{0, 0, 0};
    Coordinate coord_pi;
    Color sp{0, 0, 0};
    float pdf_pi = 0.0f;
};
// Subsurface::sample up to the diffuse lobe: false = the probe found nothing (wi = 0, bxdf = 0, pdf = 1)
bool subsurface_probe(const spt_material& mt, Rng& rng, SubsurfaceIo& io) {
    const Ctx& cx = *io.cx;
    float rand_u = rng.uniform_1d();
    float rand_x, rand_y;
    rng.uniform_2d(&rand_x, &rand_y);
    Vec3 pt = io.coord_po.to_world(v3(1, 0, 0)), pb = io.coord_po.to_world(v3(0, 1, 0)), pn = io.coord_po.to_world(v3(0, 0, 1));
    Vec3 st, sb, sn;
    if (rand_u < 0.5f) { rand_u = rand_u * 2.0f; st = pt; sb = pb; sn = pn; }
    else if (rand_u < 0.75f) { rand_u = rand_u * 4.0f - 2.0f; st = pb; sb = pn; sn = pt; }
    else { rand_u = rand_u * 4.0f - 3.0f; st = pn; sb = pt; sn = pb; }
    Color d = col(mt.c1);
    float sp_d;
    if (rand_u < 1.0f / 3.0f) { rand_u = 3.0f * rand_u; sp_d = d.r; }
    else if (rand_u < 2.0f / 3.0f) { rand_u = 3.0f * rand_u - 1.0f; sp_d = d.g; }
    else { rand_u = 3.0f * rand_u - 2.0f; sp_d = d.b; }
    float sample_r = ss_sample_r(rand_x) * sp_d;
    float r_max = ss_cdf_table().x[SPT_SS_CDF_SIZE - 1] * sp_d;
    if (sample_r < 0.0f) return false;
    float pihi = 2.0f * SPT_PI * rand_y;
    float pihi_cos = cx.m.cos(pihi), pihi_sin = cx.m.sin(pihi);
    float sample_l = spt_sqrt(r_max * r_max + sample_r * sample_r);
    Vec3 start_p = ((io.po + (st * pihi_cos) * sample_r) + (sb * pihi_sin) * sample_r) + sn * sample_l;
    Ray ray = make_ray(start_p, -sn);
    // the reference loops "until nothing is hit", but it re-uses `inter`, whose t bounds the next search from ABOVE
    // while t_min moves to just behind it: at most ONE intersection is ever collected (substrate.rs:280-291)
    Inter inter;
    inter.t = 2.0f * sample_l;
    if (!aggregate_intersect(cx, ray, inter)) return false;
    const spt_instance& in = cx.d->instances[inter.instance];
    const spt_surface& surf = cx.d->surfaces[in.surface];
    Coordinate coord_temp = surface_coord(cx, surf, ray, inter);
    Vec3 pi = point_at(ray, inter.t);
    Vec3 sample_normal = inter.normal;
    // sample_inter = min((rand_u * 1) as usize, 0) = 0
    Color sp = ss_sp(cx.m, d, length(pi - io.po));
    Vec3 offset = io.coord_po.to_local(pi - io.po);
    Vec3 nl = io.coord_po.to_local(sample_normal);
    float r_xy = spt_sqrt(offset.x * offset.x + offset.y * offset.y);
    float r_yz = spt_sqrt(offset.y * offset.y + offset.z * offset.z);
    float r_zx = spt_sqrt(offset.z * offset.z + offset.x * offset.x);
    float pdf_xy = 0.5f * spt_abs(nl.z) * avg(ss_sp(cx.m, d, r_xy));
    float pdf_yz = 0.25f * spt_abs(nl.x) * avg(ss_sp(cx.m, d, r_yz));
    float pdf_zx = 0.25f * spt_abs(nl.y) * avg(ss_sp(cx.m, d, r_zx));
    io.has = true;
    io.pi = pi;
    io.coord_pi = coord_temp;
    io.sp = sp;
    io.pdf_pi = ((pdf_xy + pdf_yz) + pdf_zx) / 1.0f;
    return true;
}

// PndfMicrofacet (src/bxdf/microfacet.rs:56-170) over the per-hit constants material_at left in the material record
inline spt_pndf_view pndf_view(const spt_scene_desc* d, const spt_material& mt) {
    return spt_pndf_view{d->pndfs + spt_f2u(mt.c1[2]), d->pndf_terms, d->pndf_nodes, d->pndf_refs, d->pndf_roots};
}
inline float pndf_half_pdf(const spt_scene_desc* d, const spt_material& mt, Vec3 half) {   // microfacet.rs:142-154
    const spt_pndf_view v = pndf_view(d, mt);
    return spt_pndf_calc(&v, mt.c1[1], spt_pndf_term_coe(v.pd, mt.c1[0]), mt.ax, mt.ay, half.x, half.y);
}
inline float pndf_ndf_visible(const spt_scene_desc* d, const spt_material& mt, Vec3 wo, Vec3 wi, Vec3 half) {   // microfacet.rs:156-169
    const float pndf = pndf_half_pdf(d, mt, half);
    const float visible = 0.25f / spt_max(wi.z * wo.z, 0.0001f);
    return pndf / spt_max(half.z, 0.0001f) * visible;
}
inline Vec3 pndf_sample_half(const spt_scene_desc* d, const spt_material& mt, Rng& rng, float* pdf) {   // microfacet.rs:98-140
    const spt_pndf_view v = pndf_view(d, mt);
    const spt_pndf& pd = *v.pd;
    const float sigma_p = mt.c1[1];
    const float sigma_p_sqr = sigma_p * sigma_p, sigma_p_sqr_inv = 1.0f / sigma_p_sqr;
    const float sigma_h_sqr = pd.sigma_hx * pd.sigma_hy, sigma_h_sqr_inv = 1.0f / sigma_h_sqr;
    const float sigma_sqr_sum_inv = 1.0f / (sigma_p_sqr + sigma_h_sqr);
    const float rand = rng.uniform_1d();
    uint32_t ti = 0xffffffffu;
    spt_pndf_uv_walk(&v, mt.ax, mt.ay, sigma_p, 1, mt.c1[0], rand, &ti);
    if (ti == 0xffffffffu) ti = pd.first_term;   // no term within reach (the reference indexes an empty list there)
    const spt_pndf_term& g = v.terms[ti];
    const float mux = sigma_sqr_sum_inv * (sigma_h_sqr * mt.ax + sigma_p_sqr * g.u[0]), muy = sigma_sqr_sum_inv * (sigma_h_sqr * mt.ay + sigma_p_sqr * g.u[1]);
    const float sigma = 1.0f / spt_sqrt(sigma_p_sqr_inv + sigma_h_sqr_inv);
    float gx, gy;
    rng.gaussian_2d(0.0f, sigma, &gx, &gy);
    const float ux = mux + gx, uy = muy + gy;
    float jx, jy;
    spt_m2_mul(g.jacobian, ux - g.u[0], uy - g.u[1], &jx, &jy);
    const float smx = g.s[0] + jx, smy = g.s[1] + jy;
    rng.gaussian_2d(0.0f, pd.sigma_r, &gx, &gy);
    const float sx = smx + gx, sy = smy + gy;
    const Vec3 half = normalize(v3(sx, sy, spt_sqrt(spt_clamp(1.0f - (sx * sx + sy * sy), 0.0f, 1.0f))));
    *pdf = spt_pndf_calc(&v, sigma_p, spt_pndf_term_coe(v.pd, mt.c1[0]), mt.ax, mt.ay, sx, sy);
    return half;
}

BxdfSample bxdf_sample(const Math& m, const spt_material& mt, Vec3 wo, Rng& rng, SubsurfaceIo* ss = nullptr, const spt_scene_desc* d = nullptr) {
    BxdfSample s;
    switch (mt.bxdf) {
    case SPT_BXDF_PNDF_CONDUCTOR: {  // MicrofacetConductor::sample (microfacet_conductor.rs:23-42) over a PndfMicrofacet
        float half_pdf;
        Vec3 half = pndf_sample_half(d, mt, rng, &half_pdf);
        Color fr = mat_fresnel(mt, wo, half);
        Vec3 wi = reflect_n(wo, half);
        s.wi = wi; s.dir = REFLECT;
        s.bxdf = fr * pndf_ndf_visible(d, mt, wo, wi, half);
        s.pdf = half_pdf / (4.0f * spt_abs(dot(wo, half)));
        return s;
    }
    case SPT_BXDF_LAMBERT: {  // src/bxdf/lambert.rs:20-36 + rng.rs:72-80
        float rx, ry;
        rng.uniform_2d(&rx, &ry);
        float phi = rx * 2.0f * SPT_PI;
        float sp, cp;
        m.sincos(phi, &sp, &cp);
        float sin_theta = spt_sqrt(ry);
        float cos_theta = spt_sqrt(1.0f - ry);
        Vec3 wi = v3(sin_theta * cp, sin_theta * sp, cos_theta);
        if (wo.z < 0.0f) wi.z = -wi.z;
        s.wi = wi; s.dir = REFLECT;
        s.bxdf = col(mt.c0) * SPT_FRAC_1_PI;
        s.pdf = spt_abs(wi.z) * SPT_FRAC_1_PI;
        return s;
    }
    case SPT_BXDF_MICROFACET_CONDUCTOR: {  // src/bxdf/microfacet_conductor.rs:23-42
        float r0, r1, half_pdf;
        rng.uniform_2d(&r0, &r1);
        Vec3 half = ggx_smith_vndf_sample(m, wo, mt.ax, mt.ay, r0, r1, &half_pdf);
        Color fr = mat_fresnel(mt, wo, half);
        Vec3 wi = reflect_n(wo, half);
        s.wi = wi; s.dir = REFLECT;
        s.bxdf = fr * ndf_visible(mt, wo, wi, half);
        s.pdf = half_pdf / (4.0f * spt_abs(dot(wo, half)));
        return s;
    }
    case SPT_BXDF_SPECULAR_CONDUCTOR: {  // src/bxdf/specular_conductor.rs:19-36
        Color fr = mat_fresnel(mt, wo, v3(0, 0, 1));
        Vec3 wi = reflect(wo);
        s.wi = wi; s.dir = REFLECT;
        s.bxdf = fr / spt_abs(wi.z);
        s.pdf = 1.0f;
        return s;
    }
    case SPT_BXDF_MICROFACET_DIELECTRIC: {  // src/bxdf/microfacet_dielectric.rs:23-86
        float r0, r1, half_pdf;
        rng.uniform_2d(&r0, &r1);
        Vec3 half = ggx_smith_vndf_sample(m, wo, mt.ax, mt.ay, r0, r1, &half_pdf);
        Color fr = mat_fresnel(mt, wo, half);
        float reflect_pdf = luminance(fr);
        Vec3 wi;
        if (rng.uniform_1d() < reflect_pdf) {
            wi = reflect_n(wo, half);
            s.wi = wi; s.dir = REFLECT;
            s.bxdf = fr * ndf_visible(mt, wo, wi, half);
            s.pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
        } else if (refract_n(wo, half, mt.ior, &wi)) {
            float ior_ratio = (wo.z >= 0.0f) ? 1.0f / mt.ior : mt.ior;
            float denom = ior_ratio * dot(wo, half) + dot(wi, half);
            denom = denom * denom;
            float num = spt_abs(dot(wi, half));
            s.pdf = (1.0f - reflect_pdf) * half_pdf * num / denom;
            num = 4.0f * spt_abs(dot(wo, half)) * spt_abs(dot(wi, half));
            s.bxdf = (gray(1.0f) - fr) * ndf_visible(mt, wo, wi, half) * num / denom;
            s.wi = wi; s.dir = TRANSMIT;
        } else {
            s.wi = v3(0, 0, 0); s.dir = TRANSMIT; s.bxdf = gray(0.0f); s.pdf = 1.0f;
        }
        return s;
    }
    case SPT_BXDF_SPECULAR_DIELECTRIC: {  // src/bxdf/specular_dielectric.rs:19-72
        Color fr = mat_fresnel(mt, wo, v3(0, 0, 1));
        float reflect_pdf = luminance(fr);
        Vec3 wi;
        if (rng.uniform_1d() < reflect_pdf) {
            wi = reflect(wo);
            s.wi = wi; s.dir = REFLECT;
            s.bxdf = fr / spt_abs(wi.z);
            s.pdf = reflect_pdf;
        } else if (refract(wo, mt.ior, &wi)) {
            float ior_ratio = (wo.z >= 0.0f) ? 1.0f / mt.ior : mt.ior;
            s.wi = wi; s.dir = TRANSMIT;
            s.bxdf = ior_ratio * ior_ratio * (gray(1.0f) - fr) / spt_abs(wi.z);
            s.pdf = 1.0f - reflect_pdf;
        } else {
            s.wi = v3(0, 0, 0); s.dir = TRANSMIT; s.bxdf = gray(0.0f); s.pdf = 1.0f;
        }
        return s;
    }
    case SPT_BXDF_PNDF_PLASTIC:
    case SPT_BXDF_MICROFACET_PLASTIC:
    case SPT_BXDF_SPECULAR_PLASTIC: {  // microfacet_plastic.rs:26-79, specular_plastic.rs:19-63
        const bool glint = mt.bxdf == SPT_BXDF_PNDF_PLASTIC;   // the same lobe over a PndfMicrofacet
        const bool rough = mt.bxdf == SPT_BXDF_MICROFACET_PLASTIC || glint;
        Color fresnel_macro = plastic_fresnel(mt, wo, v3(0, 0, 1));
        float specular_weight = luminance(fresnel_macro);
        float substrate_weight = luminance((gray(1.0f) - fresnel_macro) * substrate_reflectance(mt));
        float reflect_pdf = specular_weight / (specular_weight + substrate_weight);
        s.dir = REFLECT;
        if (rng.uniform_1d() < reflect_pdf) {
            Vec3 wi;
            Color specular_bxdf;
            float specular_pdf;
            if (rough) {
                float r0, r1, half_pdf;
                Vec3 half;
                if (glint) {
                    half = pndf_sample_half(d, mt, rng, &half_pdf);
                } else {
                    rng.uniform_2d(&r0, &r1);
                    half = ggx_smith_vndf_sample(m, wo, mt.ax, mt.ay, r0, r1, &half_pdf);
                }
                Color fr = plastic_fresnel(mt, wo, half);
                wi = reflect_n(wo, half);
                specular_bxdf = fr * (glint ? pndf_ndf_visible(d, mt, wo, wi, half) : ndf_visible(mt, wo, wi, half));
                specular_pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
            } else {
                wi = reflect(wo);
                specular_bxdf = fresnel_macro / spt_abs(wi.z);
                specular_pdf = reflect_pdf;
            }
            Color substrate_bxdf = (gray(1.0f) - fresnel_macro) * substrate_eval(mt, wo, wi);
            float sub_pdf = (1.0f - reflect_pdf) * substrate_pdf(mt, wo, wi);
            s.wi = wi;
            s.bxdf = specular_bxdf + substrate_bxdf;
            s.pdf = specular_pdf + sub_pdf;
        } else {
            // substrate.sample: cosine hemisphere (Lambert::sample / Diffuse::sample); Subsurface::sample first places
            // the exit point with a probe ray (3 draws) and hands back an all-zero sample when that finds nothing
            Vec3 wi = v3(0, 0, 0);
            Color samp_bxdf = gray(0.0f);
            float samp_pdf = 1.0f;
            bool probe_ok = true;
            if (mt.substrate == SPT_SUBSTRATE_SUBSURFACE) probe_ok = ss != nullptr && subsurface_probe(mt, rng, *ss);
            if (probe_ok) {
                float rx, ry;
                rng.uniform_2d(&rx, &ry);
                float phi = rx * 2.0f * SPT_PI;
                float sp, cp;
                m.sincos(phi, &sp, &cp);
                float sin_theta = spt_sqrt(ry);
                float cos_theta = spt_sqrt(1.0f - ry);
                wi = v3(sin_theta * cp, sin_theta * sp, cos_theta);
                if (wo.z < 0.0f) wi.z = -wi.z;
                if (mt.substrate != SPT_SUBSTRATE_LAMBERT) {
                    float fi = fresnel_n(mt.ior, wi, v3(0, 0, 1));
                    samp_bxdf = (1.0f - fi) * col(mt.c2);
                } else {
                    samp_bxdf = col(mt.c0) * SPT_FRAC_1_PI;
                }
                samp_pdf = spt_abs(wi.z) * SPT_FRAC_1_PI;
            }
            float sub_pdf = (1.0f - reflect_pdf) * samp_pdf;
            Color substrate_bxdf = (gray(1.0f) - fresnel_macro) * samp_bxdf;
            Color specular_bxdf;
            float specular_pdf;
            if (rough) {
                Vec3 half = half_from_reflect(wo, wi);
                float half_pdf = glint ? pndf_half_pdf(d, mt, half) : ggx_smith_vndf_pdf(half, wo, mt.ax, mt.ay);
                specular_pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
                Color fr = plastic_fresnel(mt, wo, half);
                specular_bxdf = fr * (glint ? pndf_ndf_visible(d, mt, wo, wi, half) : ndf_visible(mt, wo, wi, half));
            } else {
                specular_pdf = reflect_pdf;
                specular_bxdf = fresnel_macro / spt_abs(wi.z);
            }
            s.wi = wi;
            s.bxdf = substrate_bxdf + specular_bxdf;
            s.pdf = sub_pdf + specular_pdf;
        }
        return s;
    }
    default: {  // SPT_BXDF_PSEUDO, src/bxdf/pseudo.rs:14-27
        s.wi = -wo; s.dir = TRANSMIT;
        s.bxdf = gray(1.0f) / spt_abs(wo.z);
        s.pdf = 1.0f;
        return s;
    }
    }
}

float bxdf_pdf(const spt_material& mt, Vec3 wo, Vec3 wi, const spt_scene_desc* d = nullptr) {
    switch (mt.bxdf) {
    case SPT_BXDF_PNDF_CONDUCTOR:  // microfacet_conductor.rs:44-53
        if (wo.z * wi.z >= 0.0f) {
            Vec3 half = half_from_reflect(wo, wi);
            return pndf_half_pdf(d, mt, half) / (4.0f * spt_abs(dot(wo, half)));
        }
        return 1.0f;
    case SPT_BXDF_LAMBERT:  // lambert.rs:38-44 (1.0, not 0, across hemispheres: quirk Q15)
        return (wo.z * wi.z >= 0.0f) ? spt_abs(wi.z) * SPT_FRAC_1_PI : 1.0f;
    case SPT_BXDF_MICROFACET_CONDUCTOR:  // microfacet_conductor.rs:44-53
        if (wo.z * wi.z >= 0.0f) {
            Vec3 half = half_from_reflect(wo, wi);
            float half_pdf = ggx_smith_vndf_pdf(half, wo, mt.ax, mt.ay);
            return half_pdf / (4.0f * spt_abs(dot(wo, half)));
        }
        return 1.0f;
    case SPT_BXDF_MICROFACET_DIELECTRIC: {  // microfacet_dielectric.rs:88-113
        if (wo.z * wi.z >= 0.0f) {
            Vec3 half = half_from_reflect(wo, wi);
            float half_pdf = ggx_smith_vndf_pdf(half, wo, mt.ax, mt.ay);
            float reflect_pdf = luminance(mat_fresnel(mt, wo, half));
            return reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
        }
        Vec3 half = half_from_refract(wo, wi, mt.ior);
        float half_pdf = ggx_smith_vndf_pdf(half, wo, mt.ax, mt.ay);
        float reflect_pdf = luminance(mat_fresnel(mt, wo, half));
        float ior_ratio = (wo.z >= 0.0f) ? 1.0f / mt.ior : mt.ior;
        float denom = ior_ratio * dot(wo, half) + dot(wi, half);
        denom = denom * denom;
        float num = spt_abs(dot(wi, half));
        return (1.0f - reflect_pdf) * half_pdf * num / denom;
    }
    case SPT_BXDF_SPECULAR_DIELECTRIC: {  // specular_dielectric.rs:74-82
        float reflect_pdf = luminance(mat_fresnel(mt, wo, v3(0, 0, 1)));
        return (wo.z * wi.z >= 0.0f) ? reflect_pdf : 1.0f - reflect_pdf;
    }
    case SPT_BXDF_PNDF_PLASTIC:
    case SPT_BXDF_MICROFACET_PLASTIC:
    case SPT_BXDF_SPECULAR_PLASTIC: {  // microfacet_plastic.rs:81-99, specular_plastic.rs:65-80
        if (!(wo.z * wi.z >= 0.0f)) return 1.0f;
        Color fresnel_macro = plastic_fresnel(mt, wo, v3(0, 0, 1));
        float specular_weight = luminance(fresnel_macro);
        float substrate_weight = luminance((gray(1.0f) - fresnel_macro) * substrate_reflectance(mt));
        float reflect_pdf = specular_weight / (specular_weight + substrate_weight);
        float specular_pdf;
        if (mt.bxdf != SPT_BXDF_SPECULAR_PLASTIC) {
            Vec3 half = half_from_reflect(wo, wi);
            float half_pdf = mt.bxdf == SPT_BXDF_PNDF_PLASTIC ? pndf_half_pdf(d, mt, half) : ggx_smith_vndf_pdf(half, wo, mt.ax, mt.ay);
            specular_pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
        } else {
            specular_pdf = reflect_pdf;
        }
        float sub_pdf = (1.0f - reflect_pdf) * substrate_pdf(mt, wo, wi);
        return specular_pdf + sub_pdf;
    }
    default:  // specular conductor / pseudo
        return 1.0f;
    }
}

Color bxdf_eval(const spt_material& mt, Vec3 wo, Vec3 wi, const spt_scene_desc* d = nullptr) {
    switch (mt.bxdf) {
    case SPT_BXDF_PNDF_CONDUCTOR:  // microfacet_conductor.rs:55-64
        if (wo.z * wi.z >= 0.0f) {
            Vec3 half = half_from_reflect(wo, wi);
            return mat_fresnel(mt, wo, half) * pndf_ndf_visible(d, mt, wo, wi, half);
        }
        return gray(0.0f);
    case SPT_BXDF_LAMBERT:  // lambert.rs:46-52
        return (wo.z * wi.z >= 0.0f) ? col(mt.c0) * SPT_FRAC_1_PI : gray(0.0f);
    case SPT_BXDF_MICROFACET_CONDUCTOR:  // microfacet_conductor.rs:55-64
        if (wo.z * wi.z >= 0.0f) {
            Vec3 half = half_from_reflect(wo, wi);
            return mat_fresnel(mt, wo, half) * ndf_visible(mt, wo, wi, half);
        }
        return gray(0.0f);
    case SPT_BXDF_SPECULAR_CONDUCTOR: {  // specular_conductor.rs:42-50
        if (dot(wi, reflect(wo)) > 0.999f) return mat_fresnel(mt, wo, v3(0, 0, 1)) / spt_abs(wi.z);
        return gray(0.0f);
    }
    case SPT_BXDF_MICROFACET_DIELECTRIC: {  // microfacet_dielectric.rs:115-138
        if (wo.z * wi.z >= 0.0f) {
            Vec3 half = half_from_reflect(wo, wi);
            return mat_fresnel(mt, wo, half) * ndf_visible(mt, wo, wi, half);
        }
        Vec3 half = half_from_refract(wo, wi, mt.ior);
        Color fr = mat_fresnel(mt, wo, half);
        float ior_ratio = (wo.z >= 0.0f) ? 1.0f / mt.ior : mt.ior;
        float denom = ior_ratio * dot(wo, half) + dot(wi, half);
        denom = denom * denom;
        float num = 4.0f * spt_abs(dot(wo, half)) * spt_abs(dot(wi, half));
        return (gray(1.0f) - fr) * ndf_visible(mt, wo, wi, half) * num / denom;
    }
    case SPT_BXDF_SPECULAR_DIELECTRIC: {  // specular_dielectric.rs:84-107
        Color fr = mat_fresnel(mt, wo, v3(0, 0, 1));
        if (wo.z * wi.z >= 0.0f) {
            if (dot(wi, reflect(wo)) > 0.999f) return fr / spt_abs(wi.z);
            return gray(0.0f);
        }
        Vec3 ewi;
        if (refract(wo, mt.ior, &ewi)) {
            if (dot(wi, ewi) > 0.999f) {
                float ior_ratio = (wo.z >= 0.0f) ? 1.0f / mt.ior : mt.ior;
                return ior_ratio * ior_ratio * (gray(1.0f) - fr) / spt_abs(wi.z);
            }
        }
        return gray(0.0f);
    }
    case SPT_BXDF_PNDF_PLASTIC:
    case SPT_BXDF_MICROFACET_PLASTIC: {  // microfacet_plastic.rs:101-118
        if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
        Vec3 half = half_from_reflect(wo, wi);
        Color refl = plastic_fresnel(mt, wo, half) * (mt.bxdf == SPT_BXDF_PNDF_PLASTIC ? pndf_ndf_visible(d, mt, wo, wi, half) : ndf_visible(mt, wo, wi, half));
        Color fresnel_macro = plastic_fresnel(mt, wo, v3(0, 0, 1));
        Color sub = (gray(1.0f) - fresnel_macro) * substrate_eval(mt, wo, wi);
        return refl + sub;
    }
    case SPT_BXDF_SPECULAR_PLASTIC: {  // specular_plastic.rs:82-93 (the mirror term is NOT restricted to the mirror direction)
        if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
        Color fr = plastic_fresnel(mt, wo, v3(0, 0, 1));
        Color refl = fr / spt_abs(wi.z);
        Color sub = (gray(1.0f) - fr) * substrate_eval(mt, wo, wi);
        return refl + sub;
    }
    default:  // pseudo.rs:32-38
        if (dot(wo, wi) < -0.999f) return gray(1.0f) / spt_abs(wi.z);
        return gray(0.0f);
    }
}

// ---------------------------------------------------------------- AliasTable::sample (src/core/alias_table.rs:60-68)
inline uint32_t alias_sample(const spt_alias_table& a, float rand, float* prob) {
    float temp = rand * (float)a.n;
    uint32_t x = spt_f2u_sat(temp);
    float y = temp - (float)x;
    if (y < a.u[x]) { *prob = a.props[x]; return x; }
    *prob = a.props[a.k[x]];
    return a.k[x];
}

// ---------------------------------------------------------------- lights
struct LightSample {
    Vec3 dir;
    float pdf;
    Color strength;
    float dist;
    bool is_delta;
};

// EnvLight::strength_dist_pdf(theta, phi) (src/light/environment.rs:51-84), quirk Q5 kept
inline void env_lookup(const spt_env& e, float theta, float phi, Color* c_out, float* p_out) {
    int32_t W = (int32_t)e.width, H = (int32_t)e.height;
    float x = phi * 0.5f * SPT_FRAC_1_PI * (float)e.width;
    int32_t x1 = spt_f2i_sat(spt_round(x));
    int32_t x0 = (int32_t)((uint32_t)x1 - 1u);   // wrapping, as release-mode Rust
    float xt = x - (float)x0 - 0.5f;
    uint32_t ux0 = (uint32_t)(x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0));
    uint32_t ux1 = (uint32_t)(x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1));
    float y = theta * SPT_FRAC_1_PI * (float)e.height;
    int32_t y1 = spt_f2i_sat(spt_round(y));
    int32_t y0 = (int32_t)((uint32_t)y1 - 1u);
    float yt = y - (float)y0 - 0.5f;
    uint32_t uy0 = (uint32_t)(y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0));
    uint32_t uy1 = (uint32_t)(y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1));
    auto tex = [&](uint32_t yy, uint32_t xx) { return col(e.texels + 3 * ((size_t)yy * e.width + xx)); };
    Color c00 = tex(uy0, ux0), c01 = tex(uy1, ux0), c10 = tex(uy0, ux1), c11 = tex(uy1, ux1);
    Color c0 = c00 * (1.0f - yt) + c01 * yt;
    Color c1 = c10 * (1.0f - yt) + c11 * yt;
    Color c = c0 * (1.0f - xt) + c1 * xt;
    float p00 = e.alias.props[(size_t)uy0 * e.width + ux0], p01 = e.alias.props[(size_t)uy1 * e.width + ux0];
    float p10 = e.alias.props[(size_t)uy0 * e.width + ux1], p11 = e.alias.props[(size_t)uy1 * e.width + ux1];
    float p0 = p00 * (1.0f - yt) + p01 * yt;
    float p1 = p10 * (1.0f - yt) + p11 * yt;
    float p = p0 * (1.0f - xt) * p1 * xt;
    *c_out = c * col(e.scale);
    *p_out = p;
}
// LightT::strength_dist_pdf for the env (environment.rs:128-133)
inline void env_strength_pdf(const Ctx& cx, Vec3 wi, Color* c, float* pdf) {
    float theta = cx.m.acos(wi.y);
    float phi = cx.m.atan2(wi.x, wi.z) + SPT_PI;
    env_lookup(cx.d->env, theta, phi, c, pdf);
}

// Instance::sample (src/primitive/instance.rs:111-129) over Sphere::sample / BvhAccel<Triangle>::sample
struct ShapeSample {
    Vec3 position, normal, tangent, bitangent;
    float texcoords[2];
    float pdf;
};
inline ShapeSample instance_sample(const Ctx& cx, const spt_instance& in, Rng& rng) {
    ShapeSample s;
    float pdf;
    if (in.prim_type == SPT_PRIM_SPHERE) {  // sphere.rs:103-136 + rng.rs:58-65
        const spt_sphere& sp = cx.d->spheres[in.prim_id];
        float rx, ry;
        rng.uniform_2d(&rx, &ry);
        float phi = rx * 2.0f * SPT_PI;
        float sphi, cphi;
        cx.m.sincos(phi, &sphi, &cphi);
        float cos_theta = 1.0f - 2.0f * ry;
        float sin_theta = spt_sqrt(1.0f - cos_theta * cos_theta);
        Vec3 norm = v3(sin_theta * cphi, sin_theta * sphi, cos_theta);
        s.position = v3(sp.center) + norm * sp.radius;
        s.normal = norm;
        sphere_normal_to_texcoords(cx, norm, s.texcoords);
        sphere_frame(norm, &s.tangent, &s.bitangent);
        pdf = 0.25f * SPT_FRAC_1_PI;
    } else {  // bvh.rs:293-298 + triangle.rs:224-271
        const spt_mesh& mesh = cx.d->meshes[in.prim_id];
        float fi = rng.uniform_1d() * (float)mesh.tri_count;
        uint32_t idx = spt_f2u_sat(fi);
        if (idx > mesh.tri_count - 1) idx = mesh.tri_count - 1;
        uint32_t tri = mesh.tri_first + idx;
        float r0, r1;
        rng.uniform_2d(&r0, &r1);
        float r0_sqrt = spt_sqrt(r0);
        float u = 1.0f - r0_sqrt;
        float v = r0_sqrt * (1.0f - r1);
        float w = 1.0f - u - v;
        const spt_tri_pos& tp = cx.d->tri_pos[tri];
        const spt_tri_attr& a = cx.d->tri_attr[tri];
        Vec3 p0 = v3(tp.p0), p1 = v3(tp.p1), p2 = v3(tp.p2);
        s.position = (p0 * u + p1 * v) + p2 * w;
        float area = length(cross(p1 - p0, p2 - p0)) * 0.5f;
        s.normal = (v3(a.n[0]) * u + v3(a.n[1]) * v) + v3(a.n[2]) * w;
        s.tangent = (v3(a.t[0]) * u + v3(a.t[1]) * v) + v3(a.t[2]) * w;
        s.bitangent = (v3(a.b[0]) * u + v3(a.b[1]) * v) + v3(a.b[2]) * w;
        s.texcoords[0] = (a.uv[0][0] * u + a.uv[1][0] * v) + a.uv[2][0] * w;   // Vec2 * f32 sums (triangle.rs:258)
        s.texcoords[1] = (a.uv[0][1] * u + a.uv[1][1] * v) + a.uv[2][1] * w;
        pdf = (1.0f / spt_max(area, 0.001f)) / (float)mesh.tri_count;
    }
    float original_area = length(cross(s.tangent, s.bitangent));
    s.position = xf_point(in.fwd, s.position);
    s.normal = normalize(mat3_mul(in.nrm, s.normal));
    s.bitangent = xf_vector(in.fwd, s.bitangent);
    s.tangent = xf_vector(in.fwd, s.tangent);
    float transformed_area = length(cross(s.tangent, s.bitangent));
    s.pdf = pdf * original_area / transformed_area;
    return s;
}
// Instance::pdf (instance.rs:131-141) over Triangle::pdf (triangle.rs:273-281) / Sphere::pdf
inline float instance_pdf(const Ctx& cx, const spt_instance& in, const Inter& inter) {
    Vec3 tangent = xf_vector(in.inv, inter.tangent);
    Vec3 bitangent = xf_vector(in.inv, inter.bitangent);
    float original_area = length(cross(tangent, bitangent));
    float transformed_area = length(cross(inter.tangent, inter.bitangent));
    float prim_pdf;
    if (in.prim_type == SPT_PRIM_SPHERE) {
        prim_pdf = 0.25f * SPT_FRAC_1_PI;
    } else {
        const spt_mesh& mesh = cx.d->meshes[in.prim_id];
        const spt_tri_pos& tp = cx.d->tri_pos[inter.prim];
        Vec3 p0 = v3(tp.p0), p1 = v3(tp.p1), p2 = v3(tp.p2);
        float area = length(cross(p1 - p0, p2 - p0)) * 0.5f;
        prim_pdf = (1.0f / spt_max(area, 0.001f)) / (float)mesh.tri_count;
    }
    return prim_pdf * original_area / transformed_area;
}

// LightT::sample for every light type
inline void light_sample(const Ctx& cx, const spt_light& l, Vec3 position, Rng& rng, LightSample* out) {
    switch (l.type) {
    case SPT_LIGHT_DIRECTIONAL:  // directional.rs:26-29
        out->dir = -v3(l.dir); out->pdf = 1.0f; out->strength = col(l.strength); out->dist = SPT_F32_MAX; out->is_delta = true;
        return;
    case SPT_LIGHT_POINT: {  // point.rs:23-29
        Vec3 sv = v3(l.pos) - position;
        float dist_sqr = length_squared(sv);
        float dist = spt_sqrt(dist_sqr);
        out->dir = sv / dist; out->pdf = 1.0f; out->strength = col(l.strength) / dist_sqr; out->dist = dist; out->is_delta = true;
        return;
    }
    case SPT_LIGHT_SPOT: {  // spot.rs:50-65
        Vec3 sv = v3(l.pos) - position;
        float dist_sqr = length_squared(sv);
        float dist = spt_sqrt(dist_sqr);
        sv = sv / dist;
        float atten = spt_clamp((dot(v3(l.dir), -sv) - l.cos_outer) / spt_max(l.cos_inner - l.cos_outer, 0.0001f), 0.0f, 1.0f);
        out->dir = sv; out->pdf = 1.0f; out->strength = (col(l.strength) * atten) / dist_sqr; out->dist = dist; out->is_delta = true;
        return;
    }
    case SPT_LIGHT_SHAPE: {  // shape_light.rs:20-42
        const spt_instance& in = cx.d->instances[l.instance];
        const spt_surface& sf = cx.d->surfaces[in.surface];
        ShapeSample s = instance_sample(cx, in, rng);
        Inter at;   // the sampled point as the Intersection Surface::emissive sees (no differentials)
        at.position = s.position; at.normal = s.normal; at.tangent = s.tangent; at.bitangent = s.bitangent;
        at.texcoords[0] = s.texcoords[0]; at.texcoords[1] = s.texcoords[1];
        Color emissive = surface_emissive(cx, sf, at);
        Vec3 light_vec = s.position - position;
        float dist_sqr = length_squared(light_vec);
        float dist = spt_sqrt(dist_sqr);
        Vec3 light_dir = light_vec / dist;
        float cosv;
        if (sf.flags & SPT_SURF_DOUBLE_SIDED) {
            cosv = spt_abs(dot(light_dir, s.normal));
        } else {
            cosv = dot(light_dir, -s.normal);
            if (!(cosv > 0.0f)) { cosv = 1.0f; emissive = gray(0.0f); }
        }
        out->dir = light_dir; out->pdf = s.pdf * dist_sqr / spt_max(cosv, 0.001f); out->strength = emissive; out->dist = dist; out->is_delta = false;
        return;
    }
    default: {  // SPT_LIGHT_ENV, environment.rs:110-126
        const spt_env& e = cx.d->env;
        float pr;
        uint32_t ind = alias_sample(e.alias, rng.uniform_1d(), &pr);
        uint32_t x = ind % e.width, y = ind / e.width;
        float rx, ry;
        rng.uniform_2d(&rx, &ry);
        float theta = ((float)y + ry) / (float)e.height * SPT_PI;
        float phi = ((float)x + rx) / (float)e.width * 2.0f * SPT_PI;
        float st, ct, sp, cp;
        cx.m.sincos(theta, &st, &ct);
        cx.m.sincos(phi, &sp, &cp);
        out->dir = v3(st * sp, ct, st * cp);
        env_lookup(e, theta, phi, &out->strength, &out->pdf);
        out->dist = spt_inf(); out->is_delta = false;
        return;
    }
    }
}

// LightSamplerT::sample_light (uniform.rs:28-41, power_is.rs:49-59); false when the scene has no light (D5)
inline bool sample_light(const Ctx& cx, Vec3 position, Rng& rng, LightSample* out) {
    const spt_scene_desc& d = *cx.d;
    if (d.n_lights == 0) return false;
    if (d.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS) {
        float pr;
        uint32_t index = alias_sample(d.light_alias, rng.uniform_1d(), &pr);
        light_sample(cx, d.lights[index], position, rng, out);
        out->pdf = pr * out->pdf;
    } else {
        float fi = rng.uniform_1d() * (float)d.n_lights;
        uint32_t index = spt_f2u_sat(fi);
        if (index > d.n_lights - 1) index = d.n_lights - 1;
        light_sample(cx, d.lights[index], position, rng, out);
        out->pdf = out->pdf * (1.0f / (float)d.n_lights);
    }
    return true;
}
// pdf_shape_light (uniform.rs:43-68, power_is.rs:61-88)
inline float pdf_shape_light(const Ctx& cx, Vec3 position, const Inter& inter) {
    const spt_scene_desc& d = *cx.d;
    const spt_instance& in = d.instances[inter.instance];
    const spt_surface& sf = d.surfaces[in.surface];
    float primitive_pdf = instance_pdf(cx, in, inter);
    Vec3 light_vec = inter.position - position;
    float dist_sqr = length_squared(light_vec);
    Vec3 light_dir = light_vec / spt_sqrt(dist_sqr);
    float cosv;
    if (sf.flags & SPT_SURF_DOUBLE_SIDED) {
        cosv = spt_abs(dot(light_dir, inter.normal));
    } else {
        cosv = dot(light_dir, -inter.normal);
        if (!(cosv > 0.0f)) cosv = 1.0f;
    }
    float local_pdf = primitive_pdf * dist_sqr / spt_max(cosv, 0.00001f);
    if (d.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS) return local_pdf * d.light_alias.props[in.light];
    return local_pdf * (1.0f / (float)d.n_lights);
}
// pdf_env_light (uniform.rs:70-76, power_is.rs:90-96 with the true env index, quirk Q6)
inline float pdf_env_light(const Ctx& cx) {
    const spt_scene_desc& d = *cx.d;
    if (d.env_light_index < 0) return 1.0f;
    if (d.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS) return d.light_alias.props[d.env_light_index];
    return 1.0f / (float)d.n_lights;
}

// ---------------------------------------------------------------- src/medium/{homogeneous,util}.rs
inline float henyey_greenstein(float g, float cosv) {  // util.rs:1-7
    float g2 = g * g;
    float denom = 1.0f + g2 + 2.0f * g * cosv;
    denom = denom * spt_sqrt(denom);
    return 0.25f * SPT_FRAC_1_PI * (1.0f - g2) / denom;
}
inline float henyey_greenstein_cdf_inverse(float g, float rand) {  // util.rs:9-18
    if (spt_abs(g) < 0.01f) return 1.0f - 2.0f * rand;
    float g2 = g * g;
    float temp = (1.0f - g2) / (1.0f - g + 2.0f * g * rand);
    return 0.5f * (1.0f + g2 - temp * temp) / g;
}
inline Vec3 hg_local_to_world(Vec3 wo_world, Vec3 wi_local) {  // util.rs:20-30
    Vec3 v = (spt_abs(wo_world.y) < 0.99f) ? v3(0, 1, 0) : v3(1, 0, 0);
    Vec3 u = normalize(cross(v, wo_world));
    v = cross(wo_world, u);
    return (u * wi_local.x + v * wi_local.y) + wo_world * wi_local.z;
}
// Homogeneous::sample_pi (homogeneous.rs:30-58)
inline void medium_sample_pi(const Ctx& cx, const spt_medium& md, Vec3 po, Vec3 wo, float t_max, Rng& rng, Vec3* pi, bool* still_in, Color* atten_out) {
    float rx, ry;
    rng.uniform_2d(&rx, &ry);
    float sample_sigma_t = (rx < 1.0f / 3.0f) ? md.sigma_t[0] : ((rx < 2.0f / 3.0f) ? md.sigma_t[1] : md.sigma_t[2]);
    float sample_t = -cx.m.ln(1.0f - ry) / sample_sigma_t;
    Color sigma_t = col(md.sigma_t);
    float tt = spt_min(sample_t, t_max);
    Color attenuation = cx.m.exp((-sigma_t) * tt);
    *pi = po - wo * tt;
    if (sample_t < t_max) {
        float atten_pdf = avg(sigma_t * attenuation);
        *still_in = true;
        *atten_out = attenuation * col(md.sigma_s) / atten_pdf;
    } else {
        float atten_pdf = avg(attenuation);
        *still_in = false;
        *atten_out = attenuation / atten_pdf;
    }
}
// Homogeneous::sample_wi (homogeneous.rs:60-70)
inline Vec3 medium_sample_wi(const Ctx& cx, const spt_medium& md, Vec3 wo, Rng& rng, float* pdf) {
    float rx, ry;
    rng.uniform_2d(&rx, &ry);
    float cos_theta = henyey_greenstein_cdf_inverse(md.g, rx);
    float sin_theta = spt_sqrt(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * SPT_PI * ry;
    float sp, cp;
    cx.m.sincos(phi, &sp, &cp);
    Vec3 wi = hg_local_to_world(wo, v3(sin_theta * cp, sin_theta * sp, cos_theta));
    *pdf = henyey_greenstein(md.g, cos_theta);
    return wi;
}

// ---------------------------------------------------------------- src/renderer/pt.rs
inline float power_heuristic(float p0, float p1) {  // pt.rs:298-302 with n0 = n1 = 1
    float prod0 = 1.0f * p0;
    float prod1 = 1.0f * p1;
    return prod0 * prod0 / (prod0 * prod0 + prod1 * prod1);
}

// pt.rs:212-233.  `medium_primitive` is the raw Triangle / Sphere last hit, intersected with the
// WORLD-space shadow ray, i.e. without its instance transform (reference behaviour, kept).
inline Ray shadow_ray_from_medium(const Ctx& cx, Vec3 p, Vec3 light_dir, float light_dist, int prim_type, int prim, float* transported) {
    Ray shadow_ray = make_ray(p, light_dir);
    Inter temp;
    temp.t = light_dist - 0.001f;
    bool hit = false;
    if (prim_type == SPT_PRIM_MESH) hit = triangle_intersect(cx, (uint32_t)prim, shadow_ray, temp);
    else if (prim_type == SPT_PRIM_SPHERE) hit = sphere_intersect(cx, (uint32_t)prim, shadow_ray, temp);
    else if (prim_type == SPT_PRIM_BEZIER) hit = bezier_intersect(cx, (uint32_t)prim, shadow_ray, temp);
    if (hit) {
        *transported = temp.t;
        shadow_ray.t_min += temp.t;
    } else {
        *transported = light_dist;
        shadow_ray.t_min += light_dist - 0.001f;
    }
    return shadow_ray;
}

// PathTracer::trace_ray (pt.rs:39-210)
// debug_normal: the reference's cargo feature of that name (Cargo.toml:34-36, pt.rs:113-118)
Color trace_ray(const Ctx& cx, Ray ray, Rng& rng, uint32_t max_depth, bool debug_normal = false) {
    const spt_scene_desc& d = *cx.d;
    Color final_color = gray(0.0f);
    Color throughput = gray(1.0f);
    uint32_t curr_depth = 0;
    int32_t curr_medium = -1;
    Vec3 lsi_position = v3(0, 0, 0);  // light_sampler_inputs (normal is never read)
    float last_sample_pdf = 0.0f;

    while (curr_depth < max_depth) {
        Inter inter;
        bool does_hit = aggregate_intersect(cx, ray, inter);
        if (does_hit) calc_differential(inter, ray);   // pt.rs:51-53

        if (curr_medium >= 0) {  // pt.rs:56-96
            const spt_medium& md = d.mediums[curr_medium];
            Vec3 wo = -ray.direction;
            Vec3 pi;
            bool still_in;
            Color attenuation;
            medium_sample_pi(cx, md, ray.origin, wo, inter.t, rng, &pi, &still_in, &attenuation);
            throughput = throughput * attenuation;
            if (!still_in) {
                curr_medium = -1;
                continue;
            }
            Color li = gray(0.0f);
            lsi_position = pi;
            LightSample ls;
            if (sample_light(cx, lsi_position, rng, &ls)) {
                float phase = henyey_greenstein(md.g, dot(wo, ls.dir));
                float transported;
                Ray shadow_ray = shadow_ray_from_medium(cx, pi, ls.dir, ls.dist, inter.prim_type, inter.prim, &transported);
                Color atten = cx.m.exp((-col(md.sigma_t)) * transported);
                if (ls.pdf != 0.0f && spt_is_finite(ls.pdf) && !aggregate_intersect_test(cx, shadow_ray, ls.dist - 0.001f)) {
                    if (ls.is_delta) {
                        li = atten * phase * ls.strength / ls.pdf;
                    } else {
                        float weight = power_heuristic(ls.pdf, phase);
                        li = atten * phase * ls.strength * weight / ls.pdf;
                    }
                }
            }
            final_color = final_color + throughput * li;
            float pdf;
            Vec3 wi = medium_sample_wi(cx, md, wo, rng, &pdf);
            last_sample_pdf = pdf;
            ray = make_ray(pi, wi);
        } else if (!does_hit) {  // pt.rs:97-111
            if (d.env.width != 0) {
                Color env;
                float env_pdf;
                env_strength_pdf(cx, ray.direction, &env, &env_pdf);
                float weight = 1.0f;
                if (curr_depth != 0) {
                    float pdf = pdf_env_light(cx) * env_pdf;
                    weight = power_heuristic(last_sample_pdf, pdf);
                }
                final_color = final_color + throughput * env * weight;
            }
            break;
        } else {  // pt.rs:112-193
            if (debug_normal) {   // pt.rs:113-118: the colour IS the normal of the first surface the path reaches
                Color normal_color = Color{inter.normal.x, inter.normal.y, inter.normal.z};
                normal_color = normal_color * 0.5f + gray(0.5f);
                final_color = normal_color;
                break;
            }
            Vec3 po = point_at(ray, inter.t);
            const spt_instance& in = d.instances[inter.instance];
            const spt_surface& surf = d.surfaces[in.surface];
            const spt_material mt = material_at(cx, d.materials[surf.material], inter);   // Surface::scatter_and_coord
            Coordinate coord_po = surface_coord(cx, surf, ray, inter);

            Color li_emissive = surface_emissive(cx, surf, inter);
            if (luminance(li_emissive) > 0.0f) {
                float weight = 1.0f;
                if (curr_depth != 0) {
                    float pdf = pdf_shape_light(cx, lsi_position, inter);
                    weight = power_heuristic(last_sample_pdf, pdf);
                }
                final_color = final_color + throughput * li_emissive * weight;
            }

            Vec3 wo = coord_po.to_local(-ray.direction);
            SubsurfaceIo ss;
            ss.cx = &cx; ss.po = po; ss.coord_po = coord_po;
            BxdfSample samp = bxdf_sample(cx.m, mt, wo, rng, &ss, cx.d);
            if (ss.has) {  // pt.rs:147-151
                po = ss.pi;
                coord_po = ss.coord_pi;
                throughput = throughput * (ss.sp / ss.pdf_pi);
            }

            Color li = gray(0.0f);
            lsi_position = po;
            if (!bxdf_is_delta(mt)) {
                LightSample ls;
                if (sample_light(cx, lsi_position, rng, &ls)) {
                    Vec3 wi = coord_po.to_local(ls.dir);
                    Color f = bxdf_eval(mt, wo, wi, cx.d);
                    float mat_pdf = bxdf_pdf(mt, wo, wi, cx.d);
                    Ray shadow_ray = make_ray(po, ls.dir);
                    shadow_ray.t_min = T_MIN_EPS / spt_max(spt_abs(wi.z), 0.00001f);
                    if (ls.pdf != 0.0f && spt_is_finite(ls.pdf) && !aggregate_intersect_test(cx, shadow_ray, ls.dist - 0.001f)) {
                        if (ls.is_delta) {
                            li = ls.strength * f * spt_abs(wi.z) / spt_max(ls.pdf, 0.00001f);
                        } else {
                            float weight = power_heuristic(ls.pdf, mat_pdf);
                            li = ls.strength * f * spt_abs(wi.z) * weight / spt_max(ls.pdf, 0.00001f);
                        }
                    }
                }
                final_color = final_color + throughput * li;
            }

            last_sample_pdf = samp.pdf;
            Vec3 wi_world = coord_po.to_world(samp.wi);
            ray = make_ray(po, wi_world);
            ray.t_min = T_MIN_EPS / spt_max(spt_abs(samp.wi.z), 0.00001f);
            throughput = throughput * (samp.bxdf * spt_abs(samp.wi.z) / spt_max(samp.pdf, 0.00001f));
            // Coordinate::in_expected_hemisphere (coord.rs:53-59)
            float hd = dot(wi_world, coord_po.hemisphere);
            if (!((samp.dir == REFLECT) ? (hd >= 0.0f) : (hd <= 0.0f))) break;
            if (dot(wi_world, inter.normal) < 0.0f) {
                // Surface::inside_medium (surface.rs:105-111): none when double sided
                curr_medium = (surf.flags & SPT_SURF_DOUBLE_SIDED) ? -1 : surf.inside_medium;
            }
        }

        if (!is_finite(throughput)) break;
        float rr_rand = rng.uniform_1d();
        float rr_prop = spt_clamp(luminance(throughput), 0.001f, 0.95f);
        if (rr_rand > rr_prop) break;
        throughput = throughput * (1.0f / rr_prop);  // DivAssign<f32>: multiply by the reciprocal
        curr_depth += 1;
    }
    return final_color;
}

// PerspectiveCamera::generate_ray (src/camera/perspective.rs:40-47)
inline Ray camera_ray(const spt_camera& c, float x, float y) {
    Vec3 dir = normalize((v3(c.forward) * c.half_cot_half_fov + v3(c.right) * x) + v3(c.up) * y);
    return make_ray(v3(c.eye), dir);
}

// pixel sampler (src/pixel_sampler/{random,jittered,recurrence}.rs); jittered: the reference's end
// test never fires (jittered.rs:47), the evident division_x*division_y samples are produced.
inline void pixel_offset(const spt_render_params& p, uint32_t pixel, uint32_t s, Rng& rng, float* ox, float* oy) {
    if (p.sampler == SPT_SAMPLER_RECURRENCE) {
        spt_r2_offset(pixel, p.spp, s, ox, oy);
    } else if (p.sampler == SPT_SAMPLER_JITTERED) {
        uint32_t ix = s % p.division_x, iy = s / p.division_x;
        float inv_x = 1.0f / (float)p.division_x, inv_y = 1.0f / (float)p.division_y;
        *ox = ((float)ix + rng.uniform_1d()) * inv_x;
        *oy = ((float)iy + rng.uniform_1d()) * inv_y;
    } else {
        *ox = rng.uniform_1d();
        *oy = rng.uniform_1d();
    }
}

bool row_in_shard(const spt_render_params& p, uint32_t j) {
    uint32_t sc = p.shard_count ? p.shard_count : 1, sr = p.strip_rows ? p.strip_rows : 1;
    return (j / sr) % sc == p.shard_index;
}

}  // namespace

// ---------------------------------------------------------------- C interface (oracle.h)
extern "C" {

// PathTracer::render (pt.rs:237-296): threads over contiguous row bands (util.rs:6-19)
int oracle_render(const spt_scene_desc* desc, const spt_camera* cam, const spt_render_params* params, uint32_t flags,
                  int32_t n_threads, float* rgb_mean_out, oracle_stats* stats) {
    if (!desc || !cam || !params || !rgb_mean_out) return 1;
    const spt_render_params p = *params;
    if (p.width == 0 || p.height == 0 || p.spp == 0) return 1;
    if (p.sampler == SPT_SAMPLER_JITTERED && (p.division_x == 0 || p.division_x * p.division_y != p.spp)) return 1;
    std::vector<uint32_t> rows;
    for (uint32_t j = 0; j < p.height; ++j)
        if (row_in_shard(p, j)) rows.push_back(j);
    if (n_threads <= 0) n_threads = (int32_t)std::thread::hardware_concurrency() * 2;  // pt.rs:243
    if (n_threads < 1) n_threads = 1;
    uint32_t nrows = (uint32_t)rows.size();
    if ((uint32_t)n_threads > nrows && nrows > 0) n_threads = (int32_t)nrows;
    Flags f{(flags & ORACLE_SLAB_RECIPROCAL) != 0, (flags & ORACLE_BRUTE_FORCE) != 0, (flags & ORACLE_LIBM) != 0, (flags & ORACLE_TIE_MIN_ID) != 0};
    std::vector<Counters> counters((size_t)n_threads);
    const float aspect = (float)p.width / (float)p.height;
    const float width_inv = 1.0f / (float)p.width, height_inv = 1.0f / (float)p.height;
    const float spp_sqrt_inv = 1.0f / spt_sqrt((float)p.spp);                               // pt.rs:253-254
    const float aux_dx = aspect * width_inv * spp_sqrt_inv, aux_dy = height_inv * spp_sqrt_inv;
    // one camera sample (pt.rs:266-278): its colour and the sampler offsets the film keeps with it
    auto sample = [&](Ctx& cx, uint32_t i, uint32_t j, uint32_t s, float* ox, float* oy) -> Color {
        uint32_t pixel = j * p.width + i;
        Rng rng{spt_rng_seed(p.seed, pixel, s)};
        pixel_offset(p, pixel, s, rng, ox, oy);
        float x = (((float)i + *ox) * width_inv - 0.5f) * aspect;             // pt.rs:269
        float y = ((float)(p.height - j - 1) + *oy) * height_inv - 0.5f;      // pt.rs:270-271
        Ray ray = camera_ray(*cam, x, y);
        {   // generate_ray_with_aux_ray (camera/mod.rs:15-21) with the offsets of pt.rs:272-275
            Ray rx = camera_ray(*cam, x + aux_dx, y), ry = camera_ray(*cam, x, y + aux_dy);
            ray.has_aux = true;
            ray.x_origin = rx.origin; ray.x_direction = rx.direction;
            ray.y_origin = ry.origin; ray.y_direction = ry.direction;
        }
        return trace_ray(cx, ray, rng, p.max_depth, (p.flags & SPT_RENDER_DEBUG_NORMAL) != 0);
    };
    auto run_threads = [&](const std::function<void(int)>& work) {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    };
    // BoxFilter (boxf.rs:5-34); radius 0.5 without the flag
    const float radius = (p.flags & SPT_RENDER_BOX_RADIUS) ? p.filter_radius : 0.5f;
    uint64_t traced_rows = nrows;
    if (radius == 0.5f) {
        // D3: radius_int = 0 and every weight is 1 (offsets lie in [0, 1)), so filter_pixel is the in-order sum of the
        // pixel's own samples over spp and the Vec of samples need not be kept
        run_threads([&](int t) {
            Ctx cx{desc, f, Math{f.libm}, &counters[(size_t)t]};
            uint32_t per = nrows / (uint32_t)n_threads;
            uint32_t from = (uint32_t)t * per, to = (t + 1 == n_threads) ? nrows : (uint32_t)(t + 1) * per;
            for (uint32_t rr = from; rr < to; ++rr) {
                uint32_t j = rows[rr];
                for (uint32_t i = 0; i < p.width; ++i) {
                    Color sum = gray(0.0f);
                    for (uint32_t s = 0; s < p.spp; ++s) {
                        float ox, oy;
                        Color c = sample(cx, i, j, s, &ox, &oy);
                        sum = sum + c;  // film.rs:87: color += sample.color
                    }
                    Color mean = sum / (float)p.spp;  // film.rs:91: color / weight_sum
                    float* o = rgb_mean_out + ((size_t)rr * p.width + i) * 3;
                    o[0] = mean.r; o[1] = mean.g; o[2] = mean.b;
                }
            }
        });
    } else {
        // The reference's Film as it is (film.rs:11-51): every sample of every pixel with (offset - 0.5) (pt.rs:278),
        // here for the rows this shard's filter footprint reaches, then filter_pixel (film.rs:71-92) per owned pixel.
        const int32_t R = (int32_t)std::ceil(radius - 0.5f);   // boxf.rs:12
        struct SampleData { float ox, oy; Color color; };
        std::vector<int64_t> row_slot(p.height, -1);
        std::vector<uint32_t> need;
        for (uint32_t j = 0; j < p.height; ++j) {
            bool wanted = false;
            for (int32_t d = -R; d <= R && !wanted; ++d) {
                int64_t jj = (int64_t)j + d;
                wanted = jj >= 0 && jj < (int64_t)p.height && row_in_shard(p, (uint32_t)jj);
            }
            if (wanted) { row_slot[j] = (int64_t)need.size(); need.push_back(j); }
        }
        traced_rows = need.size();
        std::vector<SampleData> data(need.size() * (size_t)p.width * p.spp);
        const uint32_t nneed = (uint32_t)need.size();
        run_threads([&](int t) {
            Ctx cx{desc, f, Math{f.libm}, &counters[(size_t)t]};
            for (uint32_t rr = (uint32_t)t; rr < nneed; rr += (uint32_t)n_threads) {
                uint32_t j = need[rr];
                for (uint32_t i = 0; i < p.width; ++i)
                    for (uint32_t s = 0; s < p.spp; ++s) {
                        float ox, oy;
                        Color c = sample(cx, i, j, s, &ox, &oy);
                        data[((size_t)rr * p.width + i) * p.spp + s] = SampleData{ox - 0.5f, oy - 0.5f, c};   // film.rs:47-51
                    }
            }
        });
        for (uint32_t rr = 0; rr < nrows; ++rr) {
            const int32_t y = (int32_t)rows[rr];
            for (uint32_t x = 0; x < p.width; ++x) {
                Color color = gray(0.0f);
                float weight_sum = 0.0f;
                for (int32_t dj = -R; dj <= R; ++dj) {
                    if (y + dj < 0 || y + dj >= (int32_t)p.height) continue;
                    for (int32_t di = -R; di <= R; ++di) {
                        if ((int32_t)x + di < 0 || (int32_t)x + di >= (int32_t)p.width) continue;
                        const SampleData* sd = &data[((size_t)row_slot[(size_t)(y + dj)] * p.width + (size_t)((int32_t)x + di)) * p.spp];
                        for (uint32_t s = 0; s < p.spp; ++s) {
                            float wx = (float)di + sd[s].ox, wy = (float)dj + sd[s].oy;
                            float weight = (spt_abs(wx) <= radius && spt_abs(wy) <= radius) ? 1.0f : 0.0f;   // boxf.rs:27-33
                            color = color + sd[s].color;   // film.rs:87: the colour is not weighted
                            weight_sum += weight;
                        }
                    }
                }
                Color out = color / weight_sum;   // film.rs:91
                float* o = rgb_mean_out + ((size_t)rr * p.width + x) * 3;
                o[0] = out.r; o[1] = out.g; o[2] = out.b;
            }
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->samples = traced_rows * p.width * p.spp;
        stats->threads = (uint32_t)n_threads;
        for (auto& c : counters) {
            stats->segments_closest += c.closest; stats->segments_shadow += c.shadow;
            stats->node_tests += c.nodes; stats->tri_tests += c.tris; stats->sphere_tests += c.spheres;
            stats->instance_visits += c.insts;
        }
    }
    return 0;
}

int oracle_trace_closest(const spt_scene_desc* desc, uint32_t flags, uint32_t n, const spt_ray* rays, spt_hit* hits) {
    if (!desc || (!rays && n) || (!hits && n)) return 1;
    Flags f{(flags & ORACLE_SLAB_RECIPROCAL) != 0, (flags & ORACLE_BRUTE_FORCE) != 0, (flags & ORACLE_LIBM) != 0, (flags & ORACLE_TIE_MIN_ID) != 0};
    Counters c;
    Ctx cx{desc, f, Math{f.libm}, &c};
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{v3(rays[i].o), v3(rays[i].d), rays[i].t_min};
        Inter inter;
        inter.t = rays[i].t_max;
        bool hit = aggregate_intersect(cx, r, inter);
        hits[i].t = hit ? inter.t : SPT_F32_MAX;
        hits[i].instance = hit ? inter.instance : -1;
        hits[i].prim = hit ? inter.prim : -1;
        hits[i].v = hit ? inter.bv : 0.0f;
        hits[i].w = hit ? inter.bw : 0.0f;
    }
    return 0;
}

int oracle_trace_any(const spt_scene_desc* desc, uint32_t flags, uint32_t n, const spt_ray* rays, uint8_t* occluded) {
    if (!desc || (!rays && n) || (!occluded && n)) return 1;
    Flags f{(flags & ORACLE_SLAB_RECIPROCAL) != 0, (flags & ORACLE_BRUTE_FORCE) != 0, (flags & ORACLE_LIBM) != 0, (flags & ORACLE_TIE_MIN_ID) != 0};
    Counters c;
    Ctx cx{desc, f, Math{f.libm}, &c};
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{v3(rays[i].o), v3(rays[i].d), rays[i].t_min};
        occluded[i] = aggregate_intersect_test(cx, r, rays[i].t_max) ? 1 : 0;
    }
    return 0;
}

// ---- unit seams for known-answer tests ------------------------------------------------
void oracle_bxdf_sample(const spt_material* mt, const float wo[3], uint64_t rng_state, uint32_t flags, float wi_out[3],
                        float bxdf_out[3], float* pdf_out, int32_t* dir_out) {
    Rng rng{{rng_state}};
    BxdfSample s = bxdf_sample(Math{(flags & ORACLE_LIBM) != 0}, *mt, v3(wo), rng);
    wi_out[0] = s.wi.x; wi_out[1] = s.wi.y; wi_out[2] = s.wi.z;
    bxdf_out[0] = s.bxdf.r; bxdf_out[1] = s.bxdf.g; bxdf_out[2] = s.bxdf.b;
    *pdf_out = s.pdf;
    *dir_out = s.dir;
}
void oracle_bxdf_eval(const spt_material* mt, const float wo[3], const float wi[3], float bxdf_out[3], float* pdf_out) {
    Color f = bxdf_eval(*mt, v3(wo), v3(wi));
    bxdf_out[0] = f.r; bxdf_out[1] = f.g; bxdf_out[2] = f.b;
    *pdf_out = bxdf_pdf(*mt, v3(wo), v3(wi));
}
// array forms of the two seams above, the counterpart of spt_debug_bxdf (include/spt_abi.h); `d` (may be NULL) holds the
// tables of a position-normal-distribution lobe
void oracle_bxdf_sample_n(const spt_scene_desc* d, const spt_material* mt, uint32_t flags, uint32_t n, const float* wo, const uint64_t* rng_state,
                          float* wi_out, float* bxdf_out, float* pdf_out, int32_t* dir_out) {
    const Math m{(flags & ORACLE_LIBM) != 0};
    for (uint32_t i = 0; i < n; ++i) {
        Rng rng{{rng_state[i]}};
        BxdfSample s = bxdf_sample(m, *mt, v3(wo + 3 * i), rng, nullptr, d);
        wi_out[3 * i] = s.wi.x; wi_out[3 * i + 1] = s.wi.y; wi_out[3 * i + 2] = s.wi.z;
        bxdf_out[3 * i] = s.bxdf.r; bxdf_out[3 * i + 1] = s.bxdf.g; bxdf_out[3 * i + 2] = s.bxdf.b;
        pdf_out[i] = s.pdf;
        dir_out[i] = s.dir;
    }
}
void oracle_bxdf_eval_n(const spt_scene_desc* d, const spt_material* mt, uint32_t n, const float* wo, const float* wi, float* bxdf_out, float* pdf_out) {
    for (uint32_t i = 0; i < n; ++i) {
        Color f = bxdf_eval(*mt, v3(wo + 3 * i), v3(wi + 3 * i), d);
        bxdf_out[3 * i] = f.r; bxdf_out[3 * i + 1] = f.g; bxdf_out[3 * i + 2] = f.b;
        pdf_out[i] = bxdf_pdf(*mt, v3(wo + 3 * i), v3(wi + 3 * i), d);
    }
}
void oracle_pndf_sum(const spt_scene_desc* d, uint32_t pndf, float sigma_p, uint32_t n, const float* u, float* sum_out) {
    spt_pndf_view v{d->pndfs + pndf, d->pndf_terms, d->pndf_nodes, d->pndf_refs, d->pndf_roots};
    for (uint32_t i = 0; i < n; ++i) sum_out[i] = spt_pndf_uv_walk(&v, u[2 * i], u[2 * i + 1], sigma_p, 0, 0.0f, 0.0f, nullptr);
}
void oracle_pndf_calc(const spt_scene_desc* d, uint32_t pndf, float sigma_p, uint32_t n, const float* u, const float* s, float* out) {
    spt_pndf_view v{d->pndfs + pndf, d->pndf_terms, d->pndf_nodes, d->pndf_refs, d->pndf_roots};
    for (uint32_t i = 0; i < n; ++i) {
        const float sum = spt_pndf_uv_walk(&v, u[2 * i], u[2 * i + 1], sigma_p, 0, 0.0f, 0.0f, nullptr);
        const float term_coe = (1.0f / sum) / (2.0f * SPT_PI * v.pd->sigma_r * v.pd->sigma_r);
        out[i] = spt_pndf_calc(&v, sigma_p, term_coe, u[2 * i], u[2 * i + 1], s[2 * i], s[2 * i + 1]);
    }
}
void oracle_pndf_sample_half(const spt_scene_desc* d, uint32_t pndf, float sigma_p, const float u[2], uint64_t seed, uint32_t n, float* half_pdf_out) {
    spt_pndf_view v{d->pndfs + pndf, d->pndf_terms, d->pndf_nodes, d->pndf_refs, d->pndf_roots};
    spt_material mt;
    std::memset(&mt, 0, sizeof mt);
    const float sum = spt_pndf_uv_walk(&v, u[0], u[1], sigma_p, 0, 0.0f, 0.0f, nullptr);
    mt.bxdf = SPT_BXDF_PNDF_CONDUCTOR;
    mt.ax = u[0]; mt.ay = u[1];
    mt.c1[0] = 1.0f / sum;
    mt.c1[1] = sigma_p;
    mt.c1[2] = spt_u2f(pndf);
    for (uint32_t i = 0; i < n; ++i) {
        Rng rng;
        rng.s = spt_rng_seed(seed, i, 0u);
        float pdf;
        const Vec3 h = pndf_sample_half(d, mt, rng, &pdf);
        half_pdf_out[4 * i] = h.x; half_pdf_out[4 * i + 1] = h.y; half_pdf_out[4 * i + 2] = h.z; half_pdf_out[4 * i + 3] = pdf;
    }
}
float oracle_fresnel_dielectric(float ior, const float i[3], const float n[3]) { return fresnel_n(ior, v3(i), v3(n)); }
float oracle_henyey_greenstein(float g, float c) { return henyey_greenstein(g, c); }
float oracle_hg_cdf_inverse(float g, float r) { return henyey_greenstein_cdf_inverse(g, r); }
uint32_t oracle_alias_sample(const spt_alias_table* a, float rand, float* prob) { return alias_sample(*a, rand, prob); }
void oracle_env_lookup(const spt_scene_desc* d, const float wi[3], float rgb[3], float* pdf) {
    Counters c;
    Ctx cx{d, Flags{false, false, false, false}, Math{false}, &c};
    Color col_;
    env_strength_pdf(cx, v3(wi), &col_, pdf);
    rgb[0] = col_.r; rgb[1] = col_.g; rgb[2] = col_.b;
}
void oracle_camera_ray(const spt_camera* cam, float x, float y, float o[3], float dir[3]) {
    Ray r = camera_ray(*cam, x, y);
    o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
    dir[0] = r.direction.x; dir[1] = r.direction.y; dir[2] = r.direction.z;
}

// shared-spec functions of include/spt_detmath.h, exported for tests/test_detmath.py
void oracle_detmath(uint32_t fn, uint32_t n, const float* a, const float* b, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = spt_sin(a[i]); break;
        case 1: out[i] = spt_cos(a[i]); break;
        case 2: out[i] = spt_log(a[i]); break;
        case 3: out[i] = spt_exp(a[i]); break;
        case 4: out[i] = spt_acos(a[i]); break;
        case 5: out[i] = spt_atan2(a[i], b[i]); break;
        case 6: out[i] = spt_asin(a[i]); break;
        case 7: out[i] = spt_round(a[i]); break;
        case 8: out[i] = spt_floor(a[i]); break;
        case 9: out[i] = spt_sqrt(a[i]); break;
        case 10: out[i] = a[i] / b[i]; break;
        case 11: out[i] = spt_max(a[i], b[i]); break;
        case 12: out[i] = spt_min(a[i], b[i]); break;
        case 13: out[i] = spt_pow(a[i], b[i]); break;
        case 14: out[i] = spt_log2(a[i]); break;
        case 15: out[i] = spt_trunc(a[i]); break;
        case 16: out[i] = spt_fract(a[i]); break;
        default: out[i] = 0.0f; break;
        }
    }
}
// texture graph at explicit inputs: `in` holds 18 floats per sample
// (position, normal, tangent, bitangent, texcoords, duvdx, duvdy), out 4 (r, g, b, a)
int oracle_tex_eval(const spt_scene_desc* desc, uint32_t flags, uint32_t node, uint32_t n, const float* in, float* rgba) {
    if (!desc || node >= desc->n_textures) return 1;
    Counters cn;
    Ctx cx{desc, Flags{(flags & ORACLE_SLAB_RECIPROCAL) != 0, (flags & ORACLE_BRUTE_FORCE) != 0, (flags & ORACLE_LIBM) != 0, (flags & ORACLE_TIE_MIN_ID) != 0},
           Math{(flags & ORACLE_LIBM) != 0}, &cn};
    for (uint32_t i = 0; i < n; ++i) {
        const float* f = in + 18 * (size_t)i;
        TexInput ti;
        ti.position = v3(f); ti.normal = v3(f + 3); ti.tangent = v3(f + 6); ti.bitangent = v3(f + 9);
        ti.texcoords[0] = f[12]; ti.texcoords[1] = f[13];
        ti.duvdx[0] = f[14]; ti.duvdx[1] = f[15]; ti.duvdy[0] = f[16]; ti.duvdy[1] = f[17];
        Vec4 c = tex_eval(cx, node, ti);
        rgba[4 * i] = c.x; rgba[4 * i + 1] = c.y; rgba[4 * i + 2] = c.z; rgba[4 * i + 3] = c.w;
    }
    return 0;
}
// Intersection::calc_differential: ray (o, d, aux x_o, x_d, y_o, y_d = 18 floats), hit (t, normal, tangent, bitangent = 10 floats)
void oracle_calc_differential(const float* ray18, const float* hit10, float duvdx[2], float duvdy[2]) {
    Ray r = make_ray(v3(ray18), v3(ray18 + 3));
    r.has_aux = true;
    r.x_origin = v3(ray18 + 6); r.x_direction = v3(ray18 + 9); r.y_origin = v3(ray18 + 12); r.y_direction = v3(ray18 + 15);
    Inter it;
    it.t = hit10[0];
    it.normal = v3(hit10 + 1); it.tangent = v3(hit10 + 4); it.bitangent = v3(hit10 + 7);
    calc_differential(it, r);
    duvdx[0] = it.duvdx[0]; duvdx[1] = it.duvdx[1]; duvdy[0] = it.duvdy[0]; duvdy[1] = it.duvdy[1];
}
// Subsurface substrate seams (substrate.rs:187-229): diffusion profile Sp(r) for d, radius sampling, table entry
void oracle_ss_sp(const float d[3], float r, float out[3]) {
    Color c = ss_sp(Math{false}, col(d), r);
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
float oracle_ss_sample_r(float rand) { return ss_sample_r(rand); }
void oracle_ss_cdf(uint32_t i, float xy[2]) { xy[0] = ss_cdf_table().x[i]; xy[1] = ss_cdf_table().y[i]; }
void oracle_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out) {
    spt_rng r = spt_rng_seed(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = spt_rng_f32(&r);
}
uint64_t oracle_rng_state(uint64_t seed, uint32_t pixel, uint32_t sample) { return spt_rng_seed(seed, pixel, sample).state; }
void oracle_r2_offsets(uint32_t pixel, uint32_t spp, uint32_t n, float* out) {
    for (uint32_t s = 0; s < n; ++s) spt_r2_offset(pixel, spp, s, &out[2 * s], &out[2 * s + 1]);
}

}  // extern "C"
