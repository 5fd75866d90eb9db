// Surface / BxDF / light / medium evaluation on the device.
//
// Replaces, per path vertex:
//   Surface::{coord, emissive, inside_medium}     reference src/core/surface.rs:49-111
//   Coordinate                                    src/core/coord.rs:10-59
//   BxdfT::{sample, pdf, bxdf, is_delta} of Lambert, MicrofacetConductor, SpecularConductor,
//     MicrofacetDielectric, SpecularDielectric, Pseudo    src/bxdf/*.rs (cited per function)
//   LightSamplerT::{sample_light, pdf_shape_light, pdf_env_light}   src/light_sampler/{uniform,power_is}.rs
//   LightT::sample of Dir/Point/Spot/Shape/Env + EnvLight lookup   src/light/*.rs
//   Homogeneous medium + Henyey-Greenstein        src/medium/{homogeneous,util}.rs
// The reference builds a Bxdf enum object per hit from texture lookups
// (MaterialT::bxdf_context); with scalar textures that object is a constant of the
// material, so the kernels switch on a POD material record instead.
#pragma once
#include "trace.h"
#include "../../../include/spt_pndf.h"

struct DRng {
    spt_rng s;
    SPT_DEV float next() { return spt_rng_f32(&s); }
    // Rng::gaussian_2d (src/core/rng.rs:28-42): Box-Muller, redrawing while the first number is <= 1e-6
    SPT_DEV void gaussian_2d(float mu, float sigma, float* x, float* y) {
        float rx, ry;
        do { rx = next(); ry = next(); } while (!(rx > 1e-6f));
        const float mag = sigma * spt_sqrt(-2.0f * spt_log(rx));
        const float temp = 2.0f * SPT_PI * ry;
        *x = mag * spt_cos(temp) + mu;
        *y = mag * spt_sin(temp) + mu;
    }
};

// ---- what Triangle::intersect / Sphere::intersect + Instance::intersect leave in `Intersection`
struct DInter {
    f3 position, normal, tangent, bitangent;  // world space
    uint32_t surface, prim_type, prim_id;
    int32_t light;
    float uv[2], duvdx[2], duvdy[2];          // texcoords + differentials (textured scenes only, else dead)
};

SPT_DEV void sphere_frame(f3 norm, f3* tangent, f3* bitangent) {  // sphere.rs:70-82
    float sin_theta = spt_sqrt(1.0f - norm.y * norm.y);
    if (sin_theta != 0.0f) {
        f3 bt = norm * (-norm.y / sin_theta);
        bt.y = sin_theta;
        *bitangent = bt;
        *tangent = cross(bt, norm);
    } else if (norm.y > 0.0f) {
        *bitangent = mk3(1, 0, 0);
        *tangent = mk3(0, 0, 1);
    } else {
        *bitangent = mk3(-1, 0, 0);
        *tangent = mk3(0, 0, -1);
    }
}

// Shading-table fetch.  kL (the fused shade kernel of an LDS-resident scene): instances, triangle attributes,
// surfaces, materials and lights were staged into LDS behind the traversal geometry, so the dependent chain
// hit -> instance -> attributes -> surface -> material -> light costs LDS instead of L2 latency.
template <bool kL>
SPT_DEV float4 tab_ld(const DScene& sc, const float4* global, uint32_t lds_off, uint32_t i) {
    if (kL) return geo_lds(sc)[lds_off + i];
    return global[i];
}
template <bool kL>
SPT_DEV spt_surface load_surface(const DScene& sc, uint32_t i) {
    const float4* g = reinterpret_cast<const float4*>(sc.surfaces);
    const float4 a = tab_ld<kL>(sc, g, sc.o_surf, 2u * i), b = tab_ld<kL>(sc, g, sc.o_surf, 2u * i + 1u);
    spt_surface s;
    s.material = __float_as_uint(a.x); s.flags = __float_as_uint(a.y); s.inside_medium = __float_as_int(a.z);
    s.emissive[0] = a.w; s.emissive[1] = b.x; s.emissive[2] = b.y;
    s.normal_map = __float_as_uint(b.z); s.emissive_map = __float_as_uint(b.w);
    return s;
}
template <bool kL>
SPT_DEV spt_light load_light(const DScene& sc, uint32_t i) {
    const float4* g = reinterpret_cast<const float4*>(sc.lights);
    const float4 a = tab_ld<kL>(sc, g, sc.o_light, 4u * i), b = tab_ld<kL>(sc, g, sc.o_light, 4u * i + 1u),
                 c = tab_ld<kL>(sc, g, sc.o_light, 4u * i + 2u), d = tab_ld<kL>(sc, g, sc.o_light, 4u * i + 3u);
    spt_light l;
    l.type = __float_as_uint(a.x);
    l.pos[0] = a.y; l.pos[1] = a.z; l.pos[2] = a.w;
    l.dir[0] = b.x; l.dir[1] = b.y; l.dir[2] = b.z;
    l.strength[0] = b.w; l.strength[1] = c.x; l.strength[2] = c.y;
    l.cos_inner = c.z; l.cos_outer = c.w;
    l.instance = __float_as_uint(d.x); l.power = d.y; l.pad[0] = 0.0f; l.pad[1] = 0.0f;
    return l;
}

struct DInstance {
    float inv[12], fwd[12], nrm[9];
    uint32_t prim_type, prim_id, surface;
    int32_t light;
};
template <bool kL = false>
SPT_DEV DInstance load_instance(const DScene& sc, uint32_t inst) {
    DInstance d;
    float4 q[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) q[k] = tab_ld<kL>(sc, sc.instances, sc.o_inst, 12u * inst + (uint32_t)k);
    const float* w = reinterpret_cast<const float*>(q);
#pragma unroll
    for (int k = 0; k < 12; ++k) d.inv[k] = w[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) d.fwd[k] = w[12 + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) d.nrm[k] = w[24 + k];
    d.prim_type = __float_as_uint(w[33]);
    d.prim_id = __float_as_uint(w[34]);
    d.surface = __float_as_uint(w[35]);
    d.light = (int32_t)__float_as_uint(w[36]);
    return d;
}

SPT_DEV void sphere_normal_to_texcoords(f3 p, float* uv) {  // sphere.rs:138-145
    float theta = spt_acos(p.y);
    float phi = spt_atan2(p.x, p.z) + SPT_PI;
    uv[0] = phi * 0.5f * SPT_FRAC_1_PI;
    uv[1] = theta * SPT_FRAC_1_PI;
}

// Rebuild the shading inputs of a recorded hit (t, instance, prim, v, w):
// triangle.rs:188-212 or sphere.rs:64-83, then instance.rs:97-104.  kTex: also the texcoords.
template <bool kTex = false, bool kL = false>
SPT_DEV DInter reconstruct_hit(const DScene& sc, const DRay& ray, const DHit& h) {
    const DInstance in = load_instance<kL>(sc, (uint32_t)h.inst);
    DInter it;
    f3 n, tg, bt;
    it.uv[0] = 0.0f; it.uv[1] = 0.0f;
    it.duvdx[0] = 0.0f; it.duvdx[1] = 0.0f; it.duvdy[0] = 0.0f; it.duvdy[1] = 0.0f;
    if (in.prim_type == SPT_PRIM_SPHERE) {
        DRay orr;
        orr.o = xf_point(in.inv, ray.o);
        orr.d = xf_vector(in.inv, ray.d);
        float4 s = tab_ld<kL>(sc, sc.spheres, sc.o_sph, in.prim_id);
        n = (point_at(orr, h.t) - mk3(s)) / s.w;
        sphere_frame(n, &tg, &bt);
        if (kTex) sphere_normal_to_texcoords(n, it.uv);
#if SPT_WITH_BEZIER
    } else if (in.prim_type == SPT_PRIM_BEZIER) {   // bezier.rs:163-168: (u, v) came with the hit record
        float4 cp[16];
        for (int k = 0; k < 16; ++k) cp[k] = sc.bez[16u * in.prim_id + (uint32_t)k];
        tg = bezier_tangent_at(cp, h.v, h.w);
        bt = bezier_bitangent_at(cp, h.v, h.w);
        n = normalize(cross(tg, bt));
        it.uv[0] = h.v; it.uv[1] = h.w;
#endif
    } else {
        float4 q[kTex ? 9 : 7];
#pragma unroll
        for (int k = 0; k < (kTex ? 9 : 7); ++k) q[k] = tab_ld<kL>(sc, sc.tri_attr, sc.o_attr, 9u * (uint32_t)h.prim + (uint32_t)k);
        const float* a = reinterpret_cast<const float*>(q);  // n[3][3] t[3][3] b[3][3] uv[3][2]
        float v = h.v, w = h.w;
        float u = 1.0f - v - w;
        n = normalize((mk3(a) * u + mk3(a + 3) * v) + mk3(a + 6) * w);
        tg = (mk3(a + 9) * u + mk3(a + 12) * v) + mk3(a + 15) * w;
        bt = (mk3(a + 18) * u + mk3(a + 21) * v) + mk3(a + 24) * w;
        if (kTex) {  // lerp_point2 (triangle.rs:291-302)
            it.uv[0] = (a[27] * u + a[29] * v) + a[31] * w;
            it.uv[1] = (a[28] * u + a[30] * v) + a[32] * w;
        }
    }
    it.position = point_at(ray, h.t);
    it.normal = normalize((mk3(in.nrm) * n.x + mk3(in.nrm + 3) * n.y) + mk3(in.nrm + 6) * n.z);
    it.tangent = xf_vector(in.fwd, tg);
    it.bitangent = xf_vector(in.fwd, bt);
    it.surface = in.surface;
    it.prim_type = in.prim_type;
    it.prim_id = in.prim_id;
    it.light = in.light;
    return it;
}

// ---- Intersection::calc_differential (src/core/intersection.rs:28-84, 104-118) for a camera ray
SPT_DEV bool solve_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float* x1, float* x2) {  // columns (a00,a01), (a10,a11)
    float det = a00 * a11 - a01 * a10;
    if (det != 0.0f) {
        float temp = b1 * a00 - b0 * a01;
        *x2 = temp / det;
        *x1 = (spt_abs(a00) > spt_abs(a01)) ? (b0 - a10 * *x2) / a00 : (b1 - a11 * *x2) / a01;
        return true;
    }
    return false;
}
SPT_DEV void calc_differential(DInter& it, const DRay& ray, float t, f3 xo, f3 xd, f3 yo, f3 yd) {
    f3 p = point_at(ray, t);
    float d = dot(p, it.normal);
    float tx = (d - dot(xo, it.normal)) / dot(xd, it.normal);
    f3 px = xo + xd * tx;
    float ty = (d - dot(yo, it.normal)) / dot(yd, it.normal);
    f3 py = yo + yd * ty;
    f3 dpdx = px - p, dpdy = py - p;
    float bx0, bx1, by0, by1, a00, a01, a10, a11;
    float ax = spt_abs(it.normal.x), ay = spt_abs(it.normal.y), az = spt_abs(it.normal.z);
    if (ax >= ay && ax >= az) {
        bx0 = dpdx.y; bx1 = dpdx.z; by0 = dpdy.y; by1 = dpdy.z;
        a00 = it.tangent.y; a01 = it.tangent.z; a10 = it.bitangent.y; a11 = it.bitangent.z;
    } else if (ay >= az) {
        bx0 = dpdx.z; bx1 = dpdx.x; by0 = dpdy.z; by1 = dpdy.x;
        a00 = it.tangent.z; a01 = it.tangent.x; a10 = it.bitangent.z; a11 = it.bitangent.x;
    } else {
        bx0 = dpdx.x; bx1 = dpdx.y; by0 = dpdy.x; by1 = dpdy.y;
        a00 = it.tangent.x; a01 = it.tangent.y; a10 = it.bitangent.x; a11 = it.bitangent.y;
    }
    float x1, x2;
    if (solve_2x2(a00, a01, a10, a11, bx0, bx1, &x1, &x2)) { it.duvdx[0] = x1; it.duvdx[1] = x2; }
    if (solve_2x2(a00, a01, a10, a11, by0, by1, &x1, &x2)) { it.duvdy[0] = x1; it.duvdy[1] = x2; }
}

// ---- textures (src/texture/*.rs) ---------------------------------------------------------------
// The closed Texture enum is a graph; spt_scene_create compiles every node into a postfix program so
// that the evaluation needs no recursion: scalar / image leaves push an RGBA value, binary ops and the
// sRGB decode act on the top of a 4-deep value stack kept in registers (the host refuses deeper
// programs).  An image leaf carries the chain of TexInputModifiers above it (outermost first), which it
// applies to the hit's TextureInput in the reference's order before sampling.
enum { TEXOP_SCALAR = 0, TEXOP_IMAGE = 1, TEXOP_ADD = 2, TEXOP_SUB = 3, TEXOP_MUL = 4, TEXOP_DIV = 5, TEXOP_SRGB = 6 };

struct DTexIn {  // TextureInput (mod.rs:50-62) as From<&Intersection> fills it (mod.rs:143-157)
    f3 position, normal, tangent, bitangent;
    float uv[2], duvdx[2], duvdy[2];
};
SPT_DEV DTexIn tex_input(const DInter& it) {
    DTexIn in;
    in.position = it.position; in.normal = it.normal; in.tangent = it.tangent; in.bitangent = it.bitangent;
    in.uv[0] = it.uv[0]; in.uv[1] = it.uv[1];
    in.duvdx[0] = it.duvdx[0]; in.duvdx[1] = it.duvdx[1];
    in.duvdy[0] = it.duvdy[0]; in.duvdy[1] = it.duvdy[1];
    return in;
}
SPT_DEV float4 rgba_to_vec4(uint32_t px) {  // image_tex.rs:153-160
    return make_float4((float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f, (float)(px >> 24) / 255.0f);
}
SPT_DEV float4 v4_lerp(float4 a, float4 b, float t) {  // a * (1 - t) + b * t
    float s = 1.0f - t;
    return make_float4(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t, a.w * s + b.w * t);
}
SPT_DEV float4 sample_bilinear(const DScene& sc, uint4 L, float u, float v) {  // image_tex.rs:102-125
    const uint32_t* tx = sc.texels + L.z;
    float x = u * (float)L.x;
    int32_t x1 = spt_f2i_sat(spt_round(x));
    int32_t x0 = (int32_t)((uint32_t)x1 - 1u);   // wrapping, as release-mode Rust
    float xt = x - (float)x0 - 0.5f;
    int32_t wmax = (int32_t)L.x - 1, hmax = (int32_t)L.y - 1;
    x0 = x0 < 0 ? 0 : (x0 > wmax ? wmax : x0);
    x1 = x1 < 0 ? 0 : (x1 > wmax ? wmax : x1);
    float y = v * (float)L.y;
    int32_t y1 = spt_f2i_sat(spt_round(y));
    int32_t y0 = (int32_t)((uint32_t)y1 - 1u);
    float yt = y - (float)y0 - 0.5f;
    y0 = y0 < 0 ? 0 : (y0 > hmax ? hmax : y0);
    y1 = y1 < 0 ? 0 : (y1 > hmax ? hmax : y1);
    float4 c00 = rgba_to_vec4(tx[(uint32_t)y0 * L.x + (uint32_t)x0]), c01 = rgba_to_vec4(tx[(uint32_t)y1 * L.x + (uint32_t)x0]);
    float4 c10 = rgba_to_vec4(tx[(uint32_t)y0 * L.x + (uint32_t)x1]), c11 = rgba_to_vec4(tx[(uint32_t)y1 * L.x + (uint32_t)x1]);
    float4 c0 = v4_lerp(c00, c01, yt);
    float4 c1 = v4_lerp(c10, c11, yt);
    return v4_lerp(c0, c1, xt);
}
SPT_DEV float4 sample_trilinear(const DScene& sc, uint32_t image, float u, float v, const float* duvdx, const float* duvdy) {  // image_tex.rs:127-151
    const uint2 im = sc.images[image];
    if (im.y == 0u) return make_float4(0, 0, 0, 0);
    const uint4* lv = sc.image_levels + im.x;
    const uint4 l_base = lv[0];
    float sx = (float)l_base.x, sy = (float)l_base.y;
    float dxx = duvdx[0] * sx, dxy = duvdx[1] * sy, dyx = duvdy[0] * sx, dyy = duvdy[1] * sy;
    float lx = spt_sqrt(dxx * dxx + dxy * dxy), ly = spt_sqrt(dyx * dyx + dyy * dyy);
    float level = spt_clamp(spt_log2(spt_max(lx, ly) + 0.001f), 0.0f, (float)(im.y - 1u));
    uint32_t l0 = spt_f2u_sat(spt_floor(level));
    if (l0 + 1u == im.y) return sample_bilinear(sc, lv[l0], u, v);
    float lt = level - (float)l0;
    float4 c0 = sample_bilinear(sc, lv[l0], u, v), c1 = sample_bilinear(sc, lv[l0 + 1u], u, v);
    return v4_lerp(c0, c1, lt);
}
SPT_DEV float srgb_to_linear(float sv) {  // srgb_tex.rs:53-59
    return (sv <= 0.04045f) ? sv / 12.92f : spt_pow((sv + 0.055f) / 1.055f, 2.4f);
}
// TextureInput::value_vec2_wrapped (mod.rs:73-141) of the input after the modifier chain
SPT_DEV float tex_wrap(float x, int32_t wrap) {
    switch (wrap) {
    case SPT_TEXWRAP_REPEAT: {
        float fr = spt_fract(x);
        return (x >= 0.0f) ? fr : 1.0f + fr;
    }
    case SPT_TEXWRAP_MIRROR_REPEAT: {
        float fr = spt_fract(x);
        float xn = (x >= 0.0f) ? fr : 1.0f + fr;
        return (spt_f2i_sat(x) % 2 == 0) ? xn : 1.0f - xn;
    }
    case SPT_TEXWRAP_CLAMP: return spt_clamp(x, 0.0f, 1.0f);
    default: return spt_abs(spt_clamp(x, 0.0f, 1.0f));
    }
}
SPT_DEV float4 tex_image_leaf(const DScene& sc, const DTexIn& base, uint32_t image, uint32_t chain_first, uint32_t chain_len) {
    DTexIn in = base;
    int32_t mode = SPT_TEXMODE_TEXCOORDS, wrap = SPT_TEXWRAP_REPEAT;
    for (uint32_t c = 0; c < chain_len; ++c) {  // TexInputModifier::apply_modifier (input_modifier.rs:35-50)
        const float4* T = sc.textures + 4u * sc.tex_chain[chain_first + c];
        const float4 t1 = T[1], t2 = T[2], t3 = T[3];   // (value.xyz, mode) (wrap, tiling.xyz) (offset.xyz, -)
        const f3 tl = mk3(t2.y, t2.z, t2.w), of = mk3(t3.x, t3.y, t3.z);
        in.position = mk3(in.position.x * tl.x + of.x, in.position.y * tl.y + of.y, in.position.z * tl.z + of.z);
        in.normal = mk3(in.normal.x * tl.x + of.x, in.normal.y * tl.y + of.y, in.normal.z * tl.z + of.z);
        in.tangent = mk3(in.tangent.x * tl.x + of.x, in.tangent.y * tl.y + of.y, in.tangent.z * tl.z + of.z);
        in.bitangent = mk3(in.bitangent.x * tl.x + of.x, in.bitangent.y * tl.y + of.y, in.bitangent.z * tl.z + of.z);
        in.uv[0] = in.uv[0] * tl.x + of.x; in.uv[1] = in.uv[1] * tl.y + of.y;
        in.duvdx[0] = in.duvdx[0] * tl.x; in.duvdx[1] = in.duvdx[1] * tl.y;
        in.duvdy[0] = in.duvdy[0] * tl.x; in.duvdy[1] = in.duvdy[1] * tl.y;
        const int32_t m = __float_as_int(t1.w), w = __float_as_int(t2.x);
        if (m >= 0) mode = m;
        if (w >= 0) wrap = w;
    }
    float vx, vy;
    switch (mode) {
    case SPT_TEXMODE_TEXCOORDS: vx = in.uv[0]; vy = in.uv[1]; break;
    case SPT_TEXMODE_POSITION: vx = in.position.x; vy = in.position.y; break;
    case SPT_TEXMODE_NORMAL: vx = in.normal.x; vy = in.normal.y; break;
    case SPT_TEXMODE_TANGENT: vx = in.tangent.x; vy = in.tangent.y; break;
    case SPT_TEXMODE_BITANGENT: vx = in.bitangent.x; vy = in.bitangent.y; break;
    default: vx = 0.0f; vy = 0.0f; break;
    }
    return sample_trilinear(sc, image, tex_wrap(vx, wrap), tex_wrap(vy, wrap), in.duvdx, in.duvdy);
}
// color_at / float_at of texture `node` as one RGBA value (ScalarTex alpha = 1, SrgbTex keeps alpha)
SPT_DEV float4 tex_eval(const DScene& sc, uint32_t node, const DTexIn& in) {
    const uint2 root = sc.tex_root[node];
    float4 v0 = make_float4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;   // value stack, v0 = top
    for (uint32_t pc = root.x; pc < root.x + root.y; ++pc) {
        const uint4 ins = sc.tex_prog[pc];
        if (ins.x <= TEXOP_IMAGE) {
            float4 nv;
            if (ins.x == TEXOP_SCALAR) nv = make_float4(__uint_as_float(ins.y), __uint_as_float(ins.z), __uint_as_float(ins.w), 1.0f);
            else nv = tex_image_leaf(sc, in, ins.y, ins.z, ins.w);
            v3 = v2; v2 = v1; v1 = v0; v0 = nv;
        } else if (ins.x == TEXOP_SRGB) {
            v0 = make_float4(srgb_to_linear(v0.x), srgb_to_linear(v0.y), srgb_to_linear(v0.z), v0.w);
        } else {
            float4 a = v1, b = v0, r;   // a (op) b: t1 was pushed first
            if (ins.x == TEXOP_ADD) r = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            else if (ins.x == TEXOP_SUB) r = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
            else if (ins.x == TEXOP_MUL) r = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
            else r = make_float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w);
            v0 = r; v1 = v2; v2 = v3;
        }
    }
    return v0;
}
SPT_DEV f3 tex_color(const DScene& sc, uint32_t node, const DTexIn& in) { return mk3(tex_eval(sc, node, in)); }
SPT_DEV float tex_float(const DScene& sc, uint32_t node, const DTexIn& in, uint32_t chan) {
    float4 c = tex_eval(sc, node, in);
    return chan == SPT_CHAN_R ? c.x : (chan == SPT_CHAN_G ? c.y : (chan == SPT_CHAN_B ? c.z : c.w));
}
// Surface::emissive (surface.rs:49-55)
template <bool kTex>
SPT_DEV f3 surface_emissive(const DScene& sc, const spt_surface& sf, const DInter& it) {
    f3 e = mk3(sf.emissive);
    if (kTex && sf.emissive_map) e = e * tex_color(sc, sf.emissive_map - 1u, tex_input(it));
    return e;
}

// ---- Coordinate (coord.rs:10-59)
struct DCoord {
    f3 xw, yw, zw, hemi;
    SPT_DEV f3 to_local(f3 w) const { return mk3(dot(xw, w), dot(yw, w), dot(zw, w)); }
    SPT_DEV f3 to_world(f3 l) const { return (xw * l.x + yw * l.y) + zw * l.z; }
};
// Surface::coord (surface.rs:65-95) + Coordinate::from_tangent_normal
template <bool kTex = false>
SPT_DEV DCoord surface_coord(const DScene& sc, const spt_surface& sf, const DRay& ray, const DInter& it) {
    f3 shade_normal = it.normal;
    if (kTex && sf.normal_map) {
        f3 value = tex_color(sc, sf.normal_map - 1u, tex_input(it));
        f3 local = normalize(value * 2.0f - gray(1.0f));
        shade_normal = normalize((local.x * normalize(it.tangent) + local.y * normalize(it.bitangent)) + local.z * it.normal);
    }
    bool hit_back = dot(ray.d, it.normal) > 0.0f;
    bool ds = (sf.flags & SPT_SURF_DOUBLE_SIDED) != 0;
    DCoord c;
    c.zw = (ds && hit_back) ? -shade_normal : shade_normal;
    c.yw = normalize(cross(c.zw, it.tangent));
    c.xw = cross(c.yw, c.zw);
    c.hemi = hit_back ? -it.normal : it.normal;
    return c;
}

// ---- src/bxdf/util.rs
SPT_DEV f3 reflect_z(f3 i) { return mk3(-i.x, -i.y, i.z); }
SPT_DEV f3 reflect_n(f3 i, f3 n) { return (2.0f * dot(i, n)) * n - i; }
SPT_DEV bool refract_z(f3 i, float ior, f3* out) {  // util.rs:11-24
    float ior_ratio = (i.z >= 0.0f) ? 1.0f / ior : ior;
    float o_z_sqr = 1.0f - (1.0f - i.z * i.z) * ior_ratio * ior_ratio;
    if (o_z_sqr >= 0.0f) {
        float o_z = (i.z >= 0.0f) ? -spt_sqrt(o_z_sqr) : spt_sqrt(o_z_sqr);
        *out = mk3(-i.x * ior_ratio, -i.y * ior_ratio, o_z);
        return true;
    }
    return false;
}
SPT_DEV bool refract_n(f3 i, f3 n, float ior, f3* out) {  // util.rs:26-46
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
SPT_DEV float fresnel_n(float ior, f3 i, f3 n) {  // util.rs:56-81
    float i_ior, o_ior;
    if (dot(i, n) >= 0.0f) { i_ior = 1.0f; o_ior = ior; } else { i_ior = ior; o_ior = 1.0f; }
    f3 rf;
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
SPT_DEV f3 fresnel_conductor_n(f3 ior, f3 ior_k, f3 i, f3 n) {  // util.rs:87-112
    float cosv = dot(i, n);
    f3 ior_ratio, k_ratio;
    if (cosv >= 0.0f) { ior_ratio = ior; k_ratio = ior_k; } else { ior_ratio = cdiv(gray(1.0f), ior); k_ratio = cdiv(gray(1.0f), ior_k); }
    float cos2 = cosv * cosv;
    float sin2 = 1.0f - cos2;
    f3 ior_ratio2 = ior_ratio * ior_ratio;
    f3 k_ratio2 = k_ratio * k_ratio;
    f3 t0 = ior_ratio2 - k_ratio2 - gray(sin2);
    f3 a2_b2 = csqrt(t0 * t0 + (ior_ratio2 * 4.0f) * k_ratio2);
    f3 t1 = a2_b2 + gray(cos2);
    f3 a = csqrt((a2_b2 + t0) * 0.5f);
    f3 t2 = a * (2.0f * cosv);
    f3 rs = cdiv(t1 - t2, t1 + t2);
    f3 t3 = a2_b2 * cos2 + gray(sin2 * sin2);
    f3 t4 = t2 * sin2;
    f3 rp = cdiv(rs * (t3 - t4), t3 + t4);
    return (rs + rp) * 0.5f;
}
SPT_DEV f3 half_from_reflect(f3 i, f3 o) { return (i.z >= 0.0f) ? normalize(i + o) : -normalize(i + o); }  // util.rs:136-142
SPT_DEV f3 half_from_refract(f3 i, f3 o, float ior) {  // util.rs:144-155
    f3 h = (i.z >= 0.0f) ? normalize(i + ior * o) : normalize(ior * i + o);
    if (h.z < 0.0f) h = -h;
    return h;
}
SPT_DEV float ggx_ndf_aniso(f3 h, float ax, float ay) {  // util.rs:162-165
    return SPT_FRAC_1_PI / spt_max(ax * ay * pow2(pow2(h.x / ax) + pow2(h.y / ay) + pow2(h.z)), 0.0001f);
}
SPT_DEV float smith_g1_aniso(f3 v, float ax, float ay) {  // util.rs:172-174
    return 2.0f / (1.0f + spt_sqrt(1.0f + (pow2(ax * v.x) + pow2(ay * v.y)) / spt_max(pow2(v.z), 0.0001f)));
}
SPT_DEV float smith_visible_aniso(f3 v, f3 l, float ax, float ay) {  // util.rs:176-180
    float vv = spt_abs(v.z) + spt_sqrt(pow2(ax * v.x) + pow2(ay * v.y) + pow2(v.z));
    float ll = spt_abs(l.z) + spt_sqrt(pow2(ax * l.x) + pow2(ay * l.y) + pow2(l.z));
    return 1.0f / (vv * ll);
}
SPT_DEV float ggx_vndf_pdf(f3 h, f3 v, float ax, float ay) {  // util.rs:189-194
    if (!(v.z >= 0.0f)) v = -v;
    return smith_g1_aniso(v, ax, ay) * ggx_ndf_aniso(h, ax, ay) * spt_max(dot(v, h), 0.0f) / spt_max(v.z, 0.0001f);
}
SPT_DEV f3 ggx_vndf_sample(f3 ve, float ax, float ay, float r0, float r1, float* pdf) {  // util.rs:196-224
    if (!(ve.z >= 0.0f)) ve = -ve;
    f3 vh = normalize(mk3(ax * ve.x, ay * ve.y, ve.z));
    float len_sqr = vh.x * vh.x + vh.y * vh.y;
    f3 t_vec1 = (len_sqr > 0.0f) ? mk3(-vh.y, vh.x, 0.0f) / spt_sqrt(len_sqr) : mk3(1, 0, 0);
    f3 t_vec2 = cross(vh, t_vec1);
    float r = spt_sqrt(r0);
    float phi = 2.0f * SPT_PI * r1;
    float sp, cp;
    spt_sincos(phi, &sp, &cp);
    float t1 = r * cp;
    float t2 = r * sp;
    float s = 0.5f * (1.0f + vh.z);
    t2 = (1.0f - s) * spt_sqrt(1.0f - t1 * t1) + s * t2;
    f3 nh = (t1 * t_vec1 + t2 * t_vec2) + spt_sqrt(spt_max(1.0f - t1 * t1 - t2 * t2, 0.0f)) * vh;
    f3 ne = normalize(mk3(ax * nh.x, ay * nh.y, spt_max(nh.z, 0.0f)));
    *pdf = ggx_vndf_pdf(ne, ve, ax, ay);
    return ne;
}

// ---- Bxdf over a POD material
struct DMat {
    uint32_t bxdf;
    f3 c0, c1, c2;
    float ax, ay, ior;
    uint32_t fresnel, substrate;
};
template <bool kL = false>
SPT_DEV DMat load_material(const DScene& sc, uint32_t m) {
    const float4* p = reinterpret_cast<const float4*>(sc.materials);   // 64-byte records
    float4 a = tab_ld<kL>(sc, p, sc.o_mat, 4u * m), b = tab_ld<kL>(sc, p, sc.o_mat, 4u * m + 1u), c = tab_ld<kL>(sc, p, sc.o_mat, 4u * m + 2u),
           e = tab_ld<kL>(sc, p, sc.o_mat, 4u * m + 3u);
    DMat d;
    d.bxdf = __float_as_uint(a.x);
    d.c0 = mk3(a.y, a.z, a.w);
    d.c1 = mk3(b.x, b.y, b.z);
    d.ax = b.w;
    d.ay = c.x;
    d.ior = c.y;
    d.c2 = mk3(c.z, c.w, e.x);
    d.fresnel = __float_as_uint(e.y);
    d.substrate = __float_as_uint(e.z);
    return d;
}
SPT_DEV float fresnel_moment1(float eta) {  // src/bxdf/util.rs:123-134
    float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1.0f) return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945f * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
// MaterialT::bxdf_context at a hit (src/material/{lambert,conductor,dielectric,plastic,pbr_metallic,pbr_specular}.rs):
// constant materials were folded by the loader; a material with an image-backed parameter carries a recipe.
// kPndf: position-normal-distribution recipes can occur (k_shade<4 | 5>): only then is their footprint walk compiled in
template <bool kTex, bool kL = false, bool kPndf = false>
SPT_DEV DMat material_at(const DScene& sc, uint32_t m, const DInter& it) {
    DMat d = load_material<kL>(sc, m);
    if (!kTex) return d;
    const uint32_t recipe = __float_as_uint(reinterpret_cast<const float4*>(sc.materials + m)[3].w);
    if (recipe == 0u) return d;
    const uint4 r0 = sc.recipes[2u * (recipe - 1u)], r1 = sc.recipes[2u * (recipe - 1u) + 1u];   // (type, tex0..2) (tex3, rough_chan, metal_chan, ior)
    const float ior = __uint_as_float(r1.w);
    const DTexIn in = tex_input(it);
    d.c0 = mk3(0, 0, 0); d.c1 = mk3(0, 0, 0); d.c2 = mk3(0, 0, 0);
    d.ax = 0.0f; d.ay = 0.0f; d.ior = 0.0f; d.fresnel = 0u; d.substrate = 0u;
    bool specular = false;
    if (r0.x != SPT_MAT_LAMBERT) {
        float rx = tex_float(sc, r0.w, in, r1.y), ry = tex_float(sc, r1.x, in, r1.y);
        const bool squared = r0.x != SPT_MAT_PLASTIC;   // plastic.rs:64-65 hands the texture value over as is
        d.ax = squared ? rx * rx : rx;
        d.ay = squared ? ry * ry : ry;
        specular = d.ax < 0.0001f || d.ay < 0.0001f;
    }
    switch (r0.x) {
    case SPT_MAT_LAMBERT:
        d.bxdf = SPT_BXDF_LAMBERT;
        d.c0 = tex_color(sc, r0.y, in);
        break;
    case SPT_MAT_CONDUCTOR:
        d.c0 = tex_color(sc, r0.y, in);
        d.c1 = tex_color(sc, r0.z, in);
        d.bxdf = specular ? SPT_BXDF_SPECULAR_CONDUCTOR : SPT_BXDF_MICROFACET_CONDUCTOR;
        break;
    case SPT_MAT_DIELECTRIC:
        d.ior = ior;
        d.bxdf = specular ? SPT_BXDF_SPECULAR_DIELECTRIC : SPT_BXDF_MICROFACET_DIELECTRIC;
        break;
    case SPT_MAT_PLASTIC: {  // Diffuse::new (substrate.rs:127-137)
        f3 albedo = tex_color(sc, r0.y, in);
        d.ior = ior;
        d.bxdf = specular ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        d.fresnel = SPT_FRESNEL_DIELECTRIC;
        d.substrate = SPT_SUBSTRATE_DIFFUSE;
        d.c0 = albedo;
        float fdr = 2.0f * fresnel_moment1(1.0f / ior);
        d.c2 = cdiv(albedo * SPT_FRAC_1_PI, ((gray(1.0f) - albedo * fdr) * ior) * ior);
        break;
    }
    case SPT_MAT_PBR_METALLIC: {  // pbr_metallic.rs:75-104
        f3 base = tex_color(sc, r0.y, in);
        float metallic = tex_float(sc, r0.z, in, r1.z);
        d.c1 = metallic * base + (1.0f - metallic) * gray(0.04f);
        d.c0 = base * (1.0f - metallic);
        d.bxdf = specular ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        d.fresnel = SPT_FRESNEL_SCHLICK;
        d.substrate = SPT_SUBSTRATE_LAMBERT;
        break;
    }
    case SPT_MAT_SUBSURFACE: {  // material/subsurface.rs:66-93, bxdf::Subsurface::new (substrate.rs:199-211)
        f3 albedo = tex_color(sc, r0.y, in);
        float ld = tex_float(sc, r0.z, in, SPT_CHAN_R);
        d.ior = ior;
        d.bxdf = specular ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        d.fresnel = SPT_FRESNEL_DIELECTRIC;
        d.substrate = SPT_SUBSTRATE_SUBSURFACE;
        d.c0 = albedo;
        float fdr = 2.0f * fresnel_moment1(1.0f / ior);
        d.c2 = cdiv(albedo * SPT_FRAC_1_PI, ((gray(1.0f) - albedo * fdr) * ior) * ior);
        f3 q = albedo - gray(0.33f);
        f3 q2 = q * q;
        d.c1 = mk3(ld / (3.5f + 100.0f * (q2.x * q2.x)), ld / (3.5f + 100.0f * (q2.y * q2.y)), ld / (3.5f + 100.0f * (q2.z * q2.z)));
        break;
    }
    case SPT_MAT_PNDF_CONDUCTOR:    // pndf_conductor.rs:156-196; PndfMicrofacet::new (microfacet.rs:67-95)
    case SPT_MAT_PNDF_PLASTIC: {    // pndf_plastic.rs:163-211
        const bool plastic = r0.x == SPT_MAT_PNDF_PLASTIC;
        const f3 albedo = tex_color(sc, r0.y, in);
        d.c0 = albedo;
        if (plastic) {   // DielectricFresnel::new(ior), Diffuse::new(albedo, ior)
            d.ior = ior;
            d.fresnel = SPT_FRESNEL_DIELECTRIC;
            d.substrate = SPT_SUBSTRATE_DIFFUSE;
            float fdr = 2.0f * fresnel_moment1(1.0f / ior);
            d.c2 = cdiv(albedo * SPT_FRAC_1_PI, ((gray(1.0f) - albedo * fdr) * ior) * ior);
            d.bxdf = specular ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;      // the fallback without a footprint
        } else {
            d.fresnel = SPT_FRESNEL_SCHLICK;
            d.bxdf = specular ? SPT_BXDF_SPECULAR_CONDUCTOR : SPT_BXDF_MICROFACET_CONDUCTOR;
        }
        if (kPndf && sc.pndfs != nullptr) {
            const spt_pndf& pd = sc.pndfs[r0.z];
            const float ux = spt_pndf_wrap(it.uv[0] * pd.tiling[0] + pd.offset[0]), uy = spt_pndf_wrap(it.uv[1] * pd.tiling[1] + pd.offset[1]);
            const float dxx = it.duvdx[0] * pd.tiling[0], dxy = it.duvdx[1] * pd.tiling[1];
            const float dyx = it.duvdy[0] * pd.tiling[0], dyy = it.duvdy[1] * pd.tiling[1];
            const float sigma_p = spt_max(spt_sqrt(dxx * dxx + dxy * dxy), spt_sqrt(dyx * dyx + dyy * dyy)) / 3.0f;
            if (sigma_p > 0.0f) {
                const spt_pndf_view v{&pd, sc.pndf_terms, sc.pndf_nodes, sc.pndf_refs, sc.pndf_roots};
                const float sum = spt_pndf_uv_walk(&v, ux, uy, sigma_p, 0, 0.0f, 0.0f, nullptr);
                d.bxdf = plastic ? SPT_BXDF_PNDF_PLASTIC : SPT_BXDF_PNDF_CONDUCTOR;
                d.ax = ux; d.ay = uy;
                d.c1 = mk3(1.0f / sum, sigma_p, __uint_as_float(r0.z));   // per-hit record of the two P-NDF lobes: (1 / sum, sigma_p, table)
            }
        }
        break;
    }
    default:  // pbr_specular.rs:60-92
        d.c0 = tex_color(sc, r0.y, in);
        d.c1 = tex_color(sc, r0.z, in);
        d.bxdf = specular ? SPT_BXDF_SPECULAR_PLASTIC : SPT_BXDF_MICROFACET_PLASTIC;
        d.fresnel = SPT_FRESNEL_SCHLICK;
        d.substrate = SPT_SUBSTRATE_LAMBERT;
        break;
    }
    return d;
}
SPT_DEV bool mat_is_delta(const DMat& m) {
    return m.bxdf == SPT_BXDF_SPECULAR_CONDUCTOR || m.bxdf == SPT_BXDF_SPECULAR_DIELECTRIC || m.bxdf == SPT_BXDF_PSEUDO;
}
SPT_DEV float pow5(float x) { return x * x * x * x * x; }
SPT_DEV f3 mat_fresnel(const DMat& m, f3 i, f3 n) {  // fresnel.rs:29-59
    if (m.bxdf == SPT_BXDF_MICROFACET_CONDUCTOR || m.bxdf == SPT_BXDF_SPECULAR_CONDUCTOR || m.bxdf == SPT_BXDF_PNDF_CONDUCTOR) {
        // the conductors of pndf_conductor.rs carry SchlickFresnel::new(albedo) (fresnel.rs:49-52)
        if (m.fresnel == SPT_FRESNEL_SCHLICK) return m.c0 + (gray(1.0f) - m.c0) * pow5(1.0f - dot(i, n));
        return fresnel_conductor_n(m.c0, m.c1, i, n);
    }
    return gray(fresnel_n(m.ior, i, n));
}
// plastic lobes: SchlickFresnel / DielectricFresnel (fresnel.rs:19-59), Lambert / Diffuse substrate
// (substrate.rs:29-45, 139-180)
SPT_DEV f3 plastic_fresnel(const DMat& m, f3 i, f3 n) {
    if (m.fresnel == SPT_FRESNEL_SCHLICK) return m.c1 + (gray(1.0f) - m.c1) * pow5(1.0f - dot(i, n));
    return gray(fresnel_n(m.ior, i, n));
}
SPT_DEV float substrate_pdf(f3 wo, f3 wi) { return (wo.z * wi.z >= 0.0f) ? spt_abs(wi.z) * SPT_FRAC_1_PI : 1.0f; }
SPT_DEV f3 substrate_eval(const DMat& m, f3 wo, f3 wi) {
    if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
    if (m.substrate != SPT_SUBSTRATE_LAMBERT) return m.c2 * (1.0f - fresnel_n(m.ior, wi, mk3(0, 0, 1)));   // Diffuse, Subsurface::diffuse
    return m.c0 * SPT_FRAC_1_PI;
}
SPT_DEV float ndf_visible(const DMat& m, f3 wo, f3 wi, f3 h) {  // microfacet.rs:47-53
    return ggx_ndf_aniso(h, m.ax, m.ay) * smith_visible_aniso(wo, wi, m.ax, m.ay);
}

// PndfMicrofacet (src/bxdf/microfacet.rs:56-170) over the per-hit constants material_at left in the material record
SPT_DEV spt_pndf_view pndf_view(const DScene& sc, const DMat& m) {
    return spt_pndf_view{sc.pndfs + __float_as_uint(m.c1.z), sc.pndf_terms, sc.pndf_nodes, sc.pndf_refs, sc.pndf_roots};
}
SPT_DEV float pndf_half_pdf(const DScene& sc, const DMat& m, f3 half) {   // microfacet.rs:142-154
    const spt_pndf_view v = pndf_view(sc, m);
    return spt_pndf_calc(&v, m.c1.y, spt_pndf_term_coe(v.pd, m.c1.x), m.ax, m.ay, half.x, half.y);
}
// The light sample of a path vertex asks bxdf(wo, wi) and pdf(wo, wi) of the same lobe for the same pair: both need the
// density of the same half vector, i.e. the same walk of the block's 4-D tree with the same arguments (the reference walks
// twice, microfacet.rs:142-169).  The first of the two leaves the value here, the second takes it: same bits, one walk.
struct DPndfMemo {
    float d;
    bool have;
};
SPT_DEV float pndf_half_pdf(const DScene& sc, const DMat& m, f3 half, DPndfMemo* memo) {
    if (memo != nullptr && memo->have) return memo->d;
    const float d = pndf_half_pdf(sc, m, half);
    if (memo != nullptr) { memo->d = d; memo->have = true; }
    return d;
}
SPT_DEV float pndf_ndf_visible(const DScene& sc, const DMat& m, f3 wo, f3 wi, f3 half, DPndfMemo* memo = nullptr) {   // microfacet.rs:156-169
    const float pndf = pndf_half_pdf(sc, m, half, memo);
    const float visible = 0.25f / spt_max(wi.z * wo.z, 0.0001f);
    return pndf / spt_max(half.z, 0.0001f) * visible;
}
SPT_DEV f3 pndf_sample_half(const DScene& sc, const DMat& m, DRng& rng, float* pdf) {   // microfacet.rs:98-140
    const spt_pndf_view v = pndf_view(sc, m);
    const spt_pndf& pd = *v.pd;
    const float sigma_p = m.c1.y;
    const float sigma_p_sqr = sigma_p * sigma_p, sigma_p_sqr_inv = 1.0f / sigma_p_sqr;
    const float sigma_h_sqr = pd.sigma_hx * pd.sigma_hy, sigma_h_sqr_inv = 1.0f / sigma_h_sqr;
    const float sigma_sqr_sum_inv = 1.0f / (sigma_p_sqr + sigma_h_sqr);
    const float rand = rng.next();
    uint32_t ti = 0xffffffffu;
    spt_pndf_uv_walk(&v, m.ax, m.ay, sigma_p, 1, m.c1.x, rand, &ti);
    if (ti == 0xffffffffu) ti = pd.first_term;   // no term within reach (the reference indexes an empty list there)
    const spt_pndf_term& g = v.terms[ti];
    const float mux = sigma_sqr_sum_inv * (sigma_h_sqr * m.ax + sigma_p_sqr * g.u[0]), muy = sigma_sqr_sum_inv * (sigma_h_sqr * m.ay + sigma_p_sqr * g.u[1]);
    const float sigma = 1.0f / spt_sqrt(sigma_p_sqr_inv + sigma_h_sqr_inv);
    float gx, gy;
    rng.gaussian_2d(0.0f, sigma, &gx, &gy);
    const float ux = mux + gx, uy = muy + gy;
    float jx, jy;
    spt_m2_mul(g.jacobian, ux - g.u[0], uy - g.u[1], &jx, &jy);
    const float smx = g.s[0] + jx, smy = g.s[1] + jy;
    rng.gaussian_2d(0.0f, pd.sigma_r, &gx, &gy);
    const float sx = smx + gx, sy = smy + gy;
    const f3 half = normalize(mk3(sx, sy, spt_sqrt(spt_clamp(1.0f - (sx * sx + sy * sy), 0.0f, 1.0f))));
    *pdf = spt_pndf_calc(&v, sigma_p, spt_pndf_term_coe(v.pd, m.c1.x), m.ax, m.ay, sx, sy);
    return half;
}

struct DBxdfSample {
    f3 wi, f;
    float pdf;
    bool transmit;
};

// ---- Subsurface substrate (src/bxdf/substrate.rs:182-350): k_shade<2> only -----------------------------
// what BxdfInputs adds for the BSSRDF (bxdf/mod.rs:62-67) and what BxdfSubsurfaceSample hands back (mod.rs:69-74)
struct DSubsurfaceIo {
    f3 po;
    DCoord coord_po;
    bool has;
    f3 pi;
    DCoord coord_pi;
    f3 sp;
    float pdf_pi;
};
SPT_DEV float ss_sample_r(const DScene& sc, float rand) {  // substrate.rs:219-229 (first entry with y >= rand)
    uint32_t lo = 1u, hi = SPT_SS_CDF_SIZE;   // the table is non-decreasing (checked when it is built): lower bound
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sc.ss_cdf[mid].y >= rand) hi = mid; else lo = mid + 1u;
    }
    if (lo >= SPT_SS_CDF_SIZE) return -1.0f;
    const float2 a = sc.ss_cdf[lo - 1u], b = sc.ss_cdf[lo];
    const float w = (rand - a.y) / (b.y - a.y);
    return b.x * w + a.x * (1.0f - w);
}
SPT_DEV f3 ss_sp(f3 d, float r) {  // substrate.rs:213-217
    const f3 q = mk3(-r / d.x, -r / d.y, -r / d.z);
    const f3 e1 = cexp(q), e2 = cexp(crcp(q, 3.0f));
    return cdiv((e1 + e2) * SPT_FRAC_1_PI, (d * 8.0f) * r);
}
// Subsurface::sample up to the diffuse lobe: false = the probe found nothing (wi = 0, bxdf = 0, pdf = 1).  The
// reference's "collect every intersection" loop re-uses `inter`, whose t bounds the next search from above while
// t_min moves to just behind it, so at most ONE intersection is ever collected (substrate.rs:280-291).
template <bool kGeoLds, bool kL>
SPT_DEV bool subsurface_probe(const DScene& sc, const DMat& m, DRng& rng, DSubsurfaceIo& io) {
    float rand_u = rng.next();
    const float rand_x = rng.next(), rand_y = rng.next();
    const f3 pt = io.coord_po.to_world(mk3(1, 0, 0)), pb = io.coord_po.to_world(mk3(0, 1, 0)), pn = io.coord_po.to_world(mk3(0, 0, 1));
    f3 st, sb, sn;
    if (rand_u < 0.5f) { rand_u = rand_u * 2.0f; st = pt; sb = pb; sn = pn; }
    else if (rand_u < 0.75f) { rand_u = rand_u * 4.0f - 2.0f; st = pb; sb = pn; sn = pt; }
    else { rand_u = rand_u * 4.0f - 3.0f; st = pn; sb = pt; sn = pb; }
    const f3 d = m.c1;
    float sp_d;
    if (rand_u < 1.0f / 3.0f) { rand_u = 3.0f * rand_u; sp_d = d.x; }
    else if (rand_u < 2.0f / 3.0f) { rand_u = 3.0f * rand_u - 1.0f; sp_d = d.y; }
    else { rand_u = 3.0f * rand_u - 2.0f; sp_d = d.z; }
    const float sample_r = ss_sample_r(sc, rand_x) * sp_d;
    const float r_max = sc.ss_cdf[SPT_SS_CDF_SIZE - 1].x * sp_d;
    if (sample_r < 0.0f) return false;
    const float pihi = 2.0f * SPT_PI * rand_y;
    const float pihi_cos = spt_cos(pihi), pihi_sin = spt_sin(pihi);
    const float sample_l = spt_sqrt(r_max * r_max + sample_r * sample_r);
    DRay ray;
    ray.o = ((io.po + (st * pihi_cos) * sample_r) + (sb * pihi_sin) * sample_r) + sn * sample_l;
    ray.d = -sn;
    ray.t_min = kTMinEps;
    const DHit h = trace_closest<kGeoLds>(sc, ray, 2.0f * sample_l);
    if (h.inst < 0) return false;
    const DInter it = reconstruct_hit<true, kL>(sc, ray, h);
    const spt_surface sf = load_surface<kL>(sc, it.surface);
    const DCoord coord_temp = surface_coord<true>(sc, sf, ray, it);
    const f3 pi = it.position;
    const f3 sp = ss_sp(d, length(pi - io.po));
    const f3 offset = io.coord_po.to_local(pi - io.po);
    const f3 nl = io.coord_po.to_local(it.normal);
    const float r_xy = spt_sqrt(offset.x * offset.x + offset.y * offset.y);
    const float r_yz = spt_sqrt(offset.y * offset.y + offset.z * offset.z);
    const float r_zx = spt_sqrt(offset.z * offset.z + offset.x * offset.x);
    const float pdf_xy = 0.5f * spt_abs(nl.z) * cavg(ss_sp(d, r_xy));
    const float pdf_yz = 0.25f * spt_abs(nl.x) * cavg(ss_sp(d, r_yz));
    const float pdf_zx = 0.25f * spt_abs(nl.y) * cavg(ss_sp(d, r_zx));
    io.has = true;
    io.pi = pi;
    io.coord_pi = coord_temp;
    io.sp = sp;
    io.pdf_pi = ((pdf_xy + pdf_yz) + pdf_zx) / 1.0f;
    return true;
}

// kSS: Subsurface substrates can occur (k_shade<3 | 5>: the BSSRDF probe is traced in here); kPndf: position-normal-distribution
// lobes can occur (k_shade<4 | 5>: tree walks with a 32-entry private stack).  sc / io are only touched then.  Two flags, so
// that a scene with one of the two features does not pay the registers of the other (round 2: one kernel for both, 374 VGPRs).
template <bool kSS = false, bool kGeoLds = false, bool kL = false, bool kPndf = kSS>
SPT_DEV DBxdfSample mat_sample(const DMat& m, f3 wo, DRng& rng, const DScene* sc = nullptr, DSubsurfaceIo* io = nullptr) {
    DBxdfSample s;
    s.transmit = false;
    switch (m.bxdf) {
    case SPT_BXDF_LAMBERT: {  // lambert.rs:20-36, rng.rs:72-80
        float rx = rng.next(), ry = rng.next();
        float phi = rx * 2.0f * SPT_PI;
        float sp, cp;
        spt_sincos(phi, &sp, &cp);
        float sin_theta = spt_sqrt(ry);
        float cos_theta = spt_sqrt(1.0f - ry);
        f3 wi = mk3(sin_theta * cp, sin_theta * sp, cos_theta);
        if (wo.z < 0.0f) wi.z = -wi.z;
        s.wi = wi;
        s.f = m.c0 * SPT_FRAC_1_PI;
        s.pdf = spt_abs(wi.z) * SPT_FRAC_1_PI;
        break;
    }
    case SPT_BXDF_MICROFACET_CONDUCTOR: {  // microfacet_conductor.rs:23-42
        float r0 = rng.next(), r1 = rng.next(), half_pdf;
        f3 half = ggx_vndf_sample(wo, m.ax, m.ay, r0, r1, &half_pdf);
        f3 fr = mat_fresnel(m, wo, half);
        f3 wi = reflect_n(wo, half);
        s.wi = wi;
        s.f = fr * ndf_visible(m, wo, wi, half);
        s.pdf = half_pdf / (4.0f * spt_abs(dot(wo, half)));
        break;
    }
    case SPT_BXDF_PNDF_CONDUCTOR: {  // MicrofacetConductor::sample (microfacet_conductor.rs:23-42) over a PndfMicrofacet
        if (kPndf) {
            float half_pdf;
            f3 half = pndf_sample_half(*sc, m, rng, &half_pdf);
            f3 fr = mat_fresnel(m, wo, half);
            f3 wi = reflect_n(wo, half);
            s.wi = wi;
            s.f = fr * pndf_ndf_visible(*sc, m, wo, wi, half);
            s.pdf = half_pdf / (4.0f * spt_abs(dot(wo, half)));
        }
        break;
    }
    case SPT_BXDF_SPECULAR_CONDUCTOR: {  // specular_conductor.rs:19-36
        f3 fr = mat_fresnel(m, wo, mk3(0, 0, 1));
        f3 wi = reflect_z(wo);
        s.wi = wi;
        s.f = crcp(fr, spt_abs(wi.z));
        s.pdf = 1.0f;
        break;
    }
    case SPT_BXDF_MICROFACET_DIELECTRIC: {  // microfacet_dielectric.rs:23-86
        float r0 = rng.next(), r1 = rng.next(), half_pdf;
        f3 half = ggx_vndf_sample(wo, m.ax, m.ay, r0, r1, &half_pdf);
        f3 fr = mat_fresnel(m, wo, half);
        float reflect_pdf = luminance(fr);
        f3 wi;
        if (rng.next() < reflect_pdf) {
            wi = reflect_n(wo, half);
            s.wi = wi;
            s.f = fr * ndf_visible(m, wo, wi, half);
            s.pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
        } else if (refract_n(wo, half, m.ior, &wi)) {
            float ior_ratio = (wo.z >= 0.0f) ? 1.0f / m.ior : m.ior;
            float denom = ior_ratio * dot(wo, half) + dot(wi, half);
            denom = denom * denom;
            float num = spt_abs(dot(wi, half));
            s.pdf = (1.0f - reflect_pdf) * half_pdf * num / denom;
            num = 4.0f * spt_abs(dot(wo, half)) * spt_abs(dot(wi, half));
            s.f = crcp(((gray(1.0f) - fr) * ndf_visible(m, wo, wi, half)) * num, denom);
            s.wi = wi;
            s.transmit = true;
        } else {
            s.wi = mk3(0, 0, 0); s.f = gray(0.0f); s.pdf = 1.0f; s.transmit = true;
        }
        break;
    }
    case SPT_BXDF_SPECULAR_DIELECTRIC: {  // specular_dielectric.rs:19-72
        f3 fr = mat_fresnel(m, wo, mk3(0, 0, 1));
        float reflect_pdf = luminance(fr);
        f3 wi;
        if (rng.next() < reflect_pdf) {
            wi = reflect_z(wo);
            s.wi = wi;
            s.f = crcp(fr, spt_abs(wi.z));
            s.pdf = reflect_pdf;
        } else if (refract_z(wo, m.ior, &wi)) {
            float ior_ratio = (wo.z >= 0.0f) ? 1.0f / m.ior : m.ior;
            s.wi = wi;
            s.f = crcp((gray(1.0f) - fr) * (ior_ratio * ior_ratio), spt_abs(wi.z));
            s.pdf = 1.0f - reflect_pdf;
            s.transmit = true;
        } else {
            s.wi = mk3(0, 0, 0); s.f = gray(0.0f); s.pdf = 1.0f; s.transmit = true;
        }
        break;
    }
    case SPT_BXDF_PNDF_PLASTIC:
    case SPT_BXDF_MICROFACET_PLASTIC:
    case SPT_BXDF_SPECULAR_PLASTIC: {  // microfacet_plastic.rs:26-79, specular_plastic.rs:19-63
        const bool glint = kPndf && m.bxdf == SPT_BXDF_PNDF_PLASTIC;   // the same lobe over a PndfMicrofacet
        const bool rough = m.bxdf == SPT_BXDF_MICROFACET_PLASTIC || glint;
        f3 fresnel_macro = plastic_fresnel(m, wo, mk3(0, 0, 1));
        float specular_weight = luminance(fresnel_macro);
        float substrate_weight = luminance((gray(1.0f) - fresnel_macro) * m.c0);
        float reflect_pdf = specular_weight / (specular_weight + substrate_weight);
        if (rng.next() < reflect_pdf) {
            f3 wi, specular_bxdf;
            float specular_pdf;
            if (rough) {
                float half_pdf;
                f3 half;
                if (glint) {
                    half = pndf_sample_half(*sc, m, rng, &half_pdf);
                } else {
                    float r0 = rng.next(), r1 = rng.next();
                    half = ggx_vndf_sample(wo, m.ax, m.ay, r0, r1, &half_pdf);
                }
                f3 fr = plastic_fresnel(m, wo, half);
                wi = reflect_n(wo, half);
                specular_bxdf = fr * (glint ? pndf_ndf_visible(*sc, m, wo, wi, half) : ndf_visible(m, wo, wi, half));
                specular_pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
            } else {
                wi = reflect_z(wo);
                specular_bxdf = crcp(fresnel_macro, spt_abs(wi.z));
                specular_pdf = reflect_pdf;
            }
            f3 substrate_bxdf = (gray(1.0f) - fresnel_macro) * substrate_eval(m, wo, wi);
            float sub_pdf = (1.0f - reflect_pdf) * substrate_pdf(wo, wi);
            s.wi = wi;
            s.f = specular_bxdf + substrate_bxdf;
            s.pdf = specular_pdf + sub_pdf;
        } else {
            // substrate.sample: cosine hemisphere; Subsurface::sample first places the exit point with a probe ray
            // (3 draws) and hands back an all-zero sample when that finds nothing
            f3 wi = mk3(0, 0, 0), samp_bxdf = mk3(0, 0, 0);
            float samp_pdf = 1.0f;
            bool probe_ok = true;
            if (kSS && m.substrate == SPT_SUBSTRATE_SUBSURFACE) probe_ok = subsurface_probe<kGeoLds, kL>(*sc, m, rng, *io);
            if (probe_ok) {
                float rx = rng.next(), ry = rng.next();
                float phi = rx * 2.0f * SPT_PI;
                float sp, cp;
                spt_sincos(phi, &sp, &cp);
                float sin_theta = spt_sqrt(ry);
                float cos_theta = spt_sqrt(1.0f - ry);
                wi = mk3(sin_theta * cp, sin_theta * sp, cos_theta);
                if (wo.z < 0.0f) wi.z = -wi.z;
                if (m.substrate != SPT_SUBSTRATE_LAMBERT) samp_bxdf = m.c2 * (1.0f - fresnel_n(m.ior, wi, mk3(0, 0, 1)));
                else samp_bxdf = m.c0 * SPT_FRAC_1_PI;
                samp_pdf = spt_abs(wi.z) * SPT_FRAC_1_PI;
            }
            float sub_pdf = (1.0f - reflect_pdf) * samp_pdf;
            f3 substrate_bxdf = (gray(1.0f) - fresnel_macro) * samp_bxdf;
            f3 specular_bxdf;
            float specular_pdf;
            if (rough) {
                f3 half = half_from_reflect(wo, wi);
                float half_pdf = glint ? pndf_half_pdf(*sc, m, half) : ggx_vndf_pdf(half, wo, m.ax, m.ay);
                specular_pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
                specular_bxdf = plastic_fresnel(m, wo, half) * (glint ? pndf_ndf_visible(*sc, m, wo, wi, half) : ndf_visible(m, wo, wi, half));
            } else {
                specular_pdf = reflect_pdf;
                specular_bxdf = crcp(fresnel_macro, spt_abs(wi.z));
            }
            s.wi = wi;
            s.f = substrate_bxdf + specular_bxdf;
            s.pdf = sub_pdf + specular_pdf;
        }
        break;
    }
    default: {  // pseudo.rs:14-27
        s.wi = -wo;
        s.f = crcp(gray(1.0f), spt_abs(wo.z));
        s.pdf = 1.0f;
        s.transmit = true;
        break;
    }
    }
    return s;
}

// kPndf (k_shade<4 | 5>): position-normal distributions can occur; sc is only touched then
template <bool kPndf = false>
SPT_DEV float mat_pdf(const DMat& m, f3 wo, f3 wi, const DScene* sc = nullptr, DPndfMemo* memo = nullptr) {
    switch (m.bxdf) {
    case SPT_BXDF_PNDF_CONDUCTOR:  // microfacet_conductor.rs:44-53
        if (kPndf && wo.z * wi.z >= 0.0f) {
            f3 half = half_from_reflect(wo, wi);
            return pndf_half_pdf(*sc, m, half, memo) / (4.0f * spt_abs(dot(wo, half)));
        }
        return 1.0f;
    case SPT_BXDF_LAMBERT:  // lambert.rs:38-44
        return (wo.z * wi.z >= 0.0f) ? spt_abs(wi.z) * SPT_FRAC_1_PI : 1.0f;
    case SPT_BXDF_MICROFACET_CONDUCTOR:  // microfacet_conductor.rs:44-53
        if (wo.z * wi.z >= 0.0f) {
            f3 half = half_from_reflect(wo, wi);
            return ggx_vndf_pdf(half, wo, m.ax, m.ay) / (4.0f * spt_abs(dot(wo, half)));
        }
        return 1.0f;
    case SPT_BXDF_MICROFACET_DIELECTRIC: {  // microfacet_dielectric.rs:88-113
        if (wo.z * wi.z >= 0.0f) {
            f3 half = half_from_reflect(wo, wi);
            float half_pdf = ggx_vndf_pdf(half, wo, m.ax, m.ay);
            float reflect_pdf = luminance(mat_fresnel(m, wo, half));
            return reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
        }
        f3 half = half_from_refract(wo, wi, m.ior);
        float half_pdf = ggx_vndf_pdf(half, wo, m.ax, m.ay);
        float reflect_pdf = luminance(mat_fresnel(m, wo, half));
        float ior_ratio = (wo.z >= 0.0f) ? 1.0f / m.ior : m.ior;
        float denom = ior_ratio * dot(wo, half) + dot(wi, half);
        denom = denom * denom;
        float num = spt_abs(dot(wi, half));
        return (1.0f - reflect_pdf) * half_pdf * num / denom;
    }
    case SPT_BXDF_SPECULAR_DIELECTRIC: {  // specular_dielectric.rs:74-82
        float reflect_pdf = luminance(mat_fresnel(m, wo, mk3(0, 0, 1)));
        return (wo.z * wi.z >= 0.0f) ? reflect_pdf : 1.0f - reflect_pdf;
    }
    case SPT_BXDF_PNDF_PLASTIC:
    case SPT_BXDF_MICROFACET_PLASTIC:
    case SPT_BXDF_SPECULAR_PLASTIC: {  // microfacet_plastic.rs:81-99, specular_plastic.rs:65-80
        if (!(wo.z * wi.z >= 0.0f)) return 1.0f;
        f3 fresnel_macro = plastic_fresnel(m, wo, mk3(0, 0, 1));
        float specular_weight = luminance(fresnel_macro);
        float substrate_weight = luminance((gray(1.0f) - fresnel_macro) * m.c0);
        float reflect_pdf = specular_weight / (specular_weight + substrate_weight);
        float specular_pdf;
        if (kPndf && m.bxdf == SPT_BXDF_PNDF_PLASTIC) {
            f3 half = half_from_reflect(wo, wi);
            specular_pdf = reflect_pdf * pndf_half_pdf(*sc, m, half, memo) / (4.0f * spt_abs(dot(wo, half)));
        } else if (m.bxdf == SPT_BXDF_MICROFACET_PLASTIC) {
            f3 half = half_from_reflect(wo, wi);
            specular_pdf = reflect_pdf * ggx_vndf_pdf(half, wo, m.ax, m.ay) / (4.0f * spt_abs(dot(wo, half)));
        } else {
            specular_pdf = reflect_pdf;
        }
        return specular_pdf + (1.0f - reflect_pdf) * substrate_pdf(wo, wi);
    }
    default:
        return 1.0f;
    }
}

template <bool kPndf = false>
SPT_DEV f3 mat_eval(const DMat& m, f3 wo, f3 wi, const DScene* sc = nullptr, DPndfMemo* memo = nullptr) {
    switch (m.bxdf) {
    case SPT_BXDF_PNDF_CONDUCTOR:  // microfacet_conductor.rs:55-64
        if (kPndf && wo.z * wi.z >= 0.0f) {
            f3 half = half_from_reflect(wo, wi);
            return mat_fresnel(m, wo, half) * pndf_ndf_visible(*sc, m, wo, wi, half, memo);
        }
        return gray(0.0f);
    case SPT_BXDF_LAMBERT:  // lambert.rs:46-52
        return (wo.z * wi.z >= 0.0f) ? m.c0 * SPT_FRAC_1_PI : gray(0.0f);
    case SPT_BXDF_MICROFACET_CONDUCTOR:  // microfacet_conductor.rs:55-64
        if (wo.z * wi.z >= 0.0f) {
            f3 half = half_from_reflect(wo, wi);
            return mat_fresnel(m, wo, half) * ndf_visible(m, wo, wi, half);
        }
        return gray(0.0f);
    case SPT_BXDF_SPECULAR_CONDUCTOR:  // specular_conductor.rs:42-50
        if (dot(wi, reflect_z(wo)) > 0.999f) return crcp(mat_fresnel(m, wo, mk3(0, 0, 1)), spt_abs(wi.z));
        return gray(0.0f);
    case SPT_BXDF_MICROFACET_DIELECTRIC: {  // microfacet_dielectric.rs:115-138
        if (wo.z * wi.z >= 0.0f) {
            f3 half = half_from_reflect(wo, wi);
            return mat_fresnel(m, wo, half) * ndf_visible(m, wo, wi, half);
        }
        f3 half = half_from_refract(wo, wi, m.ior);
        f3 fr = mat_fresnel(m, wo, half);
        float ior_ratio = (wo.z >= 0.0f) ? 1.0f / m.ior : m.ior;
        float denom = ior_ratio * dot(wo, half) + dot(wi, half);
        denom = denom * denom;
        float num = 4.0f * spt_abs(dot(wo, half)) * spt_abs(dot(wi, half));
        return crcp(((gray(1.0f) - fr) * ndf_visible(m, wo, wi, half)) * num, denom);
    }
    case SPT_BXDF_SPECULAR_DIELECTRIC: {  // specular_dielectric.rs:84-107
        f3 fr = mat_fresnel(m, wo, mk3(0, 0, 1));
        if (wo.z * wi.z >= 0.0f) {
            if (dot(wi, reflect_z(wo)) > 0.999f) return crcp(fr, spt_abs(wi.z));
            return gray(0.0f);
        }
        f3 ewi;
        if (refract_z(wo, m.ior, &ewi) && dot(wi, ewi) > 0.999f) {
            float ior_ratio = (wo.z >= 0.0f) ? 1.0f / m.ior : m.ior;
            return crcp((gray(1.0f) - fr) * (ior_ratio * ior_ratio), spt_abs(wi.z));
        }
        return gray(0.0f);
    }
    case SPT_BXDF_PNDF_PLASTIC:
    case SPT_BXDF_MICROFACET_PLASTIC: {  // microfacet_plastic.rs:101-118
        if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
        f3 half = half_from_reflect(wo, wi);
        const bool glint = kPndf && m.bxdf == SPT_BXDF_PNDF_PLASTIC;
        f3 refl = plastic_fresnel(m, wo, half) * (glint ? pndf_ndf_visible(*sc, m, wo, wi, half, memo) : ndf_visible(m, wo, wi, half));
        f3 sub = (gray(1.0f) - plastic_fresnel(m, wo, mk3(0, 0, 1))) * substrate_eval(m, wo, wi);
        return refl + sub;
    }
    case SPT_BXDF_SPECULAR_PLASTIC: {  // specular_plastic.rs:82-93
        if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
        f3 fr = plastic_fresnel(m, wo, mk3(0, 0, 1));
        return crcp(fr, spt_abs(wi.z)) + (gray(1.0f) - fr) * substrate_eval(m, wo, wi);
    }
    default:  // pseudo.rs:32-38
        if (dot(wo, wi) < -0.999f) return crcp(gray(1.0f), spt_abs(wi.z));
        return gray(0.0f);
    }
}

// ---- lights
SPT_DEV uint32_t alias_sample(const float* props, const float* u, const uint32_t* k, uint32_t n, float rand, float* prob) {  // alias_table.rs:60-68
    float temp = rand * (float)n;
    uint32_t x = spt_f2u_sat(temp);
    float y = temp - (float)x;
    if (y < u[x]) { *prob = props[x]; return x; }
    uint32_t kx = k[x];
    *prob = props[kx];
    return kx;
}

// EnvLight::strength_dist_pdf(theta, phi) (environment.rs:51-84)
SPT_DEV void env_lookup(const DScene& sc, float theta, float phi, f3* c_out, float* p_out) {
    int32_t W = (int32_t)sc.env_w, H = (int32_t)sc.env_h;
    float x = phi * 0.5f * SPT_FRAC_1_PI * (float)sc.env_w;
    int32_t x1 = spt_f2i_sat(spt_round(x));
    int32_t x0 = (int32_t)((uint32_t)x1 - 1u);   // wrapping, as release-mode Rust
    float xt = x - (float)x0 - 0.5f;
    uint32_t ux0 = (uint32_t)(x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0));
    uint32_t ux1 = (uint32_t)(x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1));
    float y = theta * SPT_FRAC_1_PI * (float)sc.env_h;
    int32_t y1 = spt_f2i_sat(spt_round(y));
    int32_t y0 = (int32_t)((uint32_t)y1 - 1u);
    float yt = y - (float)y0 - 0.5f;
    uint32_t uy0 = (uint32_t)(y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0));
    uint32_t uy1 = (uint32_t)(y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1));
    size_t i00 = (size_t)uy0 * sc.env_w + ux0, i01 = (size_t)uy1 * sc.env_w + ux0;
    size_t i10 = (size_t)uy0 * sc.env_w + ux1, i11 = (size_t)uy1 * sc.env_w + ux1;
    const float4 t00 = sc.env_px[i00], t01 = sc.env_px[i01], t10 = sc.env_px[i10], t11 = sc.env_px[i11];
    f3 c00 = mk3(t00), c01 = mk3(t01), c10 = mk3(t10), c11 = mk3(t11);
    f3 c0 = c00 * (1.0f - yt) + c01 * yt;
    f3 c1 = c10 * (1.0f - yt) + c11 * yt;
    f3 c = c0 * (1.0f - xt) + c1 * xt;
    float p0 = t00.w * (1.0f - yt) + t01.w * yt;
    float p1 = t10.w * (1.0f - yt) + t11.w * yt;
    *p_out = p0 * (1.0f - xt) * p1 * xt;
    *c_out = c * mk3(sc.env_scale);
}
SPT_DEV void env_strength_pdf(const DScene& sc, f3 wi, f3* c, float* pdf) {  // environment.rs:128-133
    float theta = spt_acos(wi.y);
    float phi = spt_atan2(wi.x, wi.z) + SPT_PI;
    env_lookup(sc, theta, phi, c, pdf);
}

struct DLightSample {
    f3 dir, strength;
    float pdf, dist;
    bool is_delta;
};

// Instance::sample (instance.rs:111-129) over Sphere::sample (sphere.rs:103-136) /
// BvhAccel<Triangle>::sample (bvh.rs:293-298) + Triangle::sample (triangle.rs:224-271)
template <bool kTex = false>
SPT_DEV void instance_sample(const DScene& sc, const DInstance& in, DRng& rng, f3* pos, f3* nrm, float* pdf_out, DInter* at = nullptr) {
    f3 p, n, tg, bt;
    float pdf;
    if (in.prim_type == SPT_PRIM_SPHERE) {
        float4 s = sc.spheres[in.prim_id];
        float rx = rng.next(), ry = rng.next();
        float phi = rx * 2.0f * SPT_PI;
        float sphi, cphi;
        spt_sincos(phi, &sphi, &cphi);
        float cos_theta = 1.0f - 2.0f * ry;
        float sin_theta = spt_sqrt(1.0f - cos_theta * cos_theta);
        n = mk3(sin_theta * cphi, sin_theta * sphi, cos_theta);
        p = mk3(s) + n * s.w;
        if (kTex) sphere_normal_to_texcoords(n, at->uv);
        sphere_frame(n, &tg, &bt);
        pdf = 0.25f * SPT_FRAC_1_PI;
    } else {
        uint4 mesh = sc.meshes[in.prim_id];
        float fi = rng.next() * (float)mesh.w;
        uint32_t idx = spt_f2u_sat(fi);
        if (idx > mesh.w - 1) idx = mesh.w - 1;
        uint32_t tri = mesh.z + idx;
        float r0 = rng.next(), r1 = rng.next();
        float r0_sqrt = spt_sqrt(r0);
        float u = 1.0f - r0_sqrt;
        float v = r0_sqrt * (1.0f - r1);
        float w = 1.0f - u - v;
        f3 p0 = mk3(sc.tri_pos[3 * tri]), p1 = mk3(sc.tri_pos[3 * tri + 1]), p2 = mk3(sc.tri_pos[3 * tri + 2]);
        const float4* A = sc.tri_attr + 9 * tri;
        float4 q[kTex ? 9 : 7];
#pragma unroll
        for (int k = 0; k < (kTex ? 9 : 7); ++k) q[k] = A[k];
        const float* a = reinterpret_cast<const float*>(q);
        if (kTex) {  // triangle.rs:258
            at->uv[0] = (a[27] * u + a[29] * v) + a[31] * w;
            at->uv[1] = (a[28] * u + a[30] * v) + a[32] * w;
        }
        p = (p0 * u + p1 * v) + p2 * w;
        float area = length(cross(p1 - p0, p2 - p0)) * 0.5f;
        n = (mk3(a) * u + mk3(a + 3) * v) + mk3(a + 6) * w;
        tg = (mk3(a + 9) * u + mk3(a + 12) * v) + mk3(a + 15) * w;
        bt = (mk3(a + 18) * u + mk3(a + 21) * v) + mk3(a + 24) * w;
        pdf = (1.0f / spt_max(area, 0.001f)) / (float)mesh.w;
    }
    float original_area = length(cross(tg, bt));
    *pos = xf_point(in.fwd, p);
    *nrm = normalize((mk3(in.nrm) * n.x + mk3(in.nrm + 3) * n.y) + mk3(in.nrm + 6) * n.z);
    bt = xf_vector(in.fwd, bt);
    tg = xf_vector(in.fwd, tg);
    float transformed_area = length(cross(tg, bt));
    *pdf_out = pdf * original_area / transformed_area;
    if (kTex) {  // the sampled point as the Intersection Surface::emissive sees (no differentials)
        at->position = *pos; at->normal = *nrm; at->tangent = tg; at->bitangent = bt;
        at->duvdx[0] = 0.0f; at->duvdx[1] = 0.0f; at->duvdy[0] = 0.0f; at->duvdy[1] = 0.0f;
    }
}

// Instance::pdf (instance.rs:131-141) over Triangle::pdf / Sphere::pdf
SPT_DEV float instance_pdf(const DScene& sc, uint32_t inst, const DInter& it, int32_t prim) {
    // re-read the 3 float4 of trans_inv here instead of keeping the whole instance record live through
    // the shading code (L1 hit; keeps k_shade<false> under the next VGPR step)
    const float4* I = sc.instances + 12 * inst;
    float4 m0 = I[0], m1 = I[1], m2 = I[2];
    const float inv[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
    f3 tangent = xf_vector(inv, it.tangent);
    f3 bitangent = xf_vector(inv, it.bitangent);
    float original_area = length(cross(tangent, bitangent));
    float transformed_area = length(cross(it.tangent, it.bitangent));
    float prim_pdf;
    if (it.prim_type == SPT_PRIM_SPHERE) {
        prim_pdf = 0.25f * SPT_FRAC_1_PI;
    } else {
        uint4 mesh = sc.meshes[it.prim_id];
        f3 p0 = mk3(sc.tri_pos[3 * prim]), p1 = mk3(sc.tri_pos[3 * prim + 1]), p2 = mk3(sc.tri_pos[3 * prim + 2]);
        float area = length(cross(p1 - p0, p2 - p0)) * 0.5f;
        prim_pdf = (1.0f / spt_max(area, 0.001f)) / (float)mesh.w;
    }
    return prim_pdf * original_area / transformed_area;
}

// kDeltaOnly: the scene has only directional / point / spot lights (checked on the host), so the
// area-light and environment branches are not even compiled into the kernel.
template <bool kDeltaOnly, bool kTex = false, bool kL = false>
SPT_DEV void light_sample(const DScene& sc, const spt_light& l, f3 position, DRng& rng, DLightSample* out) {
    uint32_t type = l.type;
    if (kDeltaOnly && type > SPT_LIGHT_SPOT) type = SPT_LIGHT_DIRECTIONAL;
    switch (type) {
    case SPT_LIGHT_DIRECTIONAL:  // directional.rs:26-29
        out->dir = -mk3(l.dir); out->pdf = 1.0f; out->strength = mk3(l.strength); out->dist = SPT_F32_MAX; out->is_delta = true;
        return;
    case SPT_LIGHT_POINT: {  // point.rs:23-29
        f3 sv = mk3(l.pos) - position;
        float dist_sqr = dot(sv, sv);
        float dist = spt_sqrt(dist_sqr);
        out->dir = sv / dist; out->pdf = 1.0f; out->strength = crcp(mk3(l.strength), dist_sqr); out->dist = dist; out->is_delta = true;
        return;
    }
    case SPT_LIGHT_SPOT: {  // spot.rs:50-65
        f3 sv = mk3(l.pos) - position;
        float dist_sqr = dot(sv, sv);
        float dist = spt_sqrt(dist_sqr);
        sv = sv / dist;
        float atten = spt_clamp((dot(mk3(l.dir), -sv) - l.cos_outer) / spt_max(l.cos_inner - l.cos_outer, 0.0001f), 0.0f, 1.0f);
        out->dir = sv; out->pdf = 1.0f; out->strength = crcp(mk3(l.strength) * atten, dist_sqr); out->dist = dist; out->is_delta = true;
        return;
    }
    case SPT_LIGHT_SHAPE: {  // shape_light.rs:20-42
        if (kDeltaOnly) return;
        DInstance in = load_instance<kL>(sc, l.instance);
        const spt_surface sf = load_surface<kL>(sc, in.surface);
        f3 spos, snrm;
        float spdf;
        DInter at;
        instance_sample<kTex>(sc, in, rng, &spos, &snrm, &spdf, &at);
        f3 emissive = surface_emissive<kTex>(sc, sf, at);
        f3 light_vec = spos - position;
        float dist_sqr = dot(light_vec, light_vec);
        float dist = spt_sqrt(dist_sqr);
        f3 light_dir = light_vec / dist;
        float cosv;
        if (sf.flags & SPT_SURF_DOUBLE_SIDED) {
            cosv = spt_abs(dot(light_dir, snrm));
        } else {
            cosv = dot(light_dir, -snrm);
            if (!(cosv > 0.0f)) { cosv = 1.0f; emissive = gray(0.0f); }
        }
        out->dir = light_dir; out->pdf = spdf * dist_sqr / spt_max(cosv, 0.001f); out->strength = emissive; out->dist = dist; out->is_delta = false;
        return;
    }
    default: {  // environment.rs:110-126
        if (kDeltaOnly) return;
        uint32_t ind;   // alias_sample (alias_table.rs:60-68) on the packed table; the returned probability is
        {               // not used by EnvLight::sample (the pdf comes from the bilinear lookup below, quirk Q5)
            const float temp = rng.next() * (float)(sc.env_w * sc.env_h);
            const uint32_t x = spt_f2u_sat(temp);
            const float y = temp - (float)x;
            const uint2 uk = sc.env_uk[x];
            ind = (y < __uint_as_float(uk.x)) ? x : uk.y;
        }
        uint32_t x = ind % sc.env_w, y = ind / sc.env_w;
        float rx = rng.next(), ry = rng.next();
        float theta = ((float)y + ry) / (float)sc.env_h * SPT_PI;
        float phi = ((float)x + rx) / (float)sc.env_w * 2.0f * SPT_PI;
        float st, ct, sp, cp;
        spt_sincos(theta, &st, &ct);
        spt_sincos(phi, &sp, &cp);
        out->dir = mk3(st * sp, ct, st * cp);
        env_lookup(sc, theta, phi, &out->strength, &out->pdf);
        out->dist = spt_inf(); out->is_delta = false;
        return;
    }
    }
}

// sample_light (uniform.rs:28-41, power_is.rs:49-59); false if the scene has no light
template <bool kDeltaOnly, bool kTex = false, bool kL = false>
SPT_DEV bool sample_light(const DScene& sc, f3 position, DRng& rng, DLightSample* out) {
    if (sc.n_lights == 0) return false;
    if (sc.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS) {
        float pr;
        uint32_t index = alias_sample(sc.light_props, sc.light_u, sc.light_k, sc.n_lights, rng.next(), &pr);
        light_sample<kDeltaOnly, kTex, kL>(sc, load_light<kL>(sc, index), position, rng, out);
        out->pdf = pr * out->pdf;
    } else {
        float fi = rng.next() * (float)sc.n_lights;
        uint32_t index = spt_f2u_sat(fi);
        if (index > sc.n_lights - 1) index = sc.n_lights - 1;
        light_sample<kDeltaOnly, kTex, kL>(sc, load_light<kL>(sc, index), position, rng, out);
        out->pdf = out->pdf * (1.0f / (float)sc.n_lights);
    }
    return true;
}
// pdf_shape_light (uniform.rs:43-68, power_is.rs:61-88)
SPT_DEV float pdf_shape_light(const DScene& sc, uint32_t inst, uint32_t surf_flags, f3 position, const DInter& it, int32_t prim) {
    float primitive_pdf = instance_pdf(sc, inst, it, prim);
    f3 light_vec = it.position - position;
    float dist_sqr = dot(light_vec, light_vec);
    f3 light_dir = light_vec / spt_sqrt(dist_sqr);
    float cosv;
    if (surf_flags & SPT_SURF_DOUBLE_SIDED) {
        cosv = spt_abs(dot(light_dir, it.normal));
    } else {
        cosv = dot(light_dir, -it.normal);
        if (!(cosv > 0.0f)) cosv = 1.0f;
    }
    float local_pdf = primitive_pdf * dist_sqr / spt_max(cosv, 0.00001f);
    if (sc.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS) return local_pdf * sc.light_props[it.light];
    return local_pdf * (1.0f / (float)sc.n_lights);
}
SPT_DEV float pdf_env_light(const DScene& sc) {  // uniform.rs:70-76, power_is.rs:90-96
    if (sc.env_light_index < 0) return 1.0f;
    if (sc.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS) return sc.light_props[sc.env_light_index];
    return 1.0f / (float)sc.n_lights;
}

// ---- medium (homogeneous.rs, util.rs)
SPT_DEV float henyey_greenstein(float g, float cosv) {
    float g2 = g * g;
    float denom = 1.0f + g2 + 2.0f * g * cosv;
    denom = denom * spt_sqrt(denom);
    return 0.25f * SPT_FRAC_1_PI * (1.0f - g2) / denom;
}
SPT_DEV float hg_cdf_inverse(float g, float rand) {
    if (spt_abs(g) < 0.01f) return 1.0f - 2.0f * rand;
    float g2 = g * g;
    float temp = (1.0f - g2) / (1.0f - g + 2.0f * g * rand);
    return 0.5f * (1.0f + g2 - temp * temp) / g;
}
SPT_DEV f3 hg_local_to_world(f3 wo_world, f3 wi_local) {
    f3 v = (spt_abs(wo_world.y) < 0.99f) ? mk3(0, 1, 0) : mk3(1, 0, 0);
    f3 u = normalize(cross(v, wo_world));
    v = cross(wo_world, u);
    return (u * wi_local.x + v * wi_local.y) + wo_world * wi_local.z;
}

SPT_DEV float power_heuristic(float p0, float p1) {  // pt.rs:298-302
    float prod0 = 1.0f * p0;
    float prod1 = 1.0f * p1;
    return prod0 * prod0 / (prod0 * prod0 + prod1 * prod1);
}
