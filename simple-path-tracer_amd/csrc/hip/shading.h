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

struct DRng {
    spt_rng s;
    SPT_DEV float next() { return spt_rng_f32(&s); }
};

// ---- what Triangle::intersect / Sphere::intersect + Instance::intersect leave in `Intersection`
struct DInter {
    f3 position, normal, tangent, bitangent;  // world space
    uint32_t surface, prim_type, prim_id;
    int32_t light;
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

struct DInstance {
    float inv[12], fwd[12], nrm[9];
    uint32_t prim_type, prim_id, surface;
    int32_t light;
};
SPT_DEV DInstance load_instance(const DScene& sc, uint32_t inst) {
    const float4* I = sc.instances + 12 * inst;
    DInstance d;
    float4 q[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) q[k] = I[k];
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

// Rebuild the shading inputs of a recorded hit (t, instance, prim, v, w):
// triangle.rs:188-212 or sphere.rs:64-83, then instance.rs:97-104.
SPT_DEV DInter reconstruct_hit(const DScene& sc, const DRay& ray, const DHit& h) {
    const DInstance in = load_instance(sc, (uint32_t)h.inst);
    DInter it;
    f3 n, tg, bt;
    if (in.prim_type == SPT_PRIM_SPHERE) {
        DRay orr;
        orr.o = xf_point(in.inv, ray.o);
        orr.d = xf_vector(in.inv, ray.d);
        float4 s = sc.spheres[in.prim_id];
        n = (point_at(orr, h.t) - mk3(s)) / s.w;
        sphere_frame(n, &tg, &bt);
    } else {
        const float4* A = sc.tri_attr + 9 * (uint32_t)h.prim;
        float4 q[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = A[k];
        const float* a = reinterpret_cast<const float*>(q);  // n[3][3] t[3][3] b[3][3]
        float v = h.v, w = h.w;
        float u = 1.0f - v - w;
        n = normalize((mk3(a) * u + mk3(a + 3) * v) + mk3(a + 6) * w);
        tg = (mk3(a + 9) * u + mk3(a + 12) * v) + mk3(a + 15) * w;
        bt = (mk3(a + 18) * u + mk3(a + 21) * v) + mk3(a + 24) * w;
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

// ---- Coordinate (coord.rs:10-59)
struct DCoord {
    f3 xw, yw, zw, hemi;
    SPT_DEV f3 to_local(f3 w) const { return mk3(dot(xw, w), dot(yw, w), dot(zw, w)); }
    SPT_DEV f3 to_world(f3 l) const { return (xw * l.x + yw * l.y) + zw * l.z; }
};
// Surface::coord without normal map (surface.rs:65-95) + Coordinate::from_tangent_normal
SPT_DEV DCoord surface_coord(uint32_t surf_flags, const DRay& ray, const DInter& it) {
    bool hit_back = dot(ray.d, it.normal) > 0.0f;
    bool ds = (surf_flags & SPT_SURF_DOUBLE_SIDED) != 0;
    DCoord c;
    c.zw = (ds && hit_back) ? -it.normal : it.normal;
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
SPT_DEV DMat load_material(const DScene& sc, uint32_t m) {
    const float4* p = reinterpret_cast<const float4*>(sc.materials + m);   // 64-byte records
    float4 a = p[0], b = p[1], c = p[2], e = p[3];
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
SPT_DEV bool mat_is_delta(const DMat& m) {
    return m.bxdf == SPT_BXDF_SPECULAR_CONDUCTOR || m.bxdf == SPT_BXDF_SPECULAR_DIELECTRIC || m.bxdf == SPT_BXDF_PSEUDO;
}
SPT_DEV f3 mat_fresnel(const DMat& m, f3 i, f3 n) {  // fresnel.rs:29-59
    if (m.bxdf == SPT_BXDF_MICROFACET_CONDUCTOR || m.bxdf == SPT_BXDF_SPECULAR_CONDUCTOR) return fresnel_conductor_n(m.c0, m.c1, i, n);
    return gray(fresnel_n(m.ior, i, n));
}
// plastic lobes: SchlickFresnel / DielectricFresnel (fresnel.rs:19-59), Lambert / Diffuse substrate
// (substrate.rs:29-45, 139-180)
SPT_DEV float pow5(float x) { return x * x * x * x * x; }
SPT_DEV f3 plastic_fresnel(const DMat& m, f3 i, f3 n) {
    if (m.fresnel == SPT_FRESNEL_SCHLICK) return m.c1 + (gray(1.0f) - m.c1) * pow5(1.0f - dot(i, n));
    return gray(fresnel_n(m.ior, i, n));
}
SPT_DEV float substrate_pdf(f3 wo, f3 wi) { return (wo.z * wi.z >= 0.0f) ? spt_abs(wi.z) * SPT_FRAC_1_PI : 1.0f; }
SPT_DEV f3 substrate_eval(const DMat& m, f3 wo, f3 wi) {
    if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
    if (m.substrate == SPT_SUBSTRATE_DIFFUSE) return m.c2 * (1.0f - fresnel_n(m.ior, wi, mk3(0, 0, 1)));
    return m.c0 * SPT_FRAC_1_PI;
}
SPT_DEV float ndf_visible(const DMat& m, f3 wo, f3 wi, f3 h) {  // microfacet.rs:47-53
    return ggx_ndf_aniso(h, m.ax, m.ay) * smith_visible_aniso(wo, wi, m.ax, m.ay);
}

struct DBxdfSample {
    f3 wi, f;
    float pdf;
    bool transmit;
};

SPT_DEV DBxdfSample mat_sample(const DMat& m, f3 wo, DRng& rng) {
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
    case SPT_BXDF_MICROFACET_PLASTIC:
    case SPT_BXDF_SPECULAR_PLASTIC: {  // microfacet_plastic.rs:26-79, specular_plastic.rs:19-63
        const bool rough = m.bxdf == SPT_BXDF_MICROFACET_PLASTIC;
        f3 fresnel_macro = plastic_fresnel(m, wo, mk3(0, 0, 1));
        float specular_weight = luminance(fresnel_macro);
        float substrate_weight = luminance((gray(1.0f) - fresnel_macro) * m.c0);
        float reflect_pdf = specular_weight / (specular_weight + substrate_weight);
        if (rng.next() < reflect_pdf) {
            f3 wi, specular_bxdf;
            float specular_pdf;
            if (rough) {
                float r0 = rng.next(), r1 = rng.next(), half_pdf;
                f3 half = ggx_vndf_sample(wo, m.ax, m.ay, r0, r1, &half_pdf);
                f3 fr = plastic_fresnel(m, wo, half);
                wi = reflect_n(wo, half);
                specular_bxdf = fr * ndf_visible(m, wo, wi, half);
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
            float rx = rng.next(), ry = rng.next();
            float phi = rx * 2.0f * SPT_PI;
            float sp, cp;
            spt_sincos(phi, &sp, &cp);
            float sin_theta = spt_sqrt(ry);
            float cos_theta = spt_sqrt(1.0f - ry);
            f3 wi = mk3(sin_theta * cp, sin_theta * sp, cos_theta);
            if (wo.z < 0.0f) wi.z = -wi.z;
            f3 samp_bxdf;
            if (m.substrate == SPT_SUBSTRATE_DIFFUSE) samp_bxdf = m.c2 * (1.0f - fresnel_n(m.ior, wi, mk3(0, 0, 1)));
            else samp_bxdf = m.c0 * SPT_FRAC_1_PI;
            float samp_pdf = spt_abs(wi.z) * SPT_FRAC_1_PI;
            float sub_pdf = (1.0f - reflect_pdf) * samp_pdf;
            f3 substrate_bxdf = (gray(1.0f) - fresnel_macro) * samp_bxdf;
            f3 specular_bxdf;
            float specular_pdf;
            if (rough) {
                f3 half = half_from_reflect(wo, wi);
                float half_pdf = ggx_vndf_pdf(half, wo, m.ax, m.ay);
                specular_pdf = reflect_pdf * half_pdf / (4.0f * spt_abs(dot(wo, half)));
                specular_bxdf = plastic_fresnel(m, wo, half) * ndf_visible(m, wo, wi, half);
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

SPT_DEV float mat_pdf(const DMat& m, f3 wo, f3 wi) {
    switch (m.bxdf) {
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
    case SPT_BXDF_MICROFACET_PLASTIC:
    case SPT_BXDF_SPECULAR_PLASTIC: {  // microfacet_plastic.rs:81-99, specular_plastic.rs:65-80
        if (!(wo.z * wi.z >= 0.0f)) return 1.0f;
        f3 fresnel_macro = plastic_fresnel(m, wo, mk3(0, 0, 1));
        float specular_weight = luminance(fresnel_macro);
        float substrate_weight = luminance((gray(1.0f) - fresnel_macro) * m.c0);
        float reflect_pdf = specular_weight / (specular_weight + substrate_weight);
        float specular_pdf;
        if (m.bxdf == SPT_BXDF_MICROFACET_PLASTIC) {
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

SPT_DEV f3 mat_eval(const DMat& m, f3 wo, f3 wi) {
    switch (m.bxdf) {
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
    case SPT_BXDF_MICROFACET_PLASTIC: {  // microfacet_plastic.rs:101-118
        if (!(wo.z * wi.z >= 0.0f)) return gray(0.0f);
        f3 half = half_from_reflect(wo, wi);
        f3 refl = plastic_fresnel(m, wo, half) * ndf_visible(m, wo, wi, half);
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
    int32_t x0 = x1 - 1;
    float xt = x - (float)x0 - 0.5f;
    uint32_t ux0 = (uint32_t)(x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0));
    uint32_t ux1 = (uint32_t)(x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1));
    float y = theta * SPT_FRAC_1_PI * (float)sc.env_h;
    int32_t y1 = spt_f2i_sat(spt_round(y));
    int32_t y0 = y1 - 1;
    float yt = y - (float)y0 - 0.5f;
    uint32_t uy0 = (uint32_t)(y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0));
    uint32_t uy1 = (uint32_t)(y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1));
    size_t i00 = (size_t)uy0 * sc.env_w + ux0, i01 = (size_t)uy1 * sc.env_w + ux0;
    size_t i10 = (size_t)uy0 * sc.env_w + ux1, i11 = (size_t)uy1 * sc.env_w + ux1;
    f3 c00 = mk3(sc.env_texels + 3 * i00), c01 = mk3(sc.env_texels + 3 * i01);
    f3 c10 = mk3(sc.env_texels + 3 * i10), c11 = mk3(sc.env_texels + 3 * i11);
    f3 c0 = c00 * (1.0f - yt) + c01 * yt;
    f3 c1 = c10 * (1.0f - yt) + c11 * yt;
    f3 c = c0 * (1.0f - xt) + c1 * xt;
    float p0 = sc.env_props[i00] * (1.0f - yt) + sc.env_props[i01] * yt;
    float p1 = sc.env_props[i10] * (1.0f - yt) + sc.env_props[i11] * yt;
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
SPT_DEV void instance_sample(const DScene& sc, const DInstance& in, DRng& rng, f3* pos, f3* nrm, float* pdf_out) {
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
        float4 q[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = A[k];
        const float* a = reinterpret_cast<const float*>(q);
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
template <bool kDeltaOnly>
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
        DInstance in = load_instance(sc, l.instance);
        const spt_surface& sf = sc.surfaces[in.surface];
        f3 spos, snrm;
        float spdf;
        instance_sample(sc, in, rng, &spos, &snrm, &spdf);
        f3 emissive = mk3(sf.emissive);
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
        float pr;
        uint32_t ind = alias_sample(sc.env_props, sc.env_u, sc.env_k, sc.env_w * sc.env_h, rng.next(), &pr);
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
template <bool kDeltaOnly>
SPT_DEV bool sample_light(const DScene& sc, f3 position, DRng& rng, DLightSample* out) {
    if (sc.n_lights == 0) return false;
    if (sc.light_sampler == SPT_LIGHT_SAMPLER_POWER_IS) {
        float pr;
        uint32_t index = alias_sample(sc.light_props, sc.light_u, sc.light_k, sc.n_lights, rng.next(), &pr);
        light_sample<kDeltaOnly>(sc, sc.lights[index], position, rng, out);
        out->pdf = pr * out->pdf;
    } else {
        float fi = rng.next() * (float)sc.n_lights;
        uint32_t index = spt_f2u_sat(fi);
        if (index > sc.n_lights - 1) index = sc.n_lights - 1;
        light_sample<kDeltaOnly>(sc, sc.lights[index], position, rng, out);
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
