// Position-normal distributions ("glints"): the per-material tables of a pndf_conductor material.
//
// Replaces, on the host (scene assembly):
//   PndfConductor::new                 reference src/material/pndf_conductor.rs:31-129 (one Gaussian per cell of the normal map)
//   get_normal_bilinear                pndf_conductor.rs:198-210
//   PndfGaussTerm::new                 src/bxdf/pndf_bvh.rs:405-437
//   PndfAccel::new / PndfBvh::new / PndfUvBvh::new   pndf_bvh.rs:55-92, 124-190, 266-333
// The evaluation (PndfAccel::calc / find_terms, PndfMicrofacet) runs per hit on the device (csrc/hip/shading.h) and in the
// oracle; this file only produces the arrays of include/spt_abi.h (spt_pndf, spt_pndf_term, spt_pndf_node, refs, roots).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/spt_detmath.h"
#include "hmath.hpp"
#include "host_scene.hpp"

namespace spt_host {

namespace {

struct TexIn {
    float spec[3] = {0, 0, 0};
    float uv[2] = {0, 0};
    int mode = SPT_TEXMODE_SPECIFIED, wrap = SPT_TEXWRAP_REPEAT;
};

float wrap1(float v, int wrap) {   // TextureInput::value_vec2_wrapped (src/texture/mod.rs:85-135), one coordinate
    switch (wrap) {
    case SPT_TEXWRAP_REPEAT: { const float f = spt_fract(v); return v >= 0.0f ? f : 1.0f + f; }
    case SPT_TEXWRAP_MIRROR_REPEAT: {
        const float f = spt_fract(v);
        const float n = v >= 0.0f ? f : 1.0f + f;
        return ((int32_t)v % 2 == 0) ? n : 1.0f - n;
    }
    case SPT_TEXWRAP_CLAMP: return std::fmin(std::fmax(v, 0.0f), 1.0f);
    default: return std::fabs(std::fmin(std::fmax(v, 0.0f), 1.0f));
    }
}

// sample_blinear (src/texture/image_tex.rs:100-123) on level 0: with zero differentials sample_trilinear picks level 0
// with weight 1 (image_tex.rs:125-151)
V3 bilinear0(const HostScene& hs, uint32_t image, float u, float v) {
    const spt_image_level& L = hs.image_levels[hs.images[image].first_level];
    const int w = (int)L.width, h = (int)L.height;
    auto px = [&](int x, int y) {
        const uint32_t t = hs.texels[L.first_texel + (uint32_t)y * L.width + (uint32_t)x];
        return V3{(float)(t & 255u) / 255.0f, (float)((t >> 8) & 255u) / 255.0f, (float)((t >> 16) & 255u) / 255.0f};
    };
    auto clampi = [](int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); };
    const float x = u * (float)w;
    const int x1r = (int)spt_round(x), x0r = x1r - 1;
    const float xt = x - (float)x0r - 0.5f;
    const int x0 = clampi(x0r, 0, w - 1), x1 = clampi(x1r, 0, w - 1);
    const float y = v * (float)h;
    const int y1r = (int)spt_round(y), y0r = y1r - 1;
    const float yt = y - (float)y0r - 0.5f;
    const int y0 = clampi(y0r, 0, h - 1), y1 = clampi(y1r, 0, h - 1);
    const V3 c00 = px(x0, y0), c01 = px(x0, y1), c10 = px(x1, y0), c11 = px(x1, y1);
    const V3 c0 = c00 * (1.0f - yt) + c01 * yt, c1 = c10 * (1.0f - yt) + c11 * yt;
    return c0 * (1.0f - xt) + c1 * xt;
}

float srgb_to_linear(float s) { return s <= 0.04045f ? s / 12.92f : spt_pow((s + 0.055f) / 1.055f, 2.4f); }   // srgb_tex.rs:55-61

V3 eval(const HostScene& hs, uint32_t node, TexIn in) {   // Texture::color_at for a `specified` input (zero differentials)
    const spt_texture& t = hs.textures[node];
    switch (t.type) {
    case SPT_TEX_SCALAR: return {t.value[0], t.value[1], t.value[2]};
    case SPT_TEX_IMAGE: {
        float x = 0.0f, y = 0.0f;
        if (in.mode == SPT_TEXMODE_SPECIFIED) { x = in.spec[0]; y = in.spec[1]; }
        else if (in.mode == SPT_TEXMODE_TEXCOORDS) { x = in.uv[0]; y = in.uv[1]; }
        return bilinear0(hs, t.image, wrap1(x, in.wrap), wrap1(y, in.wrap));
    }
    case SPT_TEX_SRGB: { const V3 c = eval(hs, t.a, in); return {srgb_to_linear(c.x), srgb_to_linear(c.y), srgb_to_linear(c.z)}; }
    case SPT_TEX_MODIFIER: {   // TexInputModifier::apply_modifier (input_modifier.rs:33-47)
        for (int k = 0; k < 3; ++k) in.spec[k] = in.spec[k] * t.tiling[k] + t.offset[k];
        for (int k = 0; k < 2; ++k) in.uv[k] = in.uv[k] * t.tiling[k] + t.offset[k];
        if (t.mode >= 0) in.mode = t.mode;
        if (t.wrap >= 0) in.wrap = t.wrap;
        return eval(hs, t.a, in);
    }
    case SPT_TEX_ADD: return eval(hs, t.a, in) + eval(hs, t.b, in);
    case SPT_TEX_SUB: return eval(hs, t.a, in) - eval(hs, t.b, in);
    case SPT_TEX_MUL: return eval(hs, t.a, in) * eval(hs, t.b, in);
    default: { const V3 a = eval(hs, t.a, in), b = eval(hs, t.b, in); return {a.x / b.x, a.y / b.y, a.z / b.z}; }
    }
}
// TextureT::{dimensions, tiling, offset} (src/texture/mod.rs:185-195 and the overrides of image_tex / srgb_tex / input_modifier)
bool dimensions(const HostScene& hs, uint32_t node, uint32_t* w, uint32_t* h) {
    const spt_texture& t = hs.textures[node];
    if (t.type == SPT_TEX_IMAGE) {
        const spt_image_level& L = hs.image_levels[hs.images[t.image].first_level];
        *w = L.width; *h = L.height;
        return true;
    }
    if (t.type == SPT_TEX_SRGB || t.type == SPT_TEX_MODIFIER) return dimensions(hs, t.a, w, h);
    return false;
}
V3 tiling(const HostScene& hs, uint32_t node) {
    const spt_texture& t = hs.textures[node];
    if (t.type == SPT_TEX_SRGB) return tiling(hs, t.a);
    if (t.type == SPT_TEX_MODIFIER) return tiling(hs, t.a) * V3{t.tiling[0], t.tiling[1], t.tiling[2]};
    return {1, 1, 1};
}
V3 offset(const HostScene& hs, uint32_t node) {
    const spt_texture& t = hs.textures[node];
    if (t.type == SPT_TEX_SRGB) return offset(hs, t.a);
    if (t.type == SPT_TEX_MODIFIER) return tiling(hs, t.a) * V3{t.offset[0], t.offset[1], t.offset[2]} + offset(hs, t.a);
    return {0, 0, 0};
}

struct M2 { float c0x, c0y, c1x, c1y; };   // glam Mat2: x_axis (c0), y_axis (c1)
M2 mul(const M2& a, const M2& b) {        // glam Mat2 * Mat2: columns a * b.col
    return {a.c0x * b.c0x + a.c1x * b.c0y, a.c0y * b.c0x + a.c1y * b.c0y, a.c0x * b.c1x + a.c1x * b.c1y, a.c0y * b.c1x + a.c1y * b.c1y};
}
M2 scale(float s, const M2& m) { return {s * m.c0x, s * m.c0y, s * m.c1x, s * m.c1y}; }
M2 add(const M2& a, const M2& b) { return {a.c0x + b.c0x, a.c0y + b.c0y, a.c1x + b.c1x, a.c1y + b.c1y}; }
M2 sub(const M2& a, const M2& b) { return {a.c0x - b.c0x, a.c0y - b.c0y, a.c1x - b.c1x, a.c1y - b.c1y}; }
M2 transpose(const M2& m) { return {m.c0x, m.c1x, m.c0y, m.c1y}; }
M2 inverse(const M2& m) {                 // glam Mat2::inverse: adjugate * (1 / det)
    const float det = m.c0x * m.c1y - m.c1x * m.c0y;
    const float inv = 1.0f / det;
    return {m.c1y * inv, m.c0y * -inv, m.c1x * -inv, m.c0x * inv};
}

}  // namespace

// Builds the terms and the two kinds of trees of one material and appends them to the scene; returns the index into hs.pndfs.
uint32_t build_pndf(HostScene& hs, uint32_t base_normal, float sigma_r, float h, const std::string& label) {
    uint32_t nw = 0, nh = 0;
    if (!dimensions(hs, base_normal, &nw, &nh))
        throw HostError(SPT_HOST_ERR_SCHEMA, label + " - 'base_normal' should be a Texture with non-None dimensions");
    if (!(h > 0.0f) || !(sigma_r > 0.0f)) throw HostError(SPT_HOST_ERR_SCHEMA, label + " - 'h' and 'sigma_r' must be positive");
    const float h_inv = 1.0f / h;
    const uint32_t ny = (uint32_t)((float)nh * h_inv), nx = (uint32_t)((float)nw * h_inv);
    if (nx == 0 || ny == 0 || (uint64_t)nx * ny > (1u << 24)) throw HostError(SPT_HOST_ERR_UNSUPPORTED, label + " - 0 or more than 2^24 P-NDF terms");
    const V3 til3 = tiling(hs, base_normal), off3 = offset(hs, base_normal);
    const float tx = til3.x, ty = til3.y, ox = off3.x, oy = off3.y;
    const float hx_inv = (float)nx, hx = 1.0f / hx_inv, hy_inv = (float)ny, hy = 1.0f / hy_inv;
    const float k = std::sqrt(8.0f * std::log(2.0f));
    const float sigma_hx = hx / k, sigma_hy = hy / k;
    auto normal_at = [&](float u, float v, float* sx, float* sy) {   // get_normal_bilinear
        TexIn in;
        in.spec[0] = (u - ox) / tx;
        in.spec[1] = (v - oy) / ty;
        const V3 c = eval(hs, base_normal, in) * 2.0f - V3{1, 1, 1};
        const V3 n = normalize(c);
        *sx = n.x; *sy = n.y;
    };
    spt_pndf pd;
    std::memset(&pd, 0, sizeof pd);
    pd.first_term = (uint32_t)hs.pndf_terms.size();
    pd.n_terms = nx * ny;
    pd.sigma_r = sigma_r; pd.sigma_hx = sigma_hx; pd.sigma_hy = sigma_hy;
    pd.tiling[0] = tx; pd.tiling[1] = ty; pd.offset[0] = ox; pd.offset[1] = oy;
    const float sigma_h_sqr_inv = 1.0f / (sigma_hx * sigma_hy), sigma_r_sqr_inv = 1.0f / (sigma_r * sigma_r);
    const M2 ident{1, 0, 0, 1};
    for (uint32_t i = 0; i < ny; ++i)
        for (uint32_t j = 0; j < nx; ++j) {
            const float u = ((float)j + 0.5f) * hx, v = ((float)i + 0.5f) * hy;
            float s[2], up[2], un[2], vp[2], vn[2];
            normal_at(u, v, &s[0], &s[1]);
            normal_at(u + 0.5f * hx, v, &up[0], &up[1]);
            normal_at(u - 0.5f * hx, v, &un[0], &un[1]);
            normal_at(u, v + 0.5f * hy, &vp[0], &vp[1]);
            normal_at(u, v - 0.5f * hy, &vn[0], &vn[1]);
            const M2 jac{(up[0] - un[0]) * hx_inv, (up[1] - un[1]) * hx_inv, (vp[0] - vn[0]) * hy_inv, (vp[1] - vn[1]) * hy_inv};
            // PndfGaussTerm::new (pndf_bvh.rs:405-437)
            const M2 jt = transpose(jac);
            const M2 mat_a = add(scale(sigma_h_sqr_inv, ident), mul(scale(sigma_r_sqr_inv, jt), jac));
            const M2 mat_a_inv = inverse(mat_a);
            const M2 mat_b = scale(sigma_r_sqr_inv, jt), mat_b_t = scale(sigma_r_sqr_inv, jac);
            const M2 mat_mu = mul(mat_a_inv, mat_b);
            const M2 mat_s = sub(scale(sigma_r_sqr_inv, ident), mul(mul(mat_b_t, mat_a_inv), mat_b));
            spt_pndf_term t;
            t.u[0] = u; t.u[1] = v; t.s[0] = s[0]; t.s[1] = s[1];
            const M2* ms[4] = {&jac, &mat_a, &mat_s, &mat_mu};
            float* dst[4] = {t.jacobian, t.mat_a, t.mat_s, t.mat_mu};
            for (int q = 0; q < 4; ++q) { dst[q][0] = ms[q]->c0x; dst[q][1] = ms[q]->c0y; dst[q][2] = ms[q]->c1x; dst[q][3] = ms[q]->c1y; }
            hs.pndf_terms.push_back(t);
        }
    // one tree over a list of term indices: the range is halved in list order (no sorting), leaves hold < max_leaf terms
    const uint32_t max_leaf = 5;
    auto build_tree = [&](const std::vector<uint32_t>& list, int dims, uint32_t* first_ref) -> uint32_t {
        *first_ref = (uint32_t)hs.pndf_refs.size();
        hs.pndf_refs.insert(hs.pndf_refs.end(), list.begin(), list.end());
        if (list.empty()) return 0xffffffffu;
        auto coords = [&](uint32_t ti, float* c) {
            const spt_pndf_term& t = hs.pndf_terms[ti];
            c[0] = t.u[0]; c[1] = t.u[1]; c[2] = dims == 4 ? t.s[0] : 0.0f; c[3] = dims == 4 ? t.s[1] : 0.0f;
        };
        auto make = [&](uint32_t start, uint32_t end) -> uint32_t {
            spt_pndf_node nd;
            float c[4];
            coords(list[start], c);
            for (int q = 0; q < 4; ++q) { nd.bmin[q] = c[q]; nd.bmax[q] = c[q]; }
            for (uint32_t i = start; i < end; ++i) {
                coords(list[i], c);
                for (int q = 0; q < 4; ++q) { nd.bmin[q] = std::fmin(nd.bmin[q], c[q]); nd.bmax[q] = std::fmax(nd.bmax[q], c[q]); }
            }
            nd.start = start; nd.end = end; nd.lc = 0xffffffffu; nd.rc = 0xffffffffu;
            hs.pndf_nodes.push_back(nd);
            return (uint32_t)hs.pndf_nodes.size() - 1;
        };
        const uint32_t root = make(0, (uint32_t)list.size());
        std::vector<uint32_t> stack{root};
        while (!stack.empty()) {
            const uint32_t ni = stack.back();
            stack.pop_back();
            const uint32_t start = hs.pndf_nodes[ni].start, end = hs.pndf_nodes[ni].end;
            if (end - start < max_leaf) continue;
            const uint32_t mid = start + (end - start) / 2;
            const uint32_t lc = make(start, mid), rc = make(mid, end);
            hs.pndf_nodes[ni].lc = lc;
            hs.pndf_nodes[ni].rc = rc;
            stack.push_back(lc);
            stack.push_back(rc);
        }
        return root;
    };
    // PndfAccel::new: terms binned by s into s_block_count^2 blocks
    const uint32_t sbc = (uint32_t)std::fmin(std::fmax((float)(size_t)(2.0f / (sigma_r * 16.0f)), 1.0f), 20.0f);
    pd.s_block_count = sbc;
    std::vector<std::vector<uint32_t>> split((size_t)sbc * sbc);
    for (uint32_t ti = 0; ti < pd.n_terms; ++ti) {
        const spt_pndf_term& t = hs.pndf_terms[pd.first_term + ti];
        const float sxt = (t.s[0] + 1.0f) * 0.5f, syt = (t.s[1] + 1.0f) * 0.5f;
        const uint32_t x = (uint32_t)std::fmin((float)(size_t)std::fmax(sxt * (float)sbc, 0.0f), (float)(sbc - 1));
        const uint32_t y = (uint32_t)std::fmin((float)(size_t)std::fmax(syt * (float)sbc, 0.0f), (float)(sbc - 1));
        split[(size_t)x * sbc + y].push_back(pd.first_term + ti);
    }
    pd.first_root = (uint32_t)hs.pndf_roots.size();
    for (const auto& list : split) {
        uint32_t first_ref = 0;
        const uint32_t root = build_tree(list, 4, &first_ref);
        hs.pndf_roots.push_back(root);
        hs.pndf_roots.push_back(first_ref);
    }
    std::vector<uint32_t> all(pd.n_terms);
    for (uint32_t ti = 0; ti < pd.n_terms; ++ti) all[ti] = pd.first_term + ti;
    pd.uv_root = build_tree(all, 2, &pd.uv_first_ref);
    hs.pndfs.push_back(pd);
    return (uint32_t)hs.pndfs.size() - 1;
}

}  // namespace spt_host
