/*
 * spt_pndf.h - evaluation of a position-normal distribution ("glints") over the tables of include/spt_abi.h.
 *
 * One definition of the arithmetic, compiled into the HIP kernels (csrc/hip/shading.h) and into the CPU oracle, like
 * spt_detmath.h: the tree walks add floating-point terms in an order that both sides have to share to give the same
 * bits.  (The oracle's result is pinned independently by tests/test_pndf.py: brute-force numpy sums over all terms.)
 *
 * Replaces:
 *   PndfGaussTerm::calc + integrate_gaussian_multiplication_2d   reference src/bxdf/pndf_bvh.rs:447-466, 515-540
 *   PndfAccel::calc -> PndfBvh::calc                             pndf_bvh.rs:94-110, 192-234
 *   PndfUvBvh::find_terms                                        pndf_bvh.rs:335-377 (as two walks: the sum, then the pick)
 * Visit order is the reference's: a stack, left child pushed before the right one, so the right subtree is walked first.
 */
#ifndef SPT_PNDF_H
#define SPT_PNDF_H

#include "spt_abi.h"
#include "spt_detmath.h"

typedef struct spt_pndf_view {   /* one material's tables inside the scene's arrays */
    const spt_pndf* pd;
    const spt_pndf_term* terms;
    const spt_pndf_node* nodes;
    const uint32_t* refs;
    const uint32_t* roots;
} spt_pndf_view;

#define SPT_PNDF_STACK 32        /* a tree over <= 2^24 terms halves its range per level */

/* glam Mat2 (column-major: m[0], m[1] = x_axis; m[2], m[3] = y_axis) times a vector */
SPT_HD void spt_m2_mul(const float* m, float vx, float vy, float* ox, float* oy) {
    *ox = m[0] * vx + m[2] * vy;
    *oy = m[1] * vx + m[3] * vy;
}

/* integrate_gaussian_multiplication_2d (pndf_bvh.rs:515-540) */
SPT_HD float spt_pndf_integrate(float mu0x, float mu0y, const float* s0, float c0, float mu1x, float mu1y, const float* s1, float c1) {
    const float si[4] = {s0[0] + s1[0], s0[1] + s1[1], s0[2] + s1[2], s0[3] + s1[3]};
    const float det_i = si[0] * si[3] - si[2] * si[1];
    const float inv = 1.0f / det_i;
    const float sg[4] = {si[3] * inv, si[1] * -inv, si[2] * -inv, si[0] * inv};   /* sigma_sqr = sigma_sqr_inv.inverse() */
    float ax, ay, bx, by;
    spt_m2_mul(s0, mu0x, mu0y, &ax, &ay);
    spt_m2_mul(s1, mu1x, mu1y, &bx, &by);
    float mux, muy;
    spt_m2_mul(sg, ax + bx, ay + by, &mux, &muy);
    const float d0x = mux - mu0x, d0y = muy - mu0y;
    float tx, ty;
    spt_m2_mul(s0, d0x, d0y, &tx, &ty);
    const float val0 = c0 * spt_exp(-0.5f * (d0x * tx + d0y * ty));
    const float d1x = mux - mu1x, d1y = muy - mu1y;
    spt_m2_mul(s1, d1x, d1y, &tx, &ty);
    const float val1 = c1 * spt_exp(-0.5f * (d1x * tx + d1y * ty));
    const float c = val0 * val1;
    const float det = sg[0] * sg[3] - sg[2] * sg[1];
    return c * 2.0f * SPT_PI * spt_sqrt(det);
}

/* PndfGaussTerm::calc (pndf_bvh.rs:447-466) */
SPT_HD float spt_pndf_term_calc(const spt_pndf_term* t, float sigma_p, float term_coe, float ux, float uy, float sx, float sy) {
    const float sigma_p_sqr = sigma_p * sigma_p;
    const float sigma_p_sqr_inv = 1.0f / sigma_p_sqr;
    const float dsx = sx - t->s[0], dsy = sy - t->s[1];
    float mux, muy, qx, qy;
    spt_m2_mul(t->mat_mu, dsx, dsy, &mux, &muy);
    const float c0 = 0.5f * sigma_p_sqr_inv * SPT_FRAC_1_PI;
    spt_m2_mul(t->mat_s, dsx, dsy, &qx, &qy);
    const float c1 = term_coe * spt_exp(-0.5f * (dsx * qx + dsy * qy));
    const float ident[4] = {sigma_p_sqr_inv * 1.0f, sigma_p_sqr_inv * 0.0f, sigma_p_sqr_inv * 0.0f, sigma_p_sqr_inv * 1.0f};
    return spt_pndf_integrate(ux, uy, ident, c0, t->u[0] + mux, t->u[1] + muy, t->mat_a, c1);
}

SPT_HD float spt_pndf_box_dist(const spt_pndf_node* n, int k, float p) {   /* Tuple4fBbox::dist_to_point, one coordinate */
    return spt_max(spt_max(p - n->bmax[k], n->bmin[k] - p), 0.0f);
}

/* PndfAccel::calc (pndf_bvh.rs:94-110, 192-234): the density of the half-vector's (x, y) = s at texture position u */
SPT_HD float spt_pndf_calc(const spt_pndf_view* v, float sigma_p, float term_coe, float ux, float uy, float sx, float sy) {
    const spt_pndf* pd = v->pd;
    const float sbc = (float)pd->s_block_count;
    const float sxt = (sx + 1.0f) * 0.5f, syt = (sy + 1.0f) * 0.5f;
    /* `as usize` saturates: negative and NaN -> 0 */
    float fx = sxt * sbc, fy = syt * sbc;
    uint32_t bx = (fx > 0.0f) ? (fx >= 4294967040.0f ? 0xffffffffu : (uint32_t)fx) : 0u;
    uint32_t by = (fy > 0.0f) ? (fy >= 4294967040.0f ? 0xffffffffu : (uint32_t)fy) : 0u;
    if (bx > pd->s_block_count - 1u) bx = pd->s_block_count - 1u;
    if (by > pd->s_block_count - 1u) by = pd->s_block_count - 1u;
    const uint32_t* rr = v->roots + pd->first_root + 2u * (bx * pd->s_block_count + by);
    const uint32_t root = rr[0], first_ref = rr[1];
    if (root == 0xffffffffu) return 0.0f;
    const float lim_u = 3.0f * (pd->sigma_hx + sigma_p), lim_v = 3.0f * (pd->sigma_hy + sigma_p), lim_s = 3.0f * pd->sigma_r;
    float value = 0.0f;
    uint32_t stack[SPT_PNDF_STACK];
    uint32_t sp = 0;
    stack[sp++] = root;
    while (sp > 0) {
        const spt_pndf_node* n = v->nodes + stack[--sp];
        if (spt_pndf_box_dist(n, 0, ux) > lim_u || spt_pndf_box_dist(n, 1, uy) > lim_v || spt_pndf_box_dist(n, 2, sx) > lim_s ||
            spt_pndf_box_dist(n, 3, sy) > lim_s)
            continue;
        if (n->lc == 0xffffffffu) {
            for (uint32_t i = n->start; i < n->end; ++i) {
                const float res = spt_pndf_term_calc(v->terms + v->refs[first_ref + i], sigma_p, term_coe, ux, uy, sx, sy);
                if (spt_is_finite(res)) value += res;
            }
        } else if (sp + 2 <= SPT_PNDF_STACK) {
            stack[sp++] = n->lc;
            stack[sp++] = n->rc;
        }
    }
    return value;
}

/* PndfUvBvh::find_terms (pndf_bvh.rs:335-377) without the list: the walk calls `visit(term, value)` for every term within
 * reach, in the order the reference pushes them.  Two uses: the sum of the values, and the pick of sample_half. */
SPT_HD float spt_pndf_uv_value(const spt_pndf* pd, const spt_pndf_term* t, float ux, float uy, float sigma_p) {
    const float sigma_h_sqr = pd->sigma_hx * pd->sigma_hy, sigma_p_sqr = sigma_p * sigma_p;
    const float inv = 1.0f / (sigma_h_sqr + sigma_p_sqr);
    const float coe = sigma_h_sqr * inv;
    const float dx = ux - t->u[0], dy = uy - t->u[1];
    return spt_exp(-(dx * dx + dy * dy) * inv * 0.5f) * coe;
}
/* mode 0: returns the sum of all values.  mode 1: walks the list with rand -= value * sum_inv and returns (as a float-cast
 * index bit pattern through *picked) the first term at which rand <= 0, or the last term of the list. */
SPT_HD float spt_pndf_uv_walk(const spt_pndf_view* v, float ux, float uy, float sigma_p, int mode, float sum_inv, float rand, uint32_t* picked) {
    const spt_pndf* pd = v->pd;
    const float lim_u = 3.0f * (pd->sigma_hx + sigma_p), lim_v = 3.0f * (pd->sigma_hy + sigma_p);
    float sum = 0.0f;
    uint32_t last = 0xffffffffu;
    uint32_t stack[SPT_PNDF_STACK];
    uint32_t sp = 0;
    if (pd->uv_root != 0xffffffffu) stack[sp++] = pd->uv_root;
    while (sp > 0) {
        const spt_pndf_node* n = v->nodes + stack[--sp];
        if (spt_pndf_box_dist(n, 0, ux) > lim_u || spt_pndf_box_dist(n, 1, uy) > lim_v) continue;
        if (n->lc == 0xffffffffu) {
            for (uint32_t i = n->start; i < n->end; ++i) {
                const uint32_t ti = v->refs[pd->uv_first_ref + i];
                const float value = spt_pndf_uv_value(pd, v->terms + ti, ux, uy, sigma_p);
                if (mode == 0) {
                    sum += value;
                } else {
                    last = ti;
                    rand -= value * sum_inv;
                    if (rand <= 0.0f) { *picked = ti; return 0.0f; }
                }
            }
        } else if (sp + 2 <= SPT_PNDF_STACK) {
            stack[sp++] = n->lc;
            stack[sp++] = n->rc;
        }
    }
    if (mode != 0) *picked = last;
    return sum;
}

/* PndfMicrofacet::new (src/bxdf/microfacet.rs:67-95): the coefficient of every term from 1 / (sum of the footprint's values) */
SPT_HD float spt_pndf_term_coe(const spt_pndf* pd, float sum_inv) { return sum_inv / (2.0f * SPT_PI * pd->sigma_r * pd->sigma_r); }

/* wrap_uv (src/material/pndf_conductor.rs:212-224) */
SPT_HD float spt_pndf_wrap(float x) { return x >= 0.0f ? spt_fract(x) : 1.0f + spt_fract(x); }

#endif /* SPT_PNDF_H */
