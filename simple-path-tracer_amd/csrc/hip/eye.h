// Closest hit of a PRIMARY ray through an eye-relative copy of the LDS-resident geometry.
//
// Same functions of the reference as trace.h (BvhAccel::intersect bvh.rs:262-283, Instance::intersect instance.rs:88-109,
// Triangle::intersect_ray triangle.rs:124-147, Sphere sphere.rs:25-39,58-84, Bbox::intersect_test bbox.rs:63-93), same
// arithmetic, same bits.  What changes is WHEN part of it is done.  Every camera ray of a frame starts at the eye, so every term
// that involves the ray origin only is the same number for all of them:
//   * the slab test computes (lo - o) * (1 / d), (hi - o) * (1 / d): lo - o and hi - o are per node and per camera;
//   * the object-space origin o' = M^-1 o of an instance (12 operations per instance visit) is per instance and per camera;
//   * of the 46 operations of a triangle test, s = o' - p0, s x e1 and e2 . (s x e1) - 17 of them - do not involve the direction;
//   * a sphere test's o' - c likewise.
// The host therefore keeps, per camera position, a copy of the geometry blob in which node and root boxes hold lo - o / hi - o
// (world eye for the TLAS, the instance's object-space eye for its BLAS), the instance record of a sphere carries o' - c, and
// triangles (s, e1, e2) plus one more float4 (s x e1, e2 . (s x e1)) - each value produced on the host by the SAME f32 operations in
// the same order the kernels of trace.h perform per ray (the library is built with FP contraction off on both sides), so a
// walker that reads them instead of recomputing them returns the same hit bit for bit.  k_primary<.., kEye> stages that copy in
// LDS instead of the plain one.  About 105 of the ~900 wave-instructions of a hit sample go away.
// Built only when every mesh is used by exactly one instance (its boxes can be relative to one origin only), for the
// library's own trees, and while the copy fits the LDS budget; SPT_NO_EYE_BLOB=1 switches it off.
#pragma once
#include "trace.h"

// Bbox::intersect_test with the box already relative to the ray origin
SPT_DEV bool slab_t0_rel(float4 lo, float4 hi, f3 inv_d, float t_min, float* t0_out) {
    bool empty = (lo.x > hi.x) | (lo.y > hi.y) | (lo.z > hi.z);
    float x0 = lo.x * inv_d.x, x1 = hi.x * inv_d.x;
    float y0 = lo.y * inv_d.y, y1 = hi.y * inv_d.y;
    float z0 = lo.z * inv_d.z, z1 = hi.z * inv_d.z;
    float xa = spt_min(x0, x1), xb = spt_max(x0, x1);
    float ya = spt_min(y0, y1), yb = spt_max(y0, y1);
    float za = spt_min(z0, z1), zb = spt_max(z0, z1);
    float t0 = spt_max(xa, spt_max(ya, za));
    float t1 = spt_min(xb, spt_min(yb, zb));
    *t0_out = t0;
    return !empty & (t0 <= t1) & (t1 > t_min);
}
SPT_DEV bool root_hit_rel(float4 lo, float4 hi, f3 inv_d, float t_min, float limit) {
    float t0;
    return slab_t0_rel(lo, hi, inv_d, t_min, &t0) && t0 <= limit;
}

// tri_test_edges with s, s x e1 and e2 . (s x e1) read instead of computed: a = (s, id), b = e1, c = e2, x = (s x e1, e2 . (s x e1))
SPT_DEV bool tri_test_eye(float4 a, float4 b, float4 c, float4 x, f3 d, float* t, float* v_out, float* w_out) {
    f3 s = mk3(a), e1 = mk3(b), e2 = mk3(c), rr = mk3(x);
    f3 q = cross(d, e2);
    float det = dot(e1, q);
    float inv = 1.0f / det;
    float v = dot(s, q) * inv;
    float w = dot(d, rr) * inv;
    float u = 1.0f - v - w;
    *t = x.w * inv;
    *v_out = v;
    *w_out = w;
    return (det != 0.0f) & (v >= 0.0f) & (w >= 0.0f) & (u >= 0.0f);
}

// walk_tree<true, true, false> (two-child nodes, near-first, closest hit) over eye-relative boxes
template <class LeafFn>
SPT_DEV void eye_walk_tree(const DScene& sc, uint32_t nodes_off, uint32_t root, f3 inv_d, float t_min, const float& limit, TStack& st, LeafFn leaf) {
    const uint32_t base = st.sp;
    uint32_t cur = root;
    while (true) {
        if (cur & kLeaf) {
            leaf(leaf_first(cur), leaf_count(cur));
        } else {
            const uint32_t n = nodes_off + 4u * cur;
            float4 a = geo_ld<true>(sc, n), b = geo_ld<true>(sc, n + 1u), c = geo_ld<true>(sc, n + 2u), d = geo_ld<true>(sc, n + 3u);
            float tl, tr;
            bool hl = slab_t0_rel(a, b, inv_d, t_min, &tl);
            bool hr = slab_t0_rel(c, d, inv_d, t_min, &tr);
            hl = hl && tl <= limit;
            hr = hr && tr <= limit;
            const uint32_t rl = __float_as_uint(a.w), rr = __float_as_uint(b.w);
            if (hl && hr) {
                const bool left_first = tl <= tr;
                if (st.sp < kLdsStack + kSpillStack) st.push(left_first ? rr : rl, left_first ? tr : tl);
                cur = left_first ? rl : rr;
                continue;
            }
            if (hl) { cur = rl; continue; }
            if (hr) { cur = rr; continue; }
        }
        bool found = false;
        while (st.sp > base) {
            uint32_t ref;
            float t0;
            st.pop(&ref, &t0);
            if (t0 <= limit) { cur = ref; found = true; break; }
        }
        if (!found) return;
    }
}

// instance_closest<true> for a ray from the eye with world direction dw
SPT_DEV void eye_instance_closest(const DScene& sc, uint32_t inst, f3 dw, float t_min, DHit& h, TStack& st) {
    const uint32_t I = sc.o_inst + 12u * inst;
    const float4 m0 = geo_ld<true>(sc, I), m1 = geo_ld<true>(sc, I + 1u), m2 = geo_ld<true>(sc, I + 2u), k = geo_ld<true>(sc, I + 8u);
    const float inv[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
    const uint32_t prim_type = __float_as_uint(k.y), prim_id = __float_as_uint(k.z);
    const f3 od = xf_vector(inv, dw);   // not renormalised: t is shared between the spaces (ray.rs:33-41)
    if (prim_type == SPT_PRIM_SPHERE) {   // sphere_roots with oc = o' - c read from the instance record
        const float4 s = geo_ld<true>(sc, sc.o_sph + prim_id);
        const f3 oc = mk3(geo_ld<true>(sc, I + 11u));   // pad[1..3] of the instance record: spheres may share a primitive
        const float a = dot(od, od);
        const float b = dot(od, oc);
        const float c = dot(oc, oc) - s.w * s.w;
        const float delta = b * b - a * c;
        const float sq = spt_sqrt(delta);
        const float mn = (-b - sq) / a, mx = (-b + sq) / a;
        if (delta >= 0.0f) {
            const float t = (mn < t_min) ? mx : mn;  // sphere.rs:61
            if (t_min < t && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, (int32_t)prim_id, h)))) {
                h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = 0.0f; h.w = 0.0f;
            }
        }
        return;
    }
    const float4 rlo = geo_ld<true>(sc, sc.o_mesh + 2u * prim_id), rhi = geo_ld<true>(sc, sc.o_mesh + 2u * prim_id + 1u);
    const uint32_t root = __float_as_uint(rlo.w);
    const f3 inv_o = recip3(sc, od);
    if (!root_hit_rel(rlo, rhi, inv_o, t_min, h.t)) return;
    eye_walk_tree(sc, sc.o_blas, root, inv_o, t_min, h.t, st, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first; i < first + count; ++i) {
            const uint32_t o = sc.o_tri + 3u * i;
            const float4 a = geo_ld<true>(sc, o);
            float t, v, w;
            const bool ok = tri_test_eye(a, geo_ld<true>(sc, o + 1u), geo_ld<true>(sc, o + 2u), geo_ld<true>(sc, sc.o_eye + i), od, &t, &v, &w);
            const int32_t id = __float_as_int(a.w);
            if (ok && t > t_min && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, id, h)))) {  // triangle.rs:187
                h.t = t; h.inst = (int32_t)inst; h.prim = id; h.v = v; h.w = w;
            }
        }
    });
}

// trace_closest<true> for a ray from the eye (DScene::tlas_lo / tlas_hi of the copy handed to the kernel are eye-relative too)
SPT_DEV DHit eye_trace_closest(const DScene& sc, f3 dw, float t_min, float t_max) {
    DHit h;
    h.t = t_max; h.inst = -1; h.prim = -1; h.v = 0.0f; h.w = 0.0f;
    uint2 spill_mem[kSpillStack];
    TStack st;
    st.spill = spill_mem;
    if (sc.aggregate == SPT_AGGREGATE_GROUP) {
        for (uint32_t i = 0; i < sc.n_instances; ++i) eye_instance_closest(sc, i, dw, t_min, h, st);
    } else if (sc.n_tlas_nodes > 0) {
        const f3 inv_w = recip3(sc, dw);
        const float4 tlo = make_float4(sc.tlas_lo[0], sc.tlas_lo[1], sc.tlas_lo[2], 0.0f), thi = make_float4(sc.tlas_hi[0], sc.tlas_hi[1], sc.tlas_hi[2], 0.0f);
        if (root_hit_rel(tlo, thi, inv_w, t_min, h.t))
            eye_walk_tree(sc, sc.o_tlas, sc.tlas_root, inv_w, t_min, h.t, st, [&](uint32_t first, uint32_t count) {
                for (uint32_t i = first; i < first + count; ++i) eye_instance_closest(sc, tlas_instance<true>(sc, i), dw, t_min, h, st);
            });
    }
    return h;
}
