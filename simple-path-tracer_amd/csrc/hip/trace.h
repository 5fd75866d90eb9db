// Two-level BVH traversal and triangle / sphere intersection for gfx950.
//
// Replaces, per ray:
//   BvhAccel::{intersect, intersect_test}   reference src/primitive/bvh.rs:237-283
//   Group::{intersect, intersect_test}      src/primitive/group.rs:24-40
//   Instance::{intersect, intersect_test}   src/primitive/instance.rs:88-109
//   Triangle::intersect_ray                 src/primitive/triangle.rs:124-147
//   Sphere::intersect_ray / accept rules    src/primitive/sphere.rs:25-39,50-84
//   Bbox::intersect_test                    src/core/bbox.rs:63-93
//
// Design for the hardware: one ray per lane; nodes are 32 B and triangles 48 B so
// every fetch is one or a few 16-byte vector loads; the traversal stack lives in
// LDS, laid out [level][lane] so a wave's pushes and pops hit 64 distinct banks;
// the per-ray Vec allocation of the reference (bvh.rs:243,267) is gone.  Visit order
// is the reference's (push left, push right, pop -> right subtree first) so that the
// closest hit — including which of two equal-t candidates wins — is bit-identical to
// the CPU oracle.  The slab test multiplies by a per-ray reciprocal instead of the
// reference's six divisions per node; it only culls, the returned hit is unaffected.
#pragma once
#include "device_math.h"

struct DScene {
    const float4* tlas_nodes;  // 2 x float4 per node: (bmin, a) (bmax, b)
    const float4* blas_nodes;
    const float4* tri_pos;     // 3 x float4 per triangle
    const float4* tri_attr;    // 9 x float4 per triangle (spt_tri_attr)
    const float4* instances;   // 12 x float4 per instance (spt_instance)
    const uint4* meshes;       // root, node_count, tri_first, tri_count
    const float4* spheres;     // center, radius
    const spt_surface* surfaces;
    const spt_material* materials;
    const spt_medium* mediums;
    const spt_light* lights;
    const float* light_props; const float* light_u; const uint32_t* light_k;
    const float* env_texels; const float* env_props; const float* env_u; const uint32_t* env_k;
    uint32_t n_tlas_nodes, n_instances, n_lights, n_meshes;
    uint32_t aggregate, light_sampler;
    int32_t env_light_index;
    uint32_t env_w, env_h;
    float env_scale[3];
    uint32_t stack_cap;        // LDS stack entries per lane the scene needs
    // Traversal geometry as ONE float4 blob [tlas | instances | meshes | blas | tri_pos | spheres]
    // (offsets in float4 units).  Small scenes are staged into LDS once per workgroup
    // (k_*<true>) and every node / triangle / instance fetch becomes a ds_read_b128.
    const float4* geo;
    uint32_t geo_f4;           // blob length in float4
    uint32_t o_tlas, o_inst, o_mesh, o_blas, o_tri, o_sph;
};

struct DHit {
    float t;
    int32_t inst, prim;
    float v, w;
};

// per-lane stack in LDS: entry `level` of this lane
extern __shared__ uint32_t spt_lds_stack[];
SPT_DEV uint32_t& stack_at(uint32_t level) { return spt_lds_stack[level * blockDim.x + threadIdx.x]; }

// geometry fetch: LDS copy (kLds) or the global blob
SPT_DEV float4* geo_lds(const DScene& sc) { return reinterpret_cast<float4*>(spt_lds_stack + sc.stack_cap * blockDim.x); }
template <bool kLds>
SPT_DEV float4 geo_ld(const DScene& sc, uint32_t off) {
    if (kLds) return geo_lds(sc)[off];
    return sc.geo[off];
}
// once per workgroup, before any traversal
template <bool kLds>
SPT_DEV void stage_geometry(const DScene& sc) {
    if (kLds) {
        float4* dst = geo_lds(sc);
        for (uint32_t i = threadIdx.x; i < sc.geo_f4; i += blockDim.x) dst[i] = sc.geo[i];
        __syncthreads();
    }
}

SPT_DEV f3 recip3(f3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

// Bbox::intersect_test (bbox.rs:63-93) with o, 1/d
SPT_DEV bool slab_test(float4 lo, float4 hi, f3 o, f3 inv_d, float t_min, float t_max) {
    bool empty = (lo.x > hi.x) | (lo.y > hi.y) | (lo.z > hi.z);
    float x0 = (lo.x - o.x) * inv_d.x, x1 = (hi.x - o.x) * inv_d.x;
    float y0 = (lo.y - o.y) * inv_d.y, y1 = (hi.y - o.y) * inv_d.y;
    float z0 = (lo.z - o.z) * inv_d.z, z1 = (hi.z - o.z) * inv_d.z;
    float xa = spt_min(x0, x1), xb = spt_max(x0, x1);
    float ya = spt_min(y0, y1), yb = spt_max(y0, y1);
    float za = spt_min(z0, z1), zb = spt_max(z0, z1);
    float t0 = spt_max(xa, spt_max(ya, za));
    float t1 = spt_min(xb, spt_min(yb, zb));
    return !empty & (t0 <= t1) & (t1 > t_min) & (t0 < t_max);
}

// Triangle::intersect_ray (triangle.rs:124-147), branch-free: same values, the
// nested ifs become one predicate
SPT_DEV bool tri_test3(float4 a, float4 b, float4 c, const DRay& r, float* t, float* v_out, float* w_out) {
    f3 p0 = mk3(a), p1 = mk3(b), p2 = mk3(c);
    f3 e1 = p1 - p0;
    f3 e2 = p2 - p0;
    f3 q = cross(r.d, e2);
    float det = dot(e1, q);
    float inv = 1.0f / det;
    f3 s = r.o - p0;
    float v = dot(s, q) * inv;
    f3 rr = cross(s, e1);
    float w = dot(r.d, rr) * inv;
    float u = 1.0f - v - w;
    *t = dot(e2, rr) * inv;
    *v_out = v;
    *w_out = w;
    return (det != 0.0f) & (v >= 0.0f) & (w >= 0.0f) & (u >= 0.0f);
}

SPT_DEV bool tri_test(const float4* tri_pos, uint32_t tri, const DRay& r, float* t, float* v_out, float* w_out) {
    return tri_test3(tri_pos[3 * tri], tri_pos[3 * tri + 1], tri_pos[3 * tri + 2], r, t, v_out, w_out);
}
template <bool kLds>
SPT_DEV bool tri_test_geo(const DScene& sc, uint32_t tri, const DRay& r, float* t, float* v_out, float* w_out) {
    const uint32_t o = sc.o_tri + 3u * tri;
    return tri_test3(geo_ld<kLds>(sc, o), geo_ld<kLds>(sc, o + 1), geo_ld<kLds>(sc, o + 2), r, t, v_out, w_out);
}

// Sphere::intersect_ray (sphere.rs:25-39)
SPT_DEV bool sphere_roots(float4 s, const DRay& r, float* mn, float* mx) {
    f3 oc = r.o - mk3(s);
    float a = dot(r.d, r.d);
    float b = dot(r.d, oc);
    float c = dot(oc, oc) - s.w * s.w;
    float delta = b * b - a * c;
    float sq = spt_sqrt(delta);
    *mn = (-b - sq) / a;
    *mx = (-b + sq) / a;
    return delta >= 0.0f;
}

template <bool kLds>
SPT_DEV DRay to_object(const DScene& sc, uint32_t inst, const DRay& r, uint32_t* prim_type, uint32_t* prim_id) {
    const uint32_t I = sc.o_inst + 12u * inst;
    float4 m0 = geo_ld<kLds>(sc, I), m1 = geo_ld<kLds>(sc, I + 1), m2 = geo_ld<kLds>(sc, I + 2), k = geo_ld<kLds>(sc, I + 8);
    float inv[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
    *prim_type = __float_as_uint(k.y);
    *prim_id = __float_as_uint(k.z);
    DRay o;
    o.o = xf_point(inv, r.o);
    o.d = xf_vector(inv, r.d);  // not renormalised: t is shared between the spaces (ray.rs:33-41)
    o.t_min = r.t_min;
    return o;
}

// Closest hit of the scene aggregate.  `base` = first free stack level.
template <bool kLds>
SPT_DEV DHit trace_closest(const DScene& sc, const DRay& ray, float t_max) {
    DHit h;
    h.t = t_max;
    h.inst = -1;
    h.prim = -1;
    h.v = 0.0f;
    h.w = 0.0f;
    const bool group = (sc.aggregate == SPT_AGGREGATE_GROUP);
    f3 inv_w = recip3(ray.d);
    uint32_t sp = 0;
    uint32_t next_inst = 0, end_inst = 0;  // instance range of the current TLAS leaf
    if (group) {
        end_inst = sc.n_instances;
    } else if (sc.n_tlas_nodes > 0) {
        stack_at(sp++) = 0u;
    }
    while (true) {
        if (next_inst == end_inst) {
            // pop TLAS nodes until a leaf is entered
            if (sp == 0) break;
            uint32_t ni = stack_at(--sp);
            float4 lo = geo_ld<kLds>(sc, sc.o_tlas + 2u * ni), hi = geo_ld<kLds>(sc, sc.o_tlas + 2u * ni + 1u);
            if (!slab_test(lo, hi, ray.o, inv_w, ray.t_min, h.t)) continue;
            uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);
            if (b & SPT_LEAF_FLAG) {
                next_inst = a;
                end_inst = a + (b & ~SPT_LEAF_FLAG);
            } else if (sp + 2 <= sc.stack_cap) {
                stack_at(sp++) = a;
                stack_at(sp++) = b;
            }
            continue;
        }
        uint32_t inst = next_inst++;
        uint32_t prim_type, prim_id;
        DRay orr = to_object<kLds>(sc, inst, ray, &prim_type, &prim_id);
        if (prim_type == SPT_PRIM_SPHERE) {
            float mn, mx;
            if (sphere_roots(geo_ld<kLds>(sc, sc.o_sph + prim_id), orr, &mn, &mx)) {
                float t = (mn < orr.t_min) ? mx : mn;  // sphere.rs:61
                if (orr.t_min < t && t < h.t) {
                    h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = 0.0f; h.w = 0.0f;
                }
            }
        } else {
            const uint32_t mesh_root = __float_as_uint(geo_ld<kLds>(sc, sc.o_mesh + prim_id).x);
            f3 inv_o = recip3(orr.d);
            const uint32_t base = sp;
            stack_at(sp++) = mesh_root;
            while (sp > base) {
                uint32_t ni = stack_at(--sp);
                float4 lo = geo_ld<kLds>(sc, sc.o_blas + 2u * ni), hi = geo_ld<kLds>(sc, sc.o_blas + 2u * ni + 1u);
                if (!slab_test(lo, hi, orr.o, inv_o, orr.t_min, h.t)) continue;
                uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);
                if (b & SPT_LEAF_FLAG) {
                    uint32_t n = b & ~SPT_LEAF_FLAG;
                    for (uint32_t i = a; i < a + n; ++i) {
                        float t, v, w;
                        bool ok = tri_test_geo<kLds>(sc, i, orr, &t, &v, &w);
                        if (ok && t > orr.t_min && t < h.t) {  // triangle.rs:187
                            h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)i; h.v = v; h.w = w;
                        }
                    }
                } else if (sp + 2 <= sc.stack_cap) {
                    stack_at(sp++) = a;
                    stack_at(sp++) = b;
                }
            }
        }
    }
    return h;
}

// Any hit in (t_min, t_max): intersect_test of the aggregate
template <bool kLds>
SPT_DEV bool trace_any(const DScene& sc, const DRay& ray, float t_max) {
    const bool group = (sc.aggregate == SPT_AGGREGATE_GROUP);
    f3 inv_w = recip3(ray.d);
    uint32_t sp = 0;
    uint32_t next_inst = 0, end_inst = 0;
    if (group) {
        end_inst = sc.n_instances;
    } else if (sc.n_tlas_nodes > 0) {
        stack_at(sp++) = 0u;
    }
    while (true) {
        if (next_inst == end_inst) {
            if (sp == 0) break;
            uint32_t ni = stack_at(--sp);
            float4 lo = geo_ld<kLds>(sc, sc.o_tlas + 2u * ni), hi = geo_ld<kLds>(sc, sc.o_tlas + 2u * ni + 1u);
            if (!slab_test(lo, hi, ray.o, inv_w, ray.t_min, t_max)) continue;
            uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);
            if (b & SPT_LEAF_FLAG) {
                next_inst = a;
                end_inst = a + (b & ~SPT_LEAF_FLAG);
            } else if (sp + 2 <= sc.stack_cap) {
                stack_at(sp++) = a;
                stack_at(sp++) = b;
            }
            continue;
        }
        uint32_t inst = next_inst++;
        uint32_t prim_type, prim_id;
        DRay orr = to_object<kLds>(sc, inst, ray, &prim_type, &prim_id);
        if (prim_type == SPT_PRIM_SPHERE) {
            float mn, mx;
            if (sphere_roots(geo_ld<kLds>(sc, sc.o_sph + prim_id), orr, &mn, &mx) && mn < t_max && mx > orr.t_min) return true;  // sphere.rs:51-56
        } else {
            const uint32_t mesh_root = __float_as_uint(geo_ld<kLds>(sc, sc.o_mesh + prim_id).x);
            f3 inv_o = recip3(orr.d);
            const uint32_t base = sp;
            stack_at(sp++) = mesh_root;
            while (sp > base) {
                uint32_t ni = stack_at(--sp);
                float4 lo = geo_ld<kLds>(sc, sc.o_blas + 2u * ni), hi = geo_ld<kLds>(sc, sc.o_blas + 2u * ni + 1u);
                if (!slab_test(lo, hi, orr.o, inv_o, orr.t_min, t_max)) continue;
                uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);
                if (b & SPT_LEAF_FLAG) {
                    uint32_t n = b & ~SPT_LEAF_FLAG;
                    for (uint32_t i = a; i < a + n; ++i) {
                        float t, v, w;
                        if (tri_test_geo<kLds>(sc, i, orr, &t, &v, &w) && t > orr.t_min && t < t_max) return true;
                    }
                } else if (sp + 2 <= sc.stack_cap) {
                    stack_at(sp++) = a;
                    stack_at(sp++) = b;
                }
            }
        }
    }
    return false;
}
