// Exhaustive traversal for scenes of a handful of primitives (LDS-resident geometry only).
//
// Replaces, per ray, the same reference functions as trace.h (BvhAccel::{intersect, intersect_test} bvh.rs:237-283,
// Group::{intersect, intersect_test} group.rs:24-40, Instance::intersect instance.rs:88-109, Triangle::intersect_ray
// triangle.rs:124-147, Sphere sphere.rs:25-84) with the same hit definition: the minimum over (t, instance, prim) of every
// primitive test that accepts, resp. "any accepted test inside (t_min, t_max)" - here LITERALLY so: no tree, no stack.
//
// Why.  Most of the reference's own scenes, BASELINE configs[3] among them, are a few spheres, a cube and a floor: 4 - 8
// instances, 14 - 30 triangles.  Incoherent rays (bounce >= 1, shadow rays towards an environment map) through the
// two-level tree of such a scene run with 31 % of the lanes active (cfg4, profiles/r03_pmc_cfg4/: each lane is at another
// instance, another tree level, another leaf) in a kernel that is bound by instruction issue.  Here the control flow is
// wave-uniform throughout:
//   * the instance loop is uniform; a sphere, or a mesh of fewer than kFlatTransposeMin triangles, is tested by every lane
//     (the primitive records are LDS broadcasts: one address per wave);
//   * a larger mesh (the 12-triangle cube) gets its root-box test first.  No lane passes: next instance.  Many pass: every
//     lane tests every triangle.  FEW pass (the usual case: a cube is hit by a tenth of the rays of a wave): the test is
//     TRANSPOSED - the lanes that passed park their object-space rays in the wave's (otherwise unused) stack words of LDS,
//     then all 64 lanes work as floor(64 / T) groups of T lanes, group g testing parked ray g, lane j of it triangle j.
//     The group's closest hit is one ds_min_u64 on (t bits << 32 | triangle id) - for t > 0 the bits of a float order like
//     the float, and the low word breaks ties by the smaller id, the reference's rule (triangle.rs:187 via
//     ORACLE_TIE_MIN_ID) - and the owner lane merges it into its hit with the usual comparison.  5 rays x 12 triangles
//     per pass instead of 12 triangle tests for 64 lanes of which 6 wanted them.
// The result is the same minimum over the same accepted tests as the loops, so it is bit-identical to them and to the
// tree walk (the root boxes are the padded ones of the library's own trees: they only cull).
// The host turns the mode on when the whole scene costs at most kFlatBudget triangle tests per ray (spt_hip.hip, `flat`).
// Every function here must be called by ALL 64 lanes of a wave (`active` says which of them carry a ray).
#pragma once
#include "stream.h"   // lds_u64

constexpr uint32_t kFlatBudget = 32;         // triangle tests per ray up to which a scene is walked exhaustively
constexpr uint32_t kFlatTransposeMin = 4;    // meshes of at least this many (and at most 64) triangles: box test + transposition

SPT_DEV uint32_t flat_uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// the wave's share of the stack area in front of the staged geometry (2 * kLdsStack words per thread = 4 KiB per wave):
//   words [0, 512)    64 parked rays x (o.xyz, t_min, d.xyz, t_max)
//   words [512, 640)  64 x u64 closest key (any-hit: the low word is the "occluded" flag)
//   words [640, 768)  64 x (v, w) of the winner
// (explicit LDS pointers: through a generic one the reads after an LDS atomic become flat loads)
typedef __attribute__((address_space(3))) uint32_t flat_u32;
typedef float flat_v4 __attribute__((ext_vector_type(4)));   // (float4 is a class: no copies across address spaces)
typedef __attribute__((address_space(3))) flat_v4 flat_f4;
SPT_DEV void flat_put(flat_u32* p, float x, float y, float z, float w) { flat_v4 v = {x, y, z, w}; *(flat_f4*)p = v; }
SPT_DEV float4 flat_get(const flat_u32* p) { const flat_v4 v = *(const flat_f4*)p; return make_float4(v.x, v.y, v.z, v.w); }
SPT_DEV flat_u32* flat_area() { return (flat_u32*)spt_lds_stack + (threadIdx.x >> 6) * 1024u; }
SPT_DEV lds_u64* flat_key(flat_u32* area, uint32_t r) { return (lds_u64*)(area + 512u + 2u * r); }
// LDS operations of one wave execute in order; this only keeps the compiler from moving them across the hand-over
SPT_DEV void flat_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// which lanes work on which (parked ray, triangle) pair of a mesh with `count` triangles
struct FlatGroups {
    uint32_t rpp;      // rays per pass
    uint32_t q, tri;   // this lane: group index within a pass, triangle within the mesh (valid when q < rpp)
};
SPT_DEV FlatGroups flat_groups(uint32_t count) {
    FlatGroups g;
    const float rc = __builtin_amdgcn_rcpf((float)count);
    // floor(64 / count) and floor(lane / count) through the 1-ulp reciprocal: (x + 0.5) / count is at least 0.5 / 64 away from
    // an integer for integer x, a thousand times the error of the product
    g.rpp = flat_uniform((uint32_t)(64.5f * rc));
    const uint32_t lane = threadIdx.x & 63u;
    g.q = (uint32_t)(((float)lane + 0.5f) * rc);
    g.tri = lane - g.q * count;
    return g;
}
// transposition pays when the passes it needs cost less than `count` triangle tests by all lanes (~200 issue cycles each; a
// pass: one test + ~100 cycles of hand-over; parking and merging: ~250)
SPT_DEV bool flat_transpose_pays(uint32_t n_hit, uint32_t count) { return 3u * n_hit * count + 384u <= 128u * count; }

SPT_DEV DHit flat_closest(const DScene& sc, const DRay& ray, float t_max, bool active) {
    DHit h;
    h.t = t_max; h.inst = -1; h.prim = -1; h.v = 0.0f; h.w = 0.0f;
    // (keys compare as floats only for t > 0: rays with a negative t_min - none of the renderer's - take the plain loops)
    const bool key_order_ok = __ballot(active && !(ray.t_min >= 0.0f)) == 0ull;
    flat_u32* area = flat_area();
    for (uint32_t i = 0; i < sc.n_instances; ++i) {
        uint32_t prim_type, prim_id;
        const DRay orr = to_object<true>(sc, i, ray, &prim_type, &prim_id);
        prim_type = flat_uniform(prim_type);
        prim_id = flat_uniform(prim_id);
        if (prim_type == SPT_PRIM_SPHERE) {
            float mn, mx;
            if (sphere_roots(geo_ld<true>(sc, sc.o_sph + prim_id), orr, &mn, &mx) && active) {
                const float t = (mn < orr.t_min) ? mx : mn;  // sphere.rs:61
                if (orr.t_min < t && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)i, (int32_t)prim_id, h)))) {
                    h.t = t; h.inst = (int32_t)i; h.prim = (int32_t)prim_id; h.v = 0.0f; h.w = 0.0f;
                }
            }
            continue;
        }
        const float4 rhi = geo_ld<true>(sc, sc.o_mesh + 2u * prim_id + 1u);
        const uint32_t range = flat_uniform(__float_as_uint(rhi.w));
        const uint32_t first = range & 0xffffu, count = range >> 16;
        if (count >= kFlatTransposeMin && count <= 64u) {
            const float4 rlo = geo_ld<true>(sc, sc.o_mesh + 2u * prim_id);
            const bool box = active && root_hit<true>(rlo, rhi, orr.o, recip3(sc, orr.d), orr.t_min, h.t);
            const uint64_t m = __ballot(box);
            if (m == 0ull) continue;
            const uint32_t n_hit = (uint32_t)__popcll(m);
            if (key_order_ok && flat_transpose_pays(n_hit, count)) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (box) {
                    flat_put(area + 8u * rank, orr.o.x, orr.o.y, orr.o.z, orr.t_min);
                    flat_put(area + 8u * rank + 4u, orr.d.x, orr.d.y, orr.d.z, 0.0f);
                    *flat_key(area, rank) = ~0ull;
                }
                flat_wave_sync();
                const FlatGroups g = flat_groups(count);
                for (uint32_t base = 0; base < n_hit; base += g.rpp) {
                    const uint32_t r = base + g.q;
                    const bool work = g.q < g.rpp && r < n_hit;
                    unsigned long long key = ~0ull;
                    float v = 0.0f, w = 0.0f;
                    lds_u64* slot = flat_key(area, work ? r : 0u);
                    if (work) {
                        const float4 a = flat_get(area + 8u * r), b = flat_get(area + 8u * r + 4u);
                        DRay pr;
                        pr.o = mk3(a); pr.t_min = a.w; pr.d = mk3(b);
                        float t;
                        int32_t id;
                        const bool ok = tri_test_geo<true>(sc, first + g.tri, pr, &t, &v, &w, &id);
                        if (ok && t > pr.t_min) {
                            key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)id;
                            atomicMin((unsigned long long*)slot, key);
                        }
                    }
                    flat_wave_sync();
                    if (work && key != ~0ull && *slot == key) {
                        area[640u + 2u * r] = __float_as_uint(v);
                        area[641u + 2u * r] = __float_as_uint(w);
                    }
                }
                flat_wave_sync();
                if (box) {
                    const unsigned long long key = *flat_key(area, rank);
                    const float t = __uint_as_float((uint32_t)(key >> 32));
                    const int32_t id = (int32_t)(uint32_t)key;
                    if (key != ~0ull && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)i, id, h)))) {
                        h.t = t; h.inst = (int32_t)i; h.prim = id;
                        h.v = __uint_as_float(area[640u + 2u * rank]); h.w = __uint_as_float(area[641u + 2u * rank]);
                    }
                }
                flat_wave_sync();   // the area is free again
                continue;
            }
        }
        for (uint32_t k = first; k < first + count; ++k) {
            float t, v, w;
            int32_t id;
            const bool ok = tri_test_geo<true>(sc, k, orr, &t, &v, &w, &id);
            if (active && ok && t > orr.t_min && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)i, id, h)))) {  // triangle.rs:187
                h.t = t; h.inst = (int32_t)i; h.prim = id; h.v = v; h.w = w;
            }
        }
    }
    return h;
}

// any accepted test inside (t_min, t_max); the walk of a wave ends when every lane that carries a ray has found one
SPT_DEV bool flat_any(const DScene& sc, const DRay& ray, float t_max, bool active) {
    bool occluded = false;
    flat_u32* area = flat_area();
    for (uint32_t i = 0; i < sc.n_instances; ++i) {
        uint32_t prim_type, prim_id;
        const DRay orr = to_object<true>(sc, i, ray, &prim_type, &prim_id);
        prim_type = flat_uniform(prim_type);
        prim_id = flat_uniform(prim_id);
        if (prim_type == SPT_PRIM_SPHERE) {
            float mn, mx;
            occluded |= sphere_roots(geo_ld<true>(sc, sc.o_sph + prim_id), orr, &mn, &mx) && mn < t_max && mx > orr.t_min;  // sphere.rs:51-56
        } else {
            const float4 rhi = geo_ld<true>(sc, sc.o_mesh + 2u * prim_id + 1u);
            const uint32_t range = flat_uniform(__float_as_uint(rhi.w));
            const uint32_t first = range & 0xffffu, count = range >> 16;
            bool plain = true;
            if (count >= kFlatTransposeMin && count <= 64u) {
                const float4 rlo = geo_ld<true>(sc, sc.o_mesh + 2u * prim_id);
                const bool box = active && !occluded && root_hit<false>(rlo, rhi, orr.o, recip3(sc, orr.d), orr.t_min, t_max);
                const uint64_t m = __ballot(box);
                const uint32_t n_hit = (uint32_t)__popcll(m);
                if (m == 0ull) {
                    plain = false;
                } else if (flat_transpose_pays(n_hit, count)) {
                    plain = false;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (box) {
                        flat_put(area + 8u * rank, orr.o.x, orr.o.y, orr.o.z, orr.t_min);
                        flat_put(area + 8u * rank + 4u, orr.d.x, orr.d.y, orr.d.z, t_max);
                        area[512u + 2u * rank] = 0u;
                    }
                    flat_wave_sync();
                    const FlatGroups g = flat_groups(count);
                    for (uint32_t base = 0; base < n_hit; base += g.rpp) {
                        const uint32_t r = base + g.q;
                        if (g.q < g.rpp && r < n_hit) {
                            const float4 a = flat_get(area + 8u * r), b = flat_get(area + 8u * r + 4u);
                            DRay pr;
                            pr.o = mk3(a); pr.t_min = a.w; pr.d = mk3(b);
                            float t, v, w;
                            int32_t id;
                            if (tri_test_geo<true>(sc, first + g.tri, pr, &t, &v, &w, &id) && t > pr.t_min && t < b.w) area[512u + 2u * r] = 1u;
                        }
                    }
                    flat_wave_sync();
                    if (box) occluded |= area[512u + 2u * rank] != 0u;
                    flat_wave_sync();   // the area is free again
                }
            }
            if (plain) {
                for (uint32_t k = first; k < first + count; ++k) {
                    float t, v, w;
                    int32_t id;
                    occluded |= tri_test_geo<true>(sc, k, orr, &t, &v, &w, &id) && t > orr.t_min && t < t_max;
                }
            }
        }
        if (__ballot(active && !occluded) == 0ull) break;
    }
    return occluded;
}
