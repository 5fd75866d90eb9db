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
// the per-ray Vec allocation of the reference (bvh.rs:243,267) is gone.  Nodes are
// repacked into 64-byte wide nodes and children are visited near-first (see below).
// The slab test multiplies by a per-ray reciprocal instead of the reference's six
// divisions per node; it only culls, the returned hit is unaffected.
#pragma once
#include "device_math.h"
#include "bezier.h"

struct DScene {
    const float4* tri_pos;     // 3 x float4 per triangle
    const float4* tri_attr;    // 9 x float4 per triangle (spt_tri_attr)
    const float4* instances;   // 12 x float4 per instance (spt_instance)
    const uint4* meshes;       // root, node_count, tri_first, tri_count
    const float4* spheres;     // center, radius
    const float4* bez;         // Bezier patches: 16 control points (xyz) per patch; only read by libspt_hip_bez.so
    const spt_surface* surfaces;
    const spt_material* materials;
    const spt_medium* mediums;
    const spt_light* lights;
    const float* light_props; const float* light_u; const uint32_t* light_k;
    // environment map, packed for the two access patterns (one line instead of four per bilinear tap / alias draw):
    const float4* env_px;      // per texel: (r, g, b, alias-table probability `props`)
    const uint2* env_uk;       // per texel: alias table (u bits, k)
    uint32_t n_tlas_nodes, n_instances, n_lights, n_meshes;
    uint32_t aggregate, light_sampler;
    int32_t env_light_index;
    uint32_t env_w, env_h;
    float env_scale[3];
    uint32_t stack_cap;        // LDS stack entries per lane the scene needs
    // Traversal geometry as ONE float4 blob [tlas | instances | meshes | spheres | blas | tri_pos]
    // (offsets in float4 units).  The first lds_f4 float4 are staged into LDS once per workgroup:
    //   small scenes (k_*<true>): the whole blob, every fetch is a ds_read_b128;
    //   large scenes (k_*<false>): nothing (lds_f4 = 0).  Staging TLAS + instances + the top 256 BLAS
    //   nodes was MEASURED slower on the 1 M-triangle scene (613 -> 527 Msamples/s): those records
    //   are L1/L2-hot anyway and the per-fetch "staged or global?" branch costs more than it saves.
    const float4* geo;
    uint32_t geo_f4;           // blob length in float4
    uint32_t lds_f4;           // staged prefix length in float4 (== geo_f4 for k_*<true>)
    uint32_t o_tlas, o_inst, o_mesh, o_blas, o_tri, o_sph;
    // fused bounces only: the shading tables also sit in the staged blob (float4 offsets), see tab_ld
    uint32_t o_attr, o_surf, o_mat, o_light;
    uint32_t o_tord;           // TLAS leaf slot -> instance index (the device-built TLAS re-orders its leaves)
    uint32_t fast_slab;        // device-built (padded) trees: box tests only cull, so 1/d may be v_rcp_f32 (1 ulp)
    uint32_t tlas_root;        // ref of the TLAS root (wide-node index or leaf ref)
    uint32_t o_eye;            // eye-relative copy of the blob only (eye.h): one float4 (s x e1, e2 . (s x e1)) per triangle slot
    uint32_t flat;             // the scene is small enough for the exhaustive loops of flat.h (LDS-resident geometry only)
    // streaming walker (stream.h; scenes that do not fit LDS): its TLAS is a tree of compressed 4-wide nodes in the same
    // array as the BLAS nodes (o_blas), its leaves index 96-byte instance entry records in TLAS leaf order
    uint32_t s_root;           // ref of that TLAS' root (node index relative to o_blas, or a leaf ref); its box is tlas_lo / tlas_hi
    uint32_t o_sinst;          // 6 x float4 per record: M^-1 (3), (instance, prim_type, prim_id, -), (BLAS root lo | sphere, root ref), (root hi, -)
    float tlas_lo[3], tlas_hi[3];  // its box
    // image textures (k_shade<2, .> only; all null otherwise), see shading.h "textures"
    const float4* textures;        // 4 x float4 per spt_texture node
    const uint4* tex_prog;         // postfix programs, one instruction per uint4
    const uint2* tex_root;         // per texture node: (first instruction, count)
    const uint32_t* tex_chain;     // modifier node ids, outermost first, per image leaf
    const uint2* images;           // (first_level, n_levels)
    const uint4* image_levels;     // (width, height, first_texel, -)
    const uint32_t* texels;        // RGBA8
    const uint4* recipes;          // 2 x uint4 per spt_material_recipe
    const float2* ss_cdf;          // SPT_SS_CDF_SIZE x (x, y): BSSRDF radius table (scenes with a Subsurface substrate)
    // position-normal distributions (include/spt_pndf.h; k_shade<3, .> only, null otherwise)
    const spt_pndf* pndfs;
    const spt_pndf_term* pndf_terms;
    const spt_pndf_node* pndf_nodes;
    const uint32_t* pndf_refs;
    const uint32_t* pndf_roots;
    const uint8_t* inst_class;     // per instance: the class the hit queue of bounce >= 1 files its hits under (kernels.h, kClasses)
};

struct DHit {
    float t;
    int32_t inst, prim;
    float v, w;
};

// SPT_RENDER_COUNT_VISITS: what one lane's walks fetched (kCount instantiations only; a null pointer otherwise)
struct LaneVisits {
    uint32_t nodes, tris, insts;
};
template <bool kCount>
SPT_DEV void count_node(LaneVisits* c) { if (kCount) ++c->nodes; }
template <bool kCount>
SPT_DEV void count_tri(LaneVisits* c) { if (kCount) ++c->tris; }
template <bool kCount>
SPT_DEV void count_inst(LaneVisits* c) { if (kCount) ++c->insts; }

// per-lane stack in LDS: entry `level` of this lane
extern __shared__ uint32_t spt_lds_stack[];
SPT_DEV uint32_t& stack_at(uint32_t word) { return spt_lds_stack[word * blockDim.x + threadIdx.x]; }

// geometry fetch: LDS copy (kLds) or the global blob
SPT_DEV float4* geo_lds(const DScene&) { return reinterpret_cast<float4*>(spt_lds_stack + 2u * 8u * blockDim.x); }  // after the stack (kLdsStack)
template <bool kLds>
SPT_DEV float4 geo_ld(const DScene& sc, uint32_t off) {
    if (kLds) return geo_lds(sc)[off];
    return sc.geo[off];
}
// triangles are never in the staged prefix of a large scene
template <bool kLds>
SPT_DEV float4 geo_ld_tri(const DScene& sc, uint32_t off) {
    if (kLds) return geo_lds(sc)[off];
    return sc.geo[off];
}
// instance index behind TLAS leaf slot `slot` (a GROUP aggregate walks the instances directly)
template <bool kLds>
SPT_DEV uint32_t tlas_instance(const DScene& sc, uint32_t slot) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(kLds ? geo_lds(sc) : const_cast<float4*>(sc.geo));
    return p[4u * sc.o_tord + slot];
}
// once per workgroup, before any traversal
template <bool kLds>
SPT_DEV void stage_geometry(const DScene& sc) {
    float4* dst = geo_lds(sc);
    for (uint32_t i = threadIdx.x; i < sc.lds_f4; i += blockDim.x) dst[i] = sc.geo[i];
    __syncthreads();
}

// 1/d for the slab tests.  With the caller's exact trees (SPT_REFERENCE_BVH=1) this is the IEEE division the
// oracle's reciprocal mode performs; with the library's own padded trees the boxes only cull and the
// hardware reciprocal (1 ulp, one instruction instead of ~10 per component) is enough.
SPT_DEV f3 recip3(const DScene& sc, f3 d) {
    if (sc.fast_slab) return mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

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

// the same with the two edge vectors already subtracted (the traversal blob stores p0, p1 - p0, p2 - p0:
// the identical f32 differences, computed once at scene creation instead of per test)
SPT_DEV bool tri_test_edges(float4 a, float4 b, float4 c, const DRay& r, float* t, float* v_out, float* w_out) {
    f3 p0 = mk3(a), e1 = mk3(b), e2 = mk3(c);
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
// `slot` indexes the blob's triangle copy (p0, p1 - p0, p2 - p0), which is in the order of the DEVICE-built BLAS; the
// triangle's index in the ABI arrays (= BasicPrimitiveRef, the tie-rule key, the tri_attr index)
// travels in the pad lane of its first vertex.
template <bool kLds>
SPT_DEV bool tri_test_geo(const DScene& sc, uint32_t slot, const DRay& r, float* t, float* v_out, float* w_out, int32_t* id) {
    const uint32_t o = sc.o_tri + 3u * slot;
    const float4 a = geo_ld_tri<kLds>(sc, o);
    *id = __float_as_int(a.w);
    return tri_test_edges(a, geo_ld_tri<kLds>(sc, o + 1), geo_ld_tri<kLds>(sc, o + 2), r, t, v_out, w_out);
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

// ---- wide nodes --------------------------------------------------------------------------
// The device does not walk the 32-byte nodes of the ABI directly: at scene creation every inner
// node is repacked into a 64-byte WIDE node that carries the boxes of BOTH children,
//     f4[0] = (left.bmin,  left ref)   f4[1] = (left.bmax,  right ref)
//     f4[2] = (right.bmin, -)          f4[3] = (right.bmax, -)
// ref = inner: wide-node index | leaf: kLeaf | count << 27 | first item.  One fetch (4 x 16 B) gives
// both slab tests, leaves cost no node fetch, and the nearer child is entered first so the far
// subtree is usually culled by the hit found in the near one.  The box and ref of each tree's ROOT
// live in the mesh record (BLAS, 2 x float4 per mesh) / in DScene (TLAS), so entering a tree costs
// one slab test and no node fetch.
// Visit order therefore differs from the reference's (push left, push right, pop), which only
// matters for exactly equal hit distances: the closest hit is defined as the minimum over
// (t, instance, prim) and boxes are culled with t0 <= t_best, which is independent of the order
// (the oracle's ORACLE_TIE_MIN_ID mode applies the same rule).
constexpr uint32_t kLeaf = 0x80000000u;
SPT_DEV uint32_t leaf_first(uint32_t ref) { return ref & 0x07ffffffu; }
SPT_DEV uint32_t leaf_count(uint32_t ref) { return (ref >> 27) & 15u; }

// entry distance of the slab test, or a negative-NaN-free "miss": returns hit flag, t0 through *t0
SPT_DEV bool slab_t0(float4 lo, float4 hi, f3 o, f3 inv_d, float t_min, float* t0_out) {
    bool empty = (lo.x > hi.x) | (lo.y > hi.y) | (lo.z > hi.z);
    float x0 = (lo.x - o.x) * inv_d.x, x1 = (hi.x - o.x) * inv_d.x;
    float y0 = (lo.y - o.y) * inv_d.y, y1 = (hi.y - o.y) * inv_d.y;
    float z0 = (lo.z - o.z) * inv_d.z, z1 = (hi.z - o.z) * inv_d.z;
    float xa = spt_min(x0, x1), xb = spt_max(x0, x1);
    float ya = spt_min(y0, y1), yb = spt_max(y0, y1);
    float za = spt_min(z0, z1), zb = spt_max(z0, z1);
    float t0 = spt_max(xa, spt_max(ya, za));
    float t1 = spt_min(xb, spt_min(yb, zb));
    *t0_out = t0;
    return !empty & (t0 <= t1) & (t1 > t_min);
}

// Traversal stack: entries are (ref, t0).  The first kLdsStack levels of a lane live in LDS
// ([word][lane] layout: conflict-free), deeper levels spill to a small private array.  Near-first
// order keeps the live stack short (it only grows where BOTH children are hit), so the spill is
// rarely touched, and LDS per workgroup stays at 16 KiB whatever the tree depth — a full-depth LDS
// stack (58 KiB for the 1 M-triangle scene) would cap the CU at two workgroups.
constexpr uint32_t kLdsStack = 8;
constexpr uint32_t kSpillStack = 40;
struct TStack {
    uint32_t sp = 0;
    uint2* spill;   // kSpillStack private entries owned by the caller (kept out of this struct so that
                    // the dynamically indexed array does not drag the other fields into scratch)
    SPT_DEV void push(uint32_t ref, float t0) {
        if (sp < kLdsStack) {
            stack_at(2u * sp) = ref;
            stack_at(2u * sp + 1u) = __float_as_uint(t0);
        } else {
            spill[sp - kLdsStack] = make_uint2(ref, __float_as_uint(t0));
        }
        ++sp;
    }
    SPT_DEV void pop(uint32_t* ref, float* t0) {
        --sp;
        if (sp < kLdsStack) {
            *ref = stack_at(2u * sp);
            *t0 = __uint_as_float(stack_at(2u * sp + 1u));
        } else {
            uint2 e = spill[sp - kLdsStack];
            *ref = e.x;
            *t0 = __uint_as_float(e.y);
        }
    }
};

// ---- compressed 4-wide nodes (BLAS of large scenes) -------------------------------------------
// For scenes that do not fit LDS the BLAS is collapsed to a 4-ary tree whose 64-byte node holds FOUR
// child boxes quantised to 8 bits per coordinate relative to the node's own box:
//     f4[0] = (p.xyz = node box min,  ex | ey << 8 | ez << 16 : biased power-of-two scale exponents)
//     f4[1] = qlo.x[4] qlo.y[4] qlo.z[4] qhi.x[4]   (one byte per child)
//     f4[2] = qhi.y[4] qhi.z[4] ref0 ref1            f4[3] = ref2 ref3 - -
// child box = p + 2^e * q, rounded OUTWARD on the host with the same f32 arithmetic, so it always
// contains the exact child box: the slab test only culls, a looser box can never lose a hit.
// Versus the 2-wide nodes: half the tree depth (half the dependent fetches) and 64 B per 4 children
// instead of 128 B - the large scene is bound by random node fetches from L2 / Infinity Cache.
struct Node4 {
    float t0[4];
    uint32_t ref[4];
};
template <bool kLds>
SPT_DEV void node4_test(const DScene& sc, uint32_t off, f3 o, f3 inv_d, float t_min, Node4* out) {
    const float4 n0 = geo_ld<kLds>(sc, off), n1 = geo_ld<kLds>(sc, off + 1u), n2 = geo_ld<kLds>(sc, off + 2u), n3 = geo_ld<kLds>(sc, off + 3u);
    const uint32_t meta = __float_as_uint(n0.w);
    const float sx = spt_u2f((meta & 0xffu) << 23), sy = spt_u2f(((meta >> 8) & 0xffu) << 23), sz = spt_u2f(((meta >> 16) & 0xffu) << 23);
    const uint32_t qlx = __float_as_uint(n1.x), qly = __float_as_uint(n1.y), qlz = __float_as_uint(n1.z), qhx = __float_as_uint(n1.w);
    const uint32_t qhy = __float_as_uint(n2.x), qhz = __float_as_uint(n2.y);
    out->ref[0] = __float_as_uint(n2.z); out->ref[1] = __float_as_uint(n2.w);
    out->ref[2] = __float_as_uint(n3.x); out->ref[3] = __float_as_uint(n3.y);
#pragma unroll
    for (uint32_t c = 0; c < 4u; ++c) {
        const uint32_t sh = 8u * c;
        const float lx = n0.x + sx * (float)((qlx >> sh) & 0xffu), hx = n0.x + sx * (float)((qhx >> sh) & 0xffu);
        const float ly = n0.y + sy * (float)((qly >> sh) & 0xffu), hy = n0.y + sy * (float)((qhy >> sh) & 0xffu);
        const float lz = n0.z + sz * (float)((qlz >> sh) & 0xffu), hz = n0.z + sz * (float)((qhz >> sh) & 0xffu);
        float t0;
        const bool hit = slab_t0(make_float4(lx, ly, lz, 0.0f), make_float4(hx, hy, hz, 0.0f), o, inv_d, t_min, &t0);
        out->t0[c] = hit ? t0 : spt_inf();   // absent children are stored as empty boxes (lo > hi)
    }
}
// order the four (t0, ref) pairs by t0 (5 compare-exchanges)
SPT_DEV void node4_sort(Node4* n) {
#define SPT_CX(a, b)                                                          \
    if (n->t0[a] > n->t0[b]) {                                                \
        float tt = n->t0[a]; n->t0[a] = n->t0[b]; n->t0[b] = tt;              \
        uint32_t rr = n->ref[a]; n->ref[a] = n->ref[b]; n->ref[b] = rr;       \
    }
    SPT_CX(0, 1) SPT_CX(2, 3) SPT_CX(0, 2) SPT_CX(1, 3) SPT_CX(1, 2)
#undef SPT_CX
}

// Near-first walk of one wide-node tree.  `limit` is read on every test, so a closest-hit walk
// (kClosest: cull with t0 <= limit, the tie rule) tightens as `leaf` lowers it; an any-hit walk culls
// with t0 < limit.  leaf(first, count) returns true to stop the whole walk (any-hit found).
template <bool kLds, bool kClosest, bool kN4, bool kCount, class LeafFn>
SPT_DEV bool walk_tree(const DScene& sc, uint32_t nodes_off, uint32_t root, f3 o, f3 inv_d, float t_min, const float& limit,
                       TStack& st, LaneVisits* vc, LeafFn leaf) {
    const uint32_t base = st.sp;
    uint32_t cur = root;
    while (true) {
        if (cur & kLeaf) {
            if (leaf(leaf_first(cur), leaf_count(cur))) { st.sp = base; return true; }
        } else if (kN4) {
            Node4 n;
            count_node<kCount>(vc);
            node4_test<kLds>(sc, nodes_off + 4u * cur, o, inv_d, t_min, &n);
            node4_sort(&n);
            // farthest first onto the stack, nearest becomes `cur`
#pragma unroll
            for (int c = 3; c >= 1; --c)
                if ((kClosest ? n.t0[c] <= limit : n.t0[c] < limit) && n.t0[c] < spt_inf() && st.sp < kLdsStack + kSpillStack) st.push(n.ref[c], n.t0[c]);
            if ((kClosest ? n.t0[0] <= limit : n.t0[0] < limit) && n.t0[0] < spt_inf()) { cur = n.ref[0]; continue; }
        } else {
            const uint32_t n = nodes_off + 4u * cur;
            count_node<kCount>(vc);
            float4 a = geo_ld<kLds>(sc, n), b = geo_ld<kLds>(sc, n + 1u), c = geo_ld<kLds>(sc, n + 2u), d = geo_ld<kLds>(sc, n + 3u);
            float tl, tr;
            bool hl = slab_t0(a, b, o, inv_d, t_min, &tl);
            bool hr = slab_t0(c, d, o, inv_d, t_min, &tr);
            hl = hl && (kClosest ? tl <= limit : tl < limit);
            hr = hr && (kClosest ? tr <= limit : tr < limit);
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
        // pop the next subtree that can still matter
        bool found = false;
        while (st.sp > base) {
            uint32_t ref;
            float t0;
            st.pop(&ref, &t0);
            if (kClosest ? t0 <= limit : t0 < limit) { cur = ref; found = true; break; }
        }
        if (!found) return false;
    }
}

// root box test of a tree: true if the walk has to start (closest: t0 <= limit, any: t0 < limit)
template <bool kClosest>
SPT_DEV bool root_hit(float4 lo, float4 hi, f3 o, f3 inv_d, float t_min, float limit) {
    float t0;
    bool hit = slab_t0(lo, hi, o, inv_d, t_min, &t0);
    return hit && (kClosest ? t0 <= limit : t0 < limit);
}

SPT_DEV bool key_less(int32_t inst, int32_t prim, const DHit& h) {
    return (inst < h.inst) || (inst == h.inst && prim < h.prim);
}

// one instance against the current best hit
template <bool kLds, bool kCount = false>
SPT_DEV void instance_closest(const DScene& sc, uint32_t inst, const DRay& ray, DHit& h, TStack& st, LaneVisits* vc = nullptr) {
    uint32_t prim_type, prim_id;
    count_inst<kCount>(vc);
    DRay orr = to_object<kLds>(sc, inst, ray, &prim_type, &prim_id);
    if (prim_type == SPT_PRIM_SPHERE) {
        float mn, mx;
        if (sphere_roots(geo_ld<kLds>(sc, sc.o_sph + prim_id), orr, &mn, &mx)) {
            float t = (mn < orr.t_min) ? mx : mn;  // sphere.rs:61
            if (orr.t_min < t && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, (int32_t)prim_id, h)))) {
                h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = 0.0f; h.w = 0.0f;
            }
        }
        return;
    }
#if SPT_WITH_BEZIER
    if (prim_type == SPT_PRIM_BEZIER) {   // bezier.rs:160-174; the patch parameters ride in the hit's (v, w)
        float u, v, t;
        if (bezier_intersect_ray(sc.bez + 16u * prim_id, orr, &u, &v, &t) && t > orr.t_min &&
            (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, (int32_t)prim_id, h)))) {
            h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = u; h.w = v;
        }
        return;
    }
#endif
    const float4 rlo = geo_ld<kLds>(sc, sc.o_mesh + 2u * prim_id), rhi = geo_ld<kLds>(sc, sc.o_mesh + 2u * prim_id + 1u);
    const uint32_t root = __float_as_uint(rlo.w);
    const f3 inv_o = recip3(sc, orr.d);
    if (!root_hit<true>(rlo, rhi, orr.o, inv_o, orr.t_min, h.t)) return;
    walk_tree<kLds, true, !kLds, kCount>(sc, sc.o_blas, root, orr.o, inv_o, orr.t_min, h.t, st, vc, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first; i < first + count; ++i) {
            float t, v, w;
            int32_t id;
            count_tri<kCount>(vc);
            bool ok = tri_test_geo<kLds>(sc, i, orr, &t, &v, &w, &id);
            if (ok && t > orr.t_min && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, id, h)))) {  // triangle.rs:187
                h.t = t; h.inst = (int32_t)inst; h.prim = id; h.v = v; h.w = w;
            }
        }
        return false;
    });
}

template <bool kLds, bool kCount = false>
SPT_DEV bool instance_any(const DScene& sc, uint32_t inst, const DRay& ray, float t_max, TStack& st, LaneVisits* vc = nullptr) {
    uint32_t prim_type, prim_id;
    count_inst<kCount>(vc);
    DRay orr = to_object<kLds>(sc, inst, ray, &prim_type, &prim_id);
    if (prim_type == SPT_PRIM_SPHERE) {
        float mn, mx;
        return sphere_roots(geo_ld<kLds>(sc, sc.o_sph + prim_id), orr, &mn, &mx) && mn < t_max && mx > orr.t_min;  // sphere.rs:51-56
    }
#if SPT_WITH_BEZIER
    if (prim_type == SPT_PRIM_BEZIER) {   // bezier.rs:152-158
        float u, v, t;
        return bezier_intersect_ray(sc.bez + 16u * prim_id, orr, &u, &v, &t) && t > orr.t_min && t < t_max;
    }
#endif
    const float4 rlo = geo_ld<kLds>(sc, sc.o_mesh + 2u * prim_id), rhi = geo_ld<kLds>(sc, sc.o_mesh + 2u * prim_id + 1u);
    const uint32_t root = __float_as_uint(rlo.w);
    const f3 inv_o = recip3(sc, orr.d);
    if (!root_hit<false>(rlo, rhi, orr.o, inv_o, orr.t_min, t_max)) return false;
    return walk_tree<kLds, false, !kLds, kCount>(sc, sc.o_blas, root, orr.o, inv_o, orr.t_min, t_max, st, vc, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first; i < first + count; ++i) {
            float t, v, w;
            int32_t id;
            count_tri<kCount>(vc);
            if (tri_test_geo<kLds>(sc, i, orr, &t, &v, &w, &id) && t > orr.t_min && t < t_max) return true;
        }
        return false;
    });
}

// Closest hit of the scene aggregate: BvhAccel<Instance> / Group::intersect
template <bool kLds, bool kCount = false>
SPT_DEV DHit trace_closest(const DScene& sc, const DRay& ray, float t_max, LaneVisits* vc = nullptr) {
    DHit h;
    h.t = t_max;
    h.inst = -1;
    h.prim = -1;
    h.v = 0.0f;
    h.w = 0.0f;
    uint2 spill_mem[kSpillStack];
    TStack st;
    st.spill = spill_mem;
    if (sc.aggregate == SPT_AGGREGATE_GROUP) {
        for (uint32_t i = 0; i < sc.n_instances; ++i) instance_closest<kLds, kCount>(sc, i, ray, h, st, vc);
    } else if (sc.n_tlas_nodes > 0) {
        const f3 inv_w = recip3(sc, ray.d);
        const float4 tlo = make_float4(sc.tlas_lo[0], sc.tlas_lo[1], sc.tlas_lo[2], 0.0f), thi = make_float4(sc.tlas_hi[0], sc.tlas_hi[1], sc.tlas_hi[2], 0.0f);
        if (root_hit<true>(tlo, thi, ray.o, inv_w, ray.t_min, h.t))
        walk_tree<kLds, true, false, kCount>(sc, sc.o_tlas, sc.tlas_root, ray.o, inv_w, ray.t_min, h.t, st, vc, [&](uint32_t first, uint32_t count) {
            for (uint32_t i = first; i < first + count; ++i) instance_closest<kLds, kCount>(sc, tlas_instance<kLds>(sc, i), ray, h, st, vc);
            return false;
        });
    }
    return h;
}

// Any hit in (t_min, t_max): intersect_test of the aggregate
template <bool kLds, bool kCount = false>
SPT_DEV bool trace_any(const DScene& sc, const DRay& ray, float t_max, LaneVisits* vc = nullptr) {
    uint2 spill_mem[kSpillStack];
    TStack st;
    st.spill = spill_mem;
    if (sc.aggregate == SPT_AGGREGATE_GROUP) {
        for (uint32_t i = 0; i < sc.n_instances; ++i)
            if (instance_any<kLds, kCount>(sc, i, ray, t_max, st, vc)) return true;
        return false;
    }
    if (sc.n_tlas_nodes == 0) return false;
    const f3 inv_w = recip3(sc, ray.d);
    const float4 tlo = make_float4(sc.tlas_lo[0], sc.tlas_lo[1], sc.tlas_lo[2], 0.0f), thi = make_float4(sc.tlas_hi[0], sc.tlas_hi[1], sc.tlas_hi[2], 0.0f);
    if (!root_hit<false>(tlo, thi, ray.o, inv_w, ray.t_min, t_max)) return false;
    return walk_tree<kLds, false, false, kCount>(sc, sc.o_tlas, sc.tlas_root, ray.o, inv_w, ray.t_min, t_max, st, vc, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first; i < first + count; ++i)
            if (instance_any<kLds, kCount>(sc, tlas_instance<kLds>(sc, i), ray, t_max, st, vc)) return true;
        return false;
    });
}

// ---- resumable traversal -----------------------------------------------------------------------
// The same walk as trace_closest / trace_any (identical visit order and arithmetic per ray), written
// as a state machine that advances ONE step per call.  It exists for incoherent rays on large scenes:
// with the nested loops above a wave runs until its slowest lane is done (measured on the 1 M-triangle
// scene: 16 % of the lanes active in the shadow kernel, 30 % in extend), whereas a stepping walker
// lets a persistent wave hand a finished lane the next ray of the queue while the others continue.
template <bool kLds, bool kClosest, bool kCount = false>
struct Walker {
    LaneVisits vc;             // kCount only (dead otherwise)
    f3 o, d, inv_w;            // world ray
    float t_min;
    f3 oo, od, inv_o;          // ray in the space of the current instance
    uint32_t cur;              // ref being processed (phase 1: TLAS, phase 2: BLAS)
    uint32_t phase;            // 0: next instance of the leaf / TLAS pop, 1: TLAS ref, 2: BLAS ref
    uint32_t inst, inst_next, inst_end, blas_base;
    TStack st;
    DHit h;                    // kClosest: best hit; else h.t = t_max and h.inst >= 0 means "occluded"
    bool done;

    SPT_DEV void start(const DScene& sc, const DRay& r, float t_max, uint2* spill_mem) {
        st.spill = spill_mem;
        o = r.o; d = r.d; t_min = r.t_min;
        inv_w = recip3(sc, r.d);
        h.t = t_max; h.inst = -1; h.prim = -1; h.v = 0.0f; h.w = 0.0f;
        st.sp = 0;
        done = false;
        inst_next = 0; inst_end = 0; blas_base = 0; inst = 0; cur = 0;
        if (sc.aggregate == SPT_AGGREGATE_GROUP) {
            inst_end = sc.n_instances;
            phase = 0;
        } else if (sc.n_tlas_nodes > 0) {
            const float4 tlo = make_float4(sc.tlas_lo[0], sc.tlas_lo[1], sc.tlas_lo[2], 0.0f), thi = make_float4(sc.tlas_hi[0], sc.tlas_hi[1], sc.tlas_hi[2], 0.0f);
            cur = sc.tlas_root;
            phase = 1;
            if (!root_hit<kClosest>(tlo, thi, o, inv_w, t_min, h.t)) { phase = 0; done = true; }
        } else {
            phase = 0;
            done = true;
        }
    }
    SPT_DEV bool passes(float t0) const { return kClosest ? t0 <= h.t : t0 < h.t; }

    // descend into the children of wide node `node`; returns false if neither child is hit
    SPT_DEV bool enter(const DScene& sc, uint32_t nodes_off, f3 ro, f3 rinv) {
        const uint32_t n = nodes_off + 4u * cur;
        count_node<kCount>(&vc);
        float4 a = geo_ld<kLds>(sc, n), b = geo_ld<kLds>(sc, n + 1u), c = geo_ld<kLds>(sc, n + 2u), e = geo_ld<kLds>(sc, n + 3u);
        float tl, tr;
        bool hl = slab_t0(a, b, ro, rinv, t_min, &tl);
        bool hr = slab_t0(c, e, ro, rinv, t_min, &tr);
        hl = hl && passes(tl);
        hr = hr && passes(tr);
        const uint32_t rl = __float_as_uint(a.w), rr = __float_as_uint(b.w);
        if (hl && hr) {
            const bool left_first = tl <= tr;
            if (st.sp < kLdsStack + kSpillStack) st.push(left_first ? rr : rl, left_first ? tr : tl);
            cur = left_first ? rl : rr;
            return true;
        }
        if (hl) { cur = rl; return true; }
        if (hr) { cur = rr; return true; }
        return false;
    }
    // same for a compressed 4-wide BLAS node
    SPT_DEV bool enter4(const DScene& sc) {
        Node4 n;
        count_node<kCount>(&vc);
        node4_test<kLds>(sc, sc.o_blas + 4u * cur, oo, inv_o, t_min, &n);
        node4_sort(&n);
#pragma unroll
        for (int c = 3; c >= 1; --c)
            if (passes(n.t0[c]) && n.t0[c] < spt_inf() && st.sp < kLdsStack + kSpillStack) st.push(n.ref[c], n.t0[c]);
        if (passes(n.t0[0]) && n.t0[0] < spt_inf()) { cur = n.ref[0]; return true; }
        return false;
    }
    SPT_DEV bool pop_to(uint32_t base) {
        while (st.sp > base) {
            uint32_t ref;
            float t0;
            st.pop(&ref, &t0);
            if (passes(t0)) { cur = ref; return true; }
        }
        return false;
    }

    SPT_DEV void step(const DScene& sc) {
        if (phase == 2u) {  // inside the BLAS of `inst`
            if (cur & kLeaf) {
                const uint32_t first = leaf_first(cur), count = leaf_count(cur);
                for (uint32_t i = first; i < first + count; ++i) {
                    float t, v, w;
                    DRay orr;
                    orr.o = oo; orr.d = od; orr.t_min = t_min;
                    int32_t id;
                    count_tri<kCount>(&vc);
                    bool ok = tri_test_geo<kLds>(sc, i, orr, &t, &v, &w, &id);
                    if (kClosest) {
                        if (ok && t > t_min && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, id, h)))) {
                            h.t = t; h.inst = (int32_t)inst; h.prim = id; h.v = v; h.w = w;
                        }
                    } else if (ok && t > t_min && t < h.t) {
                        h.inst = (int32_t)inst;
                        done = true;
                        return;
                    }
                }
            } else if (kLds ? enter(sc, sc.o_blas, oo, inv_o) : enter4(sc)) {
                return;
            }
            if (!pop_to(blas_base)) phase = 0u;
            return;
        }
        if (phase == 1u) {  // a TLAS ref
            if (cur & kLeaf) {
                inst_next = leaf_first(cur);
                inst_end = inst_next + leaf_count(cur);
                phase = 0u;
                return;
            }
            if (!enter(sc, sc.o_tlas, o, inv_w)) phase = 0u;
            return;
        }
        // phase 0: next instance of the current leaf, else pop the TLAS stack
        if (inst_next < inst_end) {
            inst = tlas_instance<kLds>(sc, inst_next++);   // slot -> instance (identity for a GROUP aggregate)
            count_inst<kCount>(&vc);
            uint32_t prim_type, prim_id;
            DRay wr;
            wr.o = o; wr.d = d; wr.t_min = t_min;
            DRay orr = to_object<kLds>(sc, inst, wr, &prim_type, &prim_id);
            if (prim_type == SPT_PRIM_SPHERE) {
                float mn, mx;
                bool roots = sphere_roots(geo_ld<kLds>(sc, sc.o_sph + prim_id), orr, &mn, &mx);
                if (kClosest) {
                    if (roots) {
                        float t = (mn < orr.t_min) ? mx : mn;
                        if (orr.t_min < t && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, (int32_t)prim_id, h)))) {
                            h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = 0.0f; h.w = 0.0f;
                        }
                    }
                } else if (roots && mn < h.t && mx > orr.t_min) {
                    h.inst = (int32_t)inst;
                    done = true;
                }
                return;
            }
#if SPT_WITH_BEZIER
            if (prim_type == SPT_PRIM_BEZIER) {
                float u, v, t;
                const bool got = bezier_intersect_ray(sc.bez + 16u * prim_id, orr, &u, &v, &t) && t > orr.t_min;
                if (kClosest) {
                    if (got && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, (int32_t)prim_id, h)))) {
                        h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = u; h.w = v;
                    }
                } else if (got && t < h.t) {
                    h.inst = (int32_t)inst;
                    done = true;
                }
                return;
            }
#endif
            oo = orr.o; od = orr.d;
            inv_o = recip3(sc, orr.d);
            const float4 rlo = geo_ld<kLds>(sc, sc.o_mesh + 2u * prim_id), rhi = geo_ld<kLds>(sc, sc.o_mesh + 2u * prim_id + 1u);
            if (!root_hit<kClosest>(rlo, rhi, oo, inv_o, t_min, h.t)) return;   // stays in phase 0: next instance
            cur = __float_as_uint(rlo.w);
            blas_base = st.sp;
            phase = 2u;
            return;
        }
        if (pop_to(0u)) { phase = 1u; return; }
        done = true;
    }
};
