// Streaming traversal for scenes whose geometry does not fit LDS (the 1 M-triangle class).
//
// Replaces, per ray, the same reference functions as trace.h (BvhAccel::{intersect, intersect_test} bvh.rs:237-283,
// Instance::intersect instance.rs:88-109, Triangle::intersect_ray triangle.rs:124-147, Sphere sphere.rs:25-84) with the
// same hit definition: the minimum over (t, instance, prim) of every primitive test that accepts, resp. "any accepted
// test inside (t_min, t_max)".  Primitive tests are the bit-exact ones of trace.h; only the BOX tests differ, and
// boxes only cull.
//
// Why a second walker.  Measured on cfg5 (round 1): the one-step-per-call state machine of trace.h executes, per step
// and per wave, the leaf branch, the node branch, the TLAS branch and the instance branch one after the other, each
// with its own memory round trip and with a fraction of the lanes (30 % VALU lane utilisation, 66 % of wave cycles in
// s_waitcnt), and a node visit costs ~230 VALU instructions.  Here:
//   * while-while: all lanes that hold an inner node run the node loop together until none is left, then all lanes
//     that hold a leaf run the leaf code together - one memory round trip per loop iteration, fuller waves;
//   * ONE node format for TLAS and BLAS (the compressed 4-wide node of trace.h) in one array, so "a node" is one
//     code path whatever the level; the ray of the current space lives in (ro, rd, rinv);
//   * the node test is ~100 instructions: children planes are decoded straight into slab distances,
//         t = q * (2^e / d) + (p - o) / d            (one v_cvt_f32_ubyteN + one v_fma per plane)
//     the near / far byte planes are picked by the sign of the ray direction (no per-plane min / max), and the
//     interval test is RELAXED by 4e-7 relative on both ends, which covers the few ulps by which this arithmetic and
//     the decode-then-subtract arithmetic of trace.h differ: the set of accepted children is a superset of what the
//     exact test on the (outward-quantised, padded) boxes accepts.  A NaN plane distance is ignored by v_max3 /
//     v_min3, which drops that constraint - again a superset (zero direction components are kept finite, see slab_rcp);
//   * a leaf's triangles (<= 4, contiguous) are all requested before the first one is tested;
//   * an instance visit is one contiguous 96-byte record (M^-1, ids, root box + root ref of its BLAS).
#pragma once
#include "trace.h"

constexpr uint32_t kNoRef = 0xffffffffu;     // "nothing in hand": the next thing comes off the stack
constexpr uint32_t kInTlas = 0xffffffffu;    // blas_base while the walk is in the TLAS
constexpr float kRelaxLo = 0.9999996f;       // 1 - 4e-7 (rounded): entry distances are lowered,
constexpr float kRelaxHi = 1.0000004f;       // 1 + 4e-7: exit distances raised (both only matter when positive)

SPT_DEV float ubyte_f32(uint32_t v, int k) {   // v_cvt_f32_ubyte{k}
    return (float)((v >> (8 * k)) & 0xffu);
}
// NaN-ignoring 3-way max / min (IEEE mode: a quiet NaN operand is dropped)
SPT_DEV float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
SPT_DEV float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// 1 / d for the slab distances.  A zero (or denormal) component must not become an infinity here: q * inf + (-inf) is a
// NaN, a NaN plane is ignored, and a ray with an ignored axis walks every node its other two axes touch - measured on
// cfg5: the few rays per pass that leave the floor exactly along +y (a cosine-hemisphere sample with r = 0) each visited
// the whole 330 k-node tree and one bounce took 91 ms instead of 21.  With |d| >= 1e-20 the products stay finite for
// coordinates up to ~1e18 and such an axis behaves as it should: "inside the slab or not".
SPT_DEV float slab_rcp(float d) {
    const float a = __builtin_fabsf(d) < 1e-20f ? __builtin_copysignf(1e-20f, d) : d;
    return __builtin_amdgcn_rcpf(a);
}
SPT_DEV f3 slab_rcp3(f3 d) { return mk3(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z)); }

// stack entries as one 8-byte LDS word per (level, lane): [level][lane] layout, conflict-free ds_write_b64 / ds_read_b64.
// The pointer carries the LDS address space explicitly: through a generic pointer the compiler merges the LDS and the
// scratch branch of a pop into one flat load behind a select of pointers (and ROCm 7.2's backend then fails on the null
// check of the address-space cast: "Illegal instruction ... V_CMP_NE_U32_e32 0, $src_shared_base").
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
SPT_DEV lds_u64* sstack_ptr(uint32_t level) {
    return (lds_u64*)((__attribute__((address_space(3))) uint32_t*)spt_lds_stack) + (level * blockDim.x + threadIdx.x);
}

template <bool kClosest, bool kCount>
struct SWalker {
    f3 wo, wd;                 // world ray
    f3 ro, rd, rinv;           // ray of the current space (world in the TLAS, object space inside a BLAS)
    float t_min;
    uint32_t cur;              // node index (< kLeaf), leaf ref (kLeaf | count << 27 | first) or kNoRef
    uint32_t blas_base;        // stack height at BLAS entry, kInTlas in the TLAS
    uint32_t inst;             // instance of the current BLAS
    uint32_t sp;
    DHit h;                    // kClosest: best hit (h.t = the limit); else h.t = t_max and h.inst >= 0 means "occluded"
    bool done;
    LaneVisits vc;

    SPT_DEV bool passes(float t0) const { return kClosest ? t0 <= h.t : t0 < h.t; }
    // `spill`: kSpillStack private entries owned by the kernel, handed to every call (kept out of this struct: a stored
    // pointer to them becomes a generic pointer and every stack access a flat load with aperture checks)
    SPT_DEV void push(uint32_t ref, float t0, uint2* spill) {
        if (sp < kLdsStack) *sstack_ptr(sp) = (unsigned long long)ref | ((unsigned long long)__float_as_uint(t0) << 32);
        else if (sp < kLdsStack + kSpillStack) spill[sp - kLdsStack] = make_uint2(ref, __float_as_uint(t0));
        else return;   // cannot happen: the builder bounds the pending entries (build_n4's stack_need)
        ++sp;
    }
    SPT_DEV uint2 pop_entry(const uint2* spill) {
        --sp;
        uint2 e;     // two plain loads, one per address space (a ?: over the two lvalues would make a generic pointer)
        if (sp < kLdsStack) {
            const unsigned long long v = *sstack_ptr(sp);
            e = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
        } else {
            e = spill[sp - kLdsStack];
        }
        return e;
    }
    // next subtree that can still matter; leaving a BLAS restores the world ray; an empty stack ends the walk
    SPT_DEV void pop_next(const uint2* spill) {
        cur = kNoRef;
        while (true) {
            if (blas_base != kInTlas && sp == blas_base) {
                blas_base = kInTlas;
                ro = wo; rd = wd;
                rinv = slab_rcp3(wd);
            }
            if (sp == 0u) { done = true; return; }
            const uint2 e = pop_entry(spill);
            if (passes(__uint_as_float(e.y))) { cur = e.x; return; }
        }
    }

    SPT_DEV void begin(const DScene& sc, const DRay& r, float t_max) {
        wo = r.o; wd = r.d; t_min = r.t_min;
        ro = wo; rd = wd;
        rinv = slab_rcp3(wd);
        h.t = t_max; h.inst = -1; h.prim = -1; h.v = 0.0f; h.w = 0.0f;
        sp = 0u;
        blas_base = kInTlas;
        inst = 0u;
        done = false;
        cur = sc.s_root;
        if (sc.n_instances == 0u) { done = true; cur = kNoRef; return; }
        float t0;
        if (!box_test(make_float4(sc.tlas_lo[0], sc.tlas_lo[1], sc.tlas_lo[2], 0.0f), make_float4(sc.tlas_hi[0], sc.tlas_hi[1], sc.tlas_hi[2], 0.0f), &t0)) {
            done = true;
            cur = kNoRef;
        }
    }

    // relaxed slab test of a full-precision box against the current ray
    SPT_DEV bool box_test(float4 lo, float4 hi, float* t0_out) const {
        const float x0 = (lo.x - ro.x) * rinv.x, x1 = (hi.x - ro.x) * rinv.x;
        const float y0 = (lo.y - ro.y) * rinv.y, y1 = (hi.y - ro.y) * rinv.y;
        const float z0 = (lo.z - ro.z) * rinv.z, z1 = (hi.z - ro.z) * rinv.z;
        const float t0 = max3f(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1), __builtin_fminf(z0, z1)) * kRelaxLo;
        const float t1 = min3f(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1), __builtin_fmaxf(z0, z1)) * kRelaxHi;
        *t0_out = t0;
        const bool empty = (lo.x > hi.x) | (lo.y > hi.y) | (lo.z > hi.z);
        return !empty & (t0 <= t1) & (t1 > t_min) & passes(t0);
    }

    // one compressed 4-wide node: test the children, descend into the nearest, push the others far-to-near
    SPT_DEV void node_step(const DScene& sc, uint2* spill) {
        const float4* np = sc.geo + (sc.o_blas + 4u * cur);
        const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        count_node<kCount>(&vc);
        const uint32_t meta = __float_as_uint(n0.w);
        const uint32_t n_child = meta >> 24;
        // slab distances as affine functions of the quantised coordinate: t = q * s + a
        const float sx = spt_u2f((meta & 0xffu) << 23) * rinv.x, sy = spt_u2f(((meta >> 8) & 0xffu) << 23) * rinv.y, sz = spt_u2f(((meta >> 16) & 0xffu) << 23) * rinv.z;
        const float ax = (n0.x - ro.x) * rinv.x, ay = (n0.y - ro.y) * rinv.y, az = (n0.z - ro.z) * rinv.z;
        const uint32_t qlx = __float_as_uint(n1.x), qly = __float_as_uint(n1.y), qlz = __float_as_uint(n1.z), qhx = __float_as_uint(n1.w);
        const uint32_t qhy = __float_as_uint(n2.x), qhz = __float_as_uint(n2.y);
        // the plane a ray reaches first along an axis is the low one for a positive direction, the high one otherwise
        const bool nx = rinv.x < 0.0f, ny = rinv.y < 0.0f, nz = rinv.z < 0.0f;
        const uint32_t qnx = nx ? qhx : qlx, qfx = nx ? qlx : qhx;
        const uint32_t qny = ny ? qhy : qly, qfy = ny ? qly : qhy;
        const uint32_t qnz = nz ? qhz : qlz, qfz = nz ? qlz : qhz;
        float key[4];
        uint32_t ref[4] = {__float_as_uint(n2.z), __float_as_uint(n2.w), __float_as_uint(n3.x), __float_as_uint(n3.y)};
        // The relative relaxation covers the rounding of a distance of its own size.  q * s + a, however, is the sum of two
        // terms that can each be far larger than the distance (a ray that grazes a slab from inside the node: a = (p - o) / d
        // and q * s = (plane - p) / d cancel), and its error is an ulp of THOSE: a plane at distance ~0 could come out at
        // +2e-3 for 1 / d = 1e4 and cull a node whose hit lies closer than that (fuzz seed 3034 of round 2: two pixels of a
        // scene with surfaces resting on each other).  So the interval is also widened by an absolute bound on that error,
        // 2^-22 of the largest |a| + 255 |s| of the three axes, once per node.
        const float eabs = max3f(__builtin_fabsf(ax) + 255.0f * __builtin_fabsf(sx), __builtin_fabsf(ay) + 255.0f * __builtin_fabsf(sy),
                                 __builtin_fabsf(az) + 255.0f * __builtin_fabsf(sz)) * 2.3841858e-7f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t0 = max3f(__builtin_fmaf(ubyte_f32(qnx, c), sx, ax), __builtin_fmaf(ubyte_f32(qny, c), sy, ay), __builtin_fmaf(ubyte_f32(qnz, c), sz, az)) * kRelaxLo - eabs;
            const float t1 = min3f(__builtin_fmaf(ubyte_f32(qfx, c), sx, ax), __builtin_fmaf(ubyte_f32(qfy, c), sy, ay), __builtin_fmaf(ubyte_f32(qfz, c), sz, az)) * kRelaxHi + eabs;
            const bool hit = ((uint32_t)c < n_child) & (t0 <= t1) & (t1 > t_min) & passes(t0);
            key[c] = hit ? t0 : spt_inf();
        }
#define SPT_CX(a, b)                                                   \
        if (key[a] > key[b]) {                                         \
            const float tk = key[a]; key[a] = key[b]; key[b] = tk;     \
            const uint32_t tr = ref[a]; ref[a] = ref[b]; ref[b] = tr;  \
        }
        SPT_CX(0, 1) SPT_CX(2, 3) SPT_CX(0, 2) SPT_CX(1, 3) SPT_CX(1, 2)
#undef SPT_CX
#pragma unroll
        for (int c = 3; c >= 1; --c)
            if (key[c] < spt_inf()) push(ref[c], key[c], spill);
        if (key[0] < spt_inf()) cur = ref[0];
        else pop_next(spill);
    }

    // a BLAS leaf: its triangles are requested two at a time (a leaf of the device-built trees holds 1 - 4, most hold
    // 1 or 2; four at once were 48 registers of loads in flight, which cost the kernel its fourth wave per SIMD)
    SPT_DEV void tri_leaf_step(const DScene& sc, const uint2* spill) {
        const uint32_t first = leaf_first(cur), count = leaf_count(cur);
        const float4* tp = sc.geo + (sc.o_tri + 3u * first);
        DRay orr;
        orr.o = ro; orr.d = rd; orr.t_min = t_min;
#pragma unroll
        for (uint32_t base = 0; base < 4u; base += 2u) {
            if (base >= count || done) break;
            float4 ta[2], tb[2], tc[2];
#pragma unroll
            for (uint32_t i = 0; i < 2u; ++i)
                if (base + i < count) { ta[i] = tp[3u * (base + i)]; tb[i] = tp[3u * (base + i) + 1u]; tc[i] = tp[3u * (base + i) + 2u]; }
#pragma unroll
            for (uint32_t i = 0; i < 2u; ++i) {
                if (base + i < count && !done) {
                    float t, v, w;
                    count_tri<kCount>(&vc);
                    const bool ok = tri_test_edges(ta[i], tb[i], tc[i], orr, &t, &v, &w);
                    const int32_t id = __float_as_int(ta[i].w);
                    if (kClosest) {
                        if (ok && t > t_min && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, id, h)))) {  // triangle.rs:187
                            h.t = t; h.inst = (int32_t)inst; h.prim = id; h.v = v; h.w = w;
                        }
                    } else if (ok && t > t_min && t < h.t) {
                        h.inst = (int32_t)inst;
                        done = true;
                    }
                }
            }
        }
        if (!done) pop_next(spill);
    }
    // leaves of more than 4 triangles (only the caller's trees under SPT_REFERENCE_BVH=1 can have them)
    SPT_DEV void tri_leaf_long(const DScene& sc, const uint2* spill) {
        const uint32_t first = leaf_first(cur), count = leaf_count(cur);
        DRay orr;
        orr.o = ro; orr.d = rd; orr.t_min = t_min;
        for (uint32_t i = first; i < first + count && !done; ++i) {
            float t, v, w;
            int32_t id;
            count_tri<kCount>(&vc);
            const bool ok = tri_test_geo<false>(sc, i, orr, &t, &v, &w, &id);
            if (kClosest) {
                if (ok && t > t_min && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, id, h)))) {
                    h.t = t; h.inst = (int32_t)inst; h.prim = id; h.v = v; h.w = w;
                }
            } else if (ok && t > t_min && t < h.t) {
                h.inst = (int32_t)inst;
                done = true;
            }
        }
        if (!done) pop_next(spill);
    }

    // a TLAS leaf: transform the ray into the instance (Instance::intersect, instance.rs:88-109) and either test its
    // sphere / patch right away or enter its BLAS
    SPT_DEV void instance_step(const DScene& sc, uint2* spill) {
        const uint32_t slot = leaf_first(cur), count = leaf_count(cur);
        if (count > 1u) push(kLeaf | ((count - 1u) << 27) | (slot + 1u), -spt_inf(), spill);   // the leaf's other instances: next
        const float4* ip = sc.geo + (sc.o_sinst + 6u * slot);
        const float4 m0 = ip[0], m1 = ip[1], m2 = ip[2], k = ip[3], b0 = ip[4], b1 = ip[5];
        count_inst<kCount>(&vc);
        const float inv[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
        const uint32_t prim_type = __float_as_uint(k.y), prim_id = __float_as_uint(k.z);
        inst = __float_as_uint(k.x);
        DRay orr;
        orr.o = xf_point(inv, wo);
        orr.d = xf_vector(inv, wd);   // not renormalised: t is shared between the spaces (ray.rs:33-41)
        orr.t_min = t_min;
        if (prim_type == SPT_PRIM_SPHERE) {
            float mn, mx;
            const bool roots = sphere_roots(b0, orr, &mn, &mx);
            if (kClosest) {
                if (roots) {
                    const float t = (mn < orr.t_min) ? mx : mn;   // sphere.rs:61
                    if (orr.t_min < t && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, (int32_t)prim_id, h)))) {
                        h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = 0.0f; h.w = 0.0f;
                    }
                }
            } else if (roots && mn < h.t && mx > orr.t_min) {     // sphere.rs:51-56
                h.inst = (int32_t)inst;
                done = true;
                return;
            }
            pop_next(spill);
            return;
        }
#if SPT_WITH_BEZIER
        if (prim_type == SPT_PRIM_BEZIER) {   // bezier.rs:152-174; the patch parameters ride in the hit's (v, w)
            float u, v, t;
            const bool got = bezier_intersect_ray(sc.bez + 16u * prim_id, orr, &u, &v, &t) && t > orr.t_min;
            if (kClosest) {
                if (got && (t < h.t || (t == h.t && h.inst >= 0 && key_less((int32_t)inst, (int32_t)prim_id, h)))) {
                    h.t = t; h.inst = (int32_t)inst; h.prim = (int32_t)prim_id; h.v = u; h.w = v;
                }
            } else if (got && t < h.t) {
                h.inst = (int32_t)inst;
                done = true;
                return;
            }
            pop_next(spill);
            return;
        }
#endif
        // a mesh: the root box of its BLAS in object space
        ro = orr.o; rd = orr.d;
        rinv = slab_rcp3(rd);
        blas_base = sp;
        float t0;
        if (box_test(b0, b1, &t0)) cur = __float_as_uint(b0.w);
        else pop_next(spill);     // sp == blas_base: leaves the instance again
    }

    // if-if: one node step for every lane that holds an inner node, then one leaf step for every lane that holds a leaf
    // (including the lanes whose node step just produced one); `rounds` times
    SPT_DEV void run_ifif(const DScene& sc, uint32_t rounds, uint2* spill) {
        for (uint32_t r = 0; r < rounds; ++r) {
            if (__ballot(!done) == 0ull) break;
            if (!done && cur < kLeaf) node_step(sc, spill);
            if (!done && cur != kNoRef && cur >= kLeaf) {
                if (blas_base == kInTlas) instance_step(sc, spill);
                else if (leaf_count(cur) <= 4u) tri_leaf_step(sc, spill);
                else tri_leaf_long(sc, spill);
            }
        }
    }

    // while-while: node loop until no lane of the wave holds an inner node, then one leaf round; `rounds` times
    SPT_DEV void run(const DScene& sc, uint32_t rounds, uint2* spill) {
        if (rounds & 0x100u) { run_ifif(sc, rounds & 0xffu, spill); return; }
        for (uint32_t r = 0; r < rounds; ++r) {
            // (bounded: a descent is at most the tree depth long, pops included a few times that; the bound only makes
            // sure that a corrupt tree can never keep a wave in here for ever)
            for (uint32_t guard = 0; guard < 4096u; ++guard) {
                const bool inner = !done && cur < kLeaf;
                if (__ballot(inner) == 0ull) break;
                if (inner) node_step(sc, spill);
            }
            const bool leaf = !done && cur != kNoRef;     // every live lane holds a leaf (or nothing) here
            if (__ballot(leaf) == 0ull) break;
            if (leaf) {
                if (blas_base == kInTlas) instance_step(sc, spill);
                else if (leaf_count(cur) <= 4u) tri_leaf_step(sc, spill);
                else tri_leaf_long(sc, spill);
            }
        }
    }
};
