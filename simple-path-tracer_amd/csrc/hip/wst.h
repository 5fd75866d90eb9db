// Kind-sorted traversal for scenes whose geometry does not fit LDS ("workgroup-sorted traversal", WST).
//
// Same reference functions and the same hit definition as trace.h / stream.h (BvhAccel::{intersect, intersect_test}
// bvh.rs:237-283, Instance::intersect instance.rs:88-109, Triangle::intersect_ray triangle.rs:124-147, Sphere
// sphere.rs:25-84): primitive tests are the bit-exact ones of trace.h, box tests are the relaxed ones of stream.h.
//
// Why.  Measured on cfg5 (profiles/r02_pmc_cfg5_*): the per-lane walkers - state machine (trace.h) or while-while /
// if-if (stream.h) alike - run the extension rays at 28 % VALU lane utilisation with the VALU issue port ~80 % busy:
// a wave holds rays that need a NODE test, rays that need TRIANGLE tests and rays that need an INSTANCE transform, and
// executes the three code paths one after the other for a fraction of its lanes each.  The primary rays of the same
// scene do the same work per ray (5 nodes, 2 triangles, 1 instance) at 95 % utilisation and 5 x the rays per second.
// So the fix is not a cheaper step but fuller waves:
//   * a workgroup (4 waves) owns a pool of 256 rays whose traversal state lives in LDS, not in registers;
//   * every round each ray advances by exactly ONE step (one 64 - 96 byte record: a node, a leaf's triangles or an
//     instance) - and the rays are first SORTED BY THE KIND of step they need (ballot / popc compaction into three LDS
//     lists), so thread t of the workgroup executes item t of "all node steps, then all triangle steps, then all
//     instance steps": at most two of the four waves run more than one kind, the others run one kind with 64 lanes;
//   * finished rays are retired and replaced by their slot's owner thread at the start of the next round; the
//     workgroups of a queue shard split its rays statically (64-ray chunks dealt round-robin), so taking a ray is an
//     LDS counter, not a global atomic; the append of kept paths is the usual wave-aggregated push, issued one
//     round before its result is needed.
// Cost of the indirection: ~20 LDS instructions per step to load / store the state, two workgroup barriers per round.
#pragma once
#include "stream.h"

constexpr uint32_t kWstRays = 256;            // pool slots per workgroup = threads per workgroup
constexpr uint32_t kWstBlocksPerShard = 12;   // persistent workgroups per queue shard (grid = 64 shards x 12 = 3 per CU: 46 KB of LDS each)
constexpr uint32_t kWstStack = 16;            // stack levels per ray in LDS; deeper levels (up to kLdsStack + kSpillStack in all) in global memory
enum : uint32_t { WST_FREE = 0, WST_NODE = 1, WST_TRI = 2, WST_INST = 3, WST_DONE = 4 };

typedef float wst_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t wst_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) wst_f4 lds_f4;
typedef __attribute__((address_space(3))) wst_u4 lds_u4;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;

// LDS layout (bytes): five 16-byte state planes, the stack, three lists, two counter sets, kinds
constexpr uint32_t kWstOffA = 0;                                    // (o.xyz, t_min)          ray of the current space
constexpr uint32_t kWstOffB = kWstOffA + 16u * kWstRays;            // (d.xyz, limit)          limit = best t (closest) / t_max (any)
constexpr uint32_t kWstOffI = kWstOffB + 16u * kWstRays;            // (1/d for the slabs, current instance)
constexpr uint32_t kWstOffC = kWstOffI + 16u * kWstRays;            // (cur, sp, blas_base, queue index of the ray)
constexpr uint32_t kWstOffD = kWstOffC + 16u * kWstRays;            // (v, w, prim, inst) of the best hit; inst >= 0: hit / occluded
// The stack.  A 4-wide node leaves up to three pending children, so a walk through the 1 M-triangle BLAS holds 10 - 15
// entries: with 8 levels in LDS most walks spilled, and here - unlike a per-lane walker's scratch - a spilled pop is a
// global round trip in the middle of a step that 255 other rays wait for at the barrier (first version: 180 ms instead of
// 94 for the extension rays of cfg5, 76 % of the wave cycles waiting).  16 levels at 6 bytes: the ref and the entry
// distance as a bfloat16 rounded DOWN (it only culls: a smaller distance culls later, never wrongly).
constexpr uint32_t kWstOffStack = kWstOffD + 16u * kWstRays;        // u32 [kWstStack][kWstRays]  refs
constexpr uint32_t kWstOffStackT = kWstOffStack + 4u * kWstStack * kWstRays;   // u16 [kWstStack][kWstRays]  entry distances
constexpr uint32_t kWstOffList = kWstOffStackT + 2u * kWstStack * kWstRays;    // u16 [3][kWstRays]
constexpr uint32_t kWstOffCnt = kWstOffList + 2u * 3u * kWstRays;   // u32 [2][4]: per-kind list lengths, double-buffered by round parity
constexpr uint32_t kWstOffTake = kWstOffCnt + 32u;                  // u32: rays this workgroup has taken from its share of the queue
constexpr uint32_t kWstOffKind = kWstOffTake + 16u;                 // u8 [kWstRays]
constexpr uint32_t kWstLdsBytes = kWstOffKind + kWstRays;

SPT_DEV __attribute__((address_space(3))) char* wst_lds() { return (__attribute__((address_space(3))) char*)spt_lds_stack; }
SPT_DEV lds_f4* wst_plane(uint32_t off) { return (lds_f4*)(wst_lds() + off); }
SPT_DEV lds_u4* wst_c() { return (lds_u4*)(wst_lds() + kWstOffC); }
SPT_DEV lds_u32* wst_stack_ref(uint32_t level, uint32_t slot) { return (lds_u32*)(wst_lds() + kWstOffStack) + (level * kWstRays + slot); }
SPT_DEV lds_u16* wst_stack_t(uint32_t level, uint32_t slot) { return (lds_u16*)(wst_lds() + kWstOffStackT) + (level * kWstRays + slot); }
// entry distance -> 16 bits, never larger than the value: negative (always passes) -> -inf, else truncated mantissa
SPT_DEV uint16_t wst_t_down(float t0) { return t0 < 0.0f ? (uint16_t)0xff80u : (uint16_t)(__float_as_uint(t0) >> 16); }
SPT_DEV lds_u16* wst_list(uint32_t kind) { return (lds_u16*)(wst_lds() + kWstOffList) + (kind - 1u) * kWstRays; }
SPT_DEV lds_u32* wst_cnt(uint32_t parity) { return (lds_u32*)(wst_lds() + kWstOffCnt) + 4u * parity; }
SPT_DEV lds_u32* wst_take() { return (lds_u32*)(wst_lds() + kWstOffTake); }
SPT_DEV lds_u8* wst_kind() { return (lds_u8*)(wst_lds() + kWstOffKind); }

// One ray's state while a thread works on it (loaded from / stored to the slot's LDS planes)
template <bool kClosest, bool kCount>
struct WstRay {
    f3 ro, rd, rinv;
    float t_min, limit;
    uint32_t cur, sp, blas_base, idx, inst;
    float hv, hw;
    int32_t hprim, hinst;
    uint32_t slot;
    uint2* ovf;                // this slot's kSpillStack overflow entries in global memory
    LaneVisits* vc;

    SPT_DEV void load(uint32_t s, uint2* ovf_base) {
        slot = s;
        ovf = ovf_base + ((size_t)blockIdx.x * kWstRays + s) * kSpillStack;
        const wst_f4 a = wst_plane(kWstOffA)[s], b = wst_plane(kWstOffB)[s], iv = wst_plane(kWstOffI)[s], d = wst_plane(kWstOffD)[s];
        const wst_u4 c = wst_c()[s];
        ro = mk3(a.x, a.y, a.z); t_min = a.w;
        rd = mk3(b.x, b.y, b.z); limit = b.w;
        rinv = mk3(iv.x, iv.y, iv.z); inst = __float_as_uint(iv.w);
        cur = c.x; sp = c.y; blas_base = c.z; idx = c.w;
        hv = d.x; hw = d.y; hprim = __float_as_int(d.z); hinst = __float_as_int(d.w);
    }
    SPT_DEV void store_ray() const {   // the ray of the current space changed (instance entry / exit)
        wst_plane(kWstOffA)[slot] = wst_f4{ro.x, ro.y, ro.z, t_min};
        wst_plane(kWstOffI)[slot] = wst_f4{rinv.x, rinv.y, rinv.z, __uint_as_float(inst)};
    }
    SPT_DEV void store_dir_limit() const { wst_plane(kWstOffB)[slot] = wst_f4{rd.x, rd.y, rd.z, limit}; }
    SPT_DEV void store_hit() const { wst_plane(kWstOffD)[slot] = wst_f4{hv, hw, __int_as_float(hprim), __int_as_float(hinst)}; }
    SPT_DEV void store_ctl() const { wst_c()[slot] = wst_u4{cur, sp, blas_base, idx}; }
    // the kind of step the ray needs next
    SPT_DEV uint32_t next_kind(bool done) const {
        if (done) return WST_DONE;
        if (cur < kLeaf) return WST_NODE;
        return blas_base == kInTlas ? WST_INST : WST_TRI;
    }

    SPT_DEV bool passes(float t0) const { return kClosest ? t0 <= limit : t0 < limit; }
    SPT_DEV void push(uint32_t ref, float t0) {
        if (sp < kWstStack) { *wst_stack_ref(sp, slot) = ref; *wst_stack_t(sp, slot) = wst_t_down(t0); }
        else if (sp < kLdsStack + kSpillStack) ovf[sp - kWstStack] = make_uint2(ref, __float_as_uint(t0));
        else return;   // cannot happen: the builder bounds the pending entries (build_n4's stack_need)
        ++sp;
    }
    // Next subtree that can still matter (sets cur); leaving a BLAS restores the world ray from the ray's queue record.
    // Returns true when the walk is over.
    template <class WorldRay>
    SPT_DEV bool pop_next(WorldRay world_ray) {
        cur = kNoRef;
        while (true) {
            if (blas_base != kInTlas && sp == blas_base) {
                blas_base = kInTlas;
                if (sp == 0u) return true;          // nothing left in the TLAS either: no need for the world ray
                world_ray(idx, &ro, &rd);
                rinv = slab_rcp3(rd);
                store_ray();
                store_dir_limit();
            }
            if (sp == 0u) return true;
            --sp;
            uint2 e;
            if (sp < kWstStack) e = make_uint2(*wst_stack_ref(sp, slot), (uint32_t)*wst_stack_t(sp, slot) << 16);
            else e = ovf[sp - kWstStack];
            if (passes(__uint_as_float(e.y))) { cur = e.x; return false; }
        }
    }

    SPT_DEV bool box_test(float4 lo, float4 hi) const {   // relaxed slab test of a full-precision box (stream.h)
        const float x0 = (lo.x - ro.x) * rinv.x, x1 = (hi.x - ro.x) * rinv.x;
        const float y0 = (lo.y - ro.y) * rinv.y, y1 = (hi.y - ro.y) * rinv.y;
        const float z0 = (lo.z - ro.z) * rinv.z, z1 = (hi.z - ro.z) * rinv.z;
        const float t0 = max3f(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1), __builtin_fminf(z0, z1)) * kRelaxLo;
        const float t1 = min3f(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1), __builtin_fmaxf(z0, z1)) * kRelaxHi;
        const bool empty = (lo.x > hi.x) | (lo.y > hi.y) | (lo.z > hi.z);
        return !empty & (t0 <= t1) & (t1 > t_min) & passes(t0);
    }

    // ---- the three kinds of step.  Each returns true when the walk is over. ----

    // one compressed 4-wide node (stream.h's node test): descend into the nearest child, push the others far-to-near
    template <class WorldRay>
    SPT_DEV bool node_step(const DScene& sc, WorldRay world_ray) {
        const float4* np = sc.geo + (sc.o_blas + 4u * cur);
        const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        count_node<kCount>(vc);
        const uint32_t meta = __float_as_uint(n0.w);
        const uint32_t n_child = meta >> 24;
        const float sx = spt_u2f((meta & 0xffu) << 23) * rinv.x, sy = spt_u2f(((meta >> 8) & 0xffu) << 23) * rinv.y, sz = spt_u2f(((meta >> 16) & 0xffu) << 23) * rinv.z;
        const float ax = (n0.x - ro.x) * rinv.x, ay = (n0.y - ro.y) * rinv.y, az = (n0.z - ro.z) * rinv.z;
        const uint32_t qlx = __float_as_uint(n1.x), qly = __float_as_uint(n1.y), qlz = __float_as_uint(n1.z), qhx = __float_as_uint(n1.w);
        const uint32_t qhy = __float_as_uint(n2.x), qhz = __float_as_uint(n2.y);
        const bool nx = rinv.x < 0.0f, ny = rinv.y < 0.0f, nz = rinv.z < 0.0f;
        const uint32_t qnx = nx ? qhx : qlx, qfx = nx ? qlx : qhx;
        const uint32_t qny = ny ? qhy : qly, qfy = ny ? qly : qhy;
        const uint32_t qnz = nz ? qhz : qlz, qfz = nz ? qlz : qhz;
        float key[4];
        uint32_t ref[4] = {__float_as_uint(n2.z), __float_as_uint(n2.w), __float_as_uint(n3.x), __float_as_uint(n3.y)};
        // (absolute bound on the rounding of q * s + a: see SWalker::node_step in stream.h)
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
            if (key[c] < spt_inf()) push(ref[c], key[c]);
        if (key[0] < spt_inf()) { cur = ref[0]; return false; }
        return pop_next(world_ray);
    }

    // a BLAS leaf: all its triangles are requested before the first is tested (triangle.rs:124-147, 187)
    template <class WorldRay>
    SPT_DEV bool tri_step(const DScene& sc, WorldRay world_ray) {
        const uint32_t first = leaf_first(cur), count = leaf_count(cur);
        DRay orr;
        orr.o = ro; orr.d = rd; orr.t_min = t_min;
        bool found = false;
        if (count <= 4u) {
            const float4* tp = sc.geo + (sc.o_tri + 3u * first);
            float4 ta[4], tb[4], tc[4];
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i)
                if (i < count) { ta[i] = tp[3u * i]; tb[i] = tp[3u * i + 1u]; tc[i] = tp[3u * i + 2u]; }
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                if (i < count && !(found && !kClosest)) {
                    float t, v, w;
                    count_tri<kCount>(vc);
                    const bool ok = tri_test_edges(ta[i], tb[i], tc[i], orr, &t, &v, &w);
                    const int32_t id = __float_as_int(ta[i].w);
                    if (kClosest) {
                        if (ok && t > t_min && (t < limit || (t == limit && hinst >= 0 && ((int32_t)inst < hinst || ((int32_t)inst == hinst && id < hprim))))) {
                            limit = t; hinst = (int32_t)inst; hprim = id; hv = v; hw = w; found = true;
                        }
                    } else if (ok && t > t_min && t < limit) {
                        hinst = (int32_t)inst; found = true;
                    }
                }
            }
        } else {   // leaves of more than 4 triangles: only the caller's trees under SPT_REFERENCE_BVH=1 can have them
            for (uint32_t i = first; i < first + count && !(found && !kClosest); ++i) {
                float t, v, w;
                int32_t id;
                count_tri<kCount>(vc);
                const bool ok = tri_test_geo<false>(sc, i, orr, &t, &v, &w, &id);
                if (kClosest) {
                    if (ok && t > t_min && (t < limit || (t == limit && hinst >= 0 && ((int32_t)inst < hinst || ((int32_t)inst == hinst && id < hprim))))) {
                        limit = t; hinst = (int32_t)inst; hprim = id; hv = v; hw = w; found = true;
                    }
                } else if (ok && t > t_min && t < limit) {
                    hinst = (int32_t)inst; found = true;
                }
            }
        }
        if (found) {
            store_hit();
            if (!kClosest) return true;          // any hit ends the walk
            store_dir_limit();
        }
        return pop_next(world_ray);
    }

    // a TLAS leaf: Instance::intersect (instance.rs:88-109): the ray goes into object space; a sphere / patch is tested
    // right away, a mesh's BLAS is entered through its root box
    template <class WorldRay>
    SPT_DEV bool inst_step(const DScene& sc, WorldRay world_ray) {
        const uint32_t slot_i = leaf_first(cur), count = leaf_count(cur);
        if (count > 1u) push(kLeaf | ((count - 1u) << 27) | (slot_i + 1u), -spt_inf());   // the leaf's other instances: next
        const float4* ip = sc.geo + (sc.o_sinst + 6u * slot_i);
        const float4 m0 = ip[0], m1 = ip[1], m2 = ip[2], k = ip[3], b0 = ip[4], b1 = ip[5];
        count_inst<kCount>(vc);
        const float inv[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
        const uint32_t prim_type = __float_as_uint(k.y), prim_id = __float_as_uint(k.z);
        const uint32_t ii = __float_as_uint(k.x);
        DRay orr;                     // (ro, rd) is the world ray here: the walk is in the TLAS
        orr.o = xf_point(inv, ro);
        orr.d = xf_vector(inv, rd);   // not renormalised: t is shared between the spaces (ray.rs:33-41)
        orr.t_min = t_min;
        if (prim_type == SPT_PRIM_SPHERE) {
            float mn, mx;
            const bool roots = sphere_roots(b0, orr, &mn, &mx);
            if (kClosest) {
                if (roots) {
                    const float t = (mn < orr.t_min) ? mx : mn;   // sphere.rs:61
                    if (orr.t_min < t && (t < limit || (t == limit && hinst >= 0 && ((int32_t)ii < hinst || ((int32_t)ii == hinst && (int32_t)prim_id < hprim))))) {
                        limit = t; hinst = (int32_t)ii; hprim = (int32_t)prim_id; hv = 0.0f; hw = 0.0f;
                        store_hit();
                        store_dir_limit();
                    }
                }
            } else if (roots && mn < limit && mx > orr.t_min) {     // sphere.rs:51-56
                hinst = (int32_t)ii;
                store_hit();
                return true;
            }
            return pop_next(world_ray);
        }
#if SPT_WITH_BEZIER
        if (prim_type == SPT_PRIM_BEZIER) {   // bezier.rs:152-174; the patch parameters ride in the hit's (v, w)
            float u, v, t;
            const bool got = bezier_intersect_ray(sc.bez + 16u * prim_id, orr, &u, &v, &t) && t > orr.t_min;
            if (kClosest) {
                if (got && (t < limit || (t == limit && hinst >= 0 && ((int32_t)ii < hinst || ((int32_t)ii == hinst && (int32_t)prim_id < hprim))))) {
                    limit = t; hinst = (int32_t)ii; hprim = (int32_t)prim_id; hv = u; hw = v;
                    store_hit();
                    store_dir_limit();
                }
            } else if (got && t < limit) {
                hinst = (int32_t)ii;
                store_hit();
                return true;
            }
            return pop_next(world_ray);
        }
#endif
        // a mesh: enter its BLAS if the ray reaches the root box in object space
        const f3 wo = ro, wd = rd, winv = rinv;
        ro = orr.o; rd = orr.d;
        rinv = slab_rcp3(rd);
        if (box_test(b0, b1)) {
            inst = ii;
            blas_base = sp;
            cur = __float_as_uint(b0.w);
            store_ray();
            store_dir_limit();
            return false;
        }
        ro = wo; rd = wd; rinv = winv;    // not entered: the LDS copy still holds the world ray
        return pop_next(world_ray);
    }
};

// What differs between extension rays (closest hit, the result is a hit record of the next bounce's shade queue) and
// shadow rays (any hit, the result is the light sample's contribution added to the sample's radiance slot)
struct WstExtend {
    static constexpr bool kClosest = true;
    SPT_DEV static uint32_t count(const RenderCtx& rc, uint32_t bounce, uint32_t shard) { return *q_count(rc.counts, bounce, Q_EXT, shard); }
    SPT_DEV static void ray(const RenderCtx& rc, uint32_t idx, f3* o, f3* d, float* t_min, float* t_max) {
        const float4 a = rc.qb.o_tmin[idx], b = rc.qb.d_pdf[idx];
        *o = mk3(a); *d = mk3(b); *t_min = a.w; *t_max = SPT_F32_MAX;
    }
};
struct WstShadow {
    static constexpr bool kClosest = false;
    SPT_DEV static uint32_t count(const RenderCtx& rc, uint32_t bounce, uint32_t shard) { return *q_count(rc.counts, bounce, Q_SHADOW, shard); }
    SPT_DEV static void ray(const RenderCtx& rc, uint32_t idx, f3* o, f3* d, float* t_min, float* t_max) {
        const float4 a = rc.shadow.o_tmin[idx], b = rc.shadow.d_tmax[idx];
        *o = mk3(a); *d = mk3(b); *t_min = a.w; *t_max = b.w;
    }
};

// grid = kShards * kWstBlocksPerShard workgroups of kWstRays threads, kWstLdsBytes of dynamic LDS
template <class Policy, bool kCount>
__global__ void __launch_bounds__(256, 4) k_trace_wst(DScene sc, RenderCtx rc, uint32_t bounce, uint2* ovf_base) {
    constexpr bool kClosest = Policy::kClosest;
    const uint32_t shard = blockIdx.x % kShards, part = blockIdx.x / kShards, parts = gridDim.x / kShards;
    const uint32_t n = Policy::count(rc, bounce, shard);
    const uint32_t qbase = shard * rc.shard_cap;
    const uint32_t t = threadIdx.x;
    uint32_t* next_count = q_count(rc.counts, bounce + 1, Q_HIT, shard);
    LaneVisits vc{0u, 0u, 0u};
    wst_kind()[t] = (uint8_t)WST_FREE;
    if (t < 8u) wst_cnt(0)[t] = 0u;      // both counter sets
    if (t == 0u) *wst_take() = 0u;
    __syncthreads();
    auto world_ray = [&](uint32_t idx, f3* o, f3* d) {
        float t_min, t_max;
        Policy::ray(rc, idx, o, d, &t_min, &t_max);
    };
    // a kept path whose append was issued in the previous round (closest only)
    PendingPush pend;
    pend.mask = 0ull; pend.base_raw = 0u; pend.leader = 0u;
    bool pend_keep = false;
    float4 pend_hit = make_float4(0, 0, 0, 0);
    uint2 pend_is = make_uint2(0u, 0u);
    // rays of this shard are dealt to its workgroups in chunks of 64: workgroup `part` owns chunks part, part + parts, ...
    const uint32_t my_chunks = n == 0u ? 0u : (((n + 63u) / 64u) + parts - 1u - part) / parts;   // chunks c < ceil(n / 64) with c % parts == part
    const uint32_t my_rays = my_chunks * 64u;    // upper bound; the last chunk may be partly past n
    for (uint32_t round = 0; round < (1u << 24); ++round) {
        const uint32_t parity = round & 1u;
        // ---- (1) slot owners: retire a finished ray, take the next one
        uint32_t kind = wst_kind()[t];
        bool keep = false;
        float4 new_hit = make_float4(0, 0, 0, 0);
        uint2 new_is = make_uint2(0u, 0u);
        if (kind == WST_DONE) {
            const wst_f4 d = wst_plane(kWstOffD)[t];
            const uint32_t idx = wst_c()[t].w;
            const int32_t hinst = __float_as_int(d.w);
            if (kClosest) {
                const float limit = wst_plane(kWstOffB)[t].w;
                const bool in_medium = (__float_as_uint(rc.qb.lsi_meta[idx].w) >> 8) != 0u;
                if (hinst >= 0 || in_medium) {
                    keep = true;
                    new_hit = make_float4(hinst >= 0 ? limit : SPT_F32_MAX, d.x, d.y, d.z);
                    new_is = make_uint2((uint32_t)hinst, idx);
                } else if (sc.env_w != 0u) {  // pt.rs:97-111, curr_depth > 0 here
                    const float4 b = rc.qb.d_pdf[idx], c = rc.qb.thr_slot[idx];
                    f3 env;
                    float env_pdf;
                    env_strength_pdf(sc, mk3(b), &env, &env_pdf);
                    const float weight = power_heuristic(b.w, pdf_env_light(sc) * env_pdf);
                    rad_add(rc, __float_as_uint(c.w), (mk3(c) * env) * weight);
                }
            } else if (hinst < 0) {   // not occluded: the light sample counts
                const float4 c = rc.shadow.contrib_slot[idx];
                rad_add(rc, __float_as_uint(c.w), mk3(c));
            }
            kind = WST_FREE;
        }
        if (kClosest) {
            // last round's kept paths: their slots have arrived by now; then this round's append goes out
            const uint32_t out = qbase + wave_push_finish(pend);
            if (pend_keep) {
                rc.hits.t_v_w_prim[out] = pend_hit;
                rc.hits.inst_src[out] = pend_is;
            }
            pend = wave_push_issue(keep, next_count);
            pend_keep = keep; pend_hit = new_hit; pend_is = new_is;
        }
        if (kind == WST_FREE) {
            // take the next ray of this workgroup's share: wave-aggregated LDS counter
            const unsigned long long want = __ballot(true);
            const uint32_t leader = (uint32_t)__ffsll((long long)want) - 1u;
            uint32_t base = 0u;
            if (lane_id() == leader) base = __hip_atomic_fetch_add(wst_take(), (uint32_t)__popcll(want), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = (uint32_t)__shfl((int)base, (int)leader, 64);
            const unsigned long long lt = (lane_id() == 0u) ? 0ull : (~0ull >> (64u - lane_id()));
            const uint32_t mine = base + (uint32_t)__popcll(want & lt);
            const uint32_t i = ((mine >> 6) * parts + part) * 64u + (mine & 63u);
            if (mine < my_rays && i < n) {
                const uint32_t idx = qbase + i;
                WstRay<kClosest, kCount> r;
                r.slot = t;
                r.vc = &vc;
                float t_max;
                Policy::ray(rc, idx, &r.ro, &r.rd, &r.t_min, &t_max);
                r.rinv = slab_rcp3(r.rd);
                r.limit = t_max;
                r.cur = sc.s_root; r.sp = 0u; r.blas_base = kInTlas; r.idx = idx; r.inst = 0u;
                r.hv = 0.0f; r.hw = 0.0f; r.hprim = -1; r.hinst = -1;
                const bool alive = sc.n_instances != 0u &&
                    r.box_test(make_float4(sc.tlas_lo[0], sc.tlas_lo[1], sc.tlas_lo[2], 0.0f), make_float4(sc.tlas_hi[0], sc.tlas_hi[1], sc.tlas_hi[2], 0.0f));
                r.store_ray(); r.store_dir_limit(); r.store_hit(); r.store_ctl();
                kind = r.next_kind(!alive);
            }
        }
        wst_kind()[t] = (uint8_t)kind;
        // ---- (2) sort the slots by the kind of step they need: three compacted lists
#pragma unroll
        for (uint32_t k = WST_NODE; k <= WST_INST; ++k) {
            const unsigned long long m = __ballot(kind == k);
            if (m != 0ull) {
                const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
                uint32_t base = 0u;
                if (lane_id() == leader) base = __hip_atomic_fetch_add(wst_cnt(parity) + k, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = (uint32_t)__shfl((int)base, (int)leader, 64);
                const unsigned long long lt = (lane_id() == 0u) ? 0ull : (~0ull >> (64u - lane_id()));
                if (kind == k) wst_list(k)[base + (uint32_t)__popcll(m & lt)] = (uint16_t)t;
            }
        }
        // a ray that was culled at the root (or retired nothing) and anything DONE waits for the next round's step (1)
        const bool pending_done = kind == WST_DONE;
        __syncthreads();
        const uint32_t n_node = wst_cnt(parity)[WST_NODE], n_tri = wst_cnt(parity)[WST_TRI], n_inst = wst_cnt(parity)[WST_INST];
        if (t < 4u) wst_cnt(parity ^ 1u)[t] = 0u;     // the other set is idle between two barriers: ready for the next round
        const uint32_t n_items = n_node + n_tri + n_inst;
        // the workgroup is finished when nothing is in flight, nothing waits to be retired and its share is used up
        const bool wg_idle = n_items == 0u && __syncthreads_or(pending_done) == 0 && *wst_take() >= my_rays;   // (appends still in flight are finished after the loop)
        if (wg_idle) break;
        // ---- (3) thread t runs item t of [node steps | triangle steps | instance steps]
        if (t < n_items) {
            WstRay<kClosest, kCount> r;
            r.vc = &vc;
            bool done;
            if (t < n_node) {
                r.load(wst_list(WST_NODE)[t], ovf_base);
                done = r.node_step(sc, world_ray);
            } else if (t < n_node + n_tri) {
                r.load(wst_list(WST_TRI)[t - n_node], ovf_base);
                done = r.tri_step(sc, world_ray);
            } else {
                r.load(wst_list(WST_INST)[t - n_node - n_tri], ovf_base);
                done = r.inst_step(sc, world_ray);
            }
            r.store_ctl();
            wst_kind()[r.slot] = (uint8_t)r.next_kind(done);
        }
        __syncthreads();
    }
    if (kClosest) {   // the appends of the last round
        const uint32_t out = qbase + wave_push_finish(pend);
        if (pend_keep) {
            rc.hits.t_v_w_prim[out] = pend_hit;
            rc.hits.inst_src[out] = pend_is;
        }
    }
    if (kCount) flush_visits(rc, vc, kClosest ? 2u : 1u);
}
