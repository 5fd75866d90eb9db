// Deferred Bezier-patch tests (libspt_hip_bez.so only): the (ray, patch) pairs of a whole queue, clipped by a kernel of
// their own.
//
// Why.  The patch test (bezier.h, reference src/primitive/bezier.rs:239-422) needs 1 - 3 clipping calls for a ray that
// misses and 30 - 40 for one that hits, ~1 000 instructions each.  Inside a walker the few lanes of a wave whose ray
// reached a patch clip while the others wait, and they wait for the deepest of them: the shadow / extension kernels of
// t_bezier.json ran VALU-bound at 7 - 8 % lane utilisation (profiles/r02_pmc_bezier/).  Here a walker that reaches a
// patch instance only RECORDS the pair - the ray in the patch's object space, 48 bytes - and walks on as if the patch
// were not there.  When the queue has been walked, k_bezier_pairs runs persistent waves over all pairs: every lane
// clips one pair in resumable steps (BzWalk) and takes the next pair the moment its own is decided, so a wave is full
// whatever the mix of misses and hits.  A finish kernel then completes the rays that had pairs.
//
// Same answers.  Closest hit = minimum over (t, instance, prim) of every accepted primitive test; the walk delivers
// that minimum over the non-patch primitives (it prunes with ITS best t, a superset of what the original walk visits),
// the pairs deliver the accepted patch tests, the finish kernel takes the minimum of both under the same tie rule.
// Any hit = "some accepted test inside (t_min, t_max)": a disjunction, order-free.  Rays without pairs finish inside
// the walking kernel exactly as before.  Every pair's arithmetic is bezier_intersect_ray's (same BzWalk steps).
//
// A ray's later contributions must follow its earlier ones in its radiance slot: the stages stay stream-ordered (walk,
// pairs, commit, finish on the stream the walk ran on), and within one bounce a slot receives at most one shadow term
// and one environment term, whoever adds them.
//
// MEASURED (round 2, 512^2 @ 64 spp): the walking kernels drop from 110 ms to 13 ms (t_bezier.json) - and the clipping kernel
// takes 108 ms: it runs at 27 % lane utilisation even with every lane holding a pair (the twelve hull-crossing tests, the
// split / clip / stop paths and the 1 - 40 calls per test diverge INSIDE the clipping), at 1 - 2 waves per SIMD (253 - 265
// VGPRs), and moves 42 GB of frame pushes / pops through scratch per render.  Total 146 ms as first built, 132 ms with the
// refill cadence and occupancy below (each measured), against 130 ms inline (t_catmull.json: 196 against 181 ms - without
// the pruning by closer patch hits its rays meet 1.6 x as many patches).  So
// the design the round-1 review asked for is here, bit-identical (switch sweep, fuzz seeds), and OFF by default
// (SPT_BEZ_DEFER=1 turns it on).  What it needs next is a clipping step that keeps < 128 registers live.
//
// Overflow: the pair queue holds 4 pairs per queue entry; a push beyond it is refused and the walker tests that patch
// inline, as the other kernels do - slower, never wrong.
#pragma once
#if SPT_WITH_BEZIER

// called from SWalker::instance_step at a patch instance; false = queue full (or no deferral): test inline
SPT_DEV bool bez_defer_push(BezDefer& bd, const DRay& orr, float limit, uint32_t inst, uint32_t prim_id) {
    if (bd.rec == nullptr) return false;
    const uint32_t slot = wave_push(true, bd.count);     // (divergent callers: the ballot only sees the lanes that are here)
    if (slot >= bd.cap) return false;
    float4* r = bd.rec + 3u * (size_t)slot;
    r[0] = make_float4(orr.o.x, orr.o.y, orr.o.z, orr.t_min);
    r[1] = make_float4(orr.d.x, orr.d.y, orr.d.z, limit);
    r[2] = make_float4(__uint_as_float(prim_id), __uint_as_float(bd.ray), __uint_as_float(inst), __uint_as_float(0u));
    bd.pushed += 1u;
    return true;
}

#ifndef SPT_W_BEZ
#define SPT_W_BEZ 2
#endif
constexpr uint32_t kBezRefillBelow = 32;   // refill the wave when fewer lanes than this still clip
constexpr uint32_t kBezSteps = 8;          // clipping calls between two retire / refill checks

// Persistent waves over the pair queue.  kClosest: extension rays (best candidate per ray), else shadow rays (a flag).
template <bool kClosest>
__global__ void __launch_bounds__(256, SPT_W_BEZ) k_bezier_pairs(DScene sc, BezPairs bp) {
    const uint32_t n = min(bp.ctl[0], bp.cap);
    uint32_t* cursor = bp.ctl + 1;
    BzFrame stack[kClippingMaxTimes];
    BzWalk w;
    w.finished = true;
    bool busy = false, drained = false;
    uint32_t pair = 0, ray = 0, inst = 0, prim = 0;
    float limit = 0.0f;
    for (uint32_t guard = 0; guard < (1u << 26); ++guard) {     // every wave leaves: the queue is finite, a test ends after < 2^17 calls
        const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
        if (!drained && n_busy < kBezRefillBelow) {
            const uint32_t i = wave_pull(!busy, cursor);
            if (!busy && i < n) {
                const float4* r = bp.rec + 3u * (size_t)i;
                const float4 a = r[0], b = r[1], c = r[2];
                pair = i;
                prim = __float_as_uint(c.x); ray = __float_as_uint(c.y); inst = __float_as_uint(c.z);
                limit = b.w;
                bool wanted = true;
                if (!kClosest) wanted = bp.occluded[ray] == 0;      // another pair of this ray was accepted already
                if (wanted) {
                    DRay orr;
                    orr.o = mk3(a); orr.t_min = a.w; orr.d = mk3(b);
                    w.begin(sc.bez + 16u * prim, orr);
                    busy = true;
                }
            }
            drained = __ballot(!busy && i >= n && i != 0xffffffffu) != 0ull;
        }
        if (__ballot(busy) == 0ull) {
            if (drained) break;
            continue;      // every pulled pair was skipped: pull again
        }
        for (uint32_t k = 0; k < kBezSteps; ++k)
            if (busy && !w.finished) w.step(stack);
        if (busy && w.finished) {
            const bool got = w.found && w.t > w.t_min;               // bezier.rs:152-174
            if (kClosest) {
                if (got && w.t <= limit) {      // (beyond the walk's own best hit a patch cannot win; equal t goes to the tie rule)
                    bp.rec[3u * (size_t)pair] = make_float4(w.t, w.u, w.v, 0.0f);
                    bp.rec[3u * (size_t)pair + 2u].w = __uint_as_float(1u);
                    atomicMin(bp.key + ray, ((unsigned long long)__float_as_uint(w.t) << 32) | (unsigned long long)inst);
                }
            } else if (got && w.t < limit) {
                bp.occluded[ray] = 1;
            }
            busy = false;
        }
    }
}

// extension rays: the pair that owns a ray's key hands over its (u, v)
template <int kUnused>
__global__ void __launch_bounds__(256) k_bezier_commit(BezPairs bp) {
    const uint32_t n = min(bp.ctl[0], bp.cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 c = bp.rec[3u * (size_t)i + 2u];
        if (__float_as_uint(c.w) != 1u) continue;
        const float4 r = bp.rec[3u * (size_t)i];
        const uint32_t ray = __float_as_uint(c.y), inst = __float_as_uint(c.z);
        if (bp.key[ray] == (((unsigned long long)__float_as_uint(r.x) << 32) | (unsigned long long)inst))
            bp.def_uv[ray] = make_float4(r.y, r.z, c.x, 0.0f);      // one pair per (ray, instance): no two writers
    }
}

// shadow rays that waited for pairs: k_shadow's ending
template <int kUnused>
__global__ void __launch_bounds__(256) k_bezier_finish_shadow(RenderCtx rc, BezPairs bp) {
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = bp.ctl[2u + shard];
    const uint32_t stride = (gridDim.x / kShards) * blockDim.x;
    for (uint32_t j = (blockIdx.x / kShards) * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t idx = bp.def_list[(size_t)shard * rc.shard_cap + j];
        if (bp.occluded[idx] == 0) {
            const float4 c = rc.shadow.contrib_slot[idx];
            rad_add(rc, __float_as_uint(c.w), mk3(c));
        }
    }
}

// extension rays that waited for pairs: the closest hit over both kinds of primitive, then k_extend's ending
template <int kUnused>
__global__ void __launch_bounds__(256) k_bezier_finish_extend(DScene sc, RenderCtx rc, BezPairs bp, uint32_t bounce) {
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = bp.ctl[2u + shard];
    uint32_t* next_count = q_count(rc.counts, bounce + 1, Q_HIT, shard);
    const uint32_t qbase = shard * rc.shard_cap;
    const uint32_t stride = (gridDim.x / kShards) * blockDim.x;
    const uint32_t j0 = (blockIdx.x / kShards) * blockDim.x + (threadIdx.x & ~63u);
    for (uint32_t jw = j0; jw < n; jw += stride) {        // wave-uniform trip count: wave_push below
        const uint32_t j = jw + lane_id();
        bool keep = false;
        DHit h;
        h.t = SPT_F32_MAX; h.inst = -1; h.prim = -1; h.v = 0.0f; h.w = 0.0f;
        uint32_t idx = 0;
        if (j < n) {
            idx = bp.def_list[(size_t)qbase + j];
            const float4 hv = bp.def_hit[idx];
            h.t = hv.x; h.v = hv.y; h.w = hv.z; h.prim = __float_as_int(hv.w);
            h.inst = bp.def_inst[idx];
            const unsigned long long key = bp.key[idx];
            if (key != ~0ull) {
                const float tp = __uint_as_float((uint32_t)(key >> 32));
                const int32_t ip = (int32_t)(uint32_t)key;
                const float4 uv = bp.def_uv[idx];
                const int32_t pp = (int32_t)__float_as_uint(uv.z);
                // the walkers' acceptance: strictly closer, or equally close with the smaller (instance, prim)
                if (tp < h.t || (tp == h.t && h.inst >= 0 && key_less(ip, pp, h))) {
                    h.t = tp; h.inst = ip; h.prim = pp; h.v = uv.x; h.w = uv.y;
                }
            }
            const bool in_medium = (__float_as_uint(rc.qb.lsi_meta[idx].w) >> 8) != 0u;
            if (h.inst >= 0 || in_medium) {
                keep = true;
            } else if (sc.env_w != 0u) {  // pt.rs:97-111, curr_depth > 0 here
                const float4 b = rc.qb.d_pdf[idx], c = rc.qb.thr_slot[idx];
                f3 env;
                float env_pdf;
                env_strength_pdf(sc, mk3(b), &env, &env_pdf);
                const float weight = power_heuristic(b.w, pdf_env_light(sc) * env_pdf);
                rad_add(rc, __float_as_uint(c.w), (mk3(c) * env) * weight);
            }
        }
        const uint32_t slot = qbase + wave_push(keep, next_count);
        if (keep) {
            rc.hits.t_v_w_prim[slot] = make_float4(h.t, h.v, h.w, __int_as_float(h.prim));
            rc.hits.inst_src[slot] = make_uint2((uint32_t)h.inst, idx);
        }
    }
}

#endif  // SPT_WITH_BEZIER
