// Wavefront path-tracing kernels for gfx950 (wave64).
//
// Replaces the three nested loops of PathTracer::render / trace_ray
// (reference src/renderer/pt.rs:237-296 and 39-210) by a streaming pipeline over
// SoA queues in HBM, one stage per kernel, one queue generation per bounce:
//
//   primary : camera sample -> ray -> closest hit, fused (pt.rs:265-276 + first
//             `aggregate().intersect`).  One lane per pixel looping over the
//             samples of this pass; only HITS are written out, compacted.
//   shade   : one path vertex (pt.rs:56-96 medium, 112-193 surface, 195-206 RR):
//             emission + MIS weight, Bxdf::sample, light sample -> shadow queue,
//             continuation ray -> extension queue.
//   shadow  : aggregate().intersect_test for the shadow queue; visible
//             contributions are added to the sample's radiance slot.
//   extend  : closest hit for the extension queue; misses take the environment
//             term (pt.rs:97-111) and die, hits go to the next bounce's queue.
//   resolve : per-pixel sum of the pass's sample slots, in sample order
//             (Film::filter_pixel for the box filter, film.rs:71-91).
//
// Live-path compaction: every push uses __ballot / __popcll over the 64-lane wave
// and ONE atomicAdd per wave on the queue counter, so queues stay dense and the
// next stage runs with full waves.  Queue records are float4 / uint2 planes so each
// lane's access is a 16-byte (or 8-byte) coalesced load/store.
// All kernels are persistent-style: a fixed grid strides over a queue whose length
// is read from device memory (the host never waits for counts between stages).
//
// Determinism: a camera sample owns a radiance slot; its contributions are added
// in path order by exactly one lane at a time (stages are stream-ordered), and
// pixels are summed in sample order, so the film is bit-reproducible and identical
// for any shard layout / GPU count.
#pragma once
// waves / SIMD the latency-bound large-scene kernels are compiled for (see DESIGN.md "Registers and occupancy")
#ifndef SPT_W_EXT
#define SPT_W_EXT 5   // cfg5 extension rays (streaming walker): round 2 - 88.9 ms at 3 waves (146 VGPRs), 78.0 at 4 (128, 14 spilled), 69.6 once a leaf loads its triangles two at a time; 5 waves then meant 67 - 95 spilled VGPRs and 104 - 129 ms.  Round 3, without the SLP vectoriser's packed pairs the kernel needs 105 VGPRs unbounded: 5 waves (96, NONE spilled) 69.3 -> 62.4 ms
#endif
#ifndef SPT_W_SHD_L
#define SPT_W_SHD_L 6   // cfg4 shadow (LDS-resident): 17.2 ms unbounded (84 VGPRs, 5 waves), 17.1 at 6 (80), 21.3 at 8 (64, 30 spilled)
#endif
#ifndef SPT_W_EXT_L
#define SPT_W_EXT_L 5   // cfg4 extension rays: 27.9 ms unbounded (100 VGPRs, 4 waves), 26.8 at 5 (96, none spilled), 31.0 at 6 (80, 18 spilled)
#endif
#ifndef SPT_W_SHD
#define SPT_W_SHD 6   // cfg5 shadow: 9.9 ms at 4 waves (98 VGPRs), 9.1 at 5, 8.75 at 6 (80 VGPRs, 5 spilled)
#endif
#ifndef SPT_W_PRI_L
#define SPT_W_PRI_L 6   // k_primary of LDS-resident scenes (cfg2, cfg4): unbounded 106 VGPRs = 4 waves; MEASURED at 5 (96, 5 spilled): 0.736 -> 0.655 ms per launch on cfg2; at 6 with the camera terms pinned 0.708 (24 spilled), without the pin (SPT_PRIMARY_PIN 0: 80 VGPRs, the eye-relative instance spills none) 0.608 -> 0.581 ms: 98.0 -> 100.3 Gsamples/s
#endif
#ifndef SPT_W_PRI
#define SPT_W_PRI 6   // cfg5 primary: round 2 - 8.6 ms at 4 waves (111 VGPRs), 8.3 at 5 (96, 9 spilled), 9.6 at 6 (80, 28 spilled); round 3 (no packed pairs: 93 VGPRs unbounded) 8.0 at 5, 7.8 at 6 (80, 8 spilled)
#endif
#ifndef SPT_PRIMARY_PIN
#define SPT_PRIMARY_PIN 0   // 1: k_primary keeps its camera terms in VGPRs instead of (spilled) SGPRs, see the kernel.  Paid at 4 waves per SIMD (0.819 -> 0.790 ms);
                            // at 6 waves the 18 pinned registers are worth more as occupancy: pin + 5 waves 0.608 ms, no pin + 6 waves 0.581 ms (measured, round 3)
#endif
#include "shading.h"
#include "stream.h"
#include "flat.h"
#include "eye.h"


struct DCamera {
    f3 eye, forward, up, right;
    float half_cot;
};

struct PathQueue {            // SoA path-state queue (72 B / entry)
    float4* o_tmin;           // origin, t_min
    float4* d_pdf;            // direction, last_sample_pdf
    float4* thr_slot;         // throughput, radiance-slot index (bits)
    float4* lsi_meta;         // light_sampler_inputs.position, depth | (medium+1)<<8 (bits)
    uint2* rng;               // PCG32 state
};
struct HitQueue {             // closest-hit records = the input of the shade stage (24 B / entry)
    float4* t_v_w_prim;       // t, v, w, prim (bits)
    uint2* inst_src;          // instance (-1: an in-medium miss), index of the path's record in the PathQueue it was traced
                              // from: the extend stage does not copy the 72-byte path record next to its hit, the shade
                              // stage of the next bounce fetches it through this index
};
struct ShadowQueue {          // 48 B / entry
    float4* o_tmin;
    float4* d_tmax;
    float4* contrib_slot;
};

struct RenderCtx {
    DCamera cam;
    uint32_t width, height, spp, max_depth, sampler, division_x, division_y;
    uint64_t seed;
    uint32_t shard_index, shard_count, strip_rows;
    uint32_t row_base;        // image row of local row 0 before the strip formula (0 for a shard; the first row of a
                              // band when a wide box filter renders bands of whole rows, see spt_render)
    uint32_t n_pixels;        // pixels of this shard
    uint32_t rows;            // image rows of this shard (n_pixels = rows * width)
    uint32_t tiles_x;         // 16x16 tiles per row of tiles
    uint32_t pass_first;      // first sample index of this pass
    uint32_t pass_samples;    // samples per pixel in this pass
    PathQueue qa, qb;         // qa: paths with a hit record (input of shade); qb: paths to extend
    HitQueue hits;
    HitQueue hits_next;       // fused bounces only: hit records of the NEXT bounce (ping-pong with `hits`)
    ShadowQueue shadow;
    uint32_t* counts;         // [bounce][3 queues][kShards] lengths, one 128-B line each (see q_count)
    uint32_t shard_cap;       // entries per queue shard
    float* rad;               // 3 planes [c][s][pixel] of per-sample radiance
    size_t rad_plane;         // floats per plane: pass_samples * n_pixels, or spp * n_pixels when every sample is kept
                              // (wide box filter: rad then points at the pass's first sample inside the plane)
    float* film;              // n_pixels * 3 running sums
    uint32_t* first_slot;     // per pixel: first sample of the pass that uses a rad slot
    uint8_t* slot_bits;       // chunked k_primary only (else null): [sample / 8][pixel], bit (sample % 8) = "this sample owns a
                              // radiance slot that was written" (a hit, or a miss that saw the environment).  A missing sample
                              // writes nothing at all; k_resolve adds only the marked slots
    float aspect, width_inv, height_inv, spp_inv;
    float aux_dx, aux_dy;  // auxiliary-ray offsets (pt.rs:272-275), textured scenes only
    // conservative bounding sphere of all instances relative to the camera eye (primary early-out)
    f3 bs_oc;                 // sphere centre - eye
    float bs_c;               // |oc|^2 - R^2   (R inflated by 0.1 %)
    uint32_t bs_valid;
    // screen-space bound of the scene: pixels outside [cull_i0, cull_i1] x [cull_j0, cull_j1] (image coordinates,
    // one pixel of slack) cannot see any instance box; the full image when the bound is not available
    int32_t cull_i0, cull_i1, cull_j0, cull_j1;
    uint32_t n_tiles, primary_chunks, chunk_samples;   // k_primary<., kChunked>: tiles of the shard, chunks per tile, samples per chunk
    uint32_t dyn_refill_below, dyn_steps;   // tuning of the refilling kernels
    uint32_t stream_rounds, stream_refill_below;   // streaming kernels: while-while rounds between two retire / refill checks; refill threshold
    unsigned long long* visits;             // SPT_RENDER_COUNT_VISITS: [nodes, triangles, instances] fetched by the traversals (kCount kernels)
    uint32_t n_classes, class_cap;          // hit queue of bounce >= 1: sub-queues per shard (1: one queue, the fused pipeline) and their distance
    uint32_t pack_first;                    // 1: a bounce-0 hit record carries (instance | pass-local sample << 20, image pixel index) in inst_src, so
                                            // the shade kernel does not recover (pixel, sample) from the slot with three integer divisions (~100
                                            // cycles each); the host sets it when the scene has < 2^20 instances and a pass <= 4096 samples
    const int2* row_span;                   // per image row: first / last pixel that can see an instance (null: only the rectangle above)
    uint32_t debug_normal;                  // SPT_RENDER_DEBUG_NORMAL: the reference's `debug_normal` feature (pt.rs:113-118)
};

SPT_DEV uint32_t lane_id() { return threadIdx.x & 63u; }
// image row of a local row of this shard (row strips dealt round-robin, spt_abi.h)
SPT_DEV uint32_t global_row(const RenderCtx& rc, uint32_t row_local) {
    const uint32_t strip = row_local / rc.strip_rows;
    return rc.row_base + (strip * rc.shard_count + rc.shard_index) * rc.strip_rows + (row_local - strip * rc.strip_rows);
}

// Every queue is split into kShards sub-queues with their own length counter on their
// own 128-byte line: a block only ever appends to / consumes shard (blockIdx.x % kShards).
//  - one hot counter would serialise every wave's append in the L2 atomic unit;
//  - blocks b and b+8 run on the same XCD (round-robin dispatch), so with kShards a
//    multiple of 8 a shard is written and later read by blocks of ONE XCD and its
//    records stay in that XCD's 4 MiB L2 between stages (speed only, never correctness).
constexpr uint32_t kShards = 64;
// The hit queue of bounce >= 1 is kClasses sub-queues per shard when the scene is shaded by the general kernels (round 3): the
// extension stage files a vertex under the BxDF class of the instance it hit (DScene::inst_class; kClasses - 1 = the path travels
// inside a medium: a medium event first, whatever it hit), the shade stage consumes class after class, so that the 64 vertices of a wave run ONE branch of mat_sample / mat_eval /
// mat_pdf and agree on whether there is a light sample at all.  Measured motive (tools/shade_coherence_probe.py, cfg4): the same
// kernel runs at 0.36 lane utilisation on the mixed queue and at 0.81 when every instance has the same material.  Class 0 keeps the
// old indices (shard * shard_cap + i), class c lives class_cap = kShards * shard_cap entries further per class.
constexpr uint32_t kClasses = 8;
enum { Q_HIT = 0, Q_SHADOW = 1, Q_EXT = 2, Q_SHADOW_CURSOR = 3, Q_EXT_CURSOR = 4, Q_HIT_CLASS1 = 5, Q_KINDS = 5 + (kClasses - 1) };
SPT_DEV uint32_t q_hit_kind(uint32_t cls) { return cls == 0u ? (uint32_t)Q_HIT : (uint32_t)Q_HIT_CLASS1 + cls - 1u; }
SPT_DEV uint32_t* q_count(const uint32_t* counts, uint32_t bounce, uint32_t q, uint32_t shard) {
    return const_cast<uint32_t*>(counts) + ((size_t)(bounce * Q_KINDS + q) * kShards + shard) * 32u;
}

// Wave-aggregated append: returns this lane's slot (valid only where pred).
SPT_DEV uint32_t wave_push(bool pred, uint32_t* counter) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    uint32_t lane = lane_id();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    unsigned long long lt = (lane == 0u) ? 0ull : (~0ull >> (64u - lane));
    return base + (uint32_t)__popcll(mask & lt);
}

// The same append in two halves, so the atomic's round trip (~1-2 us to the L2 atomic unit and back)
// overlaps independent work: issue() starts the atomicAdd and returns at once (the returned value is
// only read in finish(), and the compiler places the s_waitcnt at that first use).
struct PendingPush {
    unsigned long long mask;
    uint32_t base_raw, leader;
};
SPT_DEV PendingPush wave_push_issue(bool pred, uint32_t* counter) {
    PendingPush p;
    p.mask = __ballot(pred);
    p.base_raw = 0u;
    p.leader = 0u;
    if (p.mask != 0ull) {
        p.leader = (uint32_t)__ffsll((long long)p.mask) - 1u;
        if (lane_id() == p.leader) p.base_raw = atomicAdd(counter, (uint32_t)__popcll(p.mask));
    }
    return p;
}
SPT_DEV uint32_t wave_push_finish(const PendingPush& p) {
    if (p.mask == 0ull) return 0u;
    const uint32_t lane = lane_id();
    const uint32_t base = (uint32_t)__shfl((int)p.base_raw, (int)p.leader, 64);
    const unsigned long long lt = (lane == 0u) ? 0ull : (~0ull >> (64u - lane));
    return base + (uint32_t)__popcll(p.mask & lt);
}

// PerspectiveCamera::generate_ray (camera/perspective.rs:40-47)
SPT_DEV DRay camera_ray(const DCamera& c, float x, float y) {
    DRay r;
    r.o = c.eye;
    r.d = normalize((c.forward * c.half_cot + c.right * x) + c.up * y);
    r.t_min = kTMinEps;
    return r;
}

// pixel sampler (pixel_sampler/{random,jittered,recurrence}.rs)
SPT_DEV void pixel_offset(const RenderCtx& rc, uint32_t pixel, uint32_t s, DRng& rng, float* ox, float* oy) {
    if (rc.sampler == SPT_SAMPLER_RECURRENCE) {
        spt_r2_offset(pixel, rc.spp, s, ox, oy);
    } else if (rc.sampler == SPT_SAMPLER_JITTERED) {
        uint32_t ix = s % rc.division_x, iy = s / rc.division_x;
        float inv_x = 1.0f / (float)rc.division_x, inv_y = 1.0f / (float)rc.division_y;
        *ox = ((float)ix + rng.next()) * inv_x;
        *oy = ((float)iy + rng.next()) * inv_y;
    } else {
        *ox = rng.next();
        *oy = rng.next();
    }
}

// kCount kernels: add this lane's visit counts to the render's totals (one atomic per wave and counter)
// cls: 0 primary, 1 shadow, 2 extension rays
SPT_DEV void flush_visits(const RenderCtx& rc, const LaneVisits& vc, uint32_t cls) {
    uint32_t v[3] = {vc.nodes, vc.tris, vc.insts};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        unsigned long long t = v[k];
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane_id() == 0u && t) atomicAdd(rc.visits + 3u * cls + k, t);
    }
}

// Append a kept vertex (the closest hit of an extension ray, or an in-medium miss: inst < 0) to the hit queue of bounce `bounce_next`,
// filed under its class.  ONE atomic round trip per wave whatever the number of classes present: the first lane of every class adds
// that class' count to its counter (one atomic instruction, up to kClasses addresses), the bases come back through a shuffle.  (First version: one
// wave-aggregated append per class present, one after the other - 2 - 3 dependent round trips of 1 - 2 us in a kernel that waits
// for memory two thirds of its time.)  Returns the lane's queue index (valid where keep).
// kLds: the class rides in pad[0] of the staged instance record (no global load between the walk and the append).
template <bool kLds>
SPT_DEV uint32_t hit_push(const DScene& sc, const RenderCtx& rc, bool keep, int32_t inst, uint32_t bounce_next, uint32_t shard) {
    if (rc.n_classes <= 1u) return shard * rc.shard_cap + wave_push(keep, q_count(rc.counts, bounce_next, Q_HIT, shard));
    uint32_t cls = 0xffffffffu;
    if (keep) cls = inst < 0 ? kClasses - 1u : (kLds ? __float_as_uint(geo_ld<true>(sc, sc.o_inst + 12u * (uint32_t)inst + 10u).w) : (uint32_t)sc.inst_class[inst]);
    // the lanes of one class elect their first lane; the leaders of all classes issue their atomics in the same instruction
    // (only lanes that carry a kept vertex take part: the caller may run with some lanes switched off)
    unsigned long long mine = 0ull;
#pragma unroll
    for (uint32_t c = 0; c < kClasses; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (cls == c) mine = m;
    }
    const uint32_t lane = lane_id();
    const uint32_t leader = keep ? (uint32_t)__ffsll((long long)mine) - 1u : lane;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
    uint32_t base = 0u;
    if (keep && lane == leader) base = atomicAdd(q_count(rc.counts, bounce_next, q_hit_kind(cls), shard), (uint32_t)__popcll(mine));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    return cls * rc.class_cap + shard * rc.shard_cap + base + rank;
}

SPT_DEV uint32_t pack_meta(uint32_t depth, int32_t medium) { return depth | ((uint32_t)(medium + 1) << 8); }

SPT_DEV void store_path(const PathQueue& q, uint32_t i, const DRay& ray, float last_pdf, f3 thr, uint32_t slot, f3 lsi,
                        uint32_t meta, const DRng& rng) {
    q.o_tmin[i] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.t_min);
    q.d_pdf[i] = make_float4(ray.d.x, ray.d.y, ray.d.z, last_pdf);
    q.thr_slot[i] = make_float4(thr.x, thr.y, thr.z, __uint_as_float(slot));
    q.lsi_meta[i] = make_float4(lsi.x, lsi.y, lsi.z, __uint_as_float(meta));
    q.rng[i] = make_uint2((uint32_t)rng.s.state, (uint32_t)(rng.s.state >> 32));
}

// ---------------------------------------------------------------------------- primary
// Blocks are 16x16 pixel tiles of the shard's (rows x width) image.  Tile (tx, ty) appends to
// queue shard (tx + 9 ty) mod 64: every 8x8 group of tiles touches all 64 shards once, so the
// shards (and with them the XCDs that later consume them) receive equal shares of any object
// larger than ~128 px, instead of whole image columns.
constexpr uint32_t kTile = 16;
SPT_DEV uint32_t tile_shard(uint32_t tx, uint32_t ty) { return (tx + 9u * ty) % kShards; }

// kChunked: the samples of the pass are split over rc.primary_chunks workgroups per tile (blockIdx = tile +
// n_tiles * chunk).  Needed when the shard has few (active) tiles - a rank of an 8-GPU run owns 1/8 of the rows,
// and only the tiles inside the screen-space bound do work - or the sample loops of a few hundred workgroups
// would run on an otherwise idle chip.  A chunk cannot know whether an earlier sample of its pixel hit, so here
// EVERY sample of a live pixel owns a radiance slot (first_slot = 0) and the film is only touched by k_resolve,
// which adds the slots in sample order: the same additions in the same order as the register sum of the
// un-chunked kernel (a miss adds exactly +0 or its environment term), so the film is bit-identical.
// kEye (with kLds, no counting): `sc` describes the eye-relative copy of the geometry (eye.h) and the trace reads what trace.h computes per ray
template <bool kLds, bool kChunked = false, bool kCount = false, bool kEye = false>
__global__ void __launch_bounds__(256, SPT_WITH_BEZIER ? 1 : (kLds ? SPT_W_PRI_L : SPT_W_PRI)) k_primary(DScene sc, RenderCtx rc) {
    stage_geometry<kLds>(sc);
    LaneVisits vc{0u, 0u, 0u};
    const uint32_t tile = kChunked ? blockIdx.x % rc.n_tiles : blockIdx.x, chunk = kChunked ? blockIdx.x / rc.n_tiles : 0u;
    const uint32_t tx = tile % rc.tiles_x, ty = tile / rc.tiles_x;
    const uint32_t i = tx * kTile + (threadIdx.x % kTile);
    const uint32_t row_local = ty * kTile + (threadIdx.x / kTile);
    const bool valid = (i < rc.width) && (row_local < rc.rows);
    const uint32_t lp = valid ? row_local * rc.width + i : 0u;
    const uint32_t j = global_row(rc, row_local);
    const uint32_t pixel = j * rc.width + i;
    const bool has_env = sc.env_w != 0u;
    const bool lazy_rng = rc.sampler == SPT_SAMPLER_RECURRENCE;  // the R2 sampler draws nothing: seed hits only
    const uint32_t shard = tile_shard(tx, ty);
    f3 sum = mk3(0, 0, 0);
    if (!kChunked && valid) sum = mk3(rc.film[3 * lp], rc.film[3 * lp + 1], rc.film[3 * lp + 2]);
    uint32_t first = kChunked ? 0u : rc.pass_samples;
    const size_t plane = rc.rad_plane;
    uint32_t* hit_counter = q_count(rc.counts, 0, Q_HIT, shard);
    // A pixel whose whole footprint lies outside the projected bounds of the scene cannot hit anything with any
    // of its samples.  Without an environment all of them are black and leave no trace (film += 0, no radiance
    // slot), so the sample loop is not entered; with one they still look the environment up, but skip the trace.
    bool in_bounds = (int32_t)i >= rc.cull_i0 && (int32_t)i <= rc.cull_i1 && (int32_t)j >= rc.cull_j0 && (int32_t)j <= rc.cull_j1;
    if (in_bounds && valid && rc.row_span != nullptr) {   // the row's own span inside the rectangle (projected hulls of the instances' boxes)
        const int2 span = rc.row_span[j];
        in_bounds = (int32_t)i >= span.x && (int32_t)i <= span.y;
    }
    const bool live = in_bounds || has_env;
    const uint32_t s_begin = kChunked ? chunk * rc.chunk_samples : 0u;
    const uint32_t s_end = live ? (kChunked ? min(s_begin + rc.chunk_samples, rc.pass_samples) : rc.pass_samples) : s_begin;
    uint32_t slot_mask = 0u;     // chunked: the slot bits of the current group of 8 samples (chunk_samples is a multiple of 8)
#if SPT_PRIMARY_PIN
    // The kernel needs more wave-uniform values than the 102 SGPRs hold (54 are spilled into VGPR lanes and come back through
    // v_readlane, a 4-cycle VALU instruction each, inside the sample loop), while 24 VGPRs are free below the 4-wave limit.  The
    // camera terms every sample multiplies with are therefore kept in VGPRs (a broadcast value in a VGPR is as good an operand
    // as an SGPR): same values, same operations, fewer spills.
    // (LDS-resident scenes only: the large-scene instantiations are compiled for 5 waves = 96 VGPRs and have none to spare)
    auto pin = [](float v) { if (kLds) asm volatile("" : "+v"(v)); return v; };
    const f3 cam_fwd = mk3(pin(rc.cam.forward.x * rc.cam.half_cot), pin(rc.cam.forward.y * rc.cam.half_cot), pin(rc.cam.forward.z * rc.cam.half_cot));
    const f3 cam_right = mk3(pin(rc.cam.right.x), pin(rc.cam.right.y), pin(rc.cam.right.z)), cam_up = mk3(pin(rc.cam.up.x), pin(rc.cam.up.y), pin(rc.cam.up.z));
    const f3 cam_eye = mk3(pin(rc.cam.eye.x), pin(rc.cam.eye.y), pin(rc.cam.eye.z)), bs_oc = mk3(pin(rc.bs_oc.x), pin(rc.bs_oc.y), pin(rc.bs_oc.z));
    const float bs_c = pin(rc.bs_c), width_inv = pin(rc.width_inv), height_inv = pin(rc.height_inv), aspect = pin(rc.aspect);
#else
    const f3 cam_fwd = rc.cam.forward * rc.cam.half_cot, cam_right = rc.cam.right, cam_up = rc.cam.up, cam_eye = rc.cam.eye, bs_oc = rc.bs_oc;
    const float bs_c = rc.bs_c, width_inv = rc.width_inv, height_inv = rc.height_inv, aspect = rc.aspect;
#endif
    for (uint32_t s = s_begin; s < s_end; ++s) {
        const uint32_t gs = rc.pass_first + s;
        DRng rng;
        rng.s.state = 0ull;
        if (!lazy_rng) rng.s = spt_rng_seed(rc.seed, pixel, gs);
        float ox, oy;
        pixel_offset(rc, pixel, gs, rng, &ox, &oy);
        float x = (((float)i + ox) * width_inv - 0.5f) * aspect;           // pt.rs:269
        float y = ((float)(rc.height - j - 1u) + oy) * height_inv - 0.5f;       // pt.rs:270-271
        // PerspectiveCamera::generate_ray (camera/perspective.rs:40-47), split so that a sample whose
        // un-normalised direction already misses the bounding sphere of the whole scene skips the
        // normalisation and the traversal: nothing can be hit, and without an environment the sample
        // is black.  The sphere is inflated, so the test only removes rays every box test would reject.
        const f3 du = (cam_fwd + cam_right * x) + cam_up * y;
        bool may_hit = valid && in_bounds;
        if (may_hit && rc.bs_valid && bs_c > 0.0f) {
            const float b = dot(du, bs_oc);
            may_hit = valid && (b > 0.0f) && (b * b >= dot(du, du) * bs_c);
        }
        DRay ray;
        ray.o = cam_eye;
        ray.t_min = kTMinEps;
        ray.d = du;
        if (may_hit || has_env) ray.d = normalize(du);
        DHit h;
        h.inst = -1;
        if (may_hit) h = kEye ? eye_trace_closest(sc, ray.d, ray.t_min, SPT_F32_MAX) : trace_closest<kLds, kCount>(sc, ray, SPT_F32_MAX, &vc);
        const bool hit = valid && h.inst >= 0;
        const size_t ri = (size_t)s * rc.n_pixels + lp;
        if (valid && !hit) {
            if (has_env) {  // pt.rs:98-110 at depth 0: weight 1
                f3 env;
                float env_pdf;
                env_strength_pdf(sc, ray.d, &env, &env_pdf);
                f3 c = mk3(0, 0, 0) + (gray(1.0f) * env) * 1.0f;
                if (first == rc.pass_samples) {
                    sum = sum + c;
                } else {
                    rc.rad[ri] = c.x; rc.rad[plane + ri] = c.y; rc.rad[2 * plane + ri] = c.z;
                }
            } else if (!kChunked && first != rc.pass_samples) {   // chunked: a black sample leaves no trace (slot_bits)
                rc.rad[ri] = 0.0f; rc.rad[plane + ri] = 0.0f; rc.rad[2 * plane + ri] = 0.0f;
            }
        }
        // (a hit's slot is not zeroed here: the bounce-0 shade kernel WRITES the sample's first contributions, see k_shade)
        if (hit && first == rc.pass_samples) first = s;
        if (kChunked) {
            slot_mask |= ((hit || (valid && has_env)) ? 1u : 0u) << (s & 7u);
            if ((s & 7u) == 7u || s + 1u == s_end) {
                if (valid) rc.slot_bits[(size_t)(s >> 3) * rc.n_pixels + lp] = (uint8_t)slot_mask;
                slot_mask = 0u;
            }
        }
        // (deferring this append by one iteration to overlap the counter atomic with the next sample was tried
        //  twice: at 120 VGPRs the pending state cost a wave per SIMD and it was slower, at 104 VGPRs it fits
        //  but MEASURED the same 2.70 ms - the atomic is not what the kernel waits for)
        // (general pipeline: camera hits are filed by BxDF class too, see kClasses - object boundaries inside a wave cost the bounce-0
        //  shade kernel a third of its lanes; the compact record below sits at the hit's own index, so qa.d_pdf spans all classes)
        const uint32_t slot = rc.n_classes > 1u ? hit_push<kLds>(sc, rc, hit, h.inst, 0u, shard) : shard * rc.shard_cap + wave_push(hit, hit_counter);
        if (hit) {
            // bounce-0 records are compact: origin (eye), t_min, throughput (1), last_pdf (0), depth, medium and
            // the RNG stream (a function of pixel and sample = of the slot) are constants that k_shade<.., true>
            // rebuilds, so only (direction, slot) and the hit record travel: 36 B instead of 92 B per hit
            rc.qa.d_pdf[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float((uint32_t)ri));
            rc.hits.t_v_w_prim[slot] = make_float4(h.t, h.v, h.w, __int_as_float(h.prim));
            rc.hits.inst_src[slot] = rc.pack_first ? make_uint2((uint32_t)h.inst | (s << 20), pixel) : make_uint2((uint32_t)h.inst, slot);
        }
    }
    if (kCount) flush_visits(rc, vc, 0u);
    if (kChunked) {
        if (valid && chunk == 0u) rc.first_slot[lp] = live ? 0u : rc.pass_samples;
    } else if (valid) {
        rc.film[3 * lp] = sum.x; rc.film[3 * lp + 1] = sum.y; rc.film[3 * lp + 2] = sum.z;
        rc.first_slot[lp] = first;
    }
}

SPT_DEV void rad_add(const RenderCtx& rc, uint32_t slot, f3 c) {
    const size_t plane = rc.rad_plane;
    rc.rad[slot] = rc.rad[slot] + c.x;
    rc.rad[plane + slot] = rc.rad[plane + slot] + c.y;
    rc.rad[2 * plane + slot] = rc.rad[2 * plane + slot] + c.z;
}

SPT_DEV void rad_store(const RenderCtx& rc, uint32_t slot, f3 c) {
    const size_t plane = rc.rad_plane;
    rc.rad[slot] = c.x;
    rc.rad[plane + slot] = c.y;
    rc.rad[2 * plane + slot] = c.z;
}

// ---------------------------------------------------------------------------- shade
// One iteration of the `while curr_depth < max_depth` loop of trace_ray, minus the
// two traversals, for the path vertex in qa[idx] / hits[idx].
// kSimple is a scene-feature specialisation chosen on the host (spt_scene_create): Lambert
// materials, delta lights, no emission, no environment, no media.  The general code path is
// identical arithmetic; the specialisation only removes branches that cannot be taken, which
// cuts the kernel from 226 to far fewer VGPRs (more waves per SIMD to hide queue latency).
// kFirst: bounce 0, whose queue holds the compact records written by k_primary.
// kFeat 0 = kSimple, 1 = the general path.
// kFeat 2 (textured): some material parameter, normal map or emissive map is an image texture; adds
// texcoords, the camera ray differentials of bounce 0 and the per-hit material evaluation.
// kFeat 3: 2 + the Subsurface substrate (a closest-hit probe inside mat_sample; 265 VGPRs = 1 wave / SIMD, which
// is why it is a level of its own and not part of 2).
// kFused (scenes whose traversal geometry is LDS-resident): the shadow ray and the extension ray are traced
// right here instead of going through the shadow / extend queues and kernels.  The shade queue is dense,
// so the waves are as full for the two traversals as they would be in k_shadow / k_extend, and the
// 48-B shadow + 72-B path records (written once, read once: 240 B per path vertex) never touch HBM.
// Next-bounce vertices go to (qb, hits_next); the host swaps the two buffer pairs every bounce.
// (139 VGPRs for kFeat 0 = 3 waves / SIMD; forcing 128 with a launch bound spills 12 and measured the same.)
// kTab: the shading tables are read from the LDS-staged blob (tab_ld); always with kFused, and for the un-fused
// general kernel of any scene whose geometry + tables fit LDS.
// kGeoLds (kFeat 3 only): the BSSRDF probe walks the LDS-staged traversal geometry.  It has to follow the SCENE
// (lds_geo), not kTab: a scene whose geometry fits LDS but whose shading tables do not runs the un-tabbed kernel, and
// its global blob holds the LDS node format, which only the kLds walker reads.
//
#ifndef SPT_SHADE0_WAVES
#define SPT_SHADE0_WAVES 0   // waves / SIMD the fused bounce-0 kernel is compiled for; 0 = no bound (146 VGPRs, 3 waves).
                             // MEASURED with 4 (128 VGPRs, 20 spilled): cfg2 shade_first 1.44 -> 1.96 ms per step; not kept
#endif
// kLoop (fused kernels of bounce >= 1 only): a vertex produced by the extension trace inside this kernel is shaded by the
// same lane in the next turn of an inner loop instead of travelling through the queue to the next launch, down to the
// last bounce.  The host picks it when the previous pass showed that few paths are left after bounce 0 (cfg2: a cube in
// the void): the 2 x 6 launches of bounces 2 .. 7, each ~9.5 us of dispatch for nothing, are never made.  Same
// arithmetic per path, the same additions to its radiance slot in the same order; the queue counters still count.
#ifndef SPT_SHADE_HEAVY_WAVES
#define SPT_SHADE_HEAVY_WAVES 2   // waves / SIMD the probe-only (kFeat 3) and glint-only (kFeat 4) kernels are compiled for: unbounded they need
                                  // 248 - 266 resp. 284 - 300 VGPRs, i.e. mostly ONE wave; bounded to 256 a few registers spill (measured: DESIGN.md)
#endif
#ifndef SPT_SHADE1_WAVES
#define SPT_SHADE1_WAVES 3   // waves / SIMD the BOUNCE-0 instance of the general un-fused kernel (kFeat 1) is compiled for (0 = no bound: 186 VGPRs, 2 waves).
                             // MEASURED at 3 (168 VGPRs, 5 - 9 spilled): cfg4 shade_first 20.9 -> 19.5 ms, cfg5 10.9 -> 9.5.  The later-bounce instance
                             // (206 VGPRs) would spill 110 - 140 at 3 waves: cfg4 22.1 -> 24.5 ms, so it stays unbounded
#endif
template <int kFeat, bool kFirst, bool kFused = false, bool kTab = kFused, bool kGeoLds = kTab, bool kLoop = false>
__global__ void __launch_bounds__(256, (kFeat == 0 && kFirst && kFused && SPT_SHADE0_WAVES) ? SPT_SHADE0_WAVES : ((kFeat == 3 || kFeat == 4) ? SPT_SHADE_HEAVY_WAVES : ((kFeat == 1 && kFirst && SPT_SHADE1_WAVES) ? SPT_SHADE1_WAVES : 1))) k_shade(DScene sc, RenderCtx rc, uint32_t bounce) {
    static_assert(!kLoop || (kFused && !kFirst), "the in-kernel bounce loop exists for the fused kernels of bounce >= 1");
    // kFeat 3: Subsurface substrates (the probe), 4: position-normal distributions (the glint walks), 5: both
    constexpr bool kSimple = kFeat == 0, kTex = kFeat >= 2, kSubsurface = kFeat == 3 || kFeat == 5, kPndf = kFeat == 4 || kFeat == 5;
    const uint32_t shard = blockIdx.x % kShards;
    // classes of the hit queue this launch consumes (the fused pipeline has one)
    const uint32_t n_cls = kFused ? 1u : rc.n_classes;
    uint32_t n_any = 0u;
    for (uint32_t c = 0; c < n_cls; ++c) n_any = max(n_any, *q_count(rc.counts, bounce, q_hit_kind(c), shard));
    // workgroups of an empty shard (most of them in the late bounces) leave before staging anything into LDS;
    // a workgroup whose first item is past the end has nothing to do either (block-uniform, no barrier skipped)
    if ((blockIdx.x / kShards) * blockDim.x >= n_any) return;
    if (kFused || kTab || kGeoLds) stage_geometry<true>(sc);
    uint32_t* shadow_count = q_count(rc.counts, bounce, Q_SHADOW, shard);
    uint32_t* ext_count = q_count(rc.counts, bounce, Q_EXT, shard);
    const uint32_t qbase = shard * rc.shard_cap;
    const uint32_t stride = (gridDim.x / kShards) * blockDim.x;
    // the queue record of the NEXT iteration is requested before this iteration's shading (and traversals), so
    // its HBM latency is hidden behind them (2 - 3 waves / SIMD cannot hide it otherwise)
    constexpr bool kPrefetch = true;                 // (hit, direction / slot, instance): 9 registers
    // + the rest of the 72-byte path record: 14 more (k_shade<1> would reach 260 VGPRs = 1 wave / SIMD).  Fused kernels only:
    // there the record sits at the hit's own index; un-fused it is behind the hit's source index (see HitQueue)
    constexpr bool kPrefetchPath = !kFirst && kSimple && kFused;
    const uint32_t i_first = (blockIdx.x / kShards) * blockDim.x + (threadIdx.x & ~63u);
    for (uint32_t cls = 0; cls < n_cls; ++cls) {
    const uint32_t n = *q_count(rc.counts, bounce, q_hit_kind(cls), shard);
    const uint32_t hbase = cls * rc.class_cap + qbase;      // (class 0: the old indices, where bounce 0 and the fused kernels keep records and hits aligned)
    float4 pre_hv = make_float4(0, 0, 0, 0), pre_b = make_float4(0, 0, 0, 0);
    float4 pre_a = pre_hv, pre_c = pre_hv, pre_d = pre_hv;
    uint2 pre_rs = make_uint2(0u, 0u);
    uint2 pre_is = make_uint2(0xffffffffu, 0u);
    if (kPrefetch && i_first + lane_id() < n) {
        const uint32_t k = hbase + i_first + lane_id();
        pre_hv = rc.hits.t_v_w_prim[k]; pre_is = rc.hits.inst_src[k];
        if (kFirst || kFused) pre_b = rc.qa.d_pdf[k];
        if (kPrefetchPath) { pre_a = rc.qa.o_tmin[k]; pre_c = rc.qa.thr_slot[k]; pre_d = rc.qa.lsi_meta[k]; pre_rs = rc.qa.rng[k]; }
    }
    for (uint32_t i0 = i_first; i0 < n; i0 += stride) {
        const bool active = i0 + lane_id() < n;
        const uint32_t idx = hbase + i0 + lane_id();
        const float4 cur_hv = pre_hv, cur_b = pre_b, cur_a = pre_a, cur_c = pre_c, cur_d = pre_d;
        const uint2 cur_rs = pre_rs;
        const uint2 cur_is = pre_is;
        if (kPrefetch && i0 + stride + lane_id() < n) {
            const uint32_t k = idx + stride;
            pre_hv = rc.hits.t_v_w_prim[k]; pre_is = rc.hits.inst_src[k];
            if (kFirst || kFused) pre_b = rc.qa.d_pdf[k];
            if (kPrefetchPath) { pre_a = rc.qa.o_tmin[k]; pre_c = rc.qa.thr_slot[k]; pre_d = rc.qa.lsi_meta[k]; pre_rs = rc.qa.rng[k]; }
        }
        // kLoop: a lane's vertex of the next bounce, handed from the end of one turn to the start of the next
        bool carried = false, lane_on = active;
        DRay c_ray;
        c_ray.o = mk3(0, 0, 0); c_ray.d = mk3(0, 0, 0); c_ray.t_min = 0.0f;
        float c_pdf = 0.0f;
        DHit c_h;
        c_h.t = 0.0f; c_h.inst = -1; c_h.prim = -1; c_h.v = 0.0f; c_h.w = 0.0f;
        uint32_t b_cur = bounce;
        // Bounce 0 owns the first contributions of a camera sample: they are summed here, starting from the 0 the slot
        // would hold, in the order the read-modify-writes would have been made, and the slot is WRITTEN once at the end of
        // the iteration (k_primary does not zero it).  Later bounces and k_shadow / k_extend add to the slot as before.
        f3 first_acc = mk3(0.0f, 0.0f, 0.0f);
        auto slot_add = [&](uint32_t sl, f3 c) {
            if (kFirst) first_acc = first_acc + c;
            else rad_add(rc, sl, c);
        };
        DRay shadow_ray, next_ray;
        f3 thr = gray(1.0f), lsi = mk3(0, 0, 0);
        uint32_t slot = 0u, depth = 0u;
        int32_t medium = -1;
        DRng rng;
        rng.s.state = 0ull;
        bool want_shadow = false, want_ext = false;
        float shadow_tmax = 0.0f, next_pdf = 0.0f;
        f3 contrib = mk3(0, 0, 0);
        for (;;) {   // one turn, or (kLoop) one turn per bounce
        want_shadow = false; want_ext = false;
        contrib = mk3(0, 0, 0);
        if (lane_on) {
            float4 hv = kPrefetch ? cur_hv : rc.hits.t_v_w_prim[idx];
            DRay ray;
            float last_pdf;
            f3 aux_xd = mk3(0, 0, 0), aux_yd = mk3(0, 0, 0);
            if (kLoop && carried) {   // (thr, slot, lsi, depth, medium, rng are where the previous turn left them)
                ray = c_ray;
                last_pdf = c_pdf;
            } else if (kFirst) {
                // rebuild the constants of a camera path from the slot (see k_primary)
                const float4 b = kPrefetch ? cur_b : rc.qa.d_pdf[idx];
                slot = __float_as_uint(b.w);
                uint32_t s_local, pix, j = 0u, col = 0u;
                if (rc.pack_first) {   // (wave-uniform) the record says which pixel and which sample it is
                    s_local = cur_is.x >> 20;
                    pix = cur_is.y;
                    if (kTex) { j = pix / rc.width; col = pix - j * rc.width; }
                } else {
                    s_local = slot / rc.n_pixels;
                    const uint32_t lp = slot - s_local * rc.n_pixels;
                    const uint32_t row_local = lp / rc.width;
                    col = lp - row_local * rc.width;
                    j = global_row(rc, row_local);
                    pix = j * rc.width + col;
                }
                rng.s = spt_rng_seed(rc.seed, pix, rc.pass_first + s_local);
                if (kTex) {
                    // the auxiliary rays of generate_ray_with_aux_ray (camera/mod.rs:15-21, offsets of pt.rs:272-275)
                    // are a function of the pixel offsets: redo the sampler draw instead of carrying 12 floats
                    float ox, oy;
                    pixel_offset(rc, pix, rc.pass_first + s_local, rng, &ox, &oy);
                    float x = (((float)col + ox) * rc.width_inv - 0.5f) * rc.aspect;
                    float y = ((float)(rc.height - j - 1u) + oy) * rc.height_inv - 0.5f;
                    aux_xd = normalize((rc.cam.forward * rc.cam.half_cot + rc.cam.right * (x + rc.aux_dx)) + rc.cam.up * y);
                    aux_yd = normalize((rc.cam.forward * rc.cam.half_cot + rc.cam.right * x) + rc.cam.up * (y + rc.aux_dy));
                } else if (rc.sampler != SPT_SAMPLER_RECURRENCE) {  // the pixel offsets were the first two draws
                    (void)rng.next();
                    (void)rng.next();
                }
                ray.o = rc.cam.eye; ray.t_min = kTMinEps;
                ray.d = mk3(b);
                last_pdf = 0.0f;
                thr = gray(1.0f);
                lsi = mk3(0, 0, 0);
                depth = 0u;
                medium = -1;
            } else {
                // the path's record: at the hit's own index when this kernel's predecessor wrote both (fused), else where
                // the shade stage of the previous bounce left it (the extend stage only recorded the index)
                const uint32_t src = kFused ? idx : cur_is.y;
                const float4 b = kFused ? cur_b : rc.qa.d_pdf[src];
                float4 a, c, d;
                uint2 rs;
                if (kPrefetchPath) { a = cur_a; c = cur_c; d = cur_d; rs = cur_rs; }
                else { a = rc.qa.o_tmin[src]; c = rc.qa.thr_slot[src]; d = rc.qa.lsi_meta[src]; rs = rc.qa.rng[src]; }
                ray.o = mk3(a); ray.t_min = a.w;
                ray.d = mk3(b);
                last_pdf = b.w;
                thr = mk3(c);
                slot = __float_as_uint(c.w);
                lsi = mk3(d);
                uint32_t meta = __float_as_uint(d.w);
                depth = meta & 0xffu;
                medium = (int32_t)(meta >> 8) - 1;
                rng.s.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
            }
            DHit h;
            h.t = hv.x; h.v = hv.y; h.w = hv.z; h.prim = __float_as_int(hv.w);
            h.inst = (kFirst && rc.pack_first) ? (int32_t)(cur_is.x & 0xfffffu) : (int32_t)cur_is.x;
            if (kLoop && carried) h = c_h;
            const bool does_hit = h.inst >= 0;
            bool alive = true;       // false: path ended without the RR / depth tail (`break`)
            bool scattered = false;  // a new ray was produced (tail applies)

            DInter it;
            it.prim_type = 0u; it.prim_id = 0u;
            if (does_hit) {
                it = reconstruct_hit<kTex, kTab>(sc, ray, h);
                if (kTex && kFirst) calc_differential(it, ray, h.t, rc.cam.eye, aux_xd, rc.cam.eye, aux_yd);   // pt.rs:51-53
            }

            if (!kSimple && medium >= 0) {  // pt.rs:56-96
                const spt_medium& md = sc.mediums[medium];
                f3 sigma_t = mk3(md.sigma_t);
                f3 wo = -ray.d;
                float rx = rng.next(), ry = rng.next();  // Homogeneous::sample_pi, homogeneous.rs:30-58
                float sample_sigma_t = (rx < 1.0f / 3.0f) ? md.sigma_t[0] : ((rx < 2.0f / 3.0f) ? md.sigma_t[1] : md.sigma_t[2]);
                float sample_t = -spt_log(1.0f - ry) / sample_sigma_t;
                float tt = spt_min(sample_t, h.t);
                f3 attenuation = cexp((-sigma_t) * tt);
                f3 pi = ray.o - wo * tt;
                if (sample_t < h.t) {
                    float atten_pdf = cavg(sigma_t * attenuation);
                    thr = thr * crcp(attenuation * mk3(md.sigma_s), atten_pdf);
                    // still in the medium: in-scattering from one light sample
                    lsi = pi;
                    DLightSample ls;
                    if (sample_light<false, kTex, kTab>(sc, lsi, rng, &ls)) {
                        float phase = henyey_greenstein(md.g, dot(wo, ls.dir));
                        // shadow_ray_from_medium (pt.rs:212-233): probe the last-hit basic primitive
                        // with the world-space ray, in its own object space
                        DRay sr;
                        sr.o = pi; sr.d = ls.dir; sr.t_min = kTMinEps;
                        float probe_max = ls.dist - 0.001f;
                        float transported = ls.dist;
                        bool probe_hit = false;
                        float probe_t = 0.0f;
                        if (does_hit && it.prim_type == SPT_PRIM_MESH) {
                            float t, v, w;
                            if (tri_test(sc.tri_pos, (uint32_t)h.prim, sr, &t, &v, &w) && t > sr.t_min && t < probe_max) { probe_hit = true; probe_t = t; }
#if SPT_WITH_BEZIER
                        } else if (does_hit && it.prim_type == SPT_PRIM_BEZIER) {
                            float u, v, t;
                            if (bezier_intersect_ray(sc.bez + 16u * it.prim_id, sr, &u, &v, &t) && t > sr.t_min && t < probe_max) { probe_hit = true; probe_t = t; }
#endif
                        } else if (does_hit) {
                            float mn, mx;
                            if (sphere_roots(sc.spheres[it.prim_id], sr, &mn, &mx)) {
                                float t = (mn < sr.t_min) ? mx : mn;
                                if (sr.t_min < t && t < probe_max) { probe_hit = true; probe_t = t; }
                            }
                        }
                        if (probe_hit) { transported = probe_t; sr.t_min += probe_t; }
                        else { sr.t_min += ls.dist - 0.001f; }
                        f3 atten = cexp((-sigma_t) * transported);
                        if (ls.pdf != 0.0f && spt_is_finite(ls.pdf)) {
                            f3 li;
                            if (ls.is_delta) {
                                li = crcp((atten * phase) * ls.strength, ls.pdf);
                            } else {
                                float weight = power_heuristic(ls.pdf, phase);
                                li = crcp(((atten * phase) * ls.strength) * weight, ls.pdf);
                            }
                            contrib = thr * li;
                            shadow_ray = sr;
                            shadow_tmax = ls.dist - 0.001f;
                            want_shadow = true;
                        }
                    }
                    // Homogeneous::sample_wi (homogeneous.rs:60-70)
                    float r0 = rng.next(), r1 = rng.next();
                    float cos_theta = hg_cdf_inverse(md.g, r0);
                    float sin_theta = spt_sqrt(1.0f - cos_theta * cos_theta);
                    float phi = 2.0f * SPT_PI * r1;
                    float sp, cp;
                    spt_sincos(phi, &sp, &cp);
                    f3 wi = hg_local_to_world(wo, mk3(sin_theta * cp, sin_theta * sp, cos_theta));
                    next_pdf = henyey_greenstein(md.g, cos_theta);
                    next_ray.o = pi; next_ray.d = wi; next_ray.t_min = kTMinEps;
                    scattered = true;
                } else {
                    // left the medium: `continue` (pt.rs:62-64) re-traces the SAME ray with no medium and
                    // without RR / depth change; the recorded hit is that re-trace's result.
                    float atten_pdf = cavg(attenuation);
                    thr = thr * crcp(attenuation, atten_pdf);
                    medium = -1;
                }
            }

            if (!scattered && medium < 0) {
                if (!kSimple && !does_hit) {  // pt.rs:97-111 (only reached here after leaving a medium)
                    if (sc.env_w != 0u) {
                        f3 env;
                        float env_pdf;
                        env_strength_pdf(sc, ray.d, &env, &env_pdf);
                        float weight = 1.0f;
                        if (depth != 0u) weight = power_heuristic(last_pdf, pdf_env_light(sc) * env_pdf);
                        slot_add(slot, (thr * env) * weight);
                    }
                    alive = false;
                } else if (rc.debug_normal != 0u) {  // pt.rs:113-118: final_color = normal * 0.5 + 0.5; break (wave-uniform branch)
                    const f3 nc = it.normal * 0.5f + gray(0.5f);
                    if (kFirst) first_acc = nc;        // replaces, as the reference's assignment does, whatever the path gathered before
                    else rad_store(rc, slot, nc);
                    alive = false;
                } else {  // pt.rs:112-193
                    const spt_surface sf = load_surface<kTab>(sc, it.surface);
                    const uint32_t sflags = sf.flags;
                    DMat mt = material_at<kTex, kTab, kPndf>(sc, sf.material, it);
                    if (kSimple) mt.bxdf = SPT_BXDF_LAMBERT;
                    DCoord coord = surface_coord<kTex>(sc, sf, ray, it);
                    f3 po = it.position;
                    f3 le = surface_emissive<kTex>(sc, sf, it);
                    if (!kSimple && luminance(le) > 0.0f) {
                        float weight = 1.0f;
                        if (depth != 0u) weight = power_heuristic(last_pdf, pdf_shape_light(sc, (uint32_t)h.inst, sflags, lsi, it, h.prim));
                        slot_add(slot, (thr * le) * weight);
                    }
                    f3 wo = coord.to_local(-ray.d);
                    DBxdfSample samp;
                    if (kSubsurface) {   // Subsurface substrate: the BSSRDF probe ray is traced right here
                        DSubsurfaceIo ssio;
                        ssio.has = false;
                        ssio.po = po;
                        ssio.coord_po = coord;
                        samp = mat_sample<true, kGeoLds, kTab, kPndf>(mt, wo, rng, &sc, &ssio);
                        if (ssio.has) {  // pt.rs:147-151
                            po = ssio.pi;
                            coord = ssio.coord_pi;
                            thr = thr * crcp(ssio.sp, ssio.pdf_pi);
                        }
                    } else {
#ifdef SPT_EXP_CHEAP_SAMPLE
                        samp.wi = reflect_z(wo); samp.f = gray(1.0f); samp.pdf = 1.0f; samp.transmit = false;
                        (void)rng.next();
#else
                        if (kPndf) samp = mat_sample<false, false, kTab, true>(mt, wo, rng, &sc);
                        else samp = mat_sample(mt, wo, rng);
#endif
                    }
                    lsi = po;
#ifdef SPT_EXP_NO_LIGHT   /* measurement-only builds (tools/shade_attribution.sh): the film is WRONG */
                    if (false) {
#else
                    if (!mat_is_delta(mt)) {
#endif
                        DLightSample ls;
                        if (sample_light<kSimple, kTex, kTab>(sc, lsi, rng, &ls)) {
                            f3 wi = coord.to_local(ls.dir);
                            DPndfMemo memo{0.0f, false};    // bxdf() and pdf() of a glint lobe share one tree walk
                            f3 f = mat_eval<kPndf>(mt, wo, wi, &sc, &memo);
                            float mpdf = mat_pdf<kPndf>(mt, wo, wi, &sc, &memo);
                            if (ls.pdf != 0.0f && spt_is_finite(ls.pdf)) {
                                f3 li;
                                if (ls.is_delta) {
                                    li = crcp((ls.strength * f) * spt_abs(wi.z), spt_max(ls.pdf, 0.00001f));
                                } else {
                                    float weight = power_heuristic(ls.pdf, mpdf);
                                    li = crcp(((ls.strength * f) * spt_abs(wi.z)) * weight, spt_max(ls.pdf, 0.00001f));
                                }
                                contrib = thr * li;
                                shadow_ray.o = po; shadow_ray.d = ls.dir;
                                shadow_ray.t_min = kTMinEps / spt_max(spt_abs(wi.z), 0.00001f);
                                shadow_tmax = ls.dist - 0.001f;
                                want_shadow = true;
                            }
                        }
                    }
                    if (kSubsurface && !all_finite(thr) && !mat_is_delta(mt)) {
                        // Only the BSSRDF relocation above can leave a non-finite throughput in front of the light
                        // sample (sp / pdf_pi = 0 / 0 where the profile underflows).  The reference then adds
                        // throughput * li with li = 0 for an occluded or rejected light sample (pt.rs:163-181), which
                        // is NaN, not "nothing": the one place where skipping a zero contribution is not the same
                        // thing.  Resolved right here (k_shade<3> can trace), the shadow queue never sees it.
                        f3 add = thr * 0.0f;
                        if (want_shadow && !trace_any<kGeoLds>(sc, shadow_ray, shadow_tmax)) add = contrib;
                        slot_add(slot, add);
                        want_shadow = false;
                    }
                    next_pdf = samp.pdf;
                    f3 wi_world = coord.to_world(samp.wi);
                    next_ray.o = po; next_ray.d = wi_world;
                    next_ray.t_min = kTMinEps / spt_max(spt_abs(samp.wi.z), 0.00001f);
                    thr = thr * crcp(samp.f * spt_abs(samp.wi.z), spt_max(samp.pdf, 0.00001f));
                    float hd = dot(wi_world, coord.hemi);  // Coordinate::in_expected_hemisphere
                    if (!(samp.transmit ? (hd <= 0.0f) : (hd >= 0.0f))) alive = false;
                    if (!kSimple && alive && dot(wi_world, it.normal) < 0.0f)
                        medium = (sflags & SPT_SURF_DOUBLE_SIDED) ? -1 : sf.inside_medium;  // Surface::inside_medium
                    scattered = true;
                }
            }

            // a zero contribution needs no visibility test (the reference adds 0)
            if (want_shadow && contrib.x == 0.0f && contrib.y == 0.0f && contrib.z == 0.0f) want_shadow = false;

            if (alive && scattered) {  // pt.rs:195-206
                if (all_finite(thr)) {
                    float rr_rand = rng.next();
                    float rr_prop = spt_clamp(luminance(thr), 0.001f, 0.95f);
                    if (!(rr_rand > rr_prop)) {
                        thr = thr * (1.0f / rr_prop);
                        depth += 1u;
                        want_ext = depth < rc.max_depth;
                    }
                }
            }
        }
        if (kFused) {
            // the two queue counters still count the segments (spt_render_stats); the slots are not used
            const PendingPush ps = wave_push_issue(want_shadow, kLoop ? q_count(rc.counts, b_cur, Q_SHADOW, shard) : shadow_count);
            const PendingPush pe = wave_push_issue(want_ext, kLoop ? q_count(rc.counts, b_cur, Q_EXT, shard) : ext_count);
            if (want_shadow && !trace_any<true>(sc, shadow_ray, shadow_tmax)) slot_add(slot, contrib);   // k_shadow
            bool keep = false;                                                                               // k_extend
            DHit nh;
            nh.inst = -1; nh.t = SPT_F32_MAX; nh.prim = -1; nh.v = 0.0f; nh.w = 0.0f;
            if (want_ext && b_cur + 1u < rc.max_depth) {   // the host launches no k_extend after the last bounce either
                nh = trace_closest<true>(sc, next_ray, SPT_F32_MAX);
                if (nh.inst >= 0 || medium >= 0) {
                    keep = true;
                } else if (!kSimple && sc.env_w != 0u) {  // pt.rs:97-111, curr_depth > 0 here
                    f3 env;
                    float env_pdf;
                    env_strength_pdf(sc, next_ray.d, &env, &env_pdf);
                    float weight = power_heuristic(next_pdf, pdf_env_light(sc) * env_pdf);
                    slot_add(slot, (thr * env) * weight);
                }
            }
            if (kFirst && active) rad_store(rc, slot, first_acc);
            (void)wave_push_finish(ps);
            (void)wave_push_finish(pe);
            const uint32_t ns = qbase + wave_push(keep, q_count(rc.counts, b_cur + 1, Q_HIT, shard));
            if (kLoop) {
                carried = true;
                lane_on = keep;
                if (keep) { c_ray = next_ray; c_pdf = next_pdf; c_h = nh; }
                if (__ballot(lane_on) == 0ull) break;   // wave-uniform: every lane's path has ended
                b_cur += 1u;
                continue;
            }
            if (keep) {
                store_path(rc.qb, ns, next_ray, next_pdf, thr, slot, lsi, pack_meta(depth, medium), rng);
                rc.hits_next.t_v_w_prim[ns] = make_float4(nh.t, nh.v, nh.w, __int_as_float(nh.prim));
                rc.hits_next.inst_src[ns] = make_uint2((uint32_t)nh.inst, ns);
            }
        }
        break;
        }   // turns
        if (kFused) continue;
        if (kFirst && active) rad_store(rc, slot, first_acc);   // before k_shadow / k_extend of this bounce add to it
        // both reservations in flight together: one atomic round trip per iteration instead of two
        const PendingPush ps = wave_push_issue(want_shadow, shadow_count);
        const PendingPush pe = wave_push_issue(want_ext, ext_count);
        uint32_t ss = qbase + wave_push_finish(ps);
        if (want_shadow) {
            rc.shadow.o_tmin[ss] = make_float4(shadow_ray.o.x, shadow_ray.o.y, shadow_ray.o.z, shadow_ray.t_min);
            rc.shadow.d_tmax[ss] = make_float4(shadow_ray.d.x, shadow_ray.d.y, shadow_ray.d.z, shadow_tmax);
            rc.shadow.contrib_slot[ss] = make_float4(contrib.x, contrib.y, contrib.z, __uint_as_float(slot));
        }
        uint32_t es = qbase + wave_push_finish(pe);
        if (want_ext) store_path(rc.qb, es, next_ray, next_pdf, thr, slot, lsi, pack_meta(depth, medium), rng);
    }
    }   // classes
}

// ---------------------------------------------------------------------------- shadow
// kFlat (with kLds, kCount off): the exhaustive loops of flat.h instead of the tree walk
template <bool kLds, bool kCount = false, bool kFlat = false>
__global__ void __launch_bounds__(256, (kLds && !SPT_WITH_BEZIER) ? SPT_W_SHD_L : 1) k_shadow(DScene sc, RenderCtx rc, uint32_t bounce) {
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = *q_count(rc.counts, bounce, Q_SHADOW, shard);
    if ((blockIdx.x / kShards) * blockDim.x >= n) return;   // nothing for this workgroup: skip the LDS staging too
    stage_geometry<kLds>(sc);
    LaneVisits vc{0u, 0u, 0u};
    const uint32_t qbase = shard * rc.shard_cap;
    const uint32_t stride = (gridDim.x / kShards) * blockDim.x;
    if (kFlat) {   // flat.h wants whole waves: a uniform loop, `active` marks the lanes that carry a ray
        for (uint32_t i0 = (blockIdx.x / kShards) * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
            const bool active = i0 + lane_id() < n;
            const uint32_t idx = qbase + (active ? i0 + lane_id() : 0u);
            const float4 a = rc.shadow.o_tmin[idx], b = rc.shadow.d_tmax[idx], c = rc.shadow.contrib_slot[idx];
            DRay r;
            r.o = mk3(a); r.t_min = a.w; r.d = mk3(b);
            const bool occluded = flat_any(sc, r, b.w, active);
            if (active && !occluded) rad_add(rc, __float_as_uint(c.w), mk3(c));
        }
        return;
    }
    for (uint32_t i = (blockIdx.x / kShards) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t idx = qbase + i;
        float4 a = rc.shadow.o_tmin[idx], b = rc.shadow.d_tmax[idx], c = rc.shadow.contrib_slot[idx];
        DRay r;
        r.o = mk3(a); r.t_min = a.w; r.d = mk3(b);
        if (!trace_any<kLds, kCount>(sc, r, b.w, &vc)) rad_add(rc, __float_as_uint(c.w), mk3(c));
    }
    if (kCount) flush_visits(rc, vc, 1u);
}

// ---------------------------------------------------------------------------- extend
template <bool kLds, bool kCount = false, bool kFlat = false>
__global__ void __launch_bounds__(256, (kLds && !SPT_WITH_BEZIER) ? SPT_W_EXT_L : 1) k_extend(DScene sc, RenderCtx rc, uint32_t bounce) {
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = *q_count(rc.counts, bounce, Q_EXT, shard);
    if ((blockIdx.x / kShards) * blockDim.x >= n) return;   // nothing for this workgroup: skip the LDS staging too
    stage_geometry<kLds>(sc);
    LaneVisits vc{0u, 0u, 0u};
    const uint32_t qbase = shard * rc.shard_cap;
    const uint32_t stride = (gridDim.x / kShards) * blockDim.x;
    const bool env = sc.env_w != 0u;
    for (uint32_t i0 = (blockIdx.x / kShards) * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
        const bool active = i0 + lane_id() < n;
        const uint32_t idx = qbase + (active ? i0 + lane_id() : 0u);   // (inactive lanes read entry 0 of the shard and discard what they find)
        // everything the vertex may need is requested with the ray: waiting for the throughput of a miss after the walk
        // would be a memory round trip of its own, in a kernel that spends two thirds of its time waiting for those
        const float4 a = rc.qb.o_tmin[idx], b = rc.qb.d_pdf[idx];
        const uint32_t meta = __float_as_uint(rc.qb.lsi_meta[idx].w);
        float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (env) c = rc.qb.thr_slot[idx];
        DRay r;
        r.o = mk3(a); r.t_min = a.w; r.d = mk3(b);
        DHit h;
        h.inst = -1; h.t = SPT_F32_MAX; h.prim = -1; h.v = 0.0f; h.w = 0.0f;
        if (kFlat) h = flat_closest(sc, r, SPT_F32_MAX, active);     // (whole waves, see flat.h)
        else if (active) h = trace_closest<kLds, kCount>(sc, r, SPT_F32_MAX, &vc);
        bool keep = false;
        if (active) {
            const bool in_medium = (meta >> 8) != 0u;
            if (h.inst >= 0 || in_medium) {
                keep = true;
            } else if (env) {  // pt.rs:97-111, curr_depth > 0 here
                f3 env_rgb;
                float env_pdf;
                env_strength_pdf(sc, r.d, &env_rgb, &env_pdf);
                float weight = power_heuristic(b.w, pdf_env_light(sc) * env_pdf);
                rad_add(rc, __float_as_uint(c.w), (mk3(c) * env_rgb) * weight);
            }
        }
        // a kept path leaves only its hit and the index of its record (which stays where the shade stage wrote it)
        const uint32_t slot = hit_push<kLds>(sc, rc, keep, h.inst, bounce + 1u, shard);
        if (keep) {
            rc.hits.t_v_w_prim[slot] = make_float4(h.t, h.v, h.w, __int_as_float(h.prim));
            rc.hits.inst_src[slot] = make_uint2((uint32_t)h.inst, idx);
        }
    }
    if (kCount) flush_visits(rc, vc, 2u);
}

// ---------------------------------------------------------------------------- persistent, refilling variants
// Used for scenes whose geometry does not fit LDS (incoherent, deep traversals).  Each wave keeps 64
// walkers; whenever fewer than kRefillBelow of them still hold a ray, the idle lanes pull the next
// entries of their queue shard (one atomicAdd per wave on the shard's consume cursor, __ballot /
// __popcll prefix for the per-lane index) and start new walks while the others continue.  Results
// are retired with the usual wave-aggregated push.  Per-ray arithmetic and visit order are those of
// trace_closest / trace_any, so the output is bit-identical to the non-refilling kernels.
// MEASURED (1 M-triangle scene, 1024^2 x 32 spp): shadow 8.8 ms vs 10.9 ms nested (any-hit walks end at
// very different times, so refilling pays), extend 28 ms vs 20 ms nested (every lane walks to the end
// anyway and the one-step-per-call state machine costs more than the idle lanes it saves).  Hence the
// (that was with the 2-wide nodes; with the 4-wide nodes k_extend_dyn wins too, 16.0 vs 17.7 ms).  Default: both on
// for scenes that do not fit LDS (SPT_NO_DYN_SHADOW=1 / SPT_NO_DYN_EXTEND=1 switch them off).
// First version kept the spill array inside the walker struct, which dragged the whole walker into
// scratch memory (40 scratch loads/stores per step) and made it 2.4x slower than nested.
constexpr uint32_t kRefillBelow = 40;   // refill when fewer lanes than this are walking
constexpr uint32_t kStepsPerCheck = 8;  // traversal steps between two utilisation checks

// lanes with `want` get consecutive indices from *cursor; returns the index (>= n means: queue empty)
SPT_DEV uint32_t wave_pull(bool want, uint32_t* cursor) {
    unsigned long long mask = __ballot(want);
    if (mask == 0ull) return 0xffffffffu;
    uint32_t lane = lane_id();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(cursor, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    unsigned long long lt = (lane == 0u) ? 0ull : (~0ull >> (64u - lane));
    return want ? base + (uint32_t)__popcll(mask & lt) : 0xffffffffu;
}

template <bool kCount>
__global__ void __launch_bounds__(256, SPT_WITH_BEZIER ? 2 : SPT_W_SHD) k_shadow_dyn(DScene sc, RenderCtx rc, uint32_t bounce) {
    stage_geometry<false>(sc);
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = *q_count(rc.counts, bounce, Q_SHADOW, shard);
    uint32_t* cursor = q_count(rc.counts, bounce, Q_SHADOW_CURSOR, shard);
    const uint32_t qbase = shard * rc.shard_cap;
    uint2 spill_mem[kSpillStack];
    Walker<false, false, kCount> wk;
    wk.vc = LaneVisits{0u, 0u, 0u};
    bool busy = false, drained = false;
    uint32_t idx = 0;
    while (true) {
        const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
        if (!drained && n_busy < rc.dyn_refill_below) {
            uint32_t i = wave_pull(!busy, cursor);
            if (!busy && i < n) {
                idx = qbase + i;
                float4 a = rc.shadow.o_tmin[idx], b = rc.shadow.d_tmax[idx];
                DRay r;
                r.o = mk3(a); r.t_min = a.w; r.d = mk3(b);
                wk.start(sc, r, b.w, spill_mem);
                busy = true;
            }
            // the cursor only grows: once any lane was refused the shard is empty for good
            drained = __ballot(!busy) != 0ull && __ballot(!busy && i >= n && i != 0xffffffffu) != 0ull;
        }
        if (__ballot(busy) == 0ull) break;
        for (uint32_t k = 0; k < rc.dyn_steps; ++k)
            if (busy && !wk.done) wk.step(sc);
        if (busy && wk.done) {
            if (wk.h.inst < 0) {  // not occluded
                float4 c = rc.shadow.contrib_slot[idx];
                rad_add(rc, __float_as_uint(c.w), mk3(c));
            }
            busy = false;
        }
    }
    if (kCount) flush_visits(rc, wk.vc, 1u);
}

template <bool kCount>
__global__ void __launch_bounds__(256, 2) k_extend_dyn(DScene sc, RenderCtx rc, uint32_t bounce) {
    stage_geometry<false>(sc);
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = *q_count(rc.counts, bounce, Q_EXT, shard);
    uint32_t* cursor = q_count(rc.counts, bounce, Q_EXT_CURSOR, shard);
    const uint32_t qbase = shard * rc.shard_cap;
    uint2 spill_mem[kSpillStack];
    Walker<false, true, kCount> wk;
    wk.vc = LaneVisits{0u, 0u, 0u};
    bool busy = false, drained = false, in_medium = false;
    uint32_t idx = 0;
    while (true) {
        const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
        if (!drained && n_busy < rc.dyn_refill_below) {
            uint32_t i = wave_pull(!busy, cursor);
            if (!busy && i < n) {
                idx = qbase + i;
                float4 a = rc.qb.o_tmin[idx], b = rc.qb.d_pdf[idx];
                in_medium = (__float_as_uint(rc.qb.lsi_meta[idx].w) >> 8) != 0u;
                DRay r;
                r.o = mk3(a); r.t_min = a.w; r.d = mk3(b);
                wk.start(sc, r, SPT_F32_MAX, spill_mem);
                busy = true;
            }
            drained = __ballot(!busy) != 0ull && __ballot(!busy && i >= n && i != 0xffffffffu) != 0ull;
        }
        if (__ballot(busy) == 0ull) break;
        for (uint32_t k = 0; k < rc.dyn_steps; ++k)
            if (busy && !wk.done) wk.step(sc);
        const bool retire = busy && wk.done;
        bool keep = false;
        if (retire) {
            if (wk.h.inst >= 0 || in_medium) {
                keep = true;
            } else if (sc.env_w != 0u) {  // pt.rs:97-111, curr_depth > 0 here
                const float4 b = rc.qb.d_pdf[idx], c = rc.qb.thr_slot[idx];
                f3 env;
                float env_pdf;
                env_strength_pdf(sc, mk3(b), &env, &env_pdf);
                const float weight = power_heuristic(b.w, pdf_env_light(sc) * env_pdf);
                rad_add(rc, __float_as_uint(c.w), (mk3(c) * env) * weight);
            }
            busy = false;
        }
        // a kept path leaves only its hit and the index of its record (which stays where the shade stage wrote it)
        const uint32_t slot = hit_push<false>(sc, rc, keep, wk.h.inst, bounce + 1u, shard);
        if (keep) {
            rc.hits.t_v_w_prim[slot] = make_float4(wk.h.t, wk.h.v, wk.h.w, __int_as_float(wk.h.prim));
            rc.hits.inst_src[slot] = make_uint2((uint32_t)wk.h.inst, idx);
        }
    }
    if (kCount) flush_visits(rc, wk.vc, 2u);
}

// ---------------------------------------------------------------------------- streaming kernels (stream.h)
// The large-scene counterparts of k_primary / k_shadow / k_extend: persistent waves of 64 SWalkers that refill
// idle lanes (from the queue shard, or - primary - with the lane's next camera sample) between while-while rounds.
// Per-ray arithmetic of every PRIMITIVE test is that of trace.h, so the films are bit-identical to the other kernels'.
constexpr uint32_t kStreamGuard = 1u << 24;   // outer iterations a persistent wave may run

template <bool kCount>
__global__ void __launch_bounds__(256, 2) k_shadow_stream(DScene sc, RenderCtx rc, uint32_t bounce) {
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = *q_count(rc.counts, bounce, Q_SHADOW, shard);
    uint32_t* cursor = q_count(rc.counts, bounce, Q_SHADOW_CURSOR, shard);
    const uint32_t qbase = shard * rc.shard_cap;
    uint2 spill_mem[kSpillStack];
    SWalker<false, kCount> wk;
    wk.vc = LaneVisits{0u, 0u, 0u};
    wk.done = true;
    wk.cur = kNoRef;
    bool busy = false, drained = false;
    uint32_t idx = 0;
    // every wave leaves this loop: the queue is finite and every walk ends; the bound is a guard against a corrupt scene
    for (uint32_t guard = 0; guard < kStreamGuard; ++guard) {
        // Finished lanes are retired when the wave refills, not in the round they finish in: a retirement waits for two
        // dependent loads (the contribution, then the radiance slot), and paying that once per refill instead of once per
        // round keeps the walking lanes walking.
        const uint32_t n_walking = (uint32_t)__popcll(__ballot(busy && !wk.done));
        if (n_walking < rc.stream_refill_below) {
            if (busy && wk.done) {
                if (wk.h.inst < 0) {  // not occluded
                    const float4 c = rc.shadow.contrib_slot[idx];
                    rad_add(rc, __float_as_uint(c.w), mk3(c));
                }
                busy = false;
            }
            if (!drained) {
                const uint32_t i = wave_pull(!busy, cursor);
                if (!busy && i < n) {
                    idx = qbase + i;
                    const float4 a = rc.shadow.o_tmin[idx], b = rc.shadow.d_tmax[idx];
                    DRay r;
                    r.o = mk3(a); r.t_min = a.w; r.d = mk3(b);
                    wk.begin(sc, r, b.w);
                    busy = true;
                }
                // the cursor only grows: once any lane was refused the shard is empty for good
                drained = __ballot(!busy && i >= n && i != 0xffffffffu) != 0ull;
            }
        }
        if (__ballot(busy) == 0ull) break;
        wk.run(sc, rc.stream_rounds, spill_mem);
    }
    if (kCount) flush_visits(rc, wk.vc, 1u);
}

template <bool kCount>
__global__ void __launch_bounds__(256, SPT_WITH_BEZIER ? 2 : SPT_W_EXT) k_extend_stream(DScene sc, RenderCtx rc, uint32_t bounce) {
    const uint32_t shard = blockIdx.x % kShards;
    const uint32_t n = *q_count(rc.counts, bounce, Q_EXT, shard);
    uint32_t* cursor = q_count(rc.counts, bounce, Q_EXT_CURSOR, shard);
    const uint32_t qbase = shard * rc.shard_cap;
    uint2 spill_mem[kSpillStack];
    SWalker<true, kCount> wk;
    wk.vc = LaneVisits{0u, 0u, 0u};
    wk.done = true;
    wk.cur = kNoRef;
    bool busy = false, drained = false, in_medium = false;
    uint32_t idx = 0;
    // every wave leaves this loop: the queue is finite and every walk ends; the bound is a guard against a corrupt scene
    for (uint32_t guard = 0; guard < kStreamGuard; ++guard) {
        // Finished lanes are retired when the wave refills, not in the round they finish in (see k_shadow_stream): a
        // retirement is a class lookup, an atomic append and, for a miss under an environment, a read-modify-write.
        const uint32_t n_walking = (uint32_t)__popcll(__ballot(busy && !wk.done));
        if (n_walking < rc.stream_refill_below) {
            const bool retire = busy && wk.done;
            bool keep = false;
            if (retire) {
                if (wk.h.inst >= 0 || in_medium) {
                    keep = true;
                } else if (sc.env_w != 0u) {  // pt.rs:97-111, curr_depth > 0 here
                    const float4 b = rc.qb.d_pdf[idx], c = rc.qb.thr_slot[idx];
                    f3 env;
                    float env_pdf;
                    env_strength_pdf(sc, mk3(b), &env, &env_pdf);
                    const float weight = power_heuristic(b.w, pdf_env_light(sc) * env_pdf);
                    rad_add(rc, __float_as_uint(c.w), (mk3(c) * env) * weight);
                }
                busy = false;
            }
            // a kept path leaves only its hit and the index of its record (which stays where the shade stage wrote it)
            const uint32_t slot = hit_push<false>(sc, rc, keep, wk.h.inst, bounce + 1u, shard);
            if (keep) {
                rc.hits.t_v_w_prim[slot] = make_float4(wk.h.t, wk.h.v, wk.h.w, __int_as_float(wk.h.prim));
                rc.hits.inst_src[slot] = make_uint2((uint32_t)wk.h.inst, idx);
            }
            if (!drained) {
                const uint32_t i = wave_pull(!busy, cursor);
                if (!busy && i < n) {
                    idx = qbase + i;
                    const float4 a = rc.qb.o_tmin[idx], b = rc.qb.d_pdf[idx];
                    in_medium = (__float_as_uint(rc.qb.lsi_meta[idx].w) >> 8) != 0u;
                    DRay r;
                    r.o = mk3(a); r.t_min = a.w; r.d = mk3(b);
                    wk.begin(sc, r, SPT_F32_MAX);
                    busy = true;
                }
                drained = __ballot(!busy && i >= n && i != 0xffffffffu) != 0ull;
            }
        }
        if (__ballot(busy) == 0ull) break;
        wk.run(sc, rc.stream_rounds, spill_mem);
    }
    if (kCount) flush_visits(rc, wk.vc, 2u);
}

// k_primary for large scenes: a lane owns a pixel and walks its samples one after the other; a lane whose walk is done
// retires the sample (hit record / environment term / zero slot, exactly as k_primary) and starts its next one while the
// rest of the wave keeps walking.  Same tiles, chunks, slots and film semantics as k_primary<false, kChunked>.
template <bool kChunked, bool kCount>
__global__ void __launch_bounds__(256, 2) k_primary_stream(DScene sc, RenderCtx rc) {
    const uint32_t tile = kChunked ? blockIdx.x % rc.n_tiles : blockIdx.x, chunk = kChunked ? blockIdx.x / rc.n_tiles : 0u;
    const uint32_t tx = tile % rc.tiles_x, ty = tile / rc.tiles_x;
    const uint32_t i = tx * kTile + (threadIdx.x % kTile);
    const uint32_t row_local = ty * kTile + (threadIdx.x / kTile);
    const bool valid = (i < rc.width) && (row_local < rc.rows);
    const uint32_t lp = valid ? row_local * rc.width + i : 0u;
    const uint32_t j = global_row(rc, row_local);
    const uint32_t pixel = j * rc.width + i;
    const bool has_env = sc.env_w != 0u;
    const bool lazy_rng = rc.sampler == SPT_SAMPLER_RECURRENCE;
    const uint32_t shard = tile_shard(tx, ty);
    f3 sum = mk3(0, 0, 0);
    if (!kChunked && valid) sum = mk3(rc.film[3 * lp], rc.film[3 * lp + 1], rc.film[3 * lp + 2]);
    uint32_t first = kChunked ? 0u : rc.pass_samples;
    const size_t plane = rc.rad_plane;
    uint32_t* hit_counter = q_count(rc.counts, 0, Q_HIT, shard);
    bool in_bounds = (int32_t)i >= rc.cull_i0 && (int32_t)i <= rc.cull_i1 && (int32_t)j >= rc.cull_j0 && (int32_t)j <= rc.cull_j1;
    if (in_bounds && valid && rc.row_span != nullptr) {   // the row's own span inside the rectangle (projected hulls of the instances' boxes)
        const int2 span = rc.row_span[j];
        in_bounds = (int32_t)i >= span.x && (int32_t)i <= span.y;
    }
    const bool live = valid && (in_bounds || has_env);
    const uint32_t s_begin = kChunked ? chunk * rc.chunk_samples : 0u;
    const uint32_t s_end = live ? (kChunked ? min(s_begin + rc.chunk_samples, rc.pass_samples) : rc.pass_samples) : s_begin;
    uint2 spill_mem[kSpillStack];
    SWalker<true, kCount> wk;
    wk.vc = LaneVisits{0u, 0u, 0u};
    wk.done = true;
    wk.cur = kNoRef;
    uint32_t s = s_begin;        // next sample to start
    uint32_t s_cur = 0u;         // sample being walked / retired
    uint32_t slot_mask = 0u;     // chunked: the slot bits of the current group of 8 samples (chunk_samples is a multiple of 8)
    bool busy = false;
    f3 dir = mk3(0, 0, 0);
    // every wave leaves this loop: the queue is finite and every walk ends; the bound is a guard against a corrupt scene
    for (uint32_t guard = 0; guard < kStreamGuard; ++guard) {
        // start the next sample of every idle lane that has one (a sample that cannot hit anything retires at once)
        if (!busy && s < s_end) {
            s_cur = s++;
            const uint32_t gs = rc.pass_first + s_cur;
            DRng rng;
            rng.s.state = 0ull;
            if (!lazy_rng) rng.s = spt_rng_seed(rc.seed, pixel, gs);
            float ox, oy;
            pixel_offset(rc, pixel, gs, rng, &ox, &oy);
            const float x = (((float)i + ox) * rc.width_inv - 0.5f) * rc.aspect;           // pt.rs:269
            const float y = ((float)(rc.height - j - 1u) + oy) * rc.height_inv - 0.5f;       // pt.rs:270-271
            const f3 du = (rc.cam.forward * rc.cam.half_cot + rc.cam.right * x) + rc.cam.up * y;
            bool may_hit = in_bounds;
            if (may_hit && rc.bs_valid && rc.bs_c > 0.0f) {
                const float b = dot(du, rc.bs_oc);
                may_hit = (b > 0.0f) && (b * b >= dot(du, du) * rc.bs_c);
            }
            dir = du;
            if (may_hit || has_env) dir = normalize(du);
            busy = true;
            wk.done = true;
            wk.cur = kNoRef;
            wk.h.inst = -1;
            if (may_hit) {
                DRay ray;
                ray.o = rc.cam.eye; ray.t_min = kTMinEps; ray.d = dir;
                wk.begin(sc, ray, SPT_F32_MAX);
            }
        }
        if (__ballot(busy) == 0ull) break;
        wk.run(sc, rc.stream_rounds, spill_mem);
        const bool retire = busy && wk.done;
        const bool hit = retire && wk.h.inst >= 0;
        const size_t ri = (size_t)s_cur * rc.n_pixels + lp;
        if (retire && !hit) {
            if (has_env) {  // pt.rs:98-110 at depth 0: weight 1
                f3 env;
                float env_pdf;
                env_strength_pdf(sc, dir, &env, &env_pdf);
                const f3 c = mk3(0, 0, 0) + (gray(1.0f) * env) * 1.0f;
                if (first == rc.pass_samples) {
                    sum = sum + c;
                } else {
                    rc.rad[ri] = c.x; rc.rad[plane + ri] = c.y; rc.rad[2 * plane + ri] = c.z;
                }
            } else if (!kChunked && first != rc.pass_samples) {   // chunked: a black sample leaves no trace (slot_bits)
                rc.rad[ri] = 0.0f; rc.rad[plane + ri] = 0.0f; rc.rad[2 * plane + ri] = 0.0f;
            }
        }
        // (a hit's slot is not zeroed here: the bounce-0 shade kernel WRITES the sample's first contributions, see k_shade)
        if (hit && first == rc.pass_samples) first = s_cur;
        if (kChunked && retire) {    // a lane retires its samples in order: s_cur runs through [s_begin, s_end)
            slot_mask |= ((hit || has_env) ? 1u : 0u) << (s_cur & 7u);
            if ((s_cur & 7u) == 7u || s_cur + 1u == s_end) {
                rc.slot_bits[(size_t)(s_cur >> 3) * rc.n_pixels + lp] = (uint8_t)slot_mask;
                slot_mask = 0u;
            }
        }
        const uint32_t slot = rc.n_classes > 1u ? hit_push<false>(sc, rc, hit, wk.h.inst, 0u, shard) : shard * rc.shard_cap + wave_push(hit, hit_counter);
        if (hit) {
            rc.qa.d_pdf[slot] = make_float4(dir.x, dir.y, dir.z, __uint_as_float((uint32_t)ri));
            rc.hits.t_v_w_prim[slot] = make_float4(wk.h.t, wk.h.v, wk.h.w, __int_as_float(wk.h.prim));
            rc.hits.inst_src[slot] = rc.pack_first ? make_uint2((uint32_t)wk.h.inst | (s_cur << 20), pixel) : make_uint2((uint32_t)wk.h.inst, slot);
        }
        if (retire) busy = false;
    }
    if (kCount) flush_visits(rc, wk.vc, 0u);
    if (kChunked) {
        if (valid && chunk == 0u) rc.first_slot[lp] = (in_bounds || has_env) ? 0u : rc.pass_samples;
    } else if (valid) {
        rc.film[3 * lp] = sum.x; rc.film[3 * lp + 1] = sum.y; rc.film[3 * lp + 2] = sum.z;
        rc.first_slot[lp] = first;
    }
}

