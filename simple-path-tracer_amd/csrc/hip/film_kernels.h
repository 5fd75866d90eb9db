// Film / filter kernels and the small test seams: cheap to compile, only spt_hip.hip includes this file (the heavy
// kernel templates of kernels.h are instantiated in their own translation units, see kernel_list.h).
#pragma once
#include "kernels.h"

// ---------------------------------------------------------------------------- resolve
// k_resolve for passes of the chunked k_primary (slot bits; the only form the headline workload runs).
// Only the samples marked in slot_bits own a slot; the others were black and add nothing (film.rs:87 adds +0 for them: the
// same sum).  The additions of a pixel are sequential, so what the kernel waits for is memory latency, once per batch:
// kBatch samples = their mask bytes + 3 kBatch slot loads are requested TOGETHER - the slot loads do not wait for the mask; an
// unmarked slot is read (whatever it holds) and not added.  With the per-row spans nearly every live sample is marked, so the
// extra reads are few.  (Round 2's form - mask first, then the marked slots of one byte at a time - took one round trip per 8
// samples: 0.07 ms per launch whatever the shard size, 0.17 of the 0.6 ms step of an 8-GPU rank.)
template <uint32_t kResolveBatch>
__global__ void __launch_bounds__(256, 3) k_resolve_bits(RenderCtx rc) {
    const uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= rc.n_pixels) return;
    if (rc.first_slot[lp] >= rc.pass_samples) return;
    const size_t plane = rc.rad_plane;
    const float* rp = rc.rad + lp;
    const uint8_t* bp = rc.slot_bits + lp;
    f3 sum = mk3(rc.film[3 * lp], rc.film[3 * lp + 1], rc.film[3 * lp + 2]);
    for (uint32_t s0 = 0; s0 < rc.pass_samples; s0 += kResolveBatch) {
        uint32_t m = 0u;
        float r[kResolveBatch], g[kResolveBatch], b[kResolveBatch];
#pragma unroll
        for (uint32_t q = 0; q < kResolveBatch / 8u; ++q)
            if (s0 + 8u * q < rc.pass_samples) m |= (uint32_t)bp[(size_t)((s0 >> 3) + q) * rc.n_pixels] << (8u * q);
#pragma unroll
        for (uint32_t k = 0; k < kResolveBatch; ++k) {
            if (s0 + k < rc.pass_samples) {
                const size_t ri = (size_t)(s0 + k) * rc.n_pixels;
                r[k] = rp[ri]; g[k] = rp[plane + ri]; b[k] = rp[2 * plane + ri];
            }
        }
        if (m == 0u) continue;
#pragma unroll
        for (uint32_t k = 0; k < kResolveBatch; ++k)
            if ((m >> k) & 1u) sum = sum + mk3(r[k], g[k], b[k]);
    }
    rc.film[3 * lp] = sum.x; rc.film[3 * lp + 1] = sum.y; rc.film[3 * lp + 2] = sum.z;
}

__global__ void __launch_bounds__(256) k_resolve(RenderCtx rc) {
    const uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= rc.n_pixels) return;
    const uint32_t first = rc.first_slot[lp];
    if (first >= rc.pass_samples) return;
    const size_t plane = rc.rad_plane;
    f3 sum = mk3(rc.film[3 * lp], rc.film[3 * lp + 1], rc.film[3 * lp + 2]);
    // the additions are sequential (sample order = the reference's, film.rs:87), the loads are not: 8 samples
    // (24 loads) in flight per lane, which matters when a narrow shard leaves few pixels to hide latency with
    uint32_t s = first;
    for (; s + 8u <= rc.pass_samples; s += 8u) {
        float r[8], g[8], b[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const size_t ri = (size_t)(s + k) * rc.n_pixels + lp;
            r[k] = rc.rad[ri]; g[k] = rc.rad[plane + ri]; b[k] = rc.rad[2 * plane + ri];
        }
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) sum = sum + mk3(r[k], g[k], b[k]);
    }
    for (; s < rc.pass_samples; ++s) {
        const size_t ri = (size_t)s * rc.n_pixels + lp;
        sum = sum + mk3(rc.rad[ri], rc.rad[plane + ri], rc.rad[2 * plane + ri]);  // film.rs:87
    }
    rc.film[3 * lp] = sum.x; rc.film[3 * lp + 1] = sum.y; rc.film[3 * lp + 2] = sum.z;
}

// film.rs:91: color / weight_sum  (Color / f32 = Color * (1/f32))
__global__ void __launch_bounds__(256) k_finish(RenderCtx rc, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rc.n_pixels * 3u) return;
    out[i] = rc.film[i] * rc.spp_inv;
}

// ---------------------------------------------------------------------------- general box filter
// BoxFilter::weight (src/filter/boxf.rs:27-33) of sample s of `pixel` seen from a pixel (di, dj) away: the film keeps
// (offset - 0.5) per sample (pt.rs:278) and filter_pixel adds the pixel distance (film.rs:84-85).  The offsets are the
// sampler's first draws of the sample's stream, so they are recomputed here instead of being stored.
SPT_DEV float box_weight(const RenderCtx& rc, uint32_t pixel, uint32_t s, int32_t di, int32_t dj, float radius) {
    DRng rng;
    rng.s.state = 0ull;
    if (rc.sampler != SPT_SAMPLER_RECURRENCE) rng.s = spt_rng_seed(rc.seed, pixel, s);
    float ox, oy;
    pixel_offset(rc, pixel, s, rng, &ox, &oy);
    const float wx = (float)di + (ox - 0.5f), wy = (float)dj + (oy - 0.5f);
    return (fabsf(wx) <= radius && fabsf(wy) <= radius) ? 1.0f : 0.0f;
}

// radius_int <= 0 (radius <= 0.5): the colour is the pixel's own in-order sum (rc.film), the weight sum counts the
// samples whose offset lies inside the box (all of them at radius 0.5, which is k_finish).  radius_int < 0
// (radius <= -0.5) leaves both loops of filter_pixel empty: 0 * (1 / 0).
__global__ void __launch_bounds__(256) k_finish_box(RenderCtx rc, float* out, float radius, int32_t R) {
    const uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= rc.n_pixels) return;
    f3 sum = mk3(0, 0, 0);
    float wsum = 0.0f;
    if (R == 0) {
        const uint32_t row_local = lp / rc.width, col = lp - row_local * rc.width;
        const uint32_t pixel = global_row(rc, row_local) * rc.width + col;
        sum = mk3(rc.film[3 * lp], rc.film[3 * lp + 1], rc.film[3 * lp + 2]);
        for (uint32_t s = 0; s < rc.spp; ++s) wsum += box_weight(rc, pixel, s, 0, 0, radius);
    }
    const f3 c = sum * (1.0f / wsum);   // film.rs:91, Color / f32 = Color * (1 / f32) (color.rs:125-131)
    out[3 * lp] = c.x; out[3 * lp + 1] = c.y; out[3 * lp + 2] = c.z;
}

// radius_int >= 1: Film::filter_pixel (film.rs:71-92) over the kept samples of a band of whole rows.  Rows j, then
// columns i, then the samples of that pixel in the order they were added, one running colour sum (the colour is NOT
// weighted - film.rs:87 adds sample.color as is - only weight_sum looks at the offsets).
struct BoxJob {
    const float* rad;               // 3 planes [c][sample][band pixel]
    uint32_t band_base, band_rows;  // image rows held by the planes
    uint32_t out_j0, out_rows;      // image rows to filter
    float* out;                     // first pixel of row out_j0 in the packed output of the shard
    int32_t R;
    float radius;
};
__global__ void __launch_bounds__(256) k_filter_box(RenderCtx rc, BoxJob job) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= job.out_rows * rc.width) return;
    const uint32_t row = idx / rc.width, x = idx - row * rc.width;
    const int32_t y = (int32_t)(job.out_j0 + row);
    const size_t n_band = (size_t)job.band_rows * rc.width, plane = n_band * rc.spp;
    f3 sum = mk3(0, 0, 0);
    float wsum = 0.0f;
    for (int32_t dj = -job.R; dj <= job.R; ++dj) {
        const int32_t jj = y + dj;
        if (jj < 0 || jj >= (int32_t)rc.height) continue;
        for (int32_t di = -job.R; di <= job.R; ++di) {
            const int32_t ii = (int32_t)x + di;
            if (ii < 0 || ii >= (int32_t)rc.width) continue;
            const uint32_t pixel = (uint32_t)jj * rc.width + (uint32_t)ii;
            const size_t lp = (size_t)((uint32_t)jj - job.band_base) * rc.width + (uint32_t)ii;
            for (uint32_t s = 0; s < rc.spp; ++s) {
                const size_t ri = (size_t)s * n_band + lp;
                sum = sum + mk3(job.rad[ri], job.rad[plane + ri], job.rad[2 * plane + ri]);
                wsum += box_weight(rc, pixel, s, di, dj, job.radius);
            }
        }
    }
    const f3 c = sum * (1.0f / wsum);
    job.out[3 * (size_t)idx] = c.x; job.out[3 * (size_t)idx + 1] = c.y; job.out[3 * (size_t)idx + 2] = c.z;
}

// ---------------------------------------------------------------------------- test seams
template <bool kLds>
__global__ void __launch_bounds__(256) k_trace_closest(DScene sc, uint32_t n, const spt_ray* rays, spt_hit* hits) {
    stage_geometry<kLds>(sc);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const spt_ray in = rays[active ? i : 0u];
    DRay r;
    r.o = mk3(in.o); r.d = mk3(in.d); r.t_min = in.t_min;
    DHit h;
    if (kLds && sc.flat) h = flat_closest(sc, r, in.t_max, active);   // (whole waves, see flat.h)
    if (!active) return;
    if (!(kLds && sc.flat)) h = trace_closest<kLds>(sc, r, in.t_max);
    const bool hit = h.inst >= 0;
    hits[i].t = hit ? h.t : SPT_F32_MAX;
    hits[i].instance = h.inst;
    hits[i].prim = hit ? h.prim : -1;
    hits[i].v = hit ? h.v : 0.0f;
    hits[i].w = hit ? h.w : 0.0f;
}
template <bool kLds>
__global__ void __launch_bounds__(256) k_trace_any(DScene sc, uint32_t n, const spt_ray* rays, uint8_t* occluded) {
    stage_geometry<kLds>(sc);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const spt_ray in = rays[active ? i : 0u];
    DRay r;
    r.o = mk3(in.o); r.d = mk3(in.d); r.t_min = in.t_min;
    bool occ = false;
    if (kLds && sc.flat) occ = flat_any(sc, r, in.t_max, active);
    if (!active) return;
    if (!(kLds && sc.flat)) occ = trace_any<kLds>(sc, r, in.t_max);
    occluded[i] = occ ? 1 : 0;
}

// the same seams through the streaming walker (scenes that do not fit LDS): one ray per lane, no refill
__global__ void __launch_bounds__(256) k_trace_closest_stream(DScene sc, uint32_t n, const spt_ray* rays, spt_hit* hits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint2 spill_mem[kSpillStack];
    SWalker<true, false> wk;
    wk.done = true;
    wk.cur = kNoRef;
    if (i < n) {
        DRay r;
        r.o = mk3(rays[i].o); r.d = mk3(rays[i].d); r.t_min = rays[i].t_min;
        wk.begin(sc, r, rays[i].t_max);
    }
    for (uint32_t guard = 0; guard < (1u << 20) && __ballot(!wk.done) != 0ull; ++guard) wk.run(sc, 8u, spill_mem);
    if (i >= n) return;
    const bool hit = wk.h.inst >= 0;
    hits[i].t = hit ? wk.h.t : SPT_F32_MAX;
    hits[i].instance = wk.h.inst;
    hits[i].prim = hit ? wk.h.prim : -1;
    hits[i].v = hit ? wk.h.v : 0.0f;
    hits[i].w = hit ? wk.h.w : 0.0f;
}
__global__ void __launch_bounds__(256) k_trace_any_stream(DScene sc, uint32_t n, const spt_ray* rays, uint8_t* occluded) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint2 spill_mem[kSpillStack];
    SWalker<false, false> wk;
    wk.done = true;
    wk.cur = kNoRef;
    if (i < n) {
        DRay r;
        r.o = mk3(rays[i].o); r.d = mk3(rays[i].d); r.t_min = rays[i].t_min;
        wk.begin(sc, r, rays[i].t_max);
    }
    for (uint32_t guard = 0; guard < (1u << 20) && __ballot(!wk.done) != 0ull; ++guard) wk.run(sc, 8u, spill_mem);
    if (i < n) occluded[i] = wk.h.inst >= 0 ? 1 : 0;
}

__global__ void k_detmath(uint32_t fn, uint32_t n, const float* a, const float* b, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], r;
    switch (fn) {
    case 0: r = spt_sin(x); break;
    case 1: r = spt_cos(x); break;
    case 2: r = spt_log(x); break;
    case 3: r = spt_exp(x); break;
    case 4: r = spt_acos(x); break;
    case 5: r = spt_atan2(x, y); break;
    case 6: r = spt_asin(x); break;
    case 7: r = spt_round(x); break;
    case 8: r = spt_floor(x); break;
    case 9: r = spt_sqrt(x); break;
    case 10: r = x / y; break;
    case 11: r = spt_max(x, y); break;
    case 12: r = spt_min(x, y); break;
    case 13: r = spt_pow(x, y); break;
    case 14: r = spt_log2(x); break;
    case 15: r = spt_trunc(x); break;
    case 16: r = spt_fract(x); break;
    default: r = 0.0f; break;
    }
    out[i] = r;
}

// Test seam behind spt_debug_bxdf: BxdfT::{sample, bxdf, pdf} (src/bxdf/mod.rs:80-90) of one material record, one lane per input.
// op 0: sample(wo, rng stream) -> (wi, f, pdf, transmit); op 1: bxdf(wo, wi), pdf(wo, wi).  kScene: the record may be a
// position-normal-distribution lobe (SPT_BXDF_PNDF_*), whose tables are read from the scene.
template <bool kScene>
__global__ void k_debug_bxdf(DScene sc, DMat m, uint32_t op, uint32_t n, const float* wo_in, const float* wi_in, const uint64_t* rng_state,
                             float* wi_out, float* f_out, float* pdf_out, int32_t* dir_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 wo = mk3(wo_in[3 * i], wo_in[3 * i + 1], wo_in[3 * i + 2]);
    if (op == 0u) {
        DRng rng;
        rng.s.state = rng_state[i];
        DSubsurfaceIo io;
        io.has = false;
        const DBxdfSample s = mat_sample<false, false, false, kScene>(m, wo, rng, &sc, &io);
        wi_out[3 * i] = s.wi.x; wi_out[3 * i + 1] = s.wi.y; wi_out[3 * i + 2] = s.wi.z;
        f_out[3 * i] = s.f.x; f_out[3 * i + 1] = s.f.y; f_out[3 * i + 2] = s.f.z;
        pdf_out[i] = s.pdf;
        dir_out[i] = s.transmit ? 1 : 0;
    } else {
        const f3 wi = mk3(wi_in[3 * i], wi_in[3 * i + 1], wi_in[3 * i + 2]);
        const f3 f = mat_eval<kScene>(m, wo, wi, &sc);
        f_out[3 * i] = f.x; f_out[3 * i + 1] = f.y; f_out[3 * i + 2] = f.z;
        pdf_out[i] = mat_pdf<kScene>(m, wo, wi, &sc);
    }
}
