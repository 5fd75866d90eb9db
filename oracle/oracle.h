/*
 * oracle.h — C interface of the CPU oracle (oracle/liboracle.so).
 * TEST INFRASTRUCTURE ONLY: see the header of oracle.cpp.  It consumes the same
 * flattened scene (include/spt_abi.h) that the HIP library receives.
 */
#ifndef SPT_ORACLE_H
#define SPT_ORACLE_H
#include <stdint.h>

#include "../include/spt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORACLE_SLAB_RECIPROCAL = 1u, /* slab test with a precomputed 1/d, as the kernels do (default: divide, bbox.rs:68-79) */
    ORACLE_BRUTE_FORCE = 2u,     /* ignore TLAS and BLAS: linear scan of instances and triangles */
    ORACLE_LIBM = 4u,            /* libm sin/cos/log/exp/acos/atan2 instead of include/spt_detmath.h */
    ORACLE_TIE_MIN_ID = 8u       /* closest hit = min over (t, instance, prim) lexicographically, boxes culled with
                                    t0 <= t_best: independent of the visit order.  The reference keeps the first
                                    of two equal-t hits in ITS visit order (triangle.rs:187 `t < inter.t`), which is
                                    HashMap / BVH-shape dependent; the kernels visit near children first, so the
                                    GPU parity tests use this rule on both sides. */
};

typedef struct oracle_stats {
    uint64_t samples, segments_closest, segments_shadow;
    uint64_t node_tests, tri_tests, sphere_tests, instance_visits;
    uint32_t threads;
    uint32_t pad;
} oracle_stats;

/* n_threads <= 0: 2 x hardware_concurrency, the reference's layout (src/renderer/pt.rs:243). */
int oracle_render(const spt_scene_desc* desc, const spt_camera* cam, const spt_render_params* params, uint32_t flags,
                  int32_t n_threads, float* rgb_mean_out, oracle_stats* stats);
int oracle_trace_closest(const spt_scene_desc* desc, uint32_t flags, uint32_t n, const spt_ray* rays, spt_hit* hits);
int oracle_trace_any(const spt_scene_desc* desc, uint32_t flags, uint32_t n, const spt_ray* rays, uint8_t* occluded);

void oracle_bxdf_sample(const spt_material* mt, const float wo[3], uint64_t rng_state, uint32_t flags, float wi_out[3],
                        float bxdf_out[3], float* pdf_out, int32_t* dir_out);
void oracle_bxdf_eval(const spt_material* mt, const float wo[3], const float wi[3], float bxdf_out[3], float* pdf_out);
void oracle_bxdf_sample_n(const spt_scene_desc* d /* may be NULL */, const spt_material* mt, uint32_t flags, uint32_t n, const float* wo,
                          const uint64_t* rng_state, float* wi_out, float* bxdf_out, float* pdf_out, int32_t* dir_out);
void oracle_bxdf_eval_n(const spt_scene_desc* d /* may be NULL */, const spt_material* mt, uint32_t n, const float* wo, const float* wi,
                        float* bxdf_out, float* pdf_out);
float oracle_fresnel_dielectric(float ior, const float i[3], const float n[3]);
float oracle_henyey_greenstein(float g, float c);
float oracle_hg_cdf_inverse(float g, float r);
uint32_t oracle_alias_sample(const spt_alias_table* a, float rand, float* prob);
void oracle_env_lookup(const spt_scene_desc* d, const float wi[3], float rgb[3], float* pdf);
void oracle_camera_ray(const spt_camera* cam, float x, float y, float o[3], float dir[3]);
void oracle_detmath(uint32_t fn, uint32_t n, const float* a, const float* b, float* out);
/* image-texture seams (src/texture, src/core/intersection.rs:28-84): texture graph at explicit inputs
 * (18 floats per sample: position, normal, tangent, bitangent, texcoords, duvdx, duvdy -> rgba), and
 * calc_differential (ray + aux ray = 18 floats, hit t/normal/tangent/bitangent = 10 floats) */
int oracle_tex_eval(const spt_scene_desc* desc, uint32_t flags, uint32_t node, uint32_t n, const float* in, float* rgba);
void oracle_calc_differential(const float* ray18, const float* hit10, float duvdx[2], float duvdy[2]);
/* Subsurface substrate seams (src/bxdf/substrate.rs:187-229) */
void oracle_ss_sp(const float d[3], float r, float out[3]);
float oracle_ss_sample_r(float rand);
void oracle_ss_cdf(uint32_t i, float xy[2]);
/* P-NDF seams (src/bxdf/pndf_bvh.rs): PndfUvBvh::find_terms' sum and PndfAccel::calc for n points (u = 2 floats, s = 2 floats each),
 * and PndfMicrofacet::sample_half for n random streams (out: half.xyz, pdf) */
void oracle_pndf_sum(const spt_scene_desc* d, uint32_t pndf, float sigma_p, uint32_t n, const float* u, float* sum_out);
void oracle_pndf_calc(const spt_scene_desc* d, uint32_t pndf, float sigma_p, uint32_t n, const float* u, const float* s, float* out);
void oracle_pndf_sample_half(const spt_scene_desc* d, uint32_t pndf, float sigma_p, const float u[2], uint64_t seed, uint32_t n, float* half_pdf_out);
void oracle_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out);
uint64_t oracle_rng_state(uint64_t seed, uint32_t pixel, uint32_t sample);
void oracle_r2_offsets(uint32_t pixel, uint32_t spp, uint32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif
