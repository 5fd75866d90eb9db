/*
 * spt_host.h — C ABI of the host side (libspt_host.so): the stand-in for the
 * reference's Rust host that the north star keeps (scene JSON loader, camera,
 * surface/material plumbing, image write-out).  No Rust toolchain exists in the
 * build image, so this side is C++ behind a C ABI; it produces exactly the POD
 * `spt_scene_desc` a Rust `Scene::flatten()` would hand to libspt_hip.so.
 *
 * Reference counterparts:
 *   spt_host_load_scene     loader::load_scene      src/loader/json.rs:53-199
 *                           SceneResources::to_scene src/core/scene_resources.rs:29-138
 *   spt_host_load_renderer  loader::load_renderer   src/loader/json.rs:19-51
 *   spt_host_scene_camera   Scene::get_camera       src/core/scene.rs:29-41
 *   spt_host_film_to_rgb8   color_to_rgb            src/core/film.rs:94-99
 *   spt_host_write_png      image.save              src/renderer/pt.rs:290-294
 * Errors are returned as status codes + spt_host_last_error() where the reference
 * returns anyhow::Error (load) or panics (get_camera).
 */
#ifndef SPT_HOST_H
#define SPT_HOST_H

#include "spt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spt_host_scene spt_host_scene; /* owns every array the desc points to */

enum {
    SPT_HOST_ERR_IO = 100,       /* file missing / unreadable */
    SPT_HOST_ERR_PARSE = 101,    /* JSON / OBJ / EXR syntax */
    SPT_HOST_ERR_SCHEMA = 102,   /* missing or mistyped field, unknown name/type (anyhow::bail! sites) */
    SPT_HOST_ERR_UNSUPPORTED = 103 /* valid reference feature outside the hot-path scope */
};

spt_status spt_host_load_scene(const char* scene_json_path, spt_host_scene** out);
const spt_scene_desc* spt_host_scene_desc(const spt_host_scene* scene);
/* name == NULL: the scene must have exactly one camera (reference panics otherwise). */
spt_status spt_host_scene_camera(const spt_host_scene* scene, const char* name, spt_camera* out);
void spt_host_scene_free(spt_host_scene* scene);
/* Which routine intersects the scene's Bezier patches (Catmull-Clark surfaces included): 0 = Bezier clipping, the reference's
 * default build; 1 = Newton's iteration, the reference built with `--features bezier_ni` (Cargo.toml:34-36, bezier.rs:58-103).
 * Sets spt_bezier_patch::cp[0][0][3] of every patch of THIS scene (call it before spt_scene_create); without the call the
 * process-wide default applies (clipping, or Newton when the environment variable SPT_BEZIER_NI is set to a non-zero value). */
spt_status spt_host_scene_set_bezier_newton(spt_host_scene* scene, int32_t newton);

/* Fills max_depth, spp, sampler, division_x/y of *params (other fields untouched)
 * and the box-filter radius. */
spt_status spt_host_load_renderer(const char* renderer_json_path, spt_render_params* params,
                                  float* filter_radius);

/* u8 = (clamp(c*255, 0, 255)) as u8 — truncation, no gamma (src/core/film.rs:94-99). */
void spt_host_film_to_rgb8(const float* rgb_mean, uint64_t n_pixels, uint8_t* rgb8_out);
spt_status spt_host_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);
/* baseline JPEG, 4:4:4, IJG quality scale (the image crate's JpegEncoder default is 75) */
spt_status spt_host_write_jpeg(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height, int32_t quality);
/* `image.save(path)` (src/renderer/pt.rs:292-294): the extension picks the format - png, jpg / jpeg */
spt_status spt_host_write_image(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* OpenEXR scanline I/O (RGB f32 / f16; reads NO_COMPRESSION / RLE / ZIPS / ZIP / PIZ / PXR24, writes uncompressed) for
 * `environment {type: "exr"}` (get_exr_image, src/core/loader.rs:374-390). */
spt_status spt_host_read_exr(const char* path, uint32_t* width, uint32_t* height, float** rgb_out);
spt_status spt_host_write_exr(const char* path, const float* rgb, uint32_t width, uint32_t height);
/* PNG -> RGBA8 texels (r | g<<8 | b<<16 | a<<24) the way `image::open` + `get_pixel` present an
 * `image_file` texture (get_image, src/core/loader.rs:366-371); free with spt_host_free. */
spt_status spt_host_read_png(const char* path, uint32_t* width, uint32_t* height, uint32_t** rgba8_out);
/* the same for a PNG or a JPEG file (told apart by content; baseline / progressive Huffman, IJG arithmetic) */
spt_status spt_host_read_image(const char* path, uint32_t* width, uint32_t* height, uint32_t** rgba8_out);
/* The Catmull-Clark front end on its own (CatmullClark::load, src/primitive/catmull.rs:93-101): the bicubic patches of an
 * ASCII PLY control mesh after `fas_times` rounds of feature-adaptive subdivision, 16 control points (x, y, z) per patch;
 * free with spt_host_free.  (A scene's `catmull_clark` primitive goes through the same code.) */
spt_status spt_host_catmull_clark(const char* ply_path, uint32_t fas_times, uint32_t* n_patches, float** control_points_out);
void spt_host_free(void* p);

/* ---- one call, N devices, one film -------------------------------------------------------------------------------------
 * The reference's `render` fans the image rows out over all of its worker threads and assembles ONE film
 * (PathTracer::render, src/renderer/pt.rs:243-287; UnsafeFilm, src/core/film.rs:101-116).  The counterpart here: one
 * worker thread per device, each with a full scene replica (spt_scene_create on its device), device k rendering the
 * interleaved row strips (j / strip_rows) % n == k of the image straight into the caller's film (out_strip_stride: one
 * strided DMA per device, no host-side scatter, no collective).  Seeds depend on (pixel, sample) only, so the film is
 * bit-identical for every n.  The device entry points arrive as a table, so that the fan-out (this library has no HIP in
 * it) can be driven with libspt_hip.so's functions - what `spt --gpus N` and the Python binding do - or with stand-ins. */
typedef struct spt_device_api {
    spt_status (*scene_create)(const spt_scene_desc* desc, int32_t device, spt_scene** out);
    void (*scene_destroy)(spt_scene* scene);
    spt_status (*render)(const spt_scene* scene, const spt_camera* cam, const spt_render_params* params, float* rgb_mean_out,
                         spt_render_stats* stats);
    const char* (*last_error)(void);
    /* optional (NULL: the film is handed to `render` as it is): page-lock / unlock the caller's film once, so that every
     * device's copy-out is a DMA */
    spt_status (*pin_host)(void* p, uint64_t bytes);
    void (*unpin_host)(void* p);
} spt_device_api;

typedef struct spt_host_multi spt_host_multi;
/* Creates one scene replica per entry of `devices` (concurrently, one thread each; an index may repeat: two workers then
 * share that device).  Fails as a whole if any replica fails. */
spt_status spt_host_multi_create(const spt_scene_desc* desc, const spt_device_api* api, uint32_t n_devices, const int32_t* devices,
                                 spt_host_multi** out);
/* Renders the full image of `params` (its shard fields are overwritten: shard k of n_devices, `strip_rows` rows per strip, 0 = a
 * default that keeps the shares even) into film[height][width][3].  stats: NULL or n_devices entries (one per device; each
 * is written with params->stats_size bytes as spt_render does).  Synchronous: the film is complete on return. */
spt_status spt_host_multi_render(spt_host_multi* m, const spt_camera* cam, const spt_render_params* params, uint32_t strip_rows,
                                 float* film, spt_render_stats* stats);
uint32_t spt_host_multi_device_count(const spt_host_multi* m);
void spt_host_multi_destroy(spt_host_multi* m);

const char* spt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SPT_HOST_H */
