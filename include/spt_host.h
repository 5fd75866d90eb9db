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

const char* spt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SPT_HOST_H */
