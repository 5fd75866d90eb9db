// Owning storage behind a spt_scene_desc (the arrays a Rust `Scene::flatten()`
// would own on the reference side).
#pragma once
#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../../include/spt_host.h"

namespace spt_host {

struct HostError {
    spt_status code;
    std::string msg;
    HostError(spt_status c, std::string m) : code(c), msg(std::move(m)) {}
};

struct HostScene {
    std::vector<spt_bvh_node> tlas_nodes, blas_nodes;
    std::vector<spt_instance> instances;
    std::vector<spt_mesh> meshes;
    std::vector<spt_tri_pos> tri_pos;
    std::vector<spt_tri_attr> tri_attr;
    std::vector<spt_sphere> spheres;
    std::vector<spt_bezier_patch> bezier_patches;
    std::vector<spt_pndf> pndfs;                 // position-normal distributions (pndf.cpp)
    std::vector<spt_pndf_term> pndf_terms;
    std::vector<spt_pndf_node> pndf_nodes;
    std::vector<uint32_t> pndf_refs, pndf_roots;
    std::vector<spt_surface> surfaces;
    std::vector<spt_material> materials;
    std::vector<spt_medium> mediums;
    std::vector<spt_texture> textures;
    std::vector<spt_image> images;
    std::vector<spt_image_level> image_levels;
    std::vector<uint32_t> texels;
    std::vector<spt_material_recipe> material_recipes;
    std::vector<spt_light> lights;
    std::vector<float> light_props, light_u;
    std::vector<uint32_t> light_k;
    std::vector<float> env_texels, env_props, env_u;
    std::vector<uint32_t> env_k;
    uint32_t env_w = 0, env_h = 0;
    float env_scale[3] = {1, 1, 1};
    uint32_t aggregate = SPT_AGGREGATE_BVH, light_sampler = SPT_LIGHT_SAMPLER_UNIFORM;
    int32_t env_light_index = -1;
    std::vector<spt_camera> cameras;
    std::map<std::string, size_t> camera_index;
    spt_scene_desc desc;
    void finalize_desc();
};

HostScene* load_scene_file(const std::string& path);
// pndf.cpp: the Gaussian terms and trees of one pndf_conductor material (PndfConductor::new), appended to the scene
uint32_t build_pndf(HostScene& hs, uint32_t base_normal_texture, float sigma_r, float h, const std::string& label);
// catmull.cpp: 16 control points (x, y, z) per bicubic patch of the Catmull-Clark surface of an ASCII PLY control mesh
std::vector<float> catmull_clark_patches(const std::string& ply_path, uint32_t fas_times);
void read_png_rgba8(const std::string& path, uint32_t* width, uint32_t* height, std::vector<uint32_t>* texels);
void decode_png_rgba8(const std::vector<uint8_t>& bytes, const std::string& label, uint32_t* width, uint32_t* height, std::vector<uint32_t>* texels);   // PNG or JPEG (by content)
void decode_jpeg_rgba8(const std::vector<uint8_t>& bytes, const std::string& label, uint32_t* width, uint32_t* height, std::vector<uint32_t>* texels);
void encode_jpeg_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h, int quality, std::vector<uint8_t>* out);

// exr_piz.cpp: one PIZ-compressed block of an OpenEXR scanline file -> the bytes of the uncompressed block
void exr_piz_decode(const uint8_t* src, size_t size, const std::vector<int>& words_per_pixel, int64_t width, int64_t lines, uint8_t* raw, size_t raw_bytes);

}  // namespace spt_host
