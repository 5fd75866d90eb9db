/*
 * spt_abi.h — C ABI of the MI355X path-tracing integrator (libspt_hip.so).
 *
 * This is the drop-in seam for the reference's renderer boundary
 *     pub trait RendererT { fn render(&self, scene: &Scene, config: &OutputConfig); }
 *                                            (reference src/renderer/mod.rs:16-19)
 * selected by the renderer JSON "type" string in create_renderer
 * (src/renderer/mod.rs:26-38) and called once from src/main.rs:61.  The reference
 * has no FFI of its own; a Rust host adds a `Renderer::PathTracerHip` variant that
 * flattens its `Scene` into the POD arrays below and calls these entry points
 * (binding shown in INTEGRATION.md).
 *
 * Conventions
 *  - plain C, no C++/torch types; all pointers in descriptors are HOST pointers
 *    borrowed for the duration of the call (the library copies what it keeps).
 *  - every function returns spt_status (0 = OK); nothing panics/aborts/throws
 *    across the boundary (the reference panics: src/renderer/pt.rs:287,
 *    src/core/scene.rs:29-41); spt_last_error() gives a thread-local message.
 *  - there is NO CPU fallback behind this ABI: without a usable gfx950 device
 *    spt_scene_create fails with SPT_ERR_NO_DEVICE.
 *  - all arithmetic is IEEE f32 (reference: glam Vec3A / Color f32).
 */
#ifndef SPT_ABI_H
#define SPT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPT_ABI_VERSION 13

typedef int32_t spt_status;
enum {
    SPT_OK = 0,
    SPT_ERR_INVALID_ARG = 1,   /* null pointer, bad size, inconsistent descriptor */
    SPT_ERR_NO_DEVICE = 2,     /* no HIP device / not gfx950 / device index out of range */
    SPT_ERR_HIP = 3,           /* a HIP runtime call failed (message has the HIP error string) */
    SPT_ERR_UNSUPPORTED = 4,   /* feature outside the hot-path scope (e.g. an unknown primitive type) */
    SPT_ERR_OUT_OF_MEMORY = 5
};

/* ---- geometry ----------------------------------------------------------------*/

/* One node of a binary BVH, 32 B = two 16-B loads.
 * Replaces the heap BvhNode {lc, rc, bbox, start, end} of src/primitive/bvh.rs:14-20.
 *   inner: a = index of left child, b = index of right child (absolute, same array)
 *   leaf : a = first item,  b = SPT_LEAF_FLAG | item count
 * Items are triangles (BLAS, absolute index into tri_pos) or instances (TLAS). */
#define SPT_LEAF_FLAG 0x80000000u
typedef struct spt_bvh_node {
    float bmin[3];
    uint32_t a;
    float bmax[3];
    uint32_t b;
} spt_bvh_node;

/* Triangle positions in BLAS leaf order (MeshVertex.position of the three
 * corners, src/primitive/triangle.rs:124-127). 48 B = three 16-B loads; the
 * 4th lane of each row is padding (36 algorithmic bytes). */
typedef struct spt_tri_pos {
    float p0[3]; float pad0;
    float p1[3]; float pad1;
    float p2[3]; float pad2;
} spt_tri_pos;

/* Per-corner shading attributes of the same triangle, read once per hit
 * (src/primitive/triangle.rs:188-212): 132 algorithmic bytes, padded to 144. */
typedef struct spt_tri_attr {
    float n[3][3];   /* normals        */
    float t[3][3];   /* tangents       */
    float b[3][3];   /* bitangents     */
    float uv[3][2];  /* texcoords      */
    float pad[3];
} spt_tri_attr;

typedef struct spt_sphere {   /* src/primitive/sphere.rs:8-12 */
    float center[3];
    float radius;
} spt_sphere;

typedef struct spt_mesh {     /* one TriMesh = one BLAS (src/primitive/triangle.rs:19-21) */
    uint32_t root;            /* index of the BLAS root in blas_nodes */
    uint32_t node_count;
    uint32_t tri_first;       /* first triangle (tri_pos / tri_attr index) */
    uint32_t tri_count;
} spt_mesh;

/* One bicubic Bezier patch (src/primitive/bezier.rs:19-22): cp[i][j] = control_points[i][j] (xyz; w unused, except:)
 * point_at(u, v) = sum_ij B_j(u) B_i(v) cp[i][j] (bezier.rs:40-44, 222-236).  The instance's box is the hull's.
 * ABI v12: cp[0][0][3] selects the intersection routine, as the reference's Cargo feature `bezier_ni` does at compile time
 * (Cargo.toml:34-36): 0 = Bezier clipping (bezier.rs:105-134, 239-422, the default build), SPT_BEZIER_NEWTON = Newton's
 * iteration from the middle of the patch inside its bounding box (bezier.rs:58-103). */
#define SPT_BEZIER_NEWTON 1.0f
typedef struct spt_bezier_patch {
    float cp[4][4][4];
} spt_bezier_patch;       /* 256 B */

enum { SPT_PRIM_SPHERE = 0, SPT_PRIM_MESH = 1, SPT_PRIM_BEZIER = 2 };

/* Instance = primitive + transform + surface (src/primitive/instance.rs:10-16).
 * Matrices are stored as glam stores them: three columns then the translation;
 * a point maps to ((c0*x + c1*y) + c2*z) + t, a vector without t
 * (Affine3A::transform_point3a / transform_vector3a). */
typedef struct spt_instance {
    float inv[12];        /* trans_inv : world -> object                        */
    float fwd[12];        /* trans     : object -> world                        */
    float nrm[9];         /* trans_it = transpose(inverse(M).matrix3), 3 columns */
    uint32_t prim_type;   /* SPT_PRIM_*                                          */
    uint32_t prim_id;     /* sphere, mesh or Bezier-patch index                  */
    uint32_t surface;     /* index into surfaces                                 */
    int32_t light;        /* index of this instance's ShapeLight in lights, -1 if not emissive
                             (instance_light_map, src/core/scene_resources.rs:112-120) */
    float bmin[3];        /* world bbox (Instance::bbox)                          */
    float bmax[3];
    float pad[5];
} spt_instance;           /* 48 words = 192 B */

/* ---- appearance ----------------------------------------------------------------*/

/* Materials with scalar textures evaluate to a constant Bxdf variant, so the
 * host resolves MaterialT::bxdf_context (src/material/{lambert,conductor,
 * dielectric,pseudo}.rs) once: alpha = roughness^2, alpha < 1e-4 -> specular. */
enum {
    SPT_BXDF_LAMBERT = 0,              /* src/bxdf/lambert.rs               */
    SPT_BXDF_MICROFACET_CONDUCTOR = 1, /* src/bxdf/microfacet_conductor.rs  */
    SPT_BXDF_SPECULAR_CONDUCTOR = 2,   /* src/bxdf/specular_conductor.rs    */
    SPT_BXDF_MICROFACET_DIELECTRIC = 3,/* src/bxdf/microfacet_dielectric.rs */
    SPT_BXDF_SPECULAR_DIELECTRIC = 4,  /* src/bxdf/specular_dielectric.rs   */
    SPT_BXDF_PSEUDO = 5,               /* src/bxdf/pseudo.rs                */
    SPT_BXDF_MICROFACET_PLASTIC = 6,   /* src/bxdf/microfacet_plastic.rs    */
    SPT_BXDF_SPECULAR_PLASTIC = 7,     /* src/bxdf/specular_plastic.rs      */
    SPT_BXDF_PNDF_CONDUCTOR = 8,       /* MicrofacetConductor over a PndfMicrofacet (src/bxdf/microfacet.rs:56-170); only ever
                                          the result of a per-hit recipe (SPT_MAT_PNDF_CONDUCTOR), never a constant material */
    SPT_BXDF_PNDF_PLASTIC = 9          /* MicrofacetPlastic over a PndfMicrofacet (SPT_MAT_PNDF_PLASTIC), per hit only too */
};
/* plastic lobes = Fresnel-weighted specular coat over a substrate (materials plastic, pbr_metallic,
 * pbr_specular; src/material/{plastic,pbr_metallic,pbr_specular}.rs) */
enum { SPT_FRESNEL_DIELECTRIC = 0, SPT_FRESNEL_SCHLICK = 1 };   /* src/bxdf/fresnel.rs:19-59 */
enum { SPT_SUBSTRATE_LAMBERT = 0, SPT_SUBSTRATE_DIFFUSE = 1,    /* src/bxdf/substrate.rs:22-45,120-180 */
       SPT_SUBSTRATE_SUBSURFACE = 2 };                           /* substrate.rs:182-350: Diffuse + a BSSRDF probe ray */
typedef struct spt_material {
    uint32_t bxdf;
    float c0[3];     /* lambert: reflectance; conductor: ior (eta); plastic: substrate reflectance */
    float c1[3];     /* conductor: ior_k; plastic with Schlick Fresnel: r0; Subsurface substrate: d  */
    float ax, ay;    /* GGX roughness_x / roughness_y (as the material hands them to GgxMicrofacet) */
    float ior;       /* dielectric / plastic: int_ior / ext_ior                                     */
    float c2[3];     /* Diffuse substrate: bxdf_wo_fresnel (Diffuse::new, substrate.rs:127-137)     */
    uint32_t fresnel;    /* SPT_FRESNEL_*   (plastic lobes; conductors: SCHLICK = SchlickFresnel with r0 = c0, else ConductorFresnel) */
    uint32_t substrate;  /* SPT_SUBSTRATE_* (plastic lobes) */
    uint32_t recipe;     /* 0: the constants above are the Bxdf; k > 0: material_recipes[k - 1] is
                            evaluated at every hit (some parameter is an image texture) and the
                            constants only hold the values at the textures' average colours */
} spt_material;      /* 16 words = 64 B */

/* ---- textures (the files of src/texture) -------------------------------------------------
 * The closed `Texture` enum (src/texture/mod.rs:187-197) as a node table.  A named texture of
 * the scene file is TexInputModifier(SrgbTex(base)) with the two wrappers present only when
 * asked for (create_texture_from_params, mod.rs:210-243); binary ops point at named textures. */
enum {
    SPT_TEX_SCALAR = 0,    /* scalar.rs: value[3], alpha 1                              */
    SPT_TEX_IMAGE = 1,     /* image_tex.rs: mip pyramid `image`, trilinear              */
    SPT_TEX_ADD = 2, SPT_TEX_SUB = 3, SPT_TEX_MUL = 4, SPT_TEX_DIV = 5,   /* binary_op.rs: a (op) b */
    SPT_TEX_SRGB = 6,      /* srgb_tex.rs: sRGB -> linear on r,g,b of child a            */
    SPT_TEX_MODIFIER = 7   /* input_modifier.rs: input * tiling + offset, mode / wrap override, child a */
};
enum { SPT_TEXMODE_SPECIFIED = 0, SPT_TEXMODE_TEXCOORDS = 1, SPT_TEXMODE_POSITION = 2, SPT_TEXMODE_NORMAL = 3,
       SPT_TEXMODE_TANGENT = 4, SPT_TEXMODE_BITANGENT = 5 };                /* mod.rs:20-28 */
enum { SPT_TEXWRAP_REPEAT = 0, SPT_TEXWRAP_MIRROR_REPEAT = 1, SPT_TEXWRAP_CLAMP = 2, SPT_TEXWRAP_MIRROR_CLAMP = 3 };  /* mod.rs:36-42 */
enum { SPT_CHAN_R = 0, SPT_CHAN_G = 1, SPT_CHAN_B = 2, SPT_CHAN_A = 3 };
typedef struct spt_texture {
    uint32_t type;        /* SPT_TEX_*                                              */
    uint32_t a, b;        /* child texture indices (always smaller than this node's) */
    uint32_t image;       /* IMAGE: index into images                                */
    float value[3];       /* SCALAR                                                  */
    int32_t mode, wrap;   /* MODIFIER: SPT_TEXMODE_* / SPT_TEXWRAP_*, -1 = keep the incoming one */
    float tiling[3];      /* MODIFIER                                                */
    float offset[3];
    uint32_t pad;
} spt_texture;            /* 16 words = 64 B */

/* ImageTex::images (image_tex.rs:7-9): level 0 is the file as RGBA8 (what DynamicImage::get_pixel
 * returns), the rest is generate_mipmap's box pyramid down to 1x1.  Texels are r | g<<8 | b<<16 | a<<24. */
typedef struct spt_image { uint32_t first_level, n_levels; } spt_image;
typedef struct spt_image_level { uint32_t width, height, first_texel, pad; } spt_image_level;

/* A material whose parameters are not all constant: MaterialT::bxdf_context
 * (src/material/{lambert,conductor,dielectric,plastic,pbr_metallic,pbr_specular}.rs) restated as data. */
enum { SPT_MAT_LAMBERT = 0, SPT_MAT_CONDUCTOR = 1, SPT_MAT_DIELECTRIC = 2, SPT_MAT_PLASTIC = 3,
       SPT_MAT_PBR_METALLIC = 4, SPT_MAT_PBR_SPECULAR = 5, SPT_MAT_SUBSURFACE = 6,
       SPT_MAT_PNDF_CONDUCTOR = 7,     /* pndf_conductor.rs:156-196: tex[0] albedo, tex[1] = index into pndfs (not a texture),
                                          tex[2] fallback_roughness (read when the pixel footprint sigma_p is 0) */
       SPT_MAT_PNDF_PLASTIC = 8 };     /* pndf_plastic.rs:163-211: the same slots + ior; DielectricFresnel, Diffuse substrate */
typedef struct spt_material_recipe {
    uint32_t type;         /* SPT_MAT_* */
    uint32_t tex[4];       /* texture indices: [0] albedo | ior | base_color | diffuse, [1] ior_k | metallic | specular | ld,
                              [2] roughness_x, [3] roughness_y (unused slots: 0) */
    uint32_t rough_chan;   /* SPT_CHAN_* read from tex[2], tex[3] (the JSON loader always says R) */
    uint32_t metal_chan;   /* SPT_CHAN_* read from tex[1] of PBR_METALLIC                      */
    float ior;             /* DIELECTRIC / PLASTIC: int_ior / ext_ior                          */
} spt_material_recipe;     /* 8 words */

/* ---- position-normal distributions ("glints", src/bxdf/pndf_bvh.rs, src/material/pndf_conductor.rs) -----------------
 * One Gaussian term per cell of the material's normal map (PndfGaussTerm, pndf_bvh.rs:4-11, 405-437). */
typedef struct spt_pndf_term {
    float u[2];        /* cell centre in texture space                                 */
    float s[2];        /* (x, y) of the normal there                                     */
    float jacobian[4]; /* ds/du, columns (dsdu, dsdv) (glam Mat2: x_axis, y_axis)      */
    float mat_a[4], mat_s[4], mat_mu[4];   /* PndfGaussTerm::new, column-major like glam */
} spt_pndf_term;       /* 20 words = 80 B */
/* A node of PndfBvh (4-D boxes over (u, s), pndf_bvh.rs:19-25, 124-190) or of PndfUvBvh (2-D boxes over u, the last two
 * box coordinates are 0; pndf_bvh.rs:35-41, 266-333).  Both split their index range in the middle, without sorting. */
typedef struct spt_pndf_node {
    float bmin[4], bmax[4];
    uint32_t start, end;   /* range in pndf_refs, relative to the owning tree's first ref */
    uint32_t lc, rc;       /* children (absolute node indices), 0xffffffff in a leaf       */
} spt_pndf_node;       /* 12 words = 48 B */
typedef struct spt_pndf {  /* PndfConductor + PndfAccel (pndf_conductor.rs:16-28, pndf_bvh.rs:49-92) */
    uint32_t first_term, n_terms;       /* pndf_terms */
    uint32_t s_block_count;             /* the s-plane [-1, 1]^2 is cut into s_block_count^2 blocks, one PndfBvh each */
    uint32_t first_root;                /* pndf_roots[first_root + x * s_block_count + y] = root node of that block or 0xffffffff,
                                           followed by the first ref of the block's term list */
    uint32_t uv_root, uv_first_ref;     /* PndfUvBvh over all terms */
    float sigma_r, sigma_hx, sigma_hy;
    float tiling[2], offset[2];         /* base_normal.tiling() / offset() (texcoords -> u) */
    uint32_t pad[3];
} spt_pndf;            /* 16 words = 64 B */

enum { SPT_SURF_DOUBLE_SIDED = 1u };
typedef struct spt_surface {  /* src/core/surface.rs:14-22 */
    uint32_t material;
    uint32_t flags;
    int32_t inside_medium;    /* medium index or -1 */
    float emissive[3];
    uint32_t normal_map;      /* texture index + 1, 0 = none (Surface::coord, surface.rs:65-78)      */
    uint32_t emissive_map;    /* texture index + 1, 0 = none (Surface::emissive, surface.rs:49-55)   */
} spt_surface;                /* 8 words */

typedef struct spt_medium {   /* src/medium/homogeneous.rs:11-15 */
    float sigma_t[3];
    float sigma_s[3];
    float g;
    float pad;
} spt_medium;

enum {
    SPT_LIGHT_DIRECTIONAL = 0, /* src/light/directional.rs (direction stored normalised) */
    SPT_LIGHT_POINT = 1,       /* src/light/point.rs       */
    SPT_LIGHT_SPOT = 2,        /* src/light/spot.rs        */
    SPT_LIGHT_SHAPE = 3,       /* src/light/shape_light.rs */
    SPT_LIGHT_ENV = 4          /* src/light/environment.rs */
};
typedef struct spt_light {
    uint32_t type;
    float pos[3];        /* point/spot position                        */
    float dir[3];        /* directional/spot direction                 */
    float strength[3];
    float cos_inner, cos_outer;
    uint32_t instance;   /* SHAPE: instance index                      */
    float power;         /* LightT::power() (alias-table input)        */
    float pad[2];
} spt_light;             /* 16 words */

enum { SPT_LIGHT_SAMPLER_UNIFORM = 0, SPT_LIGHT_SAMPLER_POWER_IS = 1 };

/* AliasTable (src/core/alias_table.rs:1-5): props / u / k, all length n. */
typedef struct spt_alias_table {
    uint32_t n;
    const float* props;
    const float* u;
    const uint32_t* k;
} spt_alias_table;

typedef struct spt_env {      /* EnvLight (src/light/environment.rs:10-17) */
    uint32_t width, height;   /* 0,0 = no environment */
    const float* texels;      /* height*width RGB f32, row 0 = theta 0 */
    float scale[3];
    spt_alias_table alias;    /* n = width*height */
} spt_env;

enum { SPT_AGGREGATE_GROUP = 0, SPT_AGGREGATE_BVH = 1 };

/* The flattened Scene (src/core/scene.rs:9-14 + everything it points to). */
typedef struct spt_scene_desc {
    uint32_t abi_version;          /* SPT_ABI_VERSION */
    uint32_t aggregate;            /* SPT_AGGREGATE_*: GROUP skips the top-level box test
                                      (src/primitive/group.rs:34-40) */
    uint32_t n_tlas_nodes;  const spt_bvh_node* tlas_nodes;  /* leaves index instances */
    uint32_t n_instances;   const spt_instance* instances;   /* in TLAS leaf order     */
    uint32_t n_meshes;      const spt_mesh* meshes;
    uint32_t n_blas_nodes;  const spt_bvh_node* blas_nodes;
    uint32_t n_tris;        const spt_tri_pos* tri_pos;  const spt_tri_attr* tri_attr;
    uint32_t n_spheres;     const spt_sphere* spheres;
    uint32_t n_surfaces;    const spt_surface* surfaces;
    uint32_t n_materials;   const spt_material* materials;
    uint32_t n_mediums;     const spt_medium* mediums;
    uint32_t n_lights;      const spt_light* lights;
    uint32_t light_sampler;        /* SPT_LIGHT_SAMPLER_* */
    int32_t env_light_index;       /* index of the ENV light in lights, -1 if none */
    spt_alias_table light_alias;   /* POWER_IS only (n = n_lights) */
    spt_env env;
    /* image textures (all zero / null for a scene with constant materials) */
    uint32_t n_textures;          const spt_texture* textures;
    uint32_t n_images;            const spt_image* images;
    uint32_t n_image_levels;      const spt_image_level* image_levels;
    uint32_t n_texels;            const uint32_t* texels;
    uint32_t n_material_recipes;  const spt_material_recipe* material_recipes;
    /* bicubic Bezier patches (instances with prim_type SPT_PRIM_BEZIER) */
    uint32_t n_bezier_patches;    const spt_bezier_patch* bezier_patches;
    /* position-normal distributions (materials with a SPT_MAT_PNDF_CONDUCTOR / _PLASTIC recipe; ABI v10) */
    uint32_t n_pndfs;             const spt_pndf* pndfs;
    uint32_t n_pndf_terms;        const spt_pndf_term* pndf_terms;
    uint32_t n_pndf_nodes;        const spt_pndf_node* pndf_nodes;
    uint32_t n_pndf_refs;         const uint32_t* pndf_refs;     /* term indices (absolute), the trees' term lists back to back */
    uint32_t n_pndf_roots;        const uint32_t* pndf_roots;    /* pairs (root node, first ref) per s-block                    */
} spt_scene_desc;

/* PerspectiveCamera after ::new (src/camera/perspective.rs:15-27). */
typedef struct spt_camera {
    float eye[3];
    float forward[3];   /* normalised                  */
    float up[3];        /* right x forward             */
    float right[3];     /* normalize(forward x up_in)  */
    float half_cot_half_fov;
} spt_camera;

enum { SPT_SAMPLER_RANDOM = 0, SPT_SAMPLER_JITTERED = 1, SPT_SAMPLER_RECURRENCE = 2 };

/* PathTracer{max_depth, pixel_sampler, filter} + OutputConfig{width,height}
 * (src/renderer/pt.rs:24-28, src/renderer/mod.rs:9-14) + the shard of the image
 * this call renders. */
typedef struct spt_render_params {
    uint32_t width, height;        /* full image */
    uint32_t spp;                  /* samples per pixel (jittered: division_x*division_y) */
    uint32_t max_depth;
    uint32_t sampler;              /* SPT_SAMPLER_* */
    uint32_t division_x, division_y; /* jittered only */
    uint64_t seed;
    /* row-strip sharding: this call renders rows j with (j / strip_rows) % shard_count
     * == shard_index; output rows are packed in increasing j.  shard_count=1 -> all. */
    uint32_t shard_index, shard_count, strip_rows;
    uint32_t samples_per_pass;     /* tuning: spp rendered per wavefront pass (0 = default) */
    uint32_t flags;                /* SPT_RENDER_* */
    uint64_t out_strip_stride;     /* bytes between the starts of consecutive strips of THIS shard in rgb_mean_out;
                                      0 = packed (strip_rows * width * 12).  shard_count * strip_rows * width * 12 with
                                      rgb_mean_out pointing at the shard's first row inside a full-image film makes every
                                      rank write its rows in place (one strided DMA, no host-side scatter) */
    float filter_radius;           /* BoxFilter::radius (src/filter/boxf.rs:5-14); read only with SPT_RENDER_BOX_RADIUS,
                                      otherwise 0.5 (every sample of a pixel and no other) */
    uint32_t stats_size;           /* sizeof(spt_render_stats) AS THE CALLER WAS COMPILED (ABI v9).  spt_render writes at most
                                      this many bytes of `stats`, so a caller built against an older, shorter struct is never
                                      written past its end (the struct only ever grows at the tail).  Must be set when `stats`
                                      is not NULL: 0 with a non-NULL `stats` is SPT_ERR_INVALID_ARG */
} spt_render_params;
enum {
    SPT_RENDER_PROFILE = 1u,       /* time each kernel class with HIP events */
    SPT_RENDER_BOX_RADIUS = 2u,    /* filter_radius is set */
    SPT_RENDER_COUNT_VISITS = 4u,  /* count BVH node / triangle / instance visits on the device (stats->*_visits); the counting
                                      kernels are separate instantiations, slower by a few per cent: measurement runs only */
    SPT_RENDER_ASYNC = 8u          /* ABI v11: return as soon as the work is queued.  The film's device-to-host copy runs on a
                                      copy stream of its own, so it overlaps the kernels of the NEXT spt_render on this scene
                                      (a caller that renders frame after frame pays max(kernels, copy) per frame instead of the
                                      sum).  rgb_mean_out (page-locked, or the copy is not asynchronous) is valid after
                                      spt_render_wait or after a later synchronous spt_render on the scene returns; `stats`
                                      must be NULL (counters would need the device to be idle) */
    ,
    SPT_RENDER_DEBUG_NORMAL = 16u  /* ABI v13: the reference's cargo feature `debug_normal` (Cargo.toml:34-36, src/renderer/pt.rs:113-118):
                                      a path's colour is `normal * 0.5 + 0.5` of the first surface it reaches (the world-space
                                      interpolated normal Instance::intersect leaves in the Intersection), nothing is shaded */
};

#define SPT_N_KERNELS 7
/* SHADE_FIRST: the shade launches of bounce 0 (one per pass, nearly all path vertices); SHADE: bounces >= 1 */
enum { SPT_K_PRIMARY = 0, SPT_K_SHADE = 1, SPT_K_SHADOW = 2, SPT_K_EXTEND = 3, SPT_K_RESOLVE = 4, SPT_K_OTHER = 5, SPT_K_SHADE_FIRST = 6 };
typedef struct spt_render_stats {
    uint64_t samples;              /* camera samples traced = rows*width*spp               */
    uint64_t segments_closest;     /* closest-hit ray segments (primary + extension)       */
    uint64_t segments_shadow;      /* any-hit ray segments                                 */
    double gpu_ms;                 /* HIP-event time of the whole call on the render stream */
    double kernel_ms[SPT_N_KERNELS];      /* per kernel class (SPT_RENDER_PROFILE only)     */
    uint32_t kernel_launches[SPT_N_KERNELS];
    uint64_t primary_hits;         /* camera samples whose primary ray hit something           */
    uint64_t path_vertices;        /* records consumed by the shade stage over all bounces     */
    uint64_t shadow_first;         /* any-hit segments issued by the bounce-0 shade launches   */
    uint64_t vertices_second;      /* path vertices of bounce 1 (= extension rays of bounce 0 that were kept) */
    uint64_t live_samples;         /* chunked k_primary: camera samples of pixels inside the screen-space bound, each of
                                      which owns a radiance slot; 0 when the un-chunked kernel ran */
    /* ABI v9, SPT_RENDER_COUNT_VISITS only (0 otherwise): what the traversal kernels fetched, summed over all ray segments */
    uint64_t node_visits;          /* BVH node records fetched (TLAS + BLAS; one record = one multi-child node)      */
    uint64_t tri_tests;            /* triangle records fetched and tested                                            */
    uint64_t instance_visits;      /* instance records fetched (ray transformed into object space)                   */
    uint64_t node_bytes;           /* bytes of the node records above (record sizes differ between the node formats)  */
    uint64_t class_visits[3][3];   /* the same three counters per kernel class: [primary, shadow, extend][node, triangle, instance] */
} spt_render_stats;

/* Closest-hit record: what BvhAccel/Group::intersect leave in `Intersection`
 * (src/core/intersection.rs:6-18) before shading: t, who was hit, barycentrics. */
typedef struct spt_hit {
    float t;              /* f32::MAX on miss */
    int32_t instance;     /* -1 on miss */
    int32_t prim;         /* triangle index (absolute) or sphere index */
    float v, w;           /* triangle barycentrics of p1, p2 (u = 1 - v - w) */
} spt_hit;

/* One ray: origin, t_min, direction, t_max (32 B). */
typedef struct spt_ray {
    float o[3]; float t_min;
    float d[3]; float t_max;
} spt_ray;

typedef struct spt_scene spt_scene;   /* opaque: device-resident copy of a spt_scene_desc */

spt_status spt_device_count(int32_t* count);
/* Validates desc, copies it to HBM on `device` (SoA-repacked), owns the copy. */
spt_status spt_scene_create(const spt_scene_desc* desc, int32_t device, spt_scene** out);
void spt_scene_destroy(spt_scene* scene);

/* The hot path: RendererT::render for one image shard.  rgb_mean_out receives
 * shard_rows*width*3 f32 = per-pixel mean radiance (Film::filter_pixel with the
 * box filter, src/core/film.rs:71-92: the sum of the samples of the (2 ceil(radius - 0.5) + 1)^2 pixels around it
 * over the number of those samples whose offset lies within `radius`), row 0 = top.  Synchronous. */
spt_status spt_render(const spt_scene* scene, const spt_camera* cam, const spt_render_params* params,
                      float* rgb_mean_out, spt_render_stats* stats /* may be NULL */);
/* Blocks until every spt_render queued on the scene with SPT_RENDER_ASYNC has delivered its film (ABI v11). */
spt_status spt_render_wait(const spt_scene* scene);
/* Number of image rows spt_render writes for these params. */
spt_status spt_shard_rows(const spt_render_params* params, uint32_t* rows);

/* Seams below the renderer, for parity tests of rows a4/a6/a8/a9/a10:
 * Primitive::intersect / intersect_test of the scene aggregate on caller rays. */
spt_status spt_trace_closest(const spt_scene* scene, uint32_t n, const spt_ray* rays, spt_hit* hits);
spt_status spt_trace_any(const spt_scene* scene, uint32_t n, const spt_ray* rays, uint8_t* occluded);

/* Page-locked host memory for rgb_mean_out: lets the final device-to-host copy of the film run as one
 * DMA at PCIe rate instead of being staged through the runtime's bounce buffers.  Optional: any host
 * pointer is accepted by spt_render. */
spt_status spt_alloc_pinned(uint64_t bytes, void** out);
void spt_free_pinned(void* p);
/* page-lock / unlock caller memory (e.g. a shared-memory film mapped by every rank) so that spt_render's copy-out
 * into it runs at DMA speed */
spt_status spt_pin_host(void* p, uint64_t bytes);
void spt_unpin_host(void* p);

/* Test seam: evaluates one function of include/spt_detmath.h on the device (fn: 0 sin, 1 cos,
 * 2 log, 3 exp, 4 acos, 5 atan2(a,b), 6 asin, 7 round, 8 floor, 9 sqrt, 10 a/b, 11 max(a,b),
 * 12 min(a,b)), so the tests can check gfx950 returns the same bits as x86-64. */
spt_status spt_debug_detmath(int32_t device, uint32_t fn, uint32_t n, const float* a, const float* b, float* out);

/* Test seam (ABI v13) for rows a13-a16: BxdfT::{sample, bxdf, pdf} (src/bxdf/mod.rs:80-90) of ONE constant material record
 * (`recipe` 0), evaluated on the device for n inputs in the local shading frame (z = normal).
 *   op 0  sample: wo[3n], rng_state[n] (the PCG32 state a stream starts from, include/spt_detmath.h) ->
 *         wi_out[3n], f_out[3n], pdf_out[n], dir_out[n] (0 reflect, 1 transmit: BxdfDirType, mod.rs:30-36)
 *   op 1  bxdf + pdf: wo[3n], wi_in[3n] -> f_out[3n], pdf_out[n]
 * `scene` may be NULL (then `device` says where to run) unless the record is a position-normal-distribution lobe
 * (SPT_BXDF_PNDF_*: c1 = (1 / normalisation, sigma_p, P-NDF index as bits), ax / ay = u), whose tables live in a scene.
 * A Subsurface substrate is refused (SPT_ERR_UNSUPPORTED): its sample places the exit point with a traced probe ray. */
spt_status spt_debug_bxdf(const spt_scene* scene, int32_t device, const spt_material* mt, uint32_t op, uint32_t n, const float* wo,
                          const float* wi_in, const uint64_t* rng_state, float* wi_out, float* f_out, float* pdf_out, int32_t* dir_out);

const char* spt_last_error(void);
uint32_t spt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPT_ABI_H */
