/*
 * ray_capi.h -- a plain-C view of sergcpp/Ray's public API (Ray::RendererBase / Ray::SceneBase /
 * Ray::RegionContext), so that the Python host mirror in ray_amd/api.py -- and the tests -- can drive ANY
 * backend of the library through ctypes: the reference's CPU backends and the new HIP backend alike.
 *
 * It adds no behaviour: every function forwards to the virtual of the same name (cited per function).
 * The descriptor structs are C restatements of the reference's *_desc_t (SceneBase.h:46-311) with
 * handles flattened to 64-bit integers: (_block << 32) | _index.
 *
 * Two libraries are built from this file:
 *   oracle/_ref/libray_ref.so          reference CPU backends only               (test oracle / CPU baseline)
 *   ray_amd/host/_build/libray_hip.so  reference host-side scene code + RendererHIP/SceneHIP (the drop-in)
 */
#ifndef RAY_CAPI_H
#define RAY_CAPI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ray_renderer ray_renderer;
typedef struct ray_scene ray_scene;
typedef struct ray_region ray_region;
typedef uint64_t ray_handle;
#define RAY_INVALID_HANDLE 0xffffffffull /* Invalid*Handle = {0xffffffff, 0} */

typedef struct ray_shading_node_desc { /* Ray::shading_node_desc_t, SceneBase.h:48-68 */
    uint32_t type;                     /* Ray::eShadingNode */
    float base_color[3];
    ray_handle base_texture;
    ray_handle normal_map;
    float normal_map_intensity;
    ray_handle mix_materials[2];
    float roughness;
    ray_handle roughness_texture;
    float anisotropic;
    float anisotropic_rotation;
    float sheen;
    float specular;
    float strength;
    float fresnel;
    float ior;
    float tint;
    ray_handle metallic_texture;
    int32_t importance_sample;
    int32_t mix_add;
} ray_shading_node_desc;

typedef struct ray_principled_mat_desc { /* Ray::principled_mat_desc_t, SceneBase.h:70-97 */
    float base_color[3];
    ray_handle base_texture;
    float metallic;
    ray_handle metallic_texture;
    float specular;
    ray_handle specular_texture;
    float specular_tint;
    float roughness;
    ray_handle roughness_texture;
    float anisotropic;
    float anisotropic_rotation;
    float sheen;
    float sheen_tint;
    float clearcoat;
    float clearcoat_roughness;
    float ior;
    float transmission;
    float transmission_roughness;
    float emission_color[3];
    ray_handle emission_texture;
    float emission_strength;
    float alpha;
    ray_handle alpha_texture;
    ray_handle normal_map;
    float normal_map_intensity;
    int32_t importance_sample;
} ray_principled_mat_desc;

typedef struct ray_mat_group_desc { /* Ray::mat_group_desc_t, SceneBase.h:99-112 */
    ray_handle front_mat, back_mat;
    uint64_t vtx_start, vtx_count;
} ray_mat_group_desc;

typedef struct ray_mesh_desc { /* Ray::mesh_desc_t, SceneBase.h:120-133; interleaved float attributes */
    const float *attrs;
    uint64_t attrs_count;         /* number of floats */
    int32_t stride;               /* floats per vertex */
    int32_t pos_offset, nrm_offset, uv_offset;
    int32_t bnm_offset;           /* binormals, -1 = none */
    const uint32_t *indices;
    uint64_t indices_count;
    int32_t base_vertex;
    const ray_mat_group_desc *groups;
    uint32_t groups_count;
    int32_t allow_spatial_splits;
    int32_t use_fast_bvh_build;
} ray_mesh_desc;

typedef struct ray_tex_desc { /* Ray::tex_desc_t, SceneBase.h:172-187 */
    uint32_t format;            /* Ray::eTextureFormat */
    const uint8_t *data;
    uint64_t data_size;
    int32_t w, h;
    int32_t is_srgb, is_normalmap, is_YCoCg, force_no_compression, generate_mipmaps, reconstruct_z;
    int32_t mips_count;         /* levels present in `data` (block-compressed formats; 0 = 1) */
    int32_t convention;         /* Ray::eTextureConvention (SceneBase.h:147-150): 0 OGL, 1 DX (normal maps: y inverted; block textures: stored top-down) */
} ray_tex_desc;

typedef struct ray_light_desc { /* union of the six light descriptors, SceneBase.h:189-262 */
    uint32_t kind;              /* 0 directional, 1 sphere, 2 spot, 3 rect, 4 disk, 5 line */
    float color[3];
    float direction[3], angle;  /* directional / spot direction */
    float position[3];          /* sphere / spot */
    float radius;               /* sphere / spot / line */
    float spot_size, spot_blend;
    float width, height;        /* rect ; disk uses width/height as size_x/size_y ; line: height */
    int32_t doublesided, sky_portal;
    int32_t multiple_importance, cast_shadow, diffuse_visibility, specular_visibility, refraction_visibility;
    float xform[16];            /* rect / disk / line */
} ray_light_desc;

typedef struct ray_camera_desc { /* Ray::camera_desc_t, SceneBase.h:264-311 */
    uint32_t type, filter, view_transform, ltype;
    float filter_width;
    float origin[3], fwd[3], up[3], shift[2];
    float exposure, fov, gamma, sensor_height, focus_distance, focal_length, fstop, lens_rotation, lens_ratio;
    int32_t lens_blades;
    float clip_start, clip_end;
    uint32_t mi_index, uv_index;
    int32_t lighting_only, skip_direct_lighting, skip_indirect_lighting, no_background, output_sh;
    int32_t max_diff_depth, max_spec_depth, max_refr_depth, max_transp_depth, max_total_depth;
    int32_t min_total_depth, min_transp_depth;
    float clamp_direct, clamp_indirect;
    int32_t min_samples;
    float variance_threshold, regularize_alpha;
} ray_camera_desc;

#define RAY_PHYSICAL_SKY_TEXTURE 0xfffffffeull /* Ray::PhysicalSkyTexture (SceneBase.h:35): as env_map / back_map, the analytic sky */
typedef struct ray_env_desc { /* Ray::environment_desc_t, SceneBase.h:343-353; of the atmosphere the fields tests vary, the rest at defaults */
    uint32_t struct_size; /* sizeof(ray_env_desc) as the caller compiled it (ray_default_env fills it in): the struct grew in round 4 and
                           * a caller built against the shorter one must be refused, not read past its end */
    float env_col[3];
    ray_handle env_map;
    float back_col[3];
    ray_handle back_map;
    float env_map_rotation, back_map_rotation;
    int32_t importance_sample;
    int32_t envmap_resolution;                                        /* resolution of the baked sky map (default 1024) */
    float clouds_density, cirrus_clouds_amount, stars_brightness, moon_radius; /* atmosphere_params_t (SceneBase.h:314-341) */
    float clouds_offset_x, clouds_offset_z;
} ray_env_desc;

typedef struct ray_stats { /* RendererBase::stats_t, RendererBase.h:230-244 */
    unsigned long long t[11];
} ray_stats;

/* fills the structs with the defaults the C++ descriptors have */
void ray_default_shading_node(ray_shading_node_desc *d);
void ray_default_principled(ray_principled_mat_desc *d);
void ray_default_light(ray_light_desc *d, uint32_t kind);
void ray_default_camera(ray_camera_desc *d);
void ray_default_env(ray_env_desc *d);

const char *ray_last_error(void);

/* Ray::CreateRenderer (Ray.h:25-28) restricted to ONE backend type by name ("REF","SSE41","AVX","AVX2",
 * "AVX512","HIP"); returns NULL (see ray_last_error) instead of silently falling back to another type */
ray_renderer *ray_renderer_create(const char *type_name, int w, int h, int use_tex_compression, int verbose);
void ray_renderer_destroy(ray_renderer *r);
int ray_renderer_type_name(ray_renderer *r, char *buf, int cap);   /* RendererTypeName(type()) */
int ray_renderer_device_name(ray_renderer *r, char *buf, int cap); /* RendererBase::device_name */
void ray_renderer_size(ray_renderer *r, int out_wh[2]);            /* RendererBase::size */
void ray_renderer_resize(ray_renderer *r, int w, int h);           /* RendererBase::Resize */
void ray_renderer_clear(ray_renderer *r, const float rgba[4]);     /* RendererBase::Clear */
ray_scene *ray_renderer_create_scene(ray_renderer *r);             /* RendererBase::CreateScene */
void ray_renderer_render(ray_renderer *r, ray_scene *s, ray_region *region); /* RendererBase::RenderScene */
void ray_renderer_denoise(ray_renderer *r, ray_region *region);              /* RendererBase::DenoiseImage(region): NLM */
/* RendererBase::InitUNetFilter(alias_memory = false, serial parallel_for); returns unet_filter_properties_t::pass_count */
int ray_renderer_init_unet(ray_renderer *r);
void ray_renderer_denoise_unet(ray_renderer *r, int pass, ray_region *region); /* RendererBase::DenoiseImage(pass, region) */
/* which: 0 get_pixels_ref, 1 get_raw_pixels_ref, 2 aux BaseColor, 3 aux DepthNormals; dst = w*h*4 floats */
int ray_renderer_get_pixels(ray_renderer *r, int which, float *dst);
void ray_renderer_get_stats(ray_renderer *r, ray_stats *st); /* RendererBase::GetStats */
void ray_renderer_reset_stats(ray_renderer *r);              /* RendererBase::ResetStats */
/* The documented multithreading pattern (README.md:336-356, tests/test_scene.cpp:1030-1085): `threads` workers
 * pull tile x tile regions from a shared counter and run `spp` RenderScene iterations on each.  Returns wall
 * seconds.  Only for backends where RendererSupportsMultithreading() holds. */
double ray_renderer_render_tiled_mt(ray_renderer *r, ray_scene *s, int tile, int spp, int threads);
/* the same, continuing a progressive render: every tile starts at RegionContext::iteration = iterations_done */
double ray_renderer_render_tiled_from(ray_renderer *r, ray_scene *s, int tile, int iterations_done, int spp, int threads);

ray_region *ray_region_create(int x, int y, int w, int h); /* Ray::RegionContext */
void ray_region_destroy(ray_region *g);
int ray_region_iteration(ray_region *g);
void ray_region_set_iteration(ray_region *g, int it);

void ray_scene_destroy(ray_scene *s);
/* SceneBase::SetEnvironment; a descriptor whose struct_size is not this library's is ignored and ray_last_error says so */
void ray_scene_set_environment(ray_scene *s, const ray_env_desc *d);
ray_handle ray_scene_add_texture(ray_scene *s, const ray_tex_desc *d);               /* SceneBase::AddTexture */
ray_handle ray_scene_add_material_node(ray_scene *s, const ray_shading_node_desc *d); /* AddMaterial(shading_node_desc_t) */
ray_handle ray_scene_add_material_principled(ray_scene *s, const ray_principled_mat_desc *d);
ray_handle ray_scene_add_mesh(ray_scene *s, const ray_mesh_desc *d);                  /* SceneBase::AddMesh */
ray_handle ray_scene_add_mesh_instance(ray_scene *s, ray_handle mesh, const float xform[16]);
/* mesh_instance_desc_t with its visibility flags (SceneBase.h:135-143); bits: 1 camera, 2 diffuse, 4 specular,
 * 8 refraction, 16 shadow */
ray_handle ray_scene_add_mesh_instance_vis(ray_scene *s, ray_handle mesh, const float xform[16], unsigned visibility);
void ray_scene_set_mesh_instance_transform(ray_scene *s, ray_handle mi, const float xform[16]); /* SceneBase.h:464 */
void ray_scene_remove_mesh_instance(ray_scene *s, ray_handle mi);                             /* :472 */
void ray_scene_remove_mesh(ray_scene *s, ray_handle mesh);                                    /* :424 (the reference's tests add every mesh twice and remove one copy: storage compaction) */
void ray_scene_remove_light(ray_scene *s, ray_handle light);                                  /* :440 */
ray_handle ray_scene_add_light(ray_scene *s, const ray_light_desc *d);                /* the six AddLight overloads */
ray_handle ray_scene_add_camera(ray_scene *s, const ray_camera_desc *d);              /* SceneBase::AddCamera */
void ray_scene_set_current_cam(ray_scene *s, ray_handle cam);
void ray_scene_finalize(ray_scene *s);                                                /* SceneBase::Finalize */
uint32_t ray_scene_triangle_count(ray_scene *s);
uint32_t ray_scene_node_count(ray_scene *s);

/* ---- HIP-backend extras (only in libray_hip.so) ----------------------------------------------------------------
 * A SceneHIP can be created and finalized without a renderer (and therefore without a GPU): scene construction is
 * host work.  ray_hip_export_scene serialises the flat arrays + current camera + pixel-filter table with
 * ray_amd/csrc/scene_blob.h; the blob is what rayhip_scene_upload_blob takes.  This is how one scene build is
 * replicated to the 8 GPUs of a node (one process per GPU) and how big procedural scenes are cached on disk. */
ray_scene *ray_hip_create_scene(int verbose);
const char *ray_hip_sky_baked_on(ray_scene *s); /* Ray::Hip::SkyBakedOn: "device" / "host" / "none" */
ray_scene *ray_hip_create_scene_ex(int verbose, int use_tex_compression); /* settings_t::use_tex_compression for the scene's textures */
int ray_hip_export_scene(ray_scene *s, void **out_blob, uint64_t *out_size);
void ray_hip_free(void *p);
/* the PMJ02 table RendererHIP uploads at start-up (reference internal/Core.h:363-368) */
void ray_hip_pmj_table(const uint32_t **out_ptr, uint32_t *out_count);

#ifdef __cplusplus
}
#endif

#endif /* RAY_CAPI_H */
