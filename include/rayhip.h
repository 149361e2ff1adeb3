/*
 * rayhip.h -- C ABI of librayhip: the MI355X (gfx950) path-tracer core loop behind
 * sergcpp/Ray's RendererBase / SceneBase.
 *
 * This is the drop-in boundary.  Everything on the far side of it is
 * hand-written HIP; everything on the near side is plain C: opaque handles, plain pointers and sizes,
 * `int` status (0 = ok, see rayhip_last_error), no exceptions, no torch / C++ types.
 *   - the caller owns all HOST memory it passes in; the library copies what it needs
 *   - the library owns all DEVICE memory, except buffers the caller hands to the *_device entry points
 *   - one context per GPU; calls on one context are not re-entrant (the reference's GPU renderers are
 *     single-caller too: reference RendererBase.cpp:50-62, tests/test_scene.cpp:1124-1148)
 *
 * Each entry point names the reference interface it stands in for (file:line under the reference tree).
 * The reference-side binding (RendererHIP.cpp / SceneHIP.h, a new eRendererType::HIP) is in
 * ray_amd/host/ and described in INTEGRATION.md.
 *
 * The POD structs below are byte-identical to the reference's internal/Core.h structs: the scene
 * arrays built by the reference's host-side scene code (SceneCPU.cpp) are uploaded verbatim.  The
 * Vulkan backend makes the same promise about its GLSL structs (reference internal/RendererVK.cpp:31-45).
 */
#ifndef RAYHIP_H
#define RAYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RAYHIP_API __attribute__((visibility("default")))
#else
#define RAYHIP_API
#endif

/* ------------------------------------------------------------------------------------------------
 * Scene POD layouts  (== reference internal/Core.h)
 * ---------------------------------------------------------------------------------------------- */

/* precomputed 3-plane triangle, reference Core.h:72-77 (tri_accel_t), built by Core.cpp:212-258 */
typedef struct __attribute__((aligned(16))) rayhip_tri_accel {
    float n_plane[4];
    float u_plane[4];
    float v_plane[4];
} rayhip_tri_accel; /* 48 B */

/* 2-wide BVH node holding its children's boxes, reference Core.h:107-115 (bvh2_node_t).
 * child word: top 3 bits = (prim_count-1) for leaves (0 => inner node), low 29 bits = index
 * (Constants.inl:24-25) */
typedef struct __attribute__((aligned(16))) rayhip_bvh2_node {
    float ch_data0[4]; /* [ ch0.min.x, ch0.max.x, ch0.min.y, ch0.max.y ] */
    float ch_data1[4]; /* [ ch1.min.x, ch1.max.x, ch1.min.y, ch1.max.y ] */
    float ch_data2[4]; /* [ ch0.min.z, ch0.max.z, ch1.min.z, ch1.max.z ] */
    uint32_t left_child, right_child;
    uint32_t _unused0, _unused1;
} rayhip_bvh2_node; /* 64 B */

/* reference Core.h:370-373 (vertex_t) */
typedef struct rayhip_vertex {
    float p[3], n[3], b[3], t[2];
} rayhip_vertex; /* 44 B */

/* reference Core.h:384-391 (mesh_instance_t) */
typedef struct rayhip_mesh_instance {
    uint32_t mesh_index;
    uint32_t node_index;
    uint32_t lights_index;
    uint32_t ray_visibility; /* low 8 bits: ray-type mask; upper 24 bits: lights_block */
    float xform[16], inv_xform[16];
} rayhip_mesh_instance; /* 144 B */

/* reference Core.h:166-168 (tri_mat_data_t) */
typedef struct rayhip_tri_mat_data {
    uint16_t front_mi, back_mi;
} rayhip_tri_mat_data; /* 4 B */

/* reference Core.h:170-195 (material_t); `type` is Ray::eShadingNode (SceneBase.h:46) */
typedef struct rayhip_material {
    uint32_t textures[5];
    float base_color[3];
    uint32_t flags;
    uint32_t type;
    float tangent_rotation_or_strength;
    uint16_t roughness_unorm;
    uint16_t anisotropic_unorm;
    float ior;
    uint16_t sheen_unorm;
    uint16_t sheen_tint_unorm;
    uint16_t tint_unorm;
    uint16_t metallic_unorm;
    uint16_t transmission_unorm;
    uint16_t transmission_roughness_unorm;
    uint16_t specular_unorm;
    uint16_t specular_tint_unorm;
    uint16_t clearcoat_unorm;
    uint16_t clearcoat_roughness_unorm;
    uint16_t normal_map_strength_unorm;
    uint16_t _pad;
} rayhip_material; /* 76 B */

/* reference Core.h:197-237 (light_t).  Word 0 is the bitfield
 *   type:3 | doublesided:1 | cast_shadow:1 | visible:1 | sky_portal:1 | ray_visibility:8 | unused:17
 * followed by col[3] and 12 floats whose meaning depends on `type` (sph/rect/disk/line/tri/dir). */
typedef struct rayhip_light {
    uint32_t flags;
    float col[3];
    float params[12];
} rayhip_light; /* 64 B */

/* reference Core.h:139-156 (cwbvh_node_t + light_cwbvh_node_t): 8-wide quantised light-tree node */
typedef struct rayhip_light_cwbvh_node {
    float bbox_min[3];
    float _unused0;
    float bbox_max[3];
    float _unused1;
    uint8_t ch_bbox_min[3][8];
    uint8_t ch_bbox_max[3][8];
    uint32_t child[8];
    float flux[8];
    uint32_t axis[8];
    uint32_t cos_omega_ne[8];
} rayhip_light_cwbvh_node; /* 208 B */

/* One texture = a mip chain of linear RGBA8 images inside the texel pool.  The reference keeps
 * RGBA8/RGB8/RG8/R8 in swizzled CPU tiles (TextureStorageCPU.h:229-331); that swizzle is a CPU-cache
 * device, so the boundary hands over plain row-major RGBA8 (missing channels replicated exactly like
 * TexStorageSwizzled::Fetch does).  Texture handles in materials keep the reference encoding
 * (storage<<28 | flags<<24 | index, SceneCPU.cpp:192-205); `tex_table[(handle>>28)]` gives the first
 * entry of that storage in `textures`. */
typedef struct rayhip_texture {
    uint32_t width[12], height[12]; /* per mip level; unused levels repeat the last valid one */
    uint32_t offset[12];            /* texel offset of each level in the texel pool */
} rayhip_texture;

/* subset of reference Core.h:393-409 (environment_t) that the path reads */
typedef struct rayhip_environment {
    float env_col[3];
    uint32_t env_map;
    float back_col[3];
    uint32_t back_map;
    float env_map_rotation;
    float back_map_rotation;
    uint32_t light_index;
    float sky_map_spread_angle;
    int32_t qtree_levels; /* levels of the env-map importance quadtree (rayhip_scene_desc::env_qtree), 0 = none */
    uint32_t _pad[3];
} rayhip_environment;

/* == Ray::camera_t (reference Types.h:96-108), including pass_settings_t (Types.h:82-93) */
typedef struct rayhip_pass_settings {
    uint8_t max_diff_depth, max_spec_depth, max_refr_depth, max_transp_depth, max_total_depth;
    uint8_t min_total_depth, min_transp_depth;
    uint8_t flags;
    float clamp_direct, clamp_indirect;
    int32_t min_samples;
    float variance_threshold;
    float regularize_alpha;
} rayhip_pass_settings; /* 28 B */

typedef struct rayhip_camera {
    uint8_t type;           /* Ray::eCamType */
    uint8_t filter;         /* Ray::ePixelFilter */
    uint8_t view_transform; /* Ray::eViewTransform */
    uint8_t ltype;          /* Ray::eLensUnits */
    float filter_width;
    float fov, exposure, gamma, sensor_height;
    float focus_distance, focal_length, fstop, lens_rotation, lens_ratio;
    int32_t lens_blades;
    float clip_start, clip_end;
    float origin[3], fwd[3], side[3], up[3], shift[2];
    uint32_t mi_index, uv_index;
    rayhip_pass_settings pass_settings;
} rayhip_camera;

/* Flat scene: what reference RendererCPU.h:390-413 gathers into scene_data_t before every RenderScene.
 * All pointers are HOST pointers; counts are element counts (capacity of the sparse pools, since the
 * indices stored in the arrays are pool slots). */
/* == Ray::atmosphere_params_t (reference SceneBase.h:314-341), field for field: the physical sky's parameters */
typedef struct __attribute__((aligned(16))) rayhip_atmosphere {
    float planet_radius, viewpoint_height, atmosphere_height, rayleigh_height, mie_height;
    float clouds_height_beg, clouds_height_end, clouds_variety, clouds_density;
    float clouds_offset_x, clouds_offset_z, clouds_flutter_x, clouds_flutter_z;
    float cirrus_clouds_amount, cirrus_clouds_height;
    float ozone_height_center, ozone_half_width, atmosphere_density;
    float stars_brightness, moon_radius, moon_distance;
    float moon_dir[4] __attribute__((aligned(16)));
    float rayleigh_scattering[4], mie_scattering[4], mie_extinction[4], mie_absorption[4], ozone_absorption[4], ground_albedo[4];
} rayhip_atmosphere; /* 208 B */

/* The physical sky (environment_t::env_map == PhysicalSkyTexture: env.sky_map_spread_angle > 0).  Rays wider than that angle read
 * the baked environment map like any other (the reference bakes it on the host, SceneCPU.cpp:1017-1056, and so does SceneHIP); rays
 * narrower than it -- camera rays, mirror bounces -- are evaluated analytically (ShadeSkyPrimary / ShadeSkySecondary,
 * RendererCPU.h:484-486, 555-557; AtmosphereRef.cpp), which needs what this struct names: the atmosphere, the two look-up tables
 * the scene computes from it (SceneCommon.cpp:186-283), the indices of the directional lights (the suns), and the five textures the
 * reference compiles in (internal/precomputed: weather map, 3-d noise, curl noise, moon albedo, cirrus).  One element or none. */
typedef struct rayhip_sky {
    rayhip_atmosphere atmosphere;
    int32_t transmittance_lut_w, transmittance_lut_h; /* rayhip_scene_desc::sky_transmittance_lut: w * h * 4 floats */
    int32_t multiscatter_lut_res;                     /* ... sky_multiscatter_lut: res * res * 4 floats (0: none) */
    int32_t weather_res;                              /* sky_weather_tex: res * res * 3 bytes */
    int32_t noise3d_res;                              /* sky_noise3d_tex: res^3 bytes */
    int32_t curl_res;                                 /* sky_curl_tex: res * res * 3 bytes */
    int32_t moon_w, moon_h;                           /* sky_moon_tex: w * h * 3 bytes */
    int32_t cirrus_res;                               /* sky_cirrus_tex: res * res * 2 bytes */
    int32_t _pad[3];
} rayhip_sky; /* 256 B */

typedef struct rayhip_scene_desc {
    /* sizeof(rayhip_scene_desc) as the CALLER compiled it (first member, so that it is readable whatever the rest looks like): the struct
     * has grown (round 4: the sky members) and will again; an upload whose struct_size is not the library's is refused instead of being read
     * past its end.  RAYHIP_ABI_VERSION / rayhip_abi_version() say which header a library was built from. */
    uint32_t struct_size;
    const rayhip_bvh2_node *nodes;
    uint32_t nodes_count;
    const rayhip_tri_accel *tris;
    uint32_t tris_count;
    const uint32_t *tri_indices;
    uint32_t tri_indices_count;
    const rayhip_tri_mat_data *tri_materials;
    uint32_t tri_materials_count;
    const rayhip_material *materials;
    uint32_t materials_count;
    const rayhip_vertex *vertices;
    uint32_t vertices_count;
    const uint32_t *vtx_indices;
    uint32_t vtx_indices_count;
    const rayhip_mesh_instance *mesh_instances;
    uint32_t mesh_instances_count;
    const rayhip_light *lights;
    uint32_t lights_count;
    const uint32_t *li_indices;
    uint32_t li_indices_count;
    const rayhip_light_cwbvh_node *light_cwnodes;
    uint32_t light_cwnodes_count;
    const rayhip_texture *textures;
    uint32_t textures_count;
    /* env-map importance quadtree (reference environment_t::qtree_mips, Core.h:400; built by
     * Scene::PrepareEnvMapQTree): the env.qtree_levels mips concatenated, lod 0 (finest) first; mip `lod` holds
     * 4^(levels-1-lod) quads of 4 floats (luminance of the four sub-cells), row-major.  Count in floats. */
    const float *env_qtree;
    uint32_t env_qtree_count;
    const uint32_t *texels; /* RGBA8 pool */
    uint32_t texels_count;
    uint32_t tex_table[8]; /* first `textures` entry of each reference storage (RGBA,RGB,RG,R,BC1,BC3,BC4,BC5) */
    rayhip_environment env;
    uint32_t tlas_root; /* 0xffffffff = empty scene */
    uint32_t visible_lights_count;
    uint32_t blocker_lights_count;
    float bbox_min[3], bbox_max[3]; /* Scene::GetBounds (SceneCPU.cpp:1523), feeds the ray-sort grid only */
    uint32_t texture_flags;         /* RAYHIP_TEX_* */
    /* the physical sky (see rayhip_sky): all null / 0 unless env.sky_map_spread_angle > 0 */
    const rayhip_sky *sky;
    uint32_t sky_count; /* 0 or 1 */
    const float *sky_transmittance_lut;
    uint32_t sky_transmittance_lut_count;
    const float *sky_multiscatter_lut;
    uint32_t sky_multiscatter_lut_count;
    const uint32_t *sky_dir_lights; /* indices into `lights` of the LIGHT_TYPE_DIR lights (Scene::dir_lights_) */
    uint32_t sky_dir_lights_count;
    const uint8_t *sky_weather_tex, *sky_noise3d_tex, *sky_curl_tex, *sky_moon_tex, *sky_cirrus_tex;
    uint32_t sky_weather_tex_count, sky_noise3d_tex_count, sky_curl_tex_count, sky_moon_tex_count, sky_cirrus_tex_count; /* bytes */
} rayhip_scene_desc;

/* rayhip_scene_desc::texture_flags.  RAW_BC: the textures of the four block-compressed storages (tex_table[4..7]: BC1,
 * BC3, BC4, BC5 -- reference TexStorageBCn<3|4|1|2>, TextureStorageCPU.h:364-617) are handed over as the reference keeps
 * them, 4x4 blocks of 8 / 16 / 8 / 16 bytes in row-major tile order, and decoded per fetch on the device (rt_texture.h);
 * rayhip_texture::offset[] of such a texture counts 32-bit words of `texels` like everywhere else, width / height are
 * the mip's size in texels.  Without the flag every texture is row-major RGBA8. */
#define RAYHIP_TEX_RAW_BC 1u

/* == RendererBase::stats_t (reference RendererBase.h:230-244), microseconds */
typedef struct rayhip_stats {
    unsigned long long time_primary_ray_gen_us;
    unsigned long long time_primary_trace_us;
    unsigned long long time_primary_shade_us;
    unsigned long long time_primary_shadow_us;
    unsigned long long time_secondary_sort_us;
    unsigned long long time_secondary_trace_us;
    unsigned long long time_secondary_shade_us;
    unsigned long long time_secondary_shadow_us;
    unsigned long long time_denoise_us;
    unsigned long long time_cache_update_us;
    unsigned long long time_cache_resolve_us;
} rayhip_stats;

/* Wavefront ray state as the reference defines it (CoreRef.h:57-105).  Device storage is SoA
 * (DESIGN.md); these AoS forms exist only at this boundary, for the kernel-level test hooks. */
typedef struct rayhip_ray {
    float o[3], d[3], pdf;
    float c[3];
    float ior[4];
    float cone_width, cone_spread;
    uint32_t xy;
    uint32_t depth;
} rayhip_ray; /* 72 B */

typedef struct rayhip_shadow_ray {
    float o[3];
    uint32_t depth;
    float d[3], dist;
    float c[3];
    uint32_t xy;
} rayhip_shadow_ray; /* 48 B */

typedef struct rayhip_hit {
    int32_t obj_index;
    int32_t prim_index;
    float t, u, v;
} rayhip_hit; /* 20 B */

/* per-launch traversal work counters (instrumented variant of the traversal kernels): the inputs of the
 * algorithmic-bytes formula of SURVEY.md section 8d */
typedef struct rayhip_trav_counters {
    unsigned long long rays;
    unsigned long long nodes;     /* bvh2 nodes fetched (64 B each) */
    unsigned long long tris;      /* triangles tested (48 B each) */
    unsigned long long instances; /* mesh instances entered (144 B each) */
    unsigned long long max_stack; /* deepest traversal-stack use seen (entries per ray; not a sum) */
    unsigned long long nodes4;    /* 4-wide quantised BLAS nodes fetched (64 B each): RAYHIP_FLAG_COUNT_WIDE only */
} rayhip_trav_counters;

typedef struct rayhip_ctx rayhip_ctx;

enum {
    RAYHIP_BUF_FINAL = 0,        /* tonemapped, RendererBase::get_pixels_ref (RendererBase.h:152) */
    RAYHIP_BUF_RAW = 1,          /* linear running mean, get_raw_pixels_ref (RendererBase.h:157) */
    RAYHIP_BUF_BASE_COLOR = 2,   /* get_aux_pixels_ref(eAUXBuffer::BaseColor) */
    RAYHIP_BUF_DEPTH_NORMALS = 3, /* get_aux_pixels_ref(eAUXBuffer::DepthNormals) */
    RAYHIP_BUF_VARIANCE = 4       /* per-pixel variance estimate of the last accumulate (what the reference leaves in its temp
                                     buffer for DenoiseImage, RendererCPU.h:641-645) */
};

enum {
    RAYHIP_FLAG_SORT_RAYS = 1u << 0,     /* ray sort between bounces (RendererVK.cpp:641-652) */
    RAYHIP_FLAG_COUNT_TRAVERSAL = 1u << 1, /* run the instrumented traversal kernels (slower) */
    RAYHIP_FLAG_TIME_STAGES = 1u << 2,     /* record HIP events around every stage WITHOUT synchronising; read the
                                              result later with rayhip_get_stage_times / rayhip_get_trav_timing */
    RAYHIP_FLAG_COUNT_WIDE = 1u << 3       /* run the PRODUCT traversal kernels (4-wide quantised BLAS) with visit counters:
                                              nodes = TLAS BVH2 nodes, nodes4 = 4-wide nodes, tris, instances -- the
                                              kernel's own algorithmic bytes (COUNT_TRAVERSAL counts the reference's BVH2
                                              walk on the same rays instead).  Wins over COUNT_TRAVERSAL. */
};

/* ------------------------------------------------------------------------------------------------
 * Entry points
 * ---------------------------------------------------------------------------------------------- */

/* thread-local message for the last non-zero status returned on this thread */
RAYHIP_API const char *rayhip_last_error(void);

/* RAYHIP_ABI_VERSION of the header the library was built from.  Bumped whenever a struct or a signature of this file changes shape
 * (5: rayhip_scene_desc::struct_size, round 5); a host compares it with its own RAYHIP_ABI_VERSION before the first upload. */
#define RAYHIP_ABI_VERSION 5
RAYHIP_API int rayhip_abi_version(void);

/* Number of usable gfx950 devices.  The reference GPU factories throw when no device is present and
 * the caller falls back (Ray.cpp:58-63); Hip::CreateRenderer does the same on a 0 here. */
RAYHIP_API int rayhip_device_count(void);

/* Vk::Renderer::Renderer (RendererVK.cpp:240-325): create context on `device`, own stream */
RAYHIP_API int rayhip_ctx_create(int device, rayhip_ctx **out_ctx);
RAYHIP_API void rayhip_ctx_destroy(rayhip_ctx *ctx);
/* RendererBase::device_name (RendererBase.h:143) */
RAYHIP_API int rayhip_ctx_device_name(rayhip_ctx *ctx, char *buf, int cap);

/* upload of the 32 x 4096 x 2 PMJ02 table (RendererVK.cpp:299-311; table: Core.h:363-368) */
RAYHIP_API int rayhip_upload_static(rayhip_ctx *ctx, const uint32_t *pmj02_samples, uint32_t count);

/* RendererBase::Resize / Clear (RendererBase.h:176,181; RendererGPU.h:380-488) */
RAYHIP_API int rayhip_resize(rayhip_ctx *ctx, int w, int h);
RAYHIP_API int rayhip_clear(rayhip_ctx *ctx, const float rgba[4]);

/* flat-array upload after SceneBase::Finalize (what SceneVK does buffer by buffer, SceneGPU.h:62-104) */
RAYHIP_API int rayhip_scene_upload(rayhip_ctx *ctx, const rayhip_scene_desc *desc);

/* What SceneBase::SetMeshInstanceTransform / AddMeshInstance / RemoveMeshInstance / AddLight / RemoveLight /
 * SetEnvironment / Finalize change (SceneCPU.cpp:1004-1094; RebuildTLAS SceneCPU.cpp:1103-1162; RebuildLightTree
 * SceneCPU.cpp:1411-1521), without sending the geometry again: `desc` is the scene as after those calls -- its
 * mesh_instances, lights, li_indices, light_cwnodes, env, env_qtree, tlas_root (+ nodes: only the top level is read, for
 * the live instance slots and their world-space boxes), visible/blocker light counts and bounds are used, vertices /
 * vtx_indices for the corners of triangle lights; textures / texels may be null.  The top-level tree is rebuilt on the
 * device over those boxes (ray_amd/csrc/lbvh.hip.h).  Meshes, materials and textures must be the ones of the last
 * rayhip_scene_upload.  Returns 0; 1 = error; 2 = this change needs rayhip_scene_upload (an instance of a mesh that was
 * not in use at the last upload, geometry arrays of another size) -- nothing on the device was touched.  After 1 the
 * context still holds the previous top-level tree (new trees are written to alternating halves of a reserved region),
 * but instance / light arrays may already be the new ones: re-send the scene with rayhip_scene_upload. */
RAYHIP_API int rayhip_scene_update_instances(rayhip_ctx *ctx, const rayhip_scene_desc *desc);

/* which acceleration-structure form the traversal kernels walk for the uploaded scene: 4 = the 4-wide quantised BLAS
 * (ray_amd/csrc/rt_bvh4.h, the default), 8 = the 8-wide one (rt_bvh8.h; RAYHIP_BVH_WIDTH=8), 2 = the reference's BVH2 as handed
 * over (RAYHIP_BVH_WIDTH=2, or a child box that cannot be quantised); 0 = no scene.  (What the algorithmic-bytes figure of
 * bench.py prices a node visit with: 64 / 80 / 64 bytes.) */
RAYHIP_API int rayhip_scene_bvh_width(rayhip_ctx *ctx);

/* which form of the closest-hit kernel rayhip_render[_batch] launches for the secondary bounces of the uploaded scene: 0 = one ray
 * per lane to completion (k_trace_closest), 1 = persistent wavefronts that refill finished lanes from the queue
 * (k_trace_closest_refill), 2 = ... from a pool of prepared rays in LDS (k_trace_closest_pool: scenes with one instance, 4-wide
 * BLAS; RAYHIP_REFILL selects).  All forms find the same hits bit for bit; this is what bench.py names in its roofline block. */
RAYHIP_API int rayhip_closest_hit_form(rayhip_ctx *ctx);

/* Same upload from a serialised scene (ray_amd/csrc/scene_blob.h; written by the reference-side SceneHIP or by
 * tests/golden/make_fixtures.py): uploads the arrays AND the filter table stored in the blob and returns the
 * camera stored with it.  `blob` must be 16-byte aligned. */
RAYHIP_API int rayhip_scene_upload_blob(rayhip_ctx *ctx, const void *blob, size_t size, rayhip_camera *out_cam);
/* rayhip_scene_update_instances from a serialised scene (same return values; the filter table in the blob is ignored) */
RAYHIP_API int rayhip_scene_update_instances_blob(rayhip_ctx *ctx, const void *blob, size_t size, rayhip_camera *out_cam);

/* 1024-entry inverse filter CDF (RendererCPU.h:1234-1258 UpdateFilterTable; upload RendererVK.cpp:386-424) */
RAYHIP_API int rayhip_set_filter_table(rayhip_ctx *ctx, const float *table, int count);

/* RendererBase::RenderScene (RendererBase.h:196; schedule RendererVK.cpp:368-791, arithmetic
 * RendererCPU.h:373-659) for ONE iteration over `rect` = {x,y,w,h}.  `iteration` is 1-based, i.e. the
 * value of RegionContext::iteration after the increment at RendererCPU.h:384.  `stats` may be NULL; when
 * given, per-stage GPU times of this call (HIP events on the context stream) are ADDED to it.
 * The call returns when the work is enqueued unless stats != NULL (then it synchronises). */
RAYHIP_API int rayhip_render(rayhip_ctx *ctx, const rayhip_camera *cam, const int rect[4], int iteration,
                             uint32_t flags, rayhip_stats *stats);
/* The same for `count` consecutive iterations first_iteration .. first_iteration + count - 1 of the rect -- what `count`
 * RenderScene calls with an unchanged scene and camera produce, bit for bit -- but up to rayhip_max_batch() of them share one
 * wavefront pass (more rays per launch: small frames and tile shards fill the GPU).  Falls back to one pass per
 * iteration when adaptive sampling is active (variance_threshold != 0) or RAYHIP_FLAG_SORT_RAYS is set. */
RAYHIP_API int rayhip_render_batch(rayhip_ctx *ctx, const rayhip_camera *cam, const int rect[4], int first_iteration,
                                   int count, uint32_t flags, rayhip_stats *stats);
/* Largest number of iterations one wavefront pass of the current frame can carry:
 * min(512, (65535 / width) * (65535 / height)) -- the layers of a pass are stacked side by side and on top of each other
 * in one virtual frame whose coordinates must fit the two 16-bit halves of ray_data_t::xy (internal/Core.h) -- and fewer
 * for frames so large that the virtual frame would pass 2^31 pixels.
 * rayhip_render_batch splits longer runs itself; callers that choose the run length (RendererHIP's deferred RenderScene
 * calls, bench.py) use this to cut a render into passes of equal size.  Memory: about 0.3 KB of wavefront state per ray
 * in flight plus 48 B per pixel and layer.  0 before rayhip_resize. */
RAYHIP_API int rayhip_max_batch(rayhip_ctx *ctx);
/* Allocate now what passes of `count` iterations over the whole frame need under the current shard (otherwise the
 * first such rayhip_render_batch does it, synchronising the stream). */
RAYHIP_API int rayhip_reserve_batch(rayhip_ctx *ctx, int count);

/* Look-up table of the non-Standard view transforms (camera_desc_t::view_transform: AgX, Filmic_*): dims^3 entries,
 * RGB10_A2, x fastest -- the reference's precomputed tables (internal/TonemapRef.cpp:15-26, uploaded by its GPU
 * back-ends as a 3-D texture, RendererVK.cpp:404-415); arithmetic of the look-up: TonemapRef.cpp:29-80.  The table is
 * copied.  A camera with view_transform != Standard renders only after the table of THAT transform has been set (the
 * caller owns the transform -> table mapping); scene blobs carry the table of their camera. */
RAYHIP_API int rayhip_set_tonemap_lut(rayhip_ctx *ctx, int view_transform, const uint32_t *lut, int dims);

/* RendererBase::DenoiseImage(const RegionContext &) (RendererBase.h:199; arithmetic RendererCPU.h:661-783 +
 * DenoiseRef.cpp:9-93): separable 9-tap pre-filter of the per-pixel variance estimate the last accumulate left, joint
 * non-local-means filter (7x7 window, 3x3 patches) of the reversibly tone-mapped running mean guided by the variance
 * and by the base-colour / depth-normal images, then RAW <- filtered colour, FINAL <- Tonemap(RAW) on `rect`, and the
 * adaptive-sampling flags of the rect's pixels.  `iteration` = RegionContext::iteration of the last RenderScene.
 * SURVEY.md section 8f, N2; the UNet denoiser (DenoiseImage(pass, region)) is rayhip_unet_init / rayhip_denoise_unet below. */
RAYHIP_API int rayhip_denoise_nlm(rayhip_ctx *ctx, const rayhip_camera *cam, const int rect[4], int iteration);

/* Multi-GPU tile sharding (new): this context renders only the pixels of the tile x tile
 * squares (row-major walk over the frame) whose ordinal % shard_count == shard_index; the other pixels of its
 * buffers are never written by it.  Every pixel has exactly one owner, so ONE gather of the owned tiles
 * (rayhip_comm_reduce_framebuffers, or rayhip_export_owned / rayhip_import_owned with the caller's transport)
 * assembles the full frame on the root, bit-identical to a single-GPU render (a copy, not a sum: rounds 1-2 reduced
 * zero-padded frames).  Default is (64, 1, 0) = everything. */
RAYHIP_API int rayhip_set_shard(rayhip_ctx *ctx, int tile, int shard_count, int shard_index);

/* blocking device->host copy, get_pixels_ref & co. (RendererVK.cpp:1698-1757) */
RAYHIP_API int rayhip_readback(rayhip_ctx *ctx, int which, float *dst_rgba, int pitch_px);
/* same, into DEVICE memory the caller owns (e.g. a torch tensor that is then reduced over RCCL);
 * enqueued on the context stream, followed by a stream synchronise */
RAYHIP_API int rayhip_readback_device(rayhip_ctx *ctx, int which, void *dst_device_rgba, int pitch_px);
/* overwrite the running-mean (RAW/"full") buffer from DEVICE memory and re-run the tonemap pass; used
 * after the multi-GPU tile reduce so that rank 0 holds the combined frame */
RAYHIP_API int rayhip_set_raw_device(rayhip_ctx *ctx, const void *src_device_rgba, int pitch_px,
                                     const rayhip_camera *cam);

/* The sky environment map of a physical-sky scene baked ON THE DEVICE: what Scene::PrepareSkyEnvMap_nolock produces on the host
 * (internal/SceneCPU.cpp:1017-1056 over CalcSkyEnvTexture, SceneCommon.cpp:286-361; the reference's GPU scene runs it as a compute pass,
 * SceneGPU.h:1697-1768).  `desc`: only its sky members (sky, the two tables, the five textures, sky_dir_lights, env.sky_map_spread_angle > 0)
 * and `lights` are read; `out_rgbe8`: w * h texels, shared-exponent RGBE bytes as the reference's RGBA8 texture storage takes them.  The scene
 * that is on the device is not touched.  SceneHIP::Finalize calls it when the scene belongs to a renderer (a scene built without one bakes on
 * the host). */
RAYHIP_API int rayhip_bake_sky(rayhip_ctx *ctx, const rayhip_scene_desc *desc, int w, int h, uint32_t *out_rgbe8);
/* the same from a serialised scene (ray_amd/csrc/scene_blob.h) */
RAYHIP_API int rayhip_bake_sky_blob(rayhip_ctx *ctx, const void *blob, size_t blob_size, int w, int h, uint32_t *out_rgbe8);

/* ---- UNet denoiser (RendererBase::InitUNetFilter + DenoiseImage(int pass, const RegionContext &), RendererBase.h:199-227;
 * reference implementation RendererCPU.h:790-1007, 1261-1310 over internal/Convolution.h; GPU twin shaders/convolution.comp.glsl) ----
 * Sixteen 3 x 3 convolution passes of OIDN's UNet over (running mean, base colour, depth-normals), each one implicit GEMM on
 * the f32 matrix cores (ray_amd/csrc/unet_kernels.hip).  rayhip_unet_init takes the weights exactly as the reference's own
 * SetupUNetWeights<float>(alignment, &offsets, weights) lays them out (internal/UNetFilter.h:27-47: `offsets` is
 * unet_weight_offsets_t as 32 ints) and re-packs them for the device.  rayhip_denoise_unet runs pass 0 .. 15 (or -1: all) on
 * `rect` (corner a multiple of 16); the last pass writes RAW (alpha untouched) and FINAL = Tonemap(RAW) with `cam`. */
RAYHIP_API int rayhip_unet_init(rayhip_ctx *ctx, const float *weights, int weights_count, const int32_t offsets[32], int alignment);
RAYHIP_API int rayhip_denoise_unet(rayhip_ctx *ctx, const rayhip_camera *cam, const int rect[4], int pass);
/* Which arithmetic the passes run in (takes effect with the next rayhip_denoise_unet; switch between frames, not between the passes of one):
 *   0  f32 tensors and weights on v_mfma_f32_16x16x4_f32 -- the exact form (default of the C ABI: every pass within 2e-5 of the CPU reference)
 *   1  f16 tensors and weights, f32 accumulators, on v_mfma_f32_16x16x32_f16 -- what the reference's own GPU backends run where the device has
 *      half-precision (matrix) arithmetic (internal/RendererVK.cpp:254-263, 1834-1844; RendererGPU.h:533-545): 12.8 x the matrix rate, half the
 *      tensor traffic; RendererHIP selects it like they do (RAY_HIP_UNET_F32=1 keeps the exact form). */
RAYHIP_API int rayhip_unet_set_precision(rayhip_ctx *ctx, int half);
/* test hook: activation tensor `which` (0 .. 14 in the order of unet_filter_tensors_t) incl. its one-pixel border, NHWC */
RAYHIP_API int rayhip_unet_read_tensor(rayhip_ctx *ctx, int which, float *dst, size_t capacity_floats, int out_dims[3]);

/* ---- multi-GPU: the frame exchange behind the C ABI (new, the reference has no multi-GPU mode) ----
 * N contexts (one per GPU) render the tiles rayhip_set_shard deals them; ONE exchange over RCCL/xGMI then assembles the
 * frame on `root`: the shards are disjoint tiles, so every rank sends the tiles it owns, densely packed (1/N of the
 * frame), straight to the root (ncclSend / ncclRecv in one group: N - 1 point-to-point links in parallel), which scatters
 * them into its images.  Copies are exact: the result equals a single-GPU render bit for bit and the call can be
 * repeated after more iterations (progressive refinement).
 *   rayhip_comm_create            one process drives all `ndev` GPUs (what RendererHIP does with RAY_HIP_DEVICES): the root pulls the
 *                                 senders' tiles with peer copies on its stream -- no collective library; ranks may share a device
 *   rayhip_comm_unique_id +       one process per GPU: rank 0 makes the id, hands its RAYHIP_COMM_ID_BYTES bytes to the other
 *   rayhip_comm_create_rank       processes by its own means, every process creates its rank (ncclCommInitRank) -- collective
 *   rayhip_comm_bind              attach the context of `rank` (a GPU of `devices[rank]`); also sets its shard to (tile, N, rank)
 *   rayhip_comm_reduce_framebuffers   `what`: RAYHIP_REDUCE_* mask, 0 = all four images.  On return the root context holds the
 *                                 combined running mean (RAW, and FINAL re-tonemapped with `cam`), aux images and variance
 *                                 estimate, i.e. everything DenoiseImage reads.  Blocking (synchronises the local streams).
 * RCCL is only used by the one-process-per-GPU form and loaded on first use (dlopen librccl.so.1; RAYHIP_RCCL_LIB
 * overrides); librayhip itself does not link it. */
typedef struct rayhip_comm rayhip_comm;
#define RAYHIP_COMM_ID_BYTES 128
enum {
    RAYHIP_REDUCE_RADIANCE = 1u << 0,
    RAYHIP_REDUCE_BASE_COLOR = 1u << 1,
    RAYHIP_REDUCE_DEPTH_NORMALS = 1u << 2,
    RAYHIP_REDUCE_VARIANCE = 1u << 3,
    RAYHIP_REDUCE_ALL = 15u
};
RAYHIP_API int rayhip_comm_create(int ndev, const int *devices, rayhip_comm **out_comm);
/* rayhip_comm_probe: 0 if this process can load RCCL with every symbol the exchange uses -- local, NOT collective: the ranks of a job agree on
 * it before any of them calls rayhip_comm_create_rank (which blocks until all have).  rayhip_comm_info: out[0] = ranks, out[1] = (first) local
 * rank, out[2] = ncclCommCount as RCCL reports it (-1: in-process form), out[3] = 1 for the in-process (peer copy) transport. */
RAYHIP_API int rayhip_comm_probe(void);
RAYHIP_API int rayhip_comm_info(rayhip_comm *comm, int out[4]);
RAYHIP_API int rayhip_comm_unique_id(void *out_id, size_t size);
RAYHIP_API int rayhip_comm_create_rank(const void *unique_id, int nranks, int rank, rayhip_ctx *ctx, rayhip_comm **out_comm);
RAYHIP_API int rayhip_comm_bind(rayhip_comm *comm, int rank, rayhip_ctx *ctx);
RAYHIP_API int rayhip_comm_reduce_framebuffers(rayhip_comm *comm, int root, uint32_t what, const rayhip_camera *cam);
RAYHIP_API void rayhip_comm_destroy(rayhip_comm *comm);
/* The pack step alone for hosts that bring their own collective: this rank's OWNED pixels of image `which`
 * (RAYHIP_BUF_RAW = the running mean, BASE_COLOR, DEPTH_NORMALS, VARIANCE), zero elsewhere, tightly packed [h][w][4] into
 * DEVICE memory.  Sum over ranks = the frame; hand it to the root with rayhip_set_raw_device. */
RAYHIP_API int rayhip_export_shard_device(rayhip_ctx *ctx, int which, void *dst_device_rgba);
/* The same exchange with the caller's transport (MPI, torch.distributed, ...), in the packing the RCCL path uses: the
 * tiles a rank owns, densely -- owned tile j of rank r is frame tile r + j * N, one tile x tile slot of float4 each, the
 * images of the RAYHIP_REDUCE_* mask `what` (0 = all four) one after the other.  rayhip_owned_bytes: size of rank `rank`'s
 * buffer; rayhip_export_owned: pack this context's tiles (its shard: rayhip_set_shard) into DEVICE memory;
 * rayhip_import_owned: scatter rank `from_rank`'s buffer into this context's images (the root calls it once per sender);
 * rayhip_finish_import: the assembled radiance image becomes RAW and is tonemapped into FINAL.  All three wait for the
 * context stream. */
RAYHIP_API size_t rayhip_owned_bytes(rayhip_ctx *ctx, uint32_t what, int nranks, int rank);
RAYHIP_API int rayhip_export_owned(rayhip_ctx *ctx, uint32_t what, void *dst_device, size_t capacity_bytes);
RAYHIP_API int rayhip_import_owned(rayhip_ctx *ctx, uint32_t what, int from_rank, const void *src_device, size_t bytes);
RAYHIP_API int rayhip_finish_import(rayhip_ctx *ctx, const rayhip_camera *cam);

RAYHIP_API int rayhip_sync(rayhip_ctx *ctx);

/* traversal counters accumulated by RAYHIP_FLAG_COUNT_TRAVERSAL renders since the last reset:
 * [0] closest-hit kernel (K2), [1] shadow any-hit kernel (K3) */
RAYHIP_API int rayhip_get_trav_counters(rayhip_ctx *ctx, rayhip_trav_counters out[2], int reset);
/* per-stage GPU time (us) accumulated by renders that passed stats != NULL or RAYHIP_FLAG_TIME_STAGES; this is
 * what RendererBase::GetStats (RendererBase.h:245) returns for the HIP backend.  Synchronises the stream.
 * The stages are EXCLUSIVE intervals of the context's stream and add up to the time of the pass (the reference's GPU backends report
 * per-stage timestamps the same way, RendererVK.cpp:452-487).  The shadow launch of bounce b runs on a second stream beside the closest-hit
 * launch of bounce b + 1 (RAYHIP_OVERLAP_SHADOW): its stage entry is the time the pass waited for it AFTER that closest-hit launch ended;
 * the part that ran beside the trace is inside the trace stage (its own elapsed time: rayhip_get_trav_timing, kernel 1). */
RAYHIP_API int rayhip_get_stage_times(rayhip_ctx *ctx, rayhip_stats *out, int reset);
/* GPU time (ms, HIP events on the context stream) and launch count of the closest-hit traversal kernel
 * and the shadow kernel, accumulated over renders that passed stats != NULL or RAYHIP_FLAG_TIME_STAGES; [0]=K2 [1]=K3 */
RAYHIP_API int rayhip_get_trav_timing(rayhip_ctx *ctx, double out_ms[2], unsigned long long out_launches[2],
                                      int reset);

/* ---- kernel-level hooks (tests only use these; they run the same kernels RenderScene launches) ---- */

/* Ref::GeneratePrimaryRays (CoreRef.cpp:1429-1553): fills out_rays/out_hits (host, rect.w*rect.h each) */
RAYHIP_API int rayhip_k_generate_primary_rays(rayhip_ctx *ctx, const rayhip_camera *cam, const int rect[4],
                                              int iteration, rayhip_ray *out_rays, rayhip_hit *out_hits,
                                              int *out_count);
/* Ref::IntersectScene closest hit (CoreRef.cpp:3041-3158): rays/hits are in-out host arrays */
RAYHIP_API int rayhip_k_intersect_closest(rayhip_ctx *ctx, const rayhip_camera *cam, rayhip_ray *rays,
                                          rayhip_hit *hits, int count, int iteration, uint32_t flags /* COUNT_TRAVERSAL: instrumented BVH2 kernel + counters; 0: the kernel rayhip_render uses */,
                                          rayhip_trav_counters *out_counters /* may be NULL */);
/* Ref::IntersectScene(shadow_ray_t) (CoreRef.cpp:3160-3262): out_rc[count][4] visibility * colour */
RAYHIP_API int rayhip_k_intersect_shadow(rayhip_ctx *ctx, const rayhip_camera *cam,
                                         const rayhip_shadow_ray *rays, int count, int iteration,
                                         float *out_rc, rayhip_trav_counters *out_counters /* may be NULL */);
/* Ref::ShadePrimary (bounce 0) / Ref::ShadeSecondary (bounce >= 1) (ShadeRef.cpp:1654-1738, i.e. ShadeSurface
 * :1174-1652 per ray) on `count` host (ray, hit) pairs -- the oracle's twin is oracle/ref_shim.cpp: refk_shade.
 * inout_color: the per-iteration radiance image [h][w][4] (bounce 0 assigns the pixels of the rays, later bounces add).
 * out_secondary / out_shadow: room for `count` rays each; the emitted rays come back in no particular order (every
 * pixel emits at most one of each: compare after sorting by xy).  Runs the kernels rayhip_render launches; bounce 0
 * also blends the aux images of the context like a first-bounce shade does. */
RAYHIP_API int rayhip_k_shade(rayhip_ctx *ctx, const rayhip_camera *cam, int bounce, int iteration, const rayhip_ray *rays,
                              const rayhip_hit *hits, int count, float *inout_color, rayhip_ray *out_secondary,
                              int *out_secondary_count, rayhip_shadow_ray *out_shadow, int *out_shadow_count);
/* Ref::get_scrambled_2d_rand (CoreRef.cpp:1418-1427) for `count` (dim,seed,sample) triples */
RAYHIP_API int rayhip_k_scrambled_rand(rayhip_ctx *ctx, const uint32_t *dims, const uint32_t *seeds,
                                       const int32_t *samples, int count, float *out_xy);

#ifdef __cplusplus
}
#endif

#endif /* RAYHIP_H */
