// ray_capi.cpp -- C view of the Ray public API (see ray_capi.h).  Pure forwarding.
#include "ray_capi.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "Ray.h"

#ifdef RAY_CAPI_WITH_HIP
#include "RendererHIP.h"
#include "internal/Core.h"
#endif
#ifdef RAY_CAPI_WITH_CPU
#include "internal/RendererAVX.h"
#include "internal/RendererAVX2.h"
#include "internal/RendererAVX512.h"
#include "internal/RendererRef.h"
#include "internal/RendererSSE41.h"
#include "internal/simd/detect.h"
#endif

namespace {
thread_local std::string g_err;

class CollectLog final : public Ray::ILog {
    bool verbose_;

  public:
    std::string last_error;
    explicit CollectLog(bool verbose) : verbose_(verbose) {}
    void Info(const char *fmt, ...) override {
        if (verbose_) {
            va_list vl;
            va_start(vl, fmt);
            vprintf(fmt, vl);
            va_end(vl);
            putc('\n', stdout);
        }
    }
    void Warning(const char *fmt, ...) override {
        if (verbose_) {
            va_list vl;
            va_start(vl, fmt);
            vprintf(fmt, vl);
            va_end(vl);
            putc('\n', stdout);
        }
    }
    void Error(const char *fmt, ...) override {
        char buf[1024];
        va_list vl;
        va_start(vl, fmt);
        vsnprintf(buf, sizeof(buf), fmt, vl);
        va_end(vl);
        last_error = buf;
        fprintf(stderr, "Ray error: %s\n", buf);
    }
};

template <typename H> H to_handle(ray_handle h) { return H{uint32_t(h & 0xffffffffull), uint32_t(h >> 32)}; }
template <typename H> ray_handle from_handle(H h) { return (uint64_t(h._block) << 32) | uint64_t(h._index); }
} // namespace

struct ray_renderer {
    std::unique_ptr<CollectLog> log;
    std::unique_ptr<Ray::RendererBase> r;
    std::string type_name;
};
struct ray_scene {
    std::unique_ptr<Ray::SceneBase> s;
};
struct ray_region {
    Ray::RegionContext ctx;
};

extern "C" {

const char *ray_last_error(void) { return g_err.c_str(); }

void ray_default_shading_node(ray_shading_node_desc *d) {
    const Ray::shading_node_desc_t s = {};
    memset(d, 0, sizeof(*d));
    d->type = 0;
    d->base_color[0] = d->base_color[1] = d->base_color[2] = 1.0f;
    d->base_texture = d->normal_map = d->roughness_texture = d->metallic_texture = RAY_INVALID_HANDLE;
    d->mix_materials[0] = RAY_INVALID_HANDLE;
    d->mix_materials[1] = 0; // C++ aggregate init {InvalidMaterialHandle} zero-fills the second element
    d->normal_map_intensity = s.normal_map_intensity;
    d->strength = s.strength, d->fresnel = s.fresnel, d->ior = s.ior;
}
void ray_default_principled(ray_principled_mat_desc *d) {
    const Ray::principled_mat_desc_t s = {};
    memset(d, 0, sizeof(*d));
    memcpy(d->base_color, s.base_color, 12);
    d->base_texture = d->metallic_texture = d->specular_texture = d->roughness_texture = d->emission_texture =
        d->alpha_texture = d->normal_map = RAY_INVALID_HANDLE;
    d->metallic = s.metallic, d->specular = s.specular, d->specular_tint = s.specular_tint, d->roughness = s.roughness;
    d->anisotropic = s.anisotropic, d->anisotropic_rotation = s.anisotropic_rotation, d->sheen = s.sheen;
    d->sheen_tint = s.sheen_tint, d->clearcoat = s.clearcoat, d->clearcoat_roughness = s.clearcoat_roughness;
    d->ior = s.ior, d->transmission = s.transmission, d->transmission_roughness = s.transmission_roughness;
    memcpy(d->emission_color, s.emission_color, 12);
    d->emission_strength = s.emission_strength, d->alpha = s.alpha, d->normal_map_intensity = s.normal_map_intensity;
    d->importance_sample = s.importance_sample;
}
void ray_default_light(ray_light_desc *d, uint32_t kind) {
    memset(d, 0, sizeof(*d));
    d->kind = kind;
    d->color[0] = d->color[1] = d->color[2] = 1.0f;
    d->direction[1] = -1.0f;
    d->radius = 1.0f;
    d->spot_size = 45.0f, d->spot_blend = 0.15f;
    d->width = d->height = 1.0f;
    d->multiple_importance = d->cast_shadow = d->diffuse_visibility = d->specular_visibility = d->refraction_visibility = 1;
    d->xform[0] = d->xform[5] = d->xform[10] = d->xform[15] = 1.0f;
}
void ray_default_camera(ray_camera_desc *d) {
    const Ray::camera_desc_t s = {};
    memset(d, 0, sizeof(*d));
    d->type = uint32_t(s.type), d->filter = uint32_t(s.filter), d->view_transform = uint32_t(s.view_transform);
    d->ltype = uint32_t(s.ltype);
    d->filter_width = s.filter_width;
    d->exposure = s.exposure, d->fov = s.fov, d->gamma = s.gamma, d->sensor_height = s.sensor_height;
    d->focus_distance = s.focus_distance, d->focal_length = s.focal_length, d->fstop = s.fstop;
    d->lens_rotation = s.lens_rotation, d->lens_ratio = s.lens_ratio, d->lens_blades = s.lens_blades;
    d->clip_start = s.clip_start, d->clip_end = s.clip_end;
    d->mi_index = s.mi_index, d->uv_index = s.uv_index;
    d->max_diff_depth = s.max_diff_depth, d->max_spec_depth = s.max_spec_depth, d->max_refr_depth = s.max_refr_depth;
    d->max_transp_depth = s.max_transp_depth, d->max_total_depth = s.max_total_depth;
    d->min_total_depth = s.min_total_depth, d->min_transp_depth = s.min_transp_depth;
    d->clamp_direct = s.clamp_direct, d->clamp_indirect = s.clamp_indirect;
    d->min_samples = s.min_samples, d->variance_threshold = s.variance_threshold, d->regularize_alpha = s.regularize_alpha;
}
void ray_default_env(ray_env_desc *d) {
    memset(d, 0, sizeof(*d));
    d->struct_size = uint32_t(sizeof(*d));
    d->env_map = d->back_map = RAY_INVALID_HANDLE;
    d->importance_sample = 1;
    const Ray::environment_desc_t e;
    d->envmap_resolution = e.envmap_resolution;
    d->clouds_density = e.atmosphere.clouds_density, d->cirrus_clouds_amount = e.atmosphere.cirrus_clouds_amount;
    d->stars_brightness = e.atmosphere.stars_brightness, d->moon_radius = e.atmosphere.moon_radius;
    d->clouds_offset_x = e.atmosphere.clouds_offset_x, d->clouds_offset_z = e.atmosphere.clouds_offset_z;
}

ray_renderer *ray_renderer_create(const char *type_name, int w, int h, int use_tex_compression, int verbose) {
    auto out = std::make_unique<ray_renderer>();
    out->log = std::make_unique<CollectLog>(verbose != 0);
    Ray::settings_t s;
    s.w = w, s.h = h;
    s.use_tex_compression = use_tex_compression != 0;
    const std::string name = type_name ? type_name : "";
    try {
        Ray::RendererBase *r = nullptr;
#ifdef RAY_CAPI_WITH_HIP
        if (name == "HIP") {
            r = Ray::Hip::CreateRenderer(s, out->log.get());
        }
#endif
#ifdef RAY_CAPI_WITH_CPU
        // the per-backend factories Ray::CreateRenderer dispatches to (Ray.cpp:74-122), without the fallback chain
        const Ray::CpuFeatures f = Ray::GetCpuFeatures();
        if (name == "REF") {
            r = Ray::Ref::CreateRenderer(s, out->log.get());
        } else if (name == "SSE41" && f.sse41_supported) {
            r = Ray::Sse41::CreateRenderer(s, out->log.get());
        } else if (name == "AVX" && f.avx_supported) {
            r = Ray::Avx::CreateRenderer(s, out->log.get());
        } else if (name == "AVX2" && f.avx2_supported) {
            r = Ray::Avx2::CreateRenderer(s, out->log.get());
        } else if (name == "AVX512" && f.avx512_supported) {
            r = Ray::Avx512::CreateRenderer(s, out->log.get());
        }
#endif
        if (!r) {
            g_err = "renderer type '" + name + "' is not available in this library / on this machine";
            return nullptr;
        }
        out->r.reset(r);
        out->type_name = name;
    } catch (std::exception &e) {
        g_err = std::string("failed to create '") + name + "' renderer: " + e.what();
        return nullptr;
    }
    return out.release();
}
void ray_renderer_destroy(ray_renderer *r) { delete r; }
int ray_renderer_type_name(ray_renderer *r, char *buf, int cap) {
    snprintf(buf, size_t(cap), "%s", r->type_name.c_str());
    return 0;
}
int ray_renderer_device_name(ray_renderer *r, char *buf, int cap) {
    const std::string_view n = r->r->device_name();
    snprintf(buf, size_t(cap), "%.*s", int(n.size()), n.data());
    return 0;
}
void ray_renderer_size(ray_renderer *r, int out_wh[2]) {
    const auto sz = r->r->size();
    out_wh[0] = sz.first, out_wh[1] = sz.second;
}
void ray_renderer_resize(ray_renderer *r, int w, int h) { r->r->Resize(w, h); }
void ray_renderer_clear(ray_renderer *r, const float rgba[4]) {
    r->r->Clear(Ray::color_rgba_t{rgba[0], rgba[1], rgba[2], rgba[3]});
}
ray_scene *ray_renderer_create_scene(ray_renderer *r) {
    auto out = std::make_unique<ray_scene>();
    out->s.reset(r->r->CreateScene());
    return out.release();
}
void ray_renderer_render(ray_renderer *r, ray_scene *s, ray_region *region) { r->r->RenderScene(*s->s, region->ctx); }
void ray_renderer_denoise(ray_renderer *r, ray_region *region) { r->r->DenoiseImage(region->ctx); }
int ray_renderer_init_unet(ray_renderer *r) { return r->r->InitUNetFilter(false, Ray::parallel_for_serial).pass_count; }
void ray_renderer_denoise_unet(ray_renderer *r, int pass, ray_region *region) { r->r->DenoiseImage(pass, region->ctx); }
int ray_renderer_get_pixels(ray_renderer *r, int which, float *dst) {
    Ray::color_data_rgba_t px = {};
    switch (which) {
    case 0:
        px = r->r->get_pixels_ref();
        break;
    case 1:
        px = r->r->get_raw_pixels_ref();
        break;
    case 2:
        px = r->r->get_aux_pixels_ref(Ray::eAUXBuffer::BaseColor);
        break;
    case 3:
        px = r->r->get_aux_pixels_ref(Ray::eAUXBuffer::DepthNormals);
        break;
    default:
        g_err = "bad buffer id";
        return 1;
    }
    if (!px.ptr) {
        g_err = "buffer not available";
        return 1;
    }
    const auto sz = r->r->size();
    for (int y = 0; y < sz.second; ++y) {
        memcpy(dst + size_t(y) * sz.first * 4, px.ptr + size_t(y) * px.pitch, size_t(sz.first) * 16);
    }
    return 0;
}
void ray_renderer_get_stats(ray_renderer *r, ray_stats *st) {
    Ray::RendererBase::stats_t s = {};
    r->r->GetStats(s);
    static_assert(sizeof(s) == sizeof(*st), "stats layout");
    memcpy(st, &s, sizeof(s));
}
void ray_renderer_reset_stats(ray_renderer *r) { r->r->ResetStats(); }

double ray_renderer_render_tiled_mt(ray_renderer *r, ray_scene *s, int tile, int spp, int threads) {
    return ray_renderer_render_tiled_from(r, s, tile, 0, spp, threads);
}

double ray_renderer_render_tiled_from(ray_renderer *r, ray_scene *s, int tile, int iterations_done, int spp, int threads) {
    const auto sz = r->r->size();
    std::vector<Ray::RegionContext> regions;
    for (int y = 0; y < sz.second; y += tile) {
        for (int x = 0; x < sz.first; x += tile) {
            regions.emplace_back(Ray::rect_t{x, y, std::min(tile, sz.first - x), std::min(tile, sz.second - y)});
            regions.back().iteration = iterations_done; // continue a progressive render: RenderScene increments first
        }
    }
    std::atomic_int next{0};
    const auto t0 = std::chrono::high_resolution_clock::now();
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= int(regions.size())) {
                break;
            }
            for (int k = 0; k < spp; ++k) {
                r->r->RenderScene(*s->s, regions[i]);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) {
        pool.emplace_back(worker);
    }
    worker();
    for (auto &t : pool) {
        t.join();
    }
    return std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
}

ray_region *ray_region_create(int x, int y, int w, int h) { return new ray_region{Ray::RegionContext{Ray::rect_t{x, y, w, h}}}; }
void ray_region_destroy(ray_region *g) { delete g; }
int ray_region_iteration(ray_region *g) { return g->ctx.iteration; }
void ray_region_set_iteration(ray_region *g, int it) { g->ctx.iteration = it; }

void ray_scene_destroy(ray_scene *s) { delete s; }

void ray_scene_set_environment(ray_scene *s, const ray_env_desc *d) {
    if (d->struct_size != sizeof(ray_env_desc)) { // (a caller built against the shorter struct of round 3)
        g_err = "ray_env_desc::struct_size does not match this library's struct: start from ray_default_env of THIS ray_capi.h";
        return;
    }
    Ray::environment_desc_t e;
    memcpy(e.env_col, d->env_col, 12);
    e.env_map = to_handle<Ray::TextureHandle>(d->env_map);
    memcpy(e.back_col, d->back_col, 12);
    e.back_map = to_handle<Ray::TextureHandle>(d->back_map);
    e.env_map_rotation = d->env_map_rotation, e.back_map_rotation = d->back_map_rotation;
    e.importance_sample = d->importance_sample != 0;
    e.envmap_resolution = d->envmap_resolution;
    e.atmosphere.clouds_density = d->clouds_density, e.atmosphere.cirrus_clouds_amount = d->cirrus_clouds_amount;
    e.atmosphere.stars_brightness = d->stars_brightness, e.atmosphere.moon_radius = d->moon_radius;
    e.atmosphere.clouds_offset_x = d->clouds_offset_x, e.atmosphere.clouds_offset_z = d->clouds_offset_z;
    s->s->SetEnvironment(e);
}

ray_handle ray_scene_add_texture(ray_scene *s, const ray_tex_desc *d) {
    Ray::tex_desc_t t;
    t.format = Ray::eTextureFormat(d->format);
    t.data = Ray::Span<const uint8_t>(d->data, size_t(d->data_size));
    t.w = d->w, t.h = d->h;
    t.is_srgb = d->is_srgb != 0, t.is_normalmap = d->is_normalmap != 0, t.is_YCoCg = d->is_YCoCg != 0;
    t.force_no_compression = d->force_no_compression != 0, t.generate_mipmaps = d->generate_mipmaps != 0;
    t.reconstruct_z = d->reconstruct_z != 0;
    t.mips_count = d->mips_count > 0 ? d->mips_count : 1;
    t.convention = d->convention == 1 ? Ray::eTextureConvention::DX : Ray::eTextureConvention::OGL;
    return from_handle(s->s->AddTexture(t));
}

ray_handle ray_scene_add_material_node(ray_scene *s, const ray_shading_node_desc *d) {
    Ray::shading_node_desc_t m;
    m.type = Ray::eShadingNode(d->type);
    memcpy(m.base_color, d->base_color, 12);
    m.base_texture = to_handle<Ray::TextureHandle>(d->base_texture);
    m.normal_map = to_handle<Ray::TextureHandle>(d->normal_map);
    m.normal_map_intensity = d->normal_map_intensity;
    m.mix_materials[0] = to_handle<Ray::MaterialHandle>(d->mix_materials[0]);
    m.mix_materials[1] = to_handle<Ray::MaterialHandle>(d->mix_materials[1]);
    m.roughness = d->roughness;
    m.roughness_texture = to_handle<Ray::TextureHandle>(d->roughness_texture);
    m.anisotropic = d->anisotropic, m.anisotropic_rotation = d->anisotropic_rotation, m.sheen = d->sheen;
    m.specular = d->specular, m.strength = d->strength, m.fresnel = d->fresnel, m.ior = d->ior, m.tint = d->tint;
    m.metallic_texture = to_handle<Ray::TextureHandle>(d->metallic_texture);
    m.importance_sample = d->importance_sample != 0, m.mix_add = d->mix_add != 0;
    return from_handle(s->s->AddMaterial(m));
}

ray_handle ray_scene_add_material_principled(ray_scene *s, const ray_principled_mat_desc *d) {
    Ray::principled_mat_desc_t m;
    memcpy(m.base_color, d->base_color, 12);
    m.base_texture = to_handle<Ray::TextureHandle>(d->base_texture);
    m.metallic = d->metallic;
    m.metallic_texture = to_handle<Ray::TextureHandle>(d->metallic_texture);
    m.specular = d->specular;
    m.specular_texture = to_handle<Ray::TextureHandle>(d->specular_texture);
    m.specular_tint = d->specular_tint, m.roughness = d->roughness;
    m.roughness_texture = to_handle<Ray::TextureHandle>(d->roughness_texture);
    m.anisotropic = d->anisotropic, m.anisotropic_rotation = d->anisotropic_rotation;
    m.sheen = d->sheen, m.sheen_tint = d->sheen_tint, m.clearcoat = d->clearcoat;
    m.clearcoat_roughness = d->clearcoat_roughness, m.ior = d->ior, m.transmission = d->transmission;
    m.transmission_roughness = d->transmission_roughness;
    memcpy(m.emission_color, d->emission_color, 12);
    m.emission_texture = to_handle<Ray::TextureHandle>(d->emission_texture);
    m.emission_strength = d->emission_strength, m.alpha = d->alpha;
    m.alpha_texture = to_handle<Ray::TextureHandle>(d->alpha_texture);
    m.normal_map = to_handle<Ray::TextureHandle>(d->normal_map);
    m.normal_map_intensity = d->normal_map_intensity;
    m.importance_sample = d->importance_sample != 0;
    return from_handle(s->s->AddMaterial(m));
}

ray_handle ray_scene_add_mesh(ray_scene *s, const ray_mesh_desc *d) {
    Ray::mesh_desc_t m;
    m.prim_type = Ray::ePrimType::TriangleList;
    const Ray::Span<const float> attrs(d->attrs, size_t(d->attrs_count));
    m.vtx_positions = {attrs, d->pos_offset, d->stride};
    m.vtx_normals = {attrs, d->nrm_offset, d->stride};
    if (d->bnm_offset >= 0) {
        m.vtx_binormals = {attrs, d->bnm_offset, d->stride};
    }
    m.vtx_uvs = {attrs, d->uv_offset, d->stride};
    m.vtx_indices = Ray::Span<const uint32_t>(d->indices, size_t(d->indices_count));
    m.base_vertex = d->base_vertex;
    std::vector<Ray::mat_group_desc_t> groups;
    for (uint32_t i = 0; i < d->groups_count; ++i) {
        const ray_mat_group_desc &g = d->groups[i];
        groups.emplace_back(to_handle<Ray::MaterialHandle>(g.front_mat), to_handle<Ray::MaterialHandle>(g.back_mat),
                            size_t(g.vtx_start), size_t(g.vtx_count));
    }
    m.groups = groups;
    m.allow_spatial_splits = d->allow_spatial_splits != 0;
    m.use_fast_bvh_build = d->use_fast_bvh_build != 0;
    return from_handle(s->s->AddMesh(m));
}

ray_handle ray_scene_add_mesh_instance(ray_scene *s, ray_handle mesh, const float xform[16]) {
    return from_handle(s->s->AddMeshInstance(to_handle<Ray::MeshHandle>(mesh), xform));
}

ray_handle ray_scene_add_mesh_instance_vis(ray_scene *s, ray_handle mesh, const float xform[16], unsigned visibility) {
    Ray::mesh_instance_desc_t mi;
    mi.xform = xform;
    mi.mesh = to_handle<Ray::MeshHandle>(mesh);
    mi.camera_visibility = (visibility & 1u) != 0, mi.diffuse_visibility = (visibility & 2u) != 0;
    mi.specular_visibility = (visibility & 4u) != 0, mi.refraction_visibility = (visibility & 8u) != 0;
    mi.shadow_visibility = (visibility & 16u) != 0;
    return from_handle(s->s->AddMeshInstance(mi));
}

void ray_scene_set_mesh_instance_transform(ray_scene *s, ray_handle mi, const float xform[16]) {
    s->s->SetMeshInstanceTransform(to_handle<Ray::MeshInstanceHandle>(mi), xform);
}
void ray_scene_remove_mesh_instance(ray_scene *s, ray_handle mi) { s->s->RemoveMeshInstance(to_handle<Ray::MeshInstanceHandle>(mi)); }
void ray_scene_remove_mesh(ray_scene *s, ray_handle mesh) { s->s->RemoveMesh(to_handle<Ray::MeshHandle>(mesh)); }
void ray_scene_remove_light(ray_scene *s, ray_handle light) { s->s->RemoveLight(to_handle<Ray::LightHandle>(light)); }

ray_handle ray_scene_add_light(ray_scene *s, const ray_light_desc *d) {
#define COMMON(l)                                                                                                      \
    memcpy(l.color, d->color, 12);                                                                                     \
    l.multiple_importance = d->multiple_importance != 0, l.cast_shadow = d->cast_shadow != 0;                          \
    l.diffuse_visibility = d->diffuse_visibility != 0, l.specular_visibility = d->specular_visibility != 0;            \
    l.refraction_visibility = d->refraction_visibility != 0;
    switch (d->kind) {
    case 0: {
        Ray::directional_light_desc_t l;
        COMMON(l)
        memcpy(l.direction, d->direction, 12);
        l.angle = d->angle;
        return from_handle(s->s->AddLight(l));
    }
    case 1: {
        Ray::sphere_light_desc_t l;
        COMMON(l)
        memcpy(l.position, d->position, 12);
        l.radius = d->radius;
        return from_handle(s->s->AddLight(l));
    }
    case 2: {
        Ray::spot_light_desc_t l;
        COMMON(l)
        memcpy(l.position, d->position, 12);
        memcpy(l.direction, d->direction, 12);
        l.spot_size = d->spot_size, l.spot_blend = d->spot_blend, l.radius = d->radius;
        return from_handle(s->s->AddLight(l));
    }
    case 3: {
        Ray::rect_light_desc_t l;
        COMMON(l)
        l.width = d->width, l.height = d->height;
        l.doublesided = d->doublesided != 0, l.sky_portal = d->sky_portal != 0;
        return from_handle(s->s->AddLight(l, d->xform));
    }
    case 4: {
        Ray::disk_light_desc_t l;
        COMMON(l)
        l.size_x = d->width, l.size_y = d->height;
        l.doublesided = d->doublesided != 0, l.sky_portal = d->sky_portal != 0;
        return from_handle(s->s->AddLight(l, d->xform));
    }
    case 5: {
        Ray::line_light_desc_t l;
        COMMON(l)
        l.radius = d->radius, l.height = d->height;
        l.sky_portal = d->sky_portal != 0;
        return from_handle(s->s->AddLight(l, d->xform));
    }
    default:
        g_err = "bad light kind";
        return RAY_INVALID_HANDLE;
    }
#undef COMMON
}

ray_handle ray_scene_add_camera(ray_scene *s, const ray_camera_desc *d) {
    Ray::camera_desc_t c;
    c.type = Ray::eCamType(d->type), c.filter = Ray::ePixelFilter(d->filter);
    c.view_transform = Ray::eViewTransform(d->view_transform), c.ltype = Ray::eLensUnits(d->ltype);
    c.filter_width = d->filter_width;
    memcpy(c.origin, d->origin, 12), memcpy(c.fwd, d->fwd, 12), memcpy(c.up, d->up, 12), memcpy(c.shift, d->shift, 8);
    c.exposure = d->exposure, c.fov = d->fov, c.gamma = d->gamma, c.sensor_height = d->sensor_height;
    c.focus_distance = d->focus_distance, c.focal_length = d->focal_length, c.fstop = d->fstop;
    c.lens_rotation = d->lens_rotation, c.lens_ratio = d->lens_ratio, c.lens_blades = d->lens_blades;
    c.clip_start = d->clip_start, c.clip_end = d->clip_end;
    c.mi_index = d->mi_index, c.uv_index = d->uv_index;
    c.lighting_only = d->lighting_only != 0, c.skip_direct_lighting = d->skip_direct_lighting != 0;
    c.skip_indirect_lighting = d->skip_indirect_lighting != 0, c.no_background = d->no_background != 0;
    c.output_sh = d->output_sh != 0;
    c.max_diff_depth = uint8_t(d->max_diff_depth), c.max_spec_depth = uint8_t(d->max_spec_depth);
    c.max_refr_depth = uint8_t(d->max_refr_depth), c.max_transp_depth = uint8_t(d->max_transp_depth);
    c.max_total_depth = uint8_t(d->max_total_depth), c.min_total_depth = uint8_t(d->min_total_depth);
    c.min_transp_depth = uint8_t(d->min_transp_depth);
    c.clamp_direct = d->clamp_direct, c.clamp_indirect = d->clamp_indirect;
    c.min_samples = d->min_samples, c.variance_threshold = d->variance_threshold, c.regularize_alpha = d->regularize_alpha;
    return from_handle(s->s->AddCamera(c));
}
void ray_scene_set_current_cam(ray_scene *s, ray_handle cam) { s->s->set_current_cam(to_handle<Ray::CameraHandle>(cam)); }
void ray_scene_finalize(ray_scene *s) { s->s->Finalize(); }
uint32_t ray_scene_triangle_count(ray_scene *s) { return s->s->triangle_count(); }
uint32_t ray_scene_node_count(ray_scene *s) { return s->s->node_count(); }

#ifdef RAY_CAPI_WITH_HIP
ray_scene *ray_hip_create_scene_ex(int verbose, int use_tex_compression) {
    static CollectLog quiet(false), loud(true);
    auto out = std::make_unique<ray_scene>();
    out->s.reset(Ray::Hip::CreateScene(verbose ? &loud : &quiet, use_tex_compression != 0));
    return out.release();
}
ray_scene *ray_hip_create_scene(int verbose) { return ray_hip_create_scene_ex(verbose, 0); }
int ray_hip_export_scene(ray_scene *s, void **out_blob, uint64_t *out_size) {
    try {
        const std::vector<uint8_t> blob = Ray::Hip::ExportSceneBlob(*s->s);
        void *p = nullptr;
        if (posix_memalign(&p, 64, blob.size() ? blob.size() : 64) != 0) {
            g_err = "out of memory";
            return 1;
        }
        memcpy(p, blob.data(), blob.size());
        *out_blob = p;
        *out_size = blob.size();
        return 0;
    } catch (std::exception &e) {
        g_err = e.what();
        return 1;
    }
}
const char *ray_hip_sky_baked_on(ray_scene *s) { return Ray::Hip::SkyBakedOn(*s->s); }
void ray_hip_free(void *p) { free(p); }
void ray_hip_pmj_table(const uint32_t **out_ptr, uint32_t *out_count) {
    *out_ptr = Ray::__pmj02_samples;
    *out_count = uint32_t(Ray::__pmj02_dims_count) * 2u * uint32_t(Ray::__pmj02_sample_count);
}
#endif

} // extern "C"
