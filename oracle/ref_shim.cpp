// ref_shim.cpp -- oracle-only entry points into the real reference.  TEST INFRASTRUCTURE (see ref_shim.h).
#include "ref_shim.h"

#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>

#include "Ray.h"
#include "internal/CDFUtils.h"
#include "internal/CoreRef.h"
#include "internal/SceneCPU.h"
#include "internal/ShadeRef.h"

#include "../ray_amd/csrc/scene_blob.h"
#include "../ray_amd/host/scene_export.h"

// ray_capi.cpp keeps these private; the shim needs the C++ object behind a ray_scene
struct ray_scene {
    std::unique_ptr<Ray::SceneBase> s;
};

namespace Ray {
// the reference's precomputed view-transform tables (internal/TonemapRef.cpp:4-26), indexed by eViewTransform
extern const int LUT_DIMS;
extern const uint32_t *transform_luts[];
} // namespace Ray

namespace {
using namespace Ray;

static_assert(sizeof(rayhip_ray) == sizeof(Ref::ray_data_t), "layout");
static_assert(sizeof(rayhip_shadow_ray) == sizeof(Ref::shadow_ray_t), "layout");
static_assert(sizeof(rayhip_hit) == sizeof(Ref::hit_data_t), "layout");

Cpu::Scene &cpu_scene(ray_scene *s) {
    auto *p = dynamic_cast<Cpu::Scene *>(s->s.get());
    if (!p) {
        throw std::runtime_error("not a CPU scene");
    }
    return *p;
}

// builds what RendererCPU.h:390-413 builds
class OracleAccess : public Cpu::Scene {
  public:
    struct Bundle {
        cache_grid_params_t cache_grid_params;
        std::unique_ptr<scene_data_t> sc;
        const Cpu::TexStorageBase *const *textures;
        uint32_t tlas_root;
        const camera_t *cam;
    };
    static void Make(const Cpu::Scene &_s, Bundle &b) {
        const auto &s = static_cast<const OracleAccess &>(_s);
        b.cam = &s.cams_[s.current_cam_._index];
        b.sc.reset(new scene_data_t{s.env_,
                                    s.mesh_instances_.empty() ? nullptr : &s.mesh_instances_[0],
                                    s.meshes_.empty() ? nullptr : &s.meshes_[0],
                                    s.vtx_indices_.empty() ? nullptr : &s.vtx_indices_[0],
                                    s.vertices_.empty() ? nullptr : &s.vertices_[0],
                                    s.nodes_.empty() ? nullptr : &s.nodes_[0],
                                    s.wnodes_.empty() ? nullptr : &s.wnodes_[0],
                                    s.tris_.empty() ? nullptr : &s.tris_[0],
                                    s.tri_indices_.empty() ? nullptr : &s.tri_indices_[0],
                                    s.mtris_.data(),
                                    s.tri_materials_.empty() ? nullptr : &s.tri_materials_[0],
                                    s.materials_.empty() ? nullptr : &s.materials_[0],
                                    {s.lights_.data(), s.lights_.capacity()},
                                    {s.li_indices_},
                                    {s.dir_lights_},
                                    s.visible_lights_count_,
                                    s.blocker_lights_count_,
                                    {s.light_nodes_},
                                    {s.light_cwnodes_},
                                    {s.sky_transmittance_lut_},
                                    {s.sky_multiscatter_lut_},
                                    b.cache_grid_params,
                                    {s.spatial_cache_entries_},
                                    {s.spatial_cache_voxels_prev_}});
        b.textures = s.tex_storages_;
        b.tlas_root = s.tlas_root_;
    }
};

std::vector<float> make_filter_table(ePixelFilter filter, float filter_width) {
    // RendererCPU.h:1234-1258
    float (*filter_func)(float v, float width) = nullptr;
    switch (filter) {
    case ePixelFilter::Box:
        filter_func = filter_box;
        filter_width = 1.0f;
        break;
    case ePixelFilter::Gaussian:
        filter_func = filter_gaussian;
        filter_width *= 3.0f;
        break;
    case ePixelFilter::BlackmanHarris:
        filter_func = filter_blackman_harris;
        filter_width *= 2.0f;
        break;
    default:
        throw std::runtime_error("unknown filter");
    }
    return Ray::CDFInverted(FILTER_TABLE_SIZE, 0.0f, filter_width * 0.5f,
                            std::bind(filter_func, std::placeholders::_1, filter_width), true /* make_symmetric */);
}
} // namespace

extern "C" {

void refk_pmj_table(const uint32_t **out_ptr, uint32_t *out_count) {
    *out_ptr = Ray::__pmj02_samples;
    *out_count = uint32_t(Ray::__pmj02_dims_count) * 2u * uint32_t(Ray::__pmj02_sample_count);
}

void refk_filter_table(uint32_t pixel_filter, float filter_width, float *out_table) {
    const std::vector<float> t = make_filter_table(ePixelFilter(pixel_filter), filter_width);
    memcpy(out_table, t.data(), t.size() * sizeof(float));
}

int refk_export_scene(ray_scene *s, void **out_blob, size_t *out_size) {
    try {
        Cpu::Scene &cs = cpu_scene(s);
        Hip::FlatScene flat;
        Hip::SceneAccess::Export(cs, flat);
        const camera_t &cam = Hip::SceneAccess::CurrentCamera(cs);
        rayhip_camera rc;
        memcpy(&rc, &cam, sizeof(rc));
        const std::vector<float> ft = make_filter_table(cam.filter, cam.filter_width);
        const bool lut = rc.view_transform != 0; // the blob carries the table of its camera's view transform
        const std::vector<uint8_t> blob = rayhip_blob::serialize(flat.desc, rc, ft.data(), int(ft.size()),
                                                                 lut ? transform_luts[rc.view_transform] : nullptr, lut ? LUT_DIMS : 0);
        void *p = nullptr;
        if (posix_memalign(&p, 64, blob.size() ? blob.size() : 64) != 0) {
            return 1;
        }
        memcpy(p, blob.data(), blob.size());
        *out_blob = p;
        *out_size = blob.size();
        return 0;
    } catch (std::exception &e) {
        fprintf(stderr, "refk_export_scene: %s\n", e.what());
        return 1;
    }
}
void refk_free(void *p) { free(p); }

int refk_generate_primary_rays(ray_scene *s, int w, int h, const int rect[4], int iteration, rayhip_ray *out_rays,
                               rayhip_hit *out_hits, int *out_count) {
    OracleAccess::Bundle b;
    OracleAccess::Make(cpu_scene(s), b);
    const std::vector<float> ft = make_filter_table(b.cam->filter, b.cam->filter_width);
    aligned_vector<Ref::ray_data_t> rays;
    aligned_vector<Ref::hit_data_t> hits;
    const uint32_t rand_seed = Ref::hash((iteration - 1) / RAND_SAMPLES_COUNT);
    Ref::GeneratePrimaryRays(*b.cam, rect_t{rect[0], rect[1], rect[2], rect[3]}, w, h, __pmj02_samples, rand_seed, ft.data(),
                             iteration, nullptr, rays, hits);
    memcpy(out_rays, rays.data(), rays.size() * sizeof(Ref::ray_data_t));
    memcpy(out_hits, hits.data(), hits.size() * sizeof(Ref::hit_data_t));
    *out_count = int(rays.size());
    return 0;
}

int refk_intersect_closest(ray_scene *s, rayhip_ray *rays, rayhip_hit *hits, int count, int iteration) {
    OracleAccess::Bundle b;
    OracleAccess::Make(cpu_scene(s), b);
    const uint32_t rand_seed = Ref::hash((iteration - 1) / RAND_SAMPLES_COUNT);
    Ref::IntersectScene(Span<Ref::ray_data_t>(reinterpret_cast<Ref::ray_data_t *>(rays), count),
                        b.cam->pass_settings.min_transp_depth, b.cam->pass_settings.max_transp_depth, __pmj02_samples,
                        rand_seed, iteration, *b.sc, b.tlas_root, b.textures,
                        Span<Ref::hit_data_t>(reinterpret_cast<Ref::hit_data_t *>(hits), count));
    return 0;
}

int refk_intersect_shadow(ray_scene *s, const rayhip_shadow_ray *rays, int count, int iteration, float *out_rc) {
    OracleAccess::Bundle b;
    OracleAccess::Make(cpu_scene(s), b);
    const uint32_t rand_seed = Ref::hash((iteration - 1) / RAND_SAMPLES_COUNT);
    for (int i = 0; i < count; ++i) {
        const Ref::fvec4 rc =
            Ref::IntersectScene(*reinterpret_cast<const Ref::shadow_ray_t *>(&rays[i]), b.cam->pass_settings.max_transp_depth,
                                *b.sc, b.tlas_root, __pmj02_samples, rand_seed, iteration, b.textures);
        out_rc[4 * i + 0] = rc.get<0>(), out_rc[4 * i + 1] = rc.get<1>(), out_rc[4 * i + 2] = rc.get<2>();
        out_rc[4 * i + 3] = 0.0f;
    }
    return 0;
}

void refk_scrambled_rand(const uint32_t *dims, const uint32_t *seeds, const int32_t *samples, int count, float *out_xy) {
    for (int i = 0; i < count; ++i) {
        const Ref::fvec2 r = Ref::get_scrambled_2d_rand(dims[i], seeds[i], samples[i], __pmj02_samples);
        out_xy[2 * i + 0] = r.get<0>(), out_xy[2 * i + 1] = r.get<1>();
    }
}

int refk_shade(ray_scene *s, int w, int h, int bounce, int iteration, const rayhip_ray *rays, const rayhip_hit *hits, int count,
               float *inout_color, rayhip_ray *out_secondary, int *out_secondary_count, rayhip_shadow_ray *out_shadow,
               int *out_shadow_count) {
    OracleAccess::Bundle b;
    OracleAccess::Make(cpu_scene(s), b);
    const uint32_t rand_seed = Ref::hash((iteration - 1) / RAND_SAMPLES_COUNT);
    const pass_settings_t &ps = b.cam->pass_settings;
    std::vector<uint32_t> def_sky(size_t(count) + 1);
    int def_sky_count = 0;
    *out_secondary_count = *out_shadow_count = 0;
    auto *color = reinterpret_cast<color_rgba_t *>(inout_color);
    const Span<const Ref::hit_data_t> sp_hits(reinterpret_cast<const Ref::hit_data_t *>(hits), count);
    const Span<const Ref::ray_data_t> sp_rays(reinterpret_cast<const Ref::ray_data_t *>(rays), count);
    if (bounce == 0) {
        aligned_vector<color_rgba_t, 16> base(size_t(w) * h), dn(size_t(w) * h);
        Ref::ShadePrimary(ps, sp_hits, sp_rays, __pmj02_samples, rand_seed, iteration, eSpatialCacheMode::None, *b.sc,
                          b.textures, reinterpret_cast<Ref::ray_data_t *>(out_secondary), out_secondary_count,
                          reinterpret_cast<Ref::shadow_ray_t *>(out_shadow), out_shadow_count, def_sky.data(),
                          &def_sky_count, w, 1.0f / float(iteration), color, base.data(), dn.data());
    } else {
        const float clamp_direct = (bounce == 1) ? ps.clamp_direct : ps.clamp_indirect;
        Ref::ShadeSecondary(ps, clamp_direct, sp_hits, sp_rays, __pmj02_samples, rand_seed, iteration, eSpatialCacheMode::None,
                            *b.sc, b.textures, reinterpret_cast<Ref::ray_data_t *>(out_secondary), out_secondary_count,
                            reinterpret_cast<Ref::shadow_ray_t *>(out_shadow), out_shadow_count, def_sky.data(),
                            &def_sky_count, w, color, nullptr, nullptr);
    }
    return 0;
}

} // extern "C"
