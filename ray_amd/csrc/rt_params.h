// rt_params.h -- host-side preparation of per-iteration kernel parameters from the camera.
// Restates the prologue of reference internal/RendererCPU.h:373-470,578-608 (what RenderScene derives from
// camera_t before launching stages).  Host code only (libm calls here are the reference's own host calls).
#pragma once

#include <math.h>

#include "rt_accum.h"
#include "rt_raygen.h"
#include "rt_shade.h"
#include "rt_traverse.h"

namespace rt {

inline uint32_t iteration_rand_seed(int iteration) { // RendererCPU.h:446
    return hash(uint32_t((iteration - 1) / RAND_SAMPLES_COUNT));
}

inline RayGenParams make_raygen_params(const rayhip_camera &cam, int w, int h, const int rect[4], int iteration,
                                       const Shard shard = Shard{64, 1, 0}) {
    RayGenParams p;
    p.origin = mk3(cam.origin), p.fwd = mk3(cam.fwd), p.side = mk3(cam.side), p.up = mk3(cam.up);
    p.focus_distance = cam.focus_distance;
    // CoreRef.cpp:1439-1442
    p.k = float(w) / float(h);
    const float temp = tanf(0.5f * cam.fov * PI / 180.0f);
    p.fov_k = temp * cam.focus_distance;
    p.spread_angle = atanf(2.0f * temp / float(h));
    p.shift[0] = cam.shift[0], p.shift[1] = cam.shift[1];
    p.fstop = cam.fstop, p.focal_length = cam.focal_length, p.sensor_height = cam.sensor_height;
    p.lens_rotation = cam.lens_rotation, p.lens_ratio = cam.lens_ratio, p.lens_blades = cam.lens_blades;
    p.clip_start = cam.clip_start, p.clip_end = cam.clip_end;
    p.filter_is_box = (cam.filter == 0 /* ePixelFilter::Box */);
    p.w = w, p.h = h;
    for (int i = 0; i < 4; ++i) {
        p.rect[i] = rect[i];
    }
    p.iteration = iteration;
    p.rand_seed = iteration_rand_seed(iteration);
    p.shard = shard;
    p.skip_ior = 0;
    return p;
}

inline PassLimits make_pass_limits(const rayhip_camera &cam) {
    const rayhip_pass_settings &s = cam.pass_settings;
    PassLimits ps;
    ps.max_diff_depth = s.max_diff_depth, ps.max_spec_depth = s.max_spec_depth, ps.max_refr_depth = s.max_refr_depth;
    ps.max_transp_depth = s.max_transp_depth, ps.max_total_depth = s.max_total_depth;
    ps.min_total_depth = s.min_total_depth, ps.min_transp_depth = s.min_transp_depth;
    ps.regularize_alpha = s.regularize_alpha;
    return ps;
}

inline TraceParams make_trace_params(const rayhip_camera &cam, uint32_t tlas_root, int iteration) {
    TraceParams tp;
    tp.min_transp_depth = cam.pass_settings.min_transp_depth;
    tp.max_transp_depth = cam.pass_settings.max_transp_depth;
    tp.rand_seed = iteration_rand_seed(iteration);
    tp.iteration = iteration;
    tp.root_index = tlas_root;
    return tp;
}

// ShadePrimary limits: ShadeRef.cpp:1661-1662 ; ShadeSecondary (bounce >= 1): ShadeRef.cpp:1710-1711 with
// clamp_direct chosen per bounce at RendererCPU.h:546
inline ShadeParams make_shade_params(const rayhip_camera &cam, int iteration, int bounce) {
    const rayhip_pass_settings &s = cam.pass_settings;
    ShadeParams sp;
    sp.ps = make_pass_limits(cam);
    if (bounce == 0) {
        sp.limits[0] = sp.limits[1] = (s.clamp_direct != 0.0f) ? 3.0f * s.clamp_direct : FLT_MAX;
    } else {
        const float clamp_direct = (bounce == 1) ? s.clamp_direct : s.clamp_indirect;
        sp.limits[0] = (clamp_direct != 0.0f) ? 3.0f * clamp_direct : FLT_MAX;
        sp.limits[1] = (s.clamp_indirect != 0.0f) ? 3.0f * s.clamp_indirect : FLT_MAX;
    }
    sp.rand_seed = iteration_rand_seed(iteration);
    sp.iteration = iteration;
    sp.plain_ior = 0u;
    return sp;
}

// TraceShadowRays clamp: CoreRef.cpp:4860 ; value per stage RendererCPU.h:490-492,561-564
inline float shadow_clamp_limit(const rayhip_camera &cam, int bounce) {
    const float c = (bounce == 0) ? cam.pass_settings.clamp_direct : cam.pass_settings.clamp_indirect;
    return (c != 0.0f) ? 3.0f * c : FLT_MAX;
}

inline int popcount32(uint32_t x) {
    int c = 0;
    for (; x != 0; x &= x - 1) {
        c++;
    }
    return c;
}

inline AccumParams make_accum_params(const rayhip_camera &cam, int w, const int rect[4], int iteration,
                                     const Shard shard = Shard{64, 1, 0}) {
    AccumParams p;
    p.w = w;
    for (int i = 0; i < 4; ++i) {
        p.rect[i] = rect[i];
    }
    p.iteration = iteration;
    p.exposure = powf(2.0f, cam.exposure);                           // RendererCPU.h:382 (std::pow(float,float))
    p.mix_factor = 1.0f / float(iteration);                          // :470
    p.half_mix_factor = 1.0f / float((iteration + 1) / 2);           // :608
    p.is_class_a = popcount32(uint32_t(iteration - 1) & 0xaaaaaaaa) & 1; // :607
    p.view_transform = cam.view_transform;
    p.lut = nullptr, p.lut_dims = 0; // the caller points these at the table of cam.view_transform
    p.inv_gamma = (1.0f / cam.gamma);
    p.variance_threshold = iteration > cam.pass_settings.min_samples
                               ? 0.5f * cam.pass_settings.variance_threshold * cam.pass_settings.variance_threshold
                               : 0.0f; // :583-586
    p.shard = shard;
    return p;
}

// Scene::GetBounds-derived ray sort grid, RendererCPU.h:417-421
inline void make_sort_grid(const float bbox_min[3], const float bbox_max[3], float root_min[3], float cell_size[3]) {
    for (int i = 0; i < 3; ++i) {
        root_min[i] = bbox_min[i];
        cell_size[i] = (bbox_max[i] - bbox_min[i]) / 255;
    }
}

} // namespace rt
