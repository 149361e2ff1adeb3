// rt_shade.h -- the shade stage as one call: surface -> light pick -> scatter (shade_point.h, shade_lights.h,
// shade_lobes.h).  The device runs the three stages as separate kernels with a queue of shade points between them
// (kernels.hip.h); the host build of the kernel sources (tests/hostsim) and the kernel-level hook run them back to back
// through this function, which is also the single place that states how their results combine into what the reference's
// ShadeSurface returns (internal/ShadeRef.cpp:1174-1652): one pixel contribution, at most one secondary ray, at most
// one shadow ray.
#pragma once

#include "rt_sky.h"
#include "shade_lobes.h"

namespace rt {

struct ShadeResult {
    f4 col;          // radiance to assign (primary) / add (secondary) into the pixel
    f3 base_color;   // first-hit feature images
    f4 depth_normal; // N.xyz, t
    bool emit_secondary, emit_shadow;
    bool defer_sky; // the path ended in the physical sky: shade_sky_ray (rt_sky.h) adds its radiance to the pixel afterwards
};

// radiance the scatter stage books at once (lights that cast no shadow): throughput, then the indirect clamp
RT_HD f3 direct_radiance(const ShadeParams &sp, const Scatter &sct, const f3 throughput) {
    f3 c = sct.direct;
    c *= throughput;
    return clamp_radiance_sum(c, sp.limits[1]);
}
// the random number of the light pick for the path vertex (ray.xy, ray.depth)
RT_HD float light_pick_random(const SceneView &sc, const ShadeParams &sp, const uint32_t xy, const uint32_t depth) {
    return path_random(sc, sp, xy, depth).get(RAND_DIM_LIGHT_PICK).x;
}
RT_HD LightPick no_light_pick() { return LightPick{0u, 0.0f, 0.0f}; }

RT_HD ShadeResult shade_surface(const SceneView &sc, const ShadeParams &sp, const Hit &hit, const Ray &ray, Ray &new_ray, ShadowRay &sh_r) {
    ShadeResult res;
    res.emit_secondary = res.emit_shadow = false;
    ShadePoint pt;
    SurfaceOut so;
    const bool continues = surface_stage<false>(sc, sp, hit, ray, pt, so);
    res.col = so.radiance;
    res.base_color = so.base_color;
    res.depth_normal = so.normal_depth;
    res.defer_sky = so.deferred_sky;
    if (!continues) {
        return res;
    }
    const LightPick pick = sc.light_cwnodes_count != 0 ? pick_light(sc, pt.P, light_pick_random(sc, sp, ray.xy, ray.depth)) : no_light_pick();
    Scatter sct;
    scatter_stage(sc, sp, ray, pt, pick, sct);
    res.col = mk4(direct_radiance(sp, sct, ray.c), 1.0f);
    res.emit_secondary = sct.has_next, res.emit_shadow = sct.has_shadow;
    new_ray = sct.next, sh_r = sct.shadow;
    return res;
}

} // namespace rt
