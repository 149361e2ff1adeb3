// rt_pixel.h -- how stage results land in the per-iteration radiance buffer (`temp`) and the aux buffers.
// One ray per pixel per stage, so none of these need atomics and the add order per pixel is fixed by the
// stage order.
#pragma once

#include "rt_shade.h"

namespace rt {

// ShadePrimary, ShadeRef.cpp:1664-1698: assign colour, blend aux buffers with the running-mean factor
RT_HD void write_primary_pixel(const ShadeResult &r, const uint32_t xy, const int img_w, const float mix_factor,
                               float4 *temp_buf, float4 *base_color_buf, float4 *depth_normals_buf) {
    const int x = int((xy >> 16) & 0x0000ffff), y = int(xy & 0x0000ffff);
    const int idx = y * img_w + x;
    temp_buf[idx] = mkfloat4(r.col.x, r.col.y, r.col.z, r.col.w);

    {
        const float4 o = base_color_buf[idx];
        f4 old_val = {o.x, o.y, o.z, o.w};
        f4 new_val = {r.base_color.x, r.base_color.y, r.base_color.z, 0.0f};
        const float norm_factor = fmaxf(fmaxf(new_val.x, new_val.y), fmaxf(new_val.z, 1.0f));
        new_val = new_val / norm_factor;
        old_val += (new_val - old_val) * mix_factor;
        base_color_buf[idx] = mkfloat4(old_val.x, old_val.y, old_val.z, old_val.w);
    }
    {
        const float4 o = depth_normals_buf[idx];
        f4 old_val = {o.x, o.y, o.z, o.w};
        old_val += (r.depth_normal - old_val) * mix_factor;
        depth_normals_buf[idx] = mkfloat4(old_val.x, old_val.y, old_val.z, old_val.w);
    }
}

// Batched form (Layering): the aux blends depend on the iteration order, so the primary shade only stores what it
// would have blended (the normalised base colour, depth + normal) on its layer; k_accumulate blends them in order.
RT_HD void write_primary_pixel_layered(const ShadeResult &r, const uint32_t xy_virtual, const int img_w, float4 *temp_buf,
                                       float4 *aux_base_layers, float4 *aux_dn_layers) {
    const int x = int((xy_virtual >> 16) & 0x0000ffff), y = int(xy_virtual & 0x0000ffff);
    const int idx = y * img_w + x;
    temp_buf[idx] = mkfloat4(r.col.x, r.col.y, r.col.z, r.col.w);
    f4 new_val = {r.base_color.x, r.base_color.y, r.base_color.z, 0.0f};
    const float norm_factor = fmaxf(fmaxf(new_val.x, new_val.y), fmaxf(new_val.z, 1.0f));
    new_val = new_val / norm_factor;
    aux_base_layers[idx] = mkfloat4(new_val.x, new_val.y, new_val.z, new_val.w);
    aux_dn_layers[idx] = mkfloat4(r.depth_normal.x, r.depth_normal.y, r.depth_normal.z, r.depth_normal.w);
}
RT_HD void blend_aux_pixel(const int idx, const float4 new_base, const float4 new_dn, const float mix_factor, float4 *base_color_buf,
                           float4 *depth_normals_buf) {
    {
        const float4 o = base_color_buf[idx];
        f4 old_val = {o.x, o.y, o.z, o.w};
        const f4 new_val = {new_base.x, new_base.y, new_base.z, new_base.w};
        old_val += (new_val - old_val) * mix_factor;
        base_color_buf[idx] = mkfloat4(old_val.x, old_val.y, old_val.z, old_val.w);
    }
    {
        const float4 o = depth_normals_buf[idx];
        f4 old_val = {o.x, o.y, o.z, o.w};
        const f4 new_val = {new_dn.x, new_dn.y, new_dn.z, new_dn.w};
        old_val += (new_val - old_val) * mix_factor;
        depth_normals_buf[idx] = mkfloat4(old_val.x, old_val.y, old_val.z, old_val.w);
    }
}

// ShadeSecondary, ShadeRef.cpp:1713-1729: temp += rgb (alpha untouched)
RT_HD void add_secondary_pixel(const ShadeResult &r, const uint32_t xy, const int img_w, float4 *temp_buf) {
    const int x = int((xy >> 16) & 0x0000ffff), y = int(xy & 0x0000ffff);
    const int idx = y * img_w + x;
    float4 o = temp_buf[idx];
    o.x += r.col.x, o.y += r.col.y, o.z += r.col.z, o.w += 0.0f;
    temp_buf[idx] = o;
}

// ShadeSky tail, AtmosphereRef.cpp:1004-1007: temp += rgb, alpha = 1
RT_HD void add_sky_pixel(const f3 c, const uint32_t xy, const int img_w, float4 *temp_buf) {
    const int x = int((xy >> 16) & 0x0000ffff), y = int(xy & 0x0000ffff);
    float4 o = temp_buf[y * img_w + x];
    o.x += c.x, o.y += c.y, o.z += c.z, o.w = 1.0f;
    temp_buf[y * img_w + x] = o;
}

// TraceShadowRays tail, CoreRef.cpp:4866-4880: clamp on the rgb sum, then temp += rc
RT_HD void add_shadow_pixel(f3 rc, const float limit, const uint32_t xy, const int img_w, float4 *temp_buf) {
    const int x = int((xy >> 16) & 0x0000ffff), y = int(xy & 0x0000ffff);
    const float sum = hsum(mk4(rc, 0.0f));
    if (sum > limit) {
        rc *= (limit / sum);
    }
    const int idx = y * img_w + x;
    float4 o = temp_buf[idx];
    o.x += rc.x, o.y += rc.y, o.z += rc.z, o.w += 0.0f;
    temp_buf[idx] = o;
}

} // namespace rt
