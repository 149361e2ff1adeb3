// rt_accum.h -- per-pixel running mean, tonemap and variance.  Restates reference
// internal/RendererCPU.h:607-658 (GLSL twins: shaders/mix_incremental.comp.glsl:23-62 and
// shaders/postprocess.comp.glsl:28-68, fused here into one pass over the rect).
#pragma once

#include "rt_types.h"

namespace rt {

struct AccumParams {
    int w;
    int rect[4];
    int iteration;
    float exposure;        // pow(2, cam.exposure), RendererCPU.h:382
    float mix_factor;      // 1 / iteration, :470
    float half_mix_factor; // 1 / ((iteration + 1) / 2), :608
    int is_class_a;        // popcount((iteration-1) & 0xaaaaaaaa) & 1, :607
    int view_transform;    // only Standard (0) is supported
    float inv_gamma;
    float variance_threshold;
    Shard shard;
};

// TonemapRef.h:19-28
RT_HD float tonemap_standard(float c) {
    if (c < 0.0031308f) {
        return 12.92f * c;
    }
    return 1.055f * powf(c, (1.0f / 2.4f)) - 0.055f;
}

// TonemapRef.h:7-9
RT_HD f4 reversible_tonemap(f4 c) { return c / (fmaxf(c.x, fmaxf(c.y, c.z)) + 1.0f); }

// temp_px: this pixel's radiance of the iteration; variance_px: where its variance estimate goes (the reference reuses
// the temp buffer's element)
RT_HD void accumulate_pixel(const AccumParams &p, const int x, const int y, const float4 *temp_px, float4 *variance_px,
                            float4 *full_buf, float4 *half_buf, float4 *raw_buf, float4 *final_buf, uint16_t *required_samples) {
    const int idx = y * p.w + x;

    if (!(required_samples[idx] < p.iteration)) {
        const float4 t = *temp_px;
        // new_val = temp * {exposure, exposure, exposure, 1}
        const f4 new_val = {t.x * p.exposure, t.y * p.exposure, t.z * p.exposure, t.w * 1.0f};
        const float4 ff = full_buf[idx];
        f4 cur_full = {ff.x, ff.y, ff.z, ff.w};
        cur_full += (new_val - cur_full) * p.mix_factor;
        full_buf[idx] = mkfloat4(cur_full.x, cur_full.y, cur_full.z, cur_full.w);
        if (p.is_class_a) {
            const float4 hh = half_buf[idx];
            f4 cur_half = {hh.x, hh.y, hh.z, hh.w};
            cur_half += (new_val - cur_half) * p.half_mix_factor;
            half_buf[idx] = mkfloat4(cur_half.x, cur_half.y, cur_half.z, cur_half.w);
        }
    }

    const float4 ff = full_buf[idx], hh = half_buf[idx];
    const f4 full_val = {ff.x, ff.y, ff.z, ff.w}, half_val = {hh.x, hh.y, hh.z, hh.w};

    raw_buf[idx] = ff;

    // Tonemap(), TonemapRef.h:33-45 (Standard view transform)
    f4 c = full_val;
    c.x = tonemap_standard(c.x);
    c.y = tonemap_standard(c.y);
    c.z = tonemap_standard(c.z);
    if (p.inv_gamma != 1.0f) {
        c.x = powf(c.x, p.inv_gamma);
        c.y = powf(c.y, p.inv_gamma);
        c.z = powf(c.z, p.inv_gamma);
        c.w = powf(c.w, 1.0f);
    }
    // saturate = max(0, min(c, 1)) with SSE operand order
    c.x = sse_max(0.0f, sse_min(c.x, 1.0f));
    c.y = sse_max(0.0f, sse_min(c.y, 1.0f));
    c.z = sse_max(0.0f, sse_min(c.z, 1.0f));
    c.w = sse_max(0.0f, sse_min(c.w, 1.0f));
    final_buf[idx] = mkfloat4(c.x, c.y, c.z, c.w);

    // variance estimate from the two half-sample images, RendererCPU.h:641-645
    f4 d = 2.0f * full_val - half_val;
    d = {sse_max(d.x, 0.0f), sse_max(d.y, 0.0f), sse_max(d.z, 0.0f), sse_max(d.w, 0.0f)};
    const f4 p1 = reversible_tonemap(d);
    const f4 p2 = reversible_tonemap(half_val);
    const f4 variance = 0.5f * (p1 - p2) * (p1 - p2);
    *variance_px = mkfloat4(variance.x, variance.y, variance.z, variance.w);

    if (variance.x >= p.variance_threshold || variance.y >= p.variance_threshold || variance.z >= p.variance_threshold ||
        variance.w >= p.variance_threshold) {
        required_samples[idx] = uint16_t(p.iteration + 1);
    }
}

} // namespace rt
