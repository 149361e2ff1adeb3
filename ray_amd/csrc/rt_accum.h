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
    int view_transform;    // eViewTransform: 0 = Standard (sRGB curve), others = look-up through `lut`
    const uint32_t *lut;   // [lut_dims^3] RGB10_A2 (the reference's precomputed tables, handed over through the C ABI)
    int lut_dims;
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

// TonemapFilmic + FetchLUT, TonemapRef.cpp:29-80: trilinear look-up in a dims^3 RGB10_A2 table addressed by c / (c + 1)
RT_HD f4 fetch_lut(const uint32_t *lut, const int dims, const int ix, const int iy, const int iz) {
    const uint32_t v = lut[(iz * dims + iy) * dims + ix];
    return f4{float(int(v & 0x3ffu)) * (1.0f / 1023.0f), float(int((v >> 10) & 0x3ffu)) * (1.0f / 1023.0f),
              float(int((v >> 20) & 0x3ffu)) * (1.0f / 1023.0f), float(int((v >> 30) & 0x3u)) * (1.0f / 3.0f)};
}
RT_HD f4 tonemap_filmic(const uint32_t *lut, const int dims, const f4 color) {
    const f4 encoded = color / (color + f4{1.0f, 1.0f, 1.0f, 1.0f});
    const f4 uv = encoded * float(dims - 1);
    const int ix = int(uv.x), iy = int(uv.y), iz = int(uv.z); // (ivec4(fvec4): truncation)
    const float fx = fractf(uv.x), fy = fractf(uv.y), fz = fractf(uv.z);
    const int jx = (ix + 1 < dims - 1) ? ix + 1 : dims - 1, jy = (iy + 1 < dims - 1) ? iy + 1 : dims - 1,
              jz = (iz + 1 < dims - 1) ? iz + 1 : dims - 1;
    const f4 c000 = fetch_lut(lut, dims, ix, iy, iz), c001 = fetch_lut(lut, dims, jx, iy, iz),
             c010 = fetch_lut(lut, dims, ix, jy, iz), c011 = fetch_lut(lut, dims, jx, jy, iz),
             c100 = fetch_lut(lut, dims, ix, iy, jz), c101 = fetch_lut(lut, dims, jx, iy, jz),
             c110 = fetch_lut(lut, dims, ix, jy, jz), c111 = fetch_lut(lut, dims, jx, jy, jz);
    const f4 c00x = (1.0f - fx) * c000 + fx * c001, c01x = (1.0f - fx) * c010 + fx * c011,
             c10x = (1.0f - fx) * c100 + fx * c101, c11x = (1.0f - fx) * c110 + fx * c111;
    const f4 c0xx = (1.0f - fy) * c00x + fy * c01x, c1xx = (1.0f - fy) * c10x + fy * c11x;
    f4 cxxx = (1.0f - fz) * c0xx + fz * c1xx;
    cxxx.w = color.w;
    return cxxx;
}
// Tonemap(), TonemapRef.h:33-45
RT_HD f4 tonemap(const AccumParams &p, f4 c) {
    if (p.view_transform == 0) {
        c.x = tonemap_standard(c.x), c.y = tonemap_standard(c.y), c.z = tonemap_standard(c.z);
    } else {
        c = tonemap_filmic(p.lut, p.lut_dims, c);
    }
    if (p.inv_gamma != 1.0f) {
        c.x = powf(c.x, p.inv_gamma), c.y = powf(c.y, p.inv_gamma), c.z = powf(c.z, p.inv_gamma);
        c.w = powf(c.w, 1.0f);
    }
    // saturate = max(0, min(c, 1)) with SSE operand order
    c.x = sse_max(0.0f, sse_min(c.x, 1.0f)), c.y = sse_max(0.0f, sse_min(c.y, 1.0f));
    c.z = sse_max(0.0f, sse_min(c.z, 1.0f)), c.w = sse_max(0.0f, sse_min(c.w, 1.0f));
    return c;
}

// TonemapRef.h:7-9
RT_HD f4 reversible_tonemap(f4 c) { return c / (fmaxf(c.x, fmaxf(c.y, c.z)) + 1.0f); }

// One pixel's accumulation state: what an iteration's accumulate step reads and leaves behind (RendererCPU.h:607-658).
struct AccumPixel {
    f4 full, half;     // running means over all / over the class-A iterations
    f4 variance;       // estimate left by the last iteration
    uint16_t required; // required_samples
};
// One iteration folded into the state: the radiance `t` of the iteration joins the running means (if the pixel is still being sampled),
// the variance estimate and the adaptive-sampling mark follow.  What the iteration would write to raw / final is a function of the state
// it leaves (accum_outputs), so a pass of many layers keeps the state in registers and writes once (k_accumulate).
RT_HD void accumulate_step(const AccumParams &p, const float4 t, AccumPixel &s) {
    if (!(s.required < p.iteration)) {
        // new_val = temp * {exposure, exposure, exposure, 1}
        const f4 new_val = {t.x * p.exposure, t.y * p.exposure, t.z * p.exposure, t.w * 1.0f};
        s.full += (new_val - s.full) * p.mix_factor;
        if (p.is_class_a) {
            s.half += (new_val - s.half) * p.half_mix_factor;
        }
    }
    // variance estimate from the two half-sample images, RendererCPU.h:641-645
    f4 d = 2.0f * s.full - s.half;
    d = {sse_max(d.x, 0.0f), sse_max(d.y, 0.0f), sse_max(d.z, 0.0f), sse_max(d.w, 0.0f)};
    const f4 p1 = reversible_tonemap(d);
    const f4 p2 = reversible_tonemap(s.half);
    s.variance = 0.5f * (p1 - p2) * (p1 - p2);
    if (s.variance.x >= p.variance_threshold || s.variance.y >= p.variance_threshold || s.variance.z >= p.variance_threshold ||
        s.variance.w >= p.variance_threshold) {
        s.required = uint16_t(p.iteration + 1);
    }
}
RT_HD AccumPixel load_accum_pixel(const int idx, const float4 *full_buf, const float4 *half_buf, const uint16_t *required_samples) {
    const float4 ff = full_buf[idx], hh = half_buf[idx];
    return AccumPixel{f4{ff.x, ff.y, ff.z, ff.w}, f4{hh.x, hh.y, hh.z, hh.w}, f4{0.0f, 0.0f, 0.0f, 0.0f}, required_samples[idx]};
}
// write_half / write_required: whether the half-sample mean / the adaptive-sampling mark can differ from what memory holds (the layered
// pass always writes them; the one-iteration pass only on a class-A iteration / when the mark moved -- 18 bytes per pixel and iteration for
// the passes that cannot batch: ADVICE round 4)
RT_HD void store_accum_pixel(const AccumParams &p, const int idx, const AccumPixel &s, float4 *variance_px, float4 *full_buf, float4 *half_buf,
                             float4 *raw_buf, float4 *final_buf, uint16_t *required_samples, const bool write_half = true,
                             const bool write_required = true) {
    const float4 ff = mkfloat4(s.full.x, s.full.y, s.full.z, s.full.w);
    full_buf[idx] = ff;
    if (write_half) {
        half_buf[idx] = mkfloat4(s.half.x, s.half.y, s.half.z, s.half.w);
    }
    raw_buf[idx] = ff;
    const f4 c = tonemap(p, s.full);
    final_buf[idx] = mkfloat4(c.x, c.y, c.z, c.w);
    *variance_px = mkfloat4(s.variance.x, s.variance.y, s.variance.z, s.variance.w);
    if (write_required) {
        required_samples[idx] = s.required;
    }
}

// temp_px: this pixel's radiance of the iteration; variance_px: where its variance estimate goes (the reference reuses
// the temp buffer's element)
RT_HD void accumulate_pixel(const AccumParams &p, const int x, const int y, const float4 *temp_px, float4 *variance_px,
                            float4 *full_buf, float4 *half_buf, float4 *raw_buf, float4 *final_buf, uint16_t *required_samples) {
    const int idx = y * p.w + x;
    AccumPixel s = load_accum_pixel(idx, full_buf, half_buf, required_samples);
    const uint16_t required_before = s.required;
    const bool sampled = !(s.required < p.iteration);
    accumulate_step(p, *temp_px, s);
    store_accum_pixel(p, idx, s, variance_px, full_buf, half_buf, raw_buf, final_buf, required_samples, sampled && p.is_class_a != 0,
                      s.required != required_before);
}

} // namespace rt
