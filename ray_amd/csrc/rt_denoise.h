// rt_denoise.h -- RendererBase::DenoiseImage(region): variance pre-filter + joint non-local-means filter + tonemap.
// Restates reference internal/RendererCPU.h:661-783 (orchestration) and internal/DenoiseRef.cpp:9-93
// (JointNLMFilter<7, 3> with base-colour and depth-normal guides); GLSL twins: shaders/filter_variance.comp.glsl,
// shaders/nlm_filter.comp.glsl.  SURVEY.md section 8f, N2 (the UNet denoiser stays out of scope).
//
// The reference works on a copy of the region extended by EXT_RADIUS = 8 pixels, with coordinates clamped to the frame
// when it fetches from the frame buffers.  Here the intermediate images cover the extended region too:
//   stage 1 (every pixel of the extended region)  tm = reversible_tonemap(full), var_h = max(v, gauss9_horizontal(v))
//   stage 2 (the extended region minus a 4-pixel rim)  var = max(var_h, gauss9_vertical(var_h))
//   stage 3 (the region)  required_samples from var; NLM over tm guided by var and the two feature images;
//                         raw = reversible_tonemap_invert(nlm); final = Tonemap(raw)
// Per-pixel functions, host + device, same operation order as the reference's fvec4 code.
#pragma once

#include "rt_accum.h"

namespace rt {

constexpr int NLM_EXT_RADIUS = 8;  // RendererCPU.h:668
constexpr int NLM_WINDOW_SIZE = 7; // :760-761
constexpr int NLM_NEIGHBORHOOD_SIZE = 3;

struct DenoiseParams {
    int w, h;        // frame
    int rect[4];     // region x, y, w, h
    int ext_w, ext_h; // rect + 2 * NLM_EXT_RADIUS
    int iteration;   // RegionContext::iteration (of the last RenderScene)
    float variance_threshold;
};

RT_HD f4 ld4(const float4 v) { return f4{v.x, v.y, v.z, v.w}; }
RT_HD float4 st4(const f4 v) { return mkfloat4(v.x, v.y, v.z, v.w); }
RT_HD f4 max4(const f4 a, const f4 b) { return f4{sse_max(a.x, b.x), sse_max(a.y, b.y), sse_max(a.z, b.z), sse_max(a.w, b.w)}; }
RT_HD f4 min4(const f4 a, const f4 b) { return f4{sse_min(a.x, b.x), sse_min(a.y, b.y), sse_min(a.z, b.z), sse_min(a.w, b.w)}; }
// TonemapRef.h:11-13
RT_HD f4 reversible_tonemap_invert(const f4 c) { return c / (1.0f - fmaxf(c.x, fmaxf(c.y, c.z))); }

// RendererCPU.h:688 (static const float GaussWeights[])
RT_HD float nlm_gauss_weight(const int i) {
    return i == 0 ? 0.2270270270f : (i == 1 ? 0.1945945946f : (i == 2 ? 0.1216216216f : (i == 3 ? 0.0540540541f : 0.0162162162f)));
}

// stage 1, pixel (x, y) of the extended region: RendererCPU.h:690-709.  `full` and `variance` are frame buffers
// (clamped fetches), `tm_out` / `var_h_out` are [ext_h][ext_w].
RT_HD void nlm_prepare_h(const DenoiseParams &p, const int x, const int y, const float4 *full, const float4 *variance, float4 *tm_out,
                         float4 *var_h_out) {
    const int xx = p.rect[0] - NLM_EXT_RADIUS + x, yy = p.rect[1] - NLM_EXT_RADIUS + y;
    const int cy = clampi(yy, 0, p.h - 1);
    tm_out[y * p.ext_w + x] = st4(reversible_tonemap(ld4(full[cy * p.w + clampi(xx, 0, p.w - 1)])));

    const f4 center_val = ld4(variance[cy * p.w + clampi(xx, 0, p.w - 1)]);
    f4 res = center_val * nlm_gauss_weight(0);
    for (int i = 0; i < 4; ++i) {
        res += ld4(variance[cy * p.w + clampi(xx - i + 1, 0, p.w - 1)]) * nlm_gauss_weight(i + 1);
        res += ld4(variance[cy * p.w + clampi(xx + i + 1, 0, p.w - 1)]) * nlm_gauss_weight(i + 1);
    }
    var_h_out[y * p.ext_w + x] = st4(max4(res, center_val));
}

// stage 2, pixel (x, y) of the extended region with 4 <= x < ext_w - 4, 4 <= y < ext_h - 4: RendererCPU.h:719-736
RT_HD void nlm_prepare_v(const DenoiseParams &p, const int x, const int y, const float4 *var_h, float4 *var_out) {
    const f4 center_val = ld4(var_h[y * p.ext_w + x]);
    f4 res = center_val * nlm_gauss_weight(0);
    for (int i = 0; i < 4; ++i) {
        res += ld4(var_h[(y - i + 1) * p.ext_w + x]) * nlm_gauss_weight(i + 1);
        res += ld4(var_h[(y + i + 1) * p.ext_w + x]) * nlm_gauss_weight(i + 1);
    }
    var_out[y * p.ext_w + x] = st4(max4(res, center_val));
}

// the two guide images at pixel (x, y) of the extended region: clamped fetches from the frame (RendererCPU.h:713-716)
RT_HD f4 nlm_feature(const DenoiseParams &p, const float4 *frame_buf, const int x, const int y) {
    const int xx = p.rect[0] - NLM_EXT_RADIUS + x, yy = p.rect[1] - NLM_EXT_RADIUS + y;
    return ld4(frame_buf[clampi(yy, 0, p.h - 1) * p.w + clampi(xx, 0, p.w - 1)]);
}

// one term of the patch distance, DenoiseRef.cpp:39-46 (alpha 1, damping 0.45)
RT_HD f4 nlm_pair_distance(const f4 ipx, const f4 jpx, const f4 ivar, const f4 jvar) {
    const float alpha = 1.0f, damping = 0.45f;
    const f4 min_var = min4(ivar, jvar);
    return ((ipx - jpx) * (ipx - jpx) - alpha * (ivar + min_var)) /
           (f4{0.0001f, 0.0001f, 0.0001f, 0.0001f} + (damping * damping) * (ivar + jvar));
}
// weight of window position j for pixel i from the summed patch distance and the two guides, DenoiseRef.cpp:50-77
RT_HD float nlm_weight(const f4 color_distance, const f4 f0_i, const f4 f0_j, const f4 f1_i, const f4 f1_j) {
    const float PatchDistanceNormFactor = float(NLM_NEIGHBORHOOD_SIZE * NLM_NEIGHBORHOOD_SIZE);
    const float feature0_weight = 64.0f, feature1_weight = 32.0f;
    const float patch_distance =
        0.25f * PatchDistanceNormFactor * (color_distance.x + color_distance.y + color_distance.z + color_distance.w);
    float weight = expf(-fmaxf(0.0f, patch_distance));
    f4 feature_distance = feature0_weight * (f0_i - f0_j) * (f0_i - f0_j);
    feature_distance = max4(feature_distance, feature1_weight * (f1_i - f1_j) * (f1_i - f1_j));
    const float feature_patch_distance = 0.25f * (feature_distance.x + feature_distance.y + feature_distance.z + feature_distance.w);
    const float feature_weight = expf(-fmaxf(0.0f, fminf(10000.0f, feature_patch_distance)));
    return fminf(weight, feature_weight);
}

// stage 3, pixel (x, y) of the region (frame coordinates xx = rect.x + x ...): RendererCPU.h:748-777 +
// DenoiseRef.cpp:9-93 with WINDOW 7, NEIGHBORHOOD 3, alpha 1, damping 0.45, feature weights 64 / 32.
// Fetch is a callable (ext_x, ext_y, which) -> f4 with which = 0: tm, 1: var (lets the device stage them through LDS).
template <class Fetch>
RT_HD f4 nlm_filter_pixel(const DenoiseParams &p, const int x, const int y, const float4 *base_color, const float4 *depth_normals,
                          Fetch &&fetch) {
    constexpr int WindowRadius = (NLM_WINDOW_SIZE - 1) / 2, NeighborRadius = (NLM_NEIGHBORHOOD_SIZE - 1) / 2;
    const int ix = NLM_EXT_RADIUS + x, iy = NLM_EXT_RADIUS + y;

    const f4 f0_i = nlm_feature(p, base_color, ix, iy), f1_i = nlm_feature(p, depth_normals, ix, iy);

    f4 sum_output = {0.0f, 0.0f, 0.0f, 0.0f};
    float sum_weight = 0.0f;
    for (int k = -WindowRadius; k <= WindowRadius; ++k) {
        const int jy = iy + k;
        for (int l = -WindowRadius; l <= WindowRadius; ++l) {
            const int jx = ix + l;
            f4 color_distance = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int q = -NeighborRadius; q <= NeighborRadius; ++q) {
                for (int pp = -NeighborRadius; pp <= NeighborRadius; ++pp) {
                    color_distance += nlm_pair_distance(fetch(ix + pp, iy + q, 0), fetch(jx + pp, jy + q, 0), fetch(ix + pp, iy + q, 1),
                                                        fetch(jx + pp, jy + q, 1));
                }
            }
            const float weight =
                nlm_weight(color_distance, f0_i, nlm_feature(p, base_color, jx, jy), f1_i, nlm_feature(p, depth_normals, jx, jy));
            sum_output += fetch(jx, jy, 0) * weight;
            sum_weight += weight;
        }
    }
    if (sum_weight != 0.0f) {
        sum_output = sum_output / sum_weight;
    }
    return sum_output;
}

// stage 3 epilogue for frame pixel idx: RendererCPU.h:748-757 (adaptive-sampling flag) and :771-777
RT_HD void nlm_finish_pixel(const DenoiseParams &p, const AccumParams &tone, const int idx, const f4 filtered_variance, const f4 nlm,
                            float4 *raw_buf, float4 *final_buf, uint16_t *required_samples) {
    if (filtered_variance.x >= p.variance_threshold || filtered_variance.y >= p.variance_threshold ||
        filtered_variance.z >= p.variance_threshold || filtered_variance.w >= p.variance_threshold) {
        required_samples[idx] = uint16_t(p.iteration + 1);
    }
    const f4 col = reversible_tonemap_invert(nlm);
    raw_buf[idx] = st4(col);
    final_buf[idx] = st4(tonemap(tone, col));
}

} // namespace rt
