// rt_texture.h -- stochastic 1-tap texture fetch (reference CoreRef.cpp:2838-2892, TextureStorageCPU.h:229-331).
// Device layout: every texture is a mip chain of row-major RGBA8 images in one texel pool (rayhip.h).
#pragma once

#include "rt_types.h"

namespace rt {

RT_HD const rayhip_texture &tex_entry(const SceneView &sc, uint32_t handle) {
    return sc.textures[sc.tex_table[handle >> 28] + (handle & 0x00ffffffu)];
}

// TexStorageSwizzled::Fetch(index, int x, int y, lod): wrap with C '%', expand to float, divide by 255
RT_HD f4 tex_fetch(const SceneView &sc, const rayhip_texture &t, int x, int y, int lod) {
    const int w = int(t.width[lod]), h = int(t.height[lod]);
    x %= w;
    y %= h;
    const uint32_t px = sc.texels[t.offset[lod] + uint32_t(y) * uint32_t(w) + uint32_t(x)];
    f4 ret;
    ret.x = float(px & 0xffu) / 255.0f;
    ret.y = float((px >> 8) & 0xffu) / 255.0f;
    ret.z = float((px >> 16) & 0xffu) / 255.0f;
    ret.w = float((px >> 24) & 0xffu) / 255.0f;
    return ret;
}

// CoreRef.cpp:2859-2876 (USE_STOCH_TEXTURE_FILTERING branch)
RT_HD f4 sample_bilinear(const SceneView &sc, uint32_t handle, f2 uvs, int lod, f2 rnd) {
    const rayhip_texture &t = tex_entry(sc, handle);
    const f2 img_size = {float(t.width[lod]), float(t.height[lod])};
    f2 _uvs = {uvs.x - floorf(uvs.x), uvs.y - floorf(uvs.y)};
    _uvs = {_uvs.x * img_size.x - 0.5f, _uvs.y * img_size.y - 0.5f};
    _uvs = _uvs + rnd;
    return tex_fetch(sc, t, int(_uvs.x), int(_uvs.y), lod);
}

// CoreRef.cpp:2838-2850
RT_HD float get_texture_lod(const SceneView &sc, uint32_t handle, float lambda) {
    const rayhip_texture &t = tex_entry(sc, handle);
    float lod = lambda + 0.5f * fast_log2(float(t.width[0]) * float(t.height[0]));
    lod = clampf(lod - 1.0f, 0.0f, float(MAX_MIP_LEVEL));
    return lod;
}

// CoreRef.h:201-214 (3 channels, alpha passes through)
RT_HD f4 srgb_to_linear(f4 col) {
    f4 ret;
    ret.x = (col.x > 0.04045f) ? powf((col.x + 0.055f) / 1.055f, 2.4f) : col.x / 12.92f;
    ret.y = (col.y > 0.04045f) ? powf((col.y + 0.055f) / 1.055f, 2.4f) : col.y / 12.92f;
    ret.z = (col.z > 0.04045f) ? powf((col.z + 0.055f) / 1.055f, 2.4f) : col.z / 12.92f;
    ret.w = col.w;
    return ret;
}

// CoreRef.h:234-246
RT_HD f4 YCoCg_to_RGB(f4 col) {
    const float scale = (col.z * (255.0f / 8.0f)) + 1.0f;
    const float Y = col.w;
    const float Co = (col.x - (0.5f * 256.0f / 255.0f)) / scale;
    const float Cg = (col.y - (0.5f * 256.0f / 255.0f)) / scale;
    f4 rgb;
    rgb.x = Y + Co - Cg;
    rgb.y = Y + Cg;
    rgb.z = Y - Co - Cg;
    rgb.w = 1.0f;
    // saturate(): _mm_max_ps(0, _mm_min_ps(v, 1))
    rgb.x = sse_max(0.0f, sse_min(rgb.x, 1.0f));
    rgb.y = sse_max(0.0f, sse_min(rgb.y, 1.0f));
    rgb.z = sse_max(0.0f, sse_min(rgb.z, 1.0f));
    rgb.w = sse_max(0.0f, sse_min(rgb.w, 1.0f));
    return rgb;
}

// texture colour with the YCoCg / sRGB decode flags applied (pattern repeated at ShadeRef.cpp:1316-1324 etc.)
RT_HD f4 sample_color(const SceneView &sc, uint32_t handle, f2 uvs, int lod, f2 rnd) {
    f4 c = sample_bilinear(sc, handle, uvs, lod, rnd);
    if (handle & TEX_YCOCG_BIT) {
        c = YCoCg_to_RGB(c);
    }
    if (handle & TEX_SRGB_BIT) {
        c = srgb_to_linear(c);
    }
    return c;
}

} // namespace rt
