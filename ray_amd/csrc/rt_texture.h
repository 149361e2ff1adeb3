// rt_texture.h -- stochastic 1-tap texture fetch (reference CoreRef.cpp:2838-2892, TextureStorageCPU.h:229-331).
// Device layout: every texture is a mip chain in one pool of 32-bit words (rayhip.h): row-major RGBA8 images, or -- the four
// block-compressed storages under RAYHIP_TEX_RAW_BC -- the reference's 4x4 blocks.
#pragma once

#include "rt_types.h"

namespace rt {

RT_HD const rayhip_texture &tex_entry(const SceneView &sc, uint32_t handle) {
    return sc.textures[sc.tex_table[handle >> 28] + (handle & 0x00ffffffu)];
}

// ---- block-compressed storages ------------------------------------------------------------------
// With RAYHIP_TEX_RAW_BC the BC1 / BC3 / BC4 / BC5 storages keep the reference's 4x4 blocks in HBM (a quarter to an eighth
// of the RGBA8 footprint) and a fetch decodes the one texel it needs: integer arithmetic only, the values
// TexStorageBCn<N>::Get returns (TextureStorageCPU.h:381-541).  The reference's decoder has no modes: a colour block always
// has two interpolated colours (no punch-through alpha), the alpha half of a BC3 block always the 6-step ramp; the
// single-channel blocks of BC4 / BC5 pick the 6-step or the 4-step ramp by the order of their end points.
constexpr uint32_t TEX_STORAGE_BC1 = 4, TEX_STORAGE_BC3 = 5, TEX_STORAGE_BC4 = 6, TEX_STORAGE_BC5 = 7;

// 5- or 6-bit channel of an RGB565 end point -> 8 bit (TextureStorageCPU.h:354-362)
RT_HD uint32_t bc_expand(const uint32_t c, const uint32_t from_bits) {
    const uint32_t b = (1u << (from_bits - 1u)) + c * 255u;
    return (b + (b >> from_bits)) >> from_bits;
}
// colour block (8 bytes in lo, hi): texel k of the 4x4 tile
RT_HD void bc_colour(const uint32_t lo, const uint32_t hi, const uint32_t k, uint32_t rgb[3]) {
    const uint32_t c0 = lo & 0xffffu, c1 = lo >> 16, sel = (hi >> (2u * k)) & 3u;
    const uint32_t e0[3] = {bc_expand((c0 >> 11) & 31u, 5), bc_expand((c0 >> 5) & 63u, 6), bc_expand(c0 & 31u, 5)};
    const uint32_t e1[3] = {bc_expand((c1 >> 11) & 31u, 5), bc_expand((c1 >> 5) & 63u, 6), bc_expand(c1 & 31u, 5)};
    for (int ch = 0; ch < 3; ++ch) {
        const uint32_t third = (2u * e0[ch] + e1[ch]) / 3u, two_thirds = (e0[ch] + 2u * e1[ch]) / 3u;
        rgb[ch] = sel == 0u ? e0[ch] : sel == 1u ? e1[ch] : sel == 2u ? third : two_thirds;
    }
}
// single-channel block (8 bytes in lo, hi): two end points, then sixteen 3-bit selectors
RT_HD uint32_t bc_ramp(const uint32_t lo, const uint32_t hi, const uint32_t k, const bool always_six_steps) {
    const uint32_t a0 = lo & 0xffu, a1 = (lo >> 8) & 0xffu;
    const uint32_t bit = 16u + 3u * k; // of the 64-bit little-endian block
    const uint32_t sel = (bit < 32u ? ((lo >> bit) | (bit > 29u ? hi << (32u - bit) : 0u)) : (hi >> (bit - 32u))) & 7u;
    if (sel == 0u) {
        return a0;
    }
    if (sel == 1u) {
        return a1;
    }
    if (always_six_steps || a0 > a1) {
        return ((8u - sel) * a0 + (sel - 1u) * a1) / 7u;
    }
    return sel == 6u ? 0u : sel == 7u ? 255u : ((6u - sel) * a0 + (sel - 1u) * a1) / 5u;
}
RT_HD f4 tex_fetch_blocks(const SceneView &sc, const rayhip_texture &t, const int x, const int y, const int lod, const uint32_t storage) {
    const uint32_t tiles_x = (t.width[lod] + 3u) / 4u;
    const uint32_t words = (storage == TEX_STORAGE_BC3 || storage == TEX_STORAGE_BC5) ? 4u : 2u;
    const uint32_t *blk = sc.texels + t.offset[lod] + ((uint32_t(y) / 4u) * tiles_x + uint32_t(x) / 4u) * words;
    const uint32_t k = (uint32_t(y) % 4u) * 4u + uint32_t(x) % 4u;
    uint32_t v[4];
    if (storage == TEX_STORAGE_BC1) {
        bc_colour(blk[0], blk[1], k, v);
        v[3] = v[2];
    } else if (storage == TEX_STORAGE_BC3) {
        v[3] = bc_ramp(blk[0], blk[1], k, true);
        bc_colour(blk[2], blk[3], k, v);
    } else if (storage == TEX_STORAGE_BC4) {
        v[0] = v[1] = v[2] = v[3] = bc_ramp(blk[0], blk[1], k, false);
    } else {
        v[0] = bc_ramp(blk[0], blk[1], k, false);
        v[1] = v[2] = v[3] = bc_ramp(blk[2], blk[3], k, false);
    }
    return f4{float(v[0]) / 255.0f, float(v[1]) / 255.0f, float(v[2]) / 255.0f, float(v[3]) / 255.0f};
}

// TexStorageSwizzled::Fetch / TexStorageBCn::Fetch (index, int x, int y, lod): wrap with C '%', expand to float, divide by 255
RT_HD f4 tex_fetch(const SceneView &sc, const rayhip_texture &t, int x, int y, int lod, const uint32_t storage) {
    const int w = int(t.width[lod]), h = int(t.height[lod]);
    x %= w;
    y %= h;
    if (storage >= TEX_STORAGE_BC1 && (sc.tex_flags & RAYHIP_TEX_RAW_BC) != 0u) {
        return tex_fetch_blocks(sc, t, x, y, lod, storage);
    }
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
    return tex_fetch(sc, t, int(_uvs.x), int(_uvs.y), lod, handle >> 28);
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
