// scene_validate.h -- index validation of a flat scene before any kernel dereferences it (host, once per upload).
//
// A rayhip_scene_desc crosses the boundary as raw arrays whose entries index each other (include/rayhip.h); a scene
// blob read from disk is untrusted input.  Everything the kernels use as an index WITHOUT a bound of their own is
// checked here: BVH links and leaf ranges (both levels), instance -> BLAS root, tris[] entry -> triangle, triangle ->
// vertices / materials, material -> sub-materials / textures, texture mip -> texel pool, light-tree links -> lights,
// triangle lights -> triangle + instance.  Cost: one linear pass per array (Bistro-class scene: ~20 ms).
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rayhip.h"

namespace rayhip_validate {

inline bool fail(std::string &err, const char *what, uint64_t i, uint64_t v, uint64_t bound) {
    char buf[256];
    snprintf(buf, sizeof(buf), "scene validation: %s[%llu] = %llu is outside [0, %llu)", what, (unsigned long long)i, (unsigned long long)v,
             (unsigned long long)bound);
    err = buf;
    return false;
}

// walks one BVH2 tree from `root`; `leaf(word)` validates a leaf word.  `seen` marks visited nodes across calls, so a
// BLAS shared by many instances is walked once and a cycle (a node reached twice inside one tree) is detected.
template <class LeafFn>
inline bool walk_tree(const rayhip_scene_desc &d, const uint32_t root, std::vector<uint8_t> &seen, const uint8_t tag, std::string &err,
                      LeafFn &&leaf) {
    constexpr uint32_t COUNT_BITS = 7u << 29, INDEX_BITS = ~COUNT_BITS;
    std::vector<uint32_t> stack(1, root);
    while (!stack.empty()) {
        const uint32_t cur = stack.back();
        stack.pop_back();
        if (cur & COUNT_BITS) {
            if (!leaf(cur)) {
                return false;
            }
            continue;
        }
        const uint32_t n = cur & INDEX_BITS;
        if (n >= d.nodes_count) {
            return fail(err, "bvh node link", n, n, d.nodes_count);
        }
        if (seen[n] == tag) {
            err = "scene validation: the BVH is not a tree (a node is reachable twice)";
            return false;
        }
        if (seen[n] == 1 && tag != 1) {
            // a bottom-level tree that links into the top-level tree: the device would read instance leaves as triangle ranges
            err = "scene validation: a mesh's BVH links into the top-level tree";
            return false;
        }
        if (seen[n] != 0) {
            continue; // a subtree validated through another instance
        }
        seen[n] = tag;
        stack.push_back(d.nodes[n].left_child);
        stack.push_back(d.nodes[n].right_child);
    }
    return true;
}

inline bool validate_lights(const rayhip_scene_desc &d, std::string &err);
inline bool validate_sky(const rayhip_scene_desc &d, std::string &err);

// `all_sides_solid` (optional): is every side of every REACHABLE triangle plainly solid?  (the triangle pool is sparse: unused slots hold
// zeros, which would read as "not solid")
inline bool validate(const rayhip_scene_desc &d, std::string &err, bool *all_sides_solid = nullptr) {
    if (!validate_sky(d, err)) {
        return false;
    }
    constexpr uint32_t COUNT_BITS = 7u << 29, INDEX_BITS = ~COUNT_BITS, NONE = 0xffffffffu;
    constexpr uint32_t NODE_MIX = 4, NODE_PRINCIPLED = 6;
    constexpr uint32_t MAT_INDEX_BITS = 16383;
    const uint32_t n_real_tris = d.vtx_indices_count / 3;

    if (d.tri_indices_count < d.tris_count) {
        err = "scene validation: tri_indices is shorter than tris";
        return false;
    }
    // ---- both BVH levels ----
    std::vector<uint8_t> entry_used(d.tris_count, 0); // tris[] / tri_indices[] are sparse pools too: only leaf ranges count
    if (d.tlas_root != NONE) {
        std::vector<uint8_t> seen(d.nodes_count, 0);
        std::vector<uint8_t> inst_seen(d.mesh_instances_count, 0);
        bool ok = walk_tree(d, d.tlas_root, seen, 1, err, [&](const uint32_t w) {
            const uint32_t mi = w & INDEX_BITS;
            if (mi >= d.mesh_instances_count) {
                return fail(err, "TLAS leaf -> mesh instance", mi, mi, d.mesh_instances_count);
            }
            inst_seen[mi] = 1;
            return true;
        });
        if (!ok) {
            return false;
        }
        uint8_t tag = 2;
        for (uint32_t mi = 0; mi < d.mesh_instances_count; ++mi) {
            if (!inst_seen[mi]) {
                continue;
            }
            const uint32_t root = d.mesh_instances[mi].node_index;
            if ((root & COUNT_BITS) == 0 && (root & INDEX_BITS) < d.nodes_count && seen[root & INDEX_BITS] >= 2) {
                continue; // this BLAS was walked for an earlier instance
            }
            ok = walk_tree(d, root, seen, tag, err, [&](const uint32_t w) {
                const uint64_t first = w & INDEX_BITS, count = ((w & COUNT_BITS) >> 29) + 1;
                if (first + count > d.tris_count) {
                    return fail(err, "BLAS leaf range end", first, first + count, uint64_t(d.tris_count) + 1);
                }
                for (uint64_t k = first; k < first + count; ++k) {
                    entry_used[k] = 1;
                }
                return true;
            });
            if (!ok) {
                return false;
            }
            tag = tag == 255 ? 2 : uint8_t(tag + 1); // (tags only need to differ from the tree being walked)
        }
    }
    // ---- tris[] entry -> triangle -> vertices / materials ----
    // (the triangle, material and texture arrays are sparse pools on the host: only what is reachable is checked)
    std::vector<uint8_t> tri_used(n_real_tris, 0), mat_used(d.materials_count, 0), tex_used(d.textures_count, 0);
    std::vector<uint32_t> mat_work;
    auto use_material = [&](const uint32_t m) {
        if (!mat_used[m]) {
            mat_used[m] = 1;
            mat_work.push_back(m);
        }
    };
    for (uint32_t i = 0; i < d.tris_count; ++i) {
        if (!entry_used[i]) {
            continue;
        }
        const uint32_t t = d.tri_indices[i];
        if (t >= n_real_tris || t >= d.tri_materials_count) {
            return fail(err, "tri_indices", i, t, n_real_tris < d.tri_materials_count ? n_real_tris : d.tri_materials_count);
        }
        tri_used[t] = 1;
    }
    bool solid = true;
    if (all_sides_solid) {
        *all_sides_solid = false; // (until the loop below has seen every reachable triangle)
    }
    for (uint32_t t = 0; t < n_real_tris; ++t) {
        if (!tri_used[t]) {
            continue;
        }
        solid = solid && (d.tri_materials[t].front_mi & 0x8000u) != 0 && (d.tri_materials[t].back_mi & 0x8000u) != 0; // MATERIAL_SOLID_BIT
        for (int k = 0; k < 3; ++k) {
            if (d.vtx_indices[size_t(t) * 3 + k] >= d.vertices_count) {
                return fail(err, "vtx_indices", size_t(t) * 3 + k, d.vtx_indices[size_t(t) * 3 + k], d.vertices_count);
            }
        }
        const rayhip_tri_mat_data &md = d.tri_materials[t];
        if ((md.front_mi & MAT_INDEX_BITS) >= d.materials_count) {
            return fail(err, "tri_materials.front_mi", t, md.front_mi & MAT_INDEX_BITS, d.materials_count);
        }
        use_material(md.front_mi & MAT_INDEX_BITS);
        if (md.back_mi != 0xffff) {
            if ((md.back_mi & MAT_INDEX_BITS) >= d.materials_count) {
                return fail(err, "tri_materials.back_mi", t, md.back_mi & MAT_INDEX_BITS, d.materials_count);
            }
            use_material(md.back_mi & MAT_INDEX_BITS);
        }
    }
    if (all_sides_solid) {
        *all_sides_solid = solid;
    }
    // ---- materials -> sub-materials, textures ; textures -> texel pool ----
    auto use_texture = [&](const uint32_t handle) {
        if (handle == NONE) {
            return true;
        }
        if ((handle >> 28) >= 8u) {
            return false; // tex_table has eight storages; the kernels index it with these four bits unchecked
        }
        const uint64_t t = uint64_t(d.tex_table[handle >> 28]) + (handle & 0x00ffffffu);
        if (t >= d.textures_count) {
            return false;
        }
        tex_used[t] = 1;
        return true;
    };
    while (!mat_work.empty()) {
        const uint32_t m = mat_work.back();
        mat_work.pop_back();
        const rayhip_material &mat = d.materials[m];
        // slots the shade stage reads (rt_shade.h): a mix node its factor texture (slot 1; slots 3, 4 are its operands), every
        // other node normals / base / roughness, Principled also metallic / specular; the rest may hold anything
        const int k0 = mat.type == NODE_MIX ? 1 : 0, k1 = mat.type == NODE_MIX ? 2 : (mat.type == NODE_PRINCIPLED ? 5 : 3);
        for (int k = k0; k < k1; ++k) {
            if (!use_texture(mat.textures[k])) {
                return fail(err, "material texture handle", m, mat.textures[k], d.textures_count);
            }
        }
        if (mat.type == NODE_MIX) {
            for (int k = 3; k < 5; ++k) {
                if (mat.textures[k] >= d.materials_count) {
                    return fail(err, "mix material operand", m, mat.textures[k], d.materials_count);
                }
                use_material(mat.textures[k]);
            }
        }
    }
    if (!use_texture(d.env.env_map) || !use_texture(d.env.back_map)) {
        err = "scene validation: environment map handle outside the texture table";
        return false;
    }
    for (uint32_t t = 0; t < d.textures_count; ++t) {
        if (!tex_used[t]) {
            continue;
        }
        // the storage a table entry belongs to: the last one whose first entry is not behind it (block-compressed storages
        // handed over as blocks occupy 2 or 4 words per 4x4 tile, everything else one word per texel)
        uint32_t storage = 0;
        for (uint32_t k = 0; k < 8; ++k) {
            if (d.tex_table[k] <= t) {
                storage = k;
            }
        }
        const bool blocks = storage >= 4 && (d.texture_flags & RAYHIP_TEX_RAW_BC) != 0u;
        const uint64_t words_per_block = (storage == 5 || storage == 7) ? 4 : 2;
        for (int l = 0; l < 12; ++l) {
            const uint64_t w = d.textures[t].width[l], h = d.textures[t].height[l];
            const uint64_t end = uint64_t(d.textures[t].offset[l]) + (blocks ? ((w + 3) / 4) * ((h + 3) / 4) * words_per_block : w * h);
            if (end > d.texels_count) {
                return fail(err, "texture mip end", t, end, uint64_t(d.texels_count) + 1);
            }
        }
    }
    return validate_lights(d, err);
}

// the part an instance / light update replaces (rayhip_scene_update_instances): lights, light tree, env light
// the physical sky (rayhip_sky): array sizes against the dimensions the struct states, power-of-two sizes where the samplers wrap with a
// mask, the directional-light list against the light array
inline bool validate_sky(const rayhip_scene_desc &d, std::string &err) {
    if (!(d.env.sky_map_spread_angle > 0.0f)) {
        return true; // (not a physical sky: the fields are ignored)
    }
    if (d.sky_count != 1 || d.sky == nullptr) {
        err = "scene validation: the environment is the physical sky (sky_map_spread_angle > 0) but rayhip_scene_desc::sky is missing";
        return false;
    }
    const rayhip_sky &k = *d.sky;
    // arrays that state a size must exist (a direct C-ABI caller can pass a count with a null pointer; the blob path cannot -- ADVICE round 4)
    const struct { const void *p; uint64_t n; const char *what; } arrays[] = {
        {d.sky_transmittance_lut, d.sky_transmittance_lut_count, "sky_transmittance_lut"}, {d.sky_multiscatter_lut, d.sky_multiscatter_lut_count, "sky_multiscatter_lut"},
        {d.sky_dir_lights, d.sky_dir_lights_count, "sky_dir_lights"}, {d.sky_weather_tex, d.sky_weather_tex_count, "sky_weather_tex"},
        {d.sky_noise3d_tex, d.sky_noise3d_tex_count, "sky_noise3d_tex"}, {d.sky_curl_tex, d.sky_curl_tex_count, "sky_curl_tex"},
        {d.sky_moon_tex, d.sky_moon_tex_count, "sky_moon_tex"}, {d.sky_cirrus_tex, d.sky_cirrus_tex_count, "sky_cirrus_tex"}};
    for (const auto &a : arrays) {
        if (a.n != 0 && a.p == nullptr) {
            err = std::string("scene validation: ") + a.what + " states " + std::to_string(a.n) + " elements and is a null pointer";
            return false;
        }
    }
    // ... and the atmosphere must describe one: a zero or NaN radius or scale height gives NaN radiance, not an error, further down
    const rayhip_atmosphere &at = k.atmosphere;
    const auto positive = [](const float v) { return v > 0.0f && v < 3.0e38f; }; // (false for NaN)
    if (!positive(at.planet_radius) || !positive(at.atmosphere_height) || !positive(at.rayleigh_height) || !positive(at.mie_height) ||
        !(at.clouds_height_beg < at.clouds_height_end) || !(at.viewpoint_height == at.viewpoint_height)) {
        err = "scene validation: sky atmosphere needs finite positive planet_radius, atmosphere_height, rayleigh_height, mie_height and clouds_height_beg < clouds_height_end";
        return false;
    }
    const auto pow2 = [](const int32_t v) { return v > 0 && (v & (v - 1)) == 0; };
    const auto sized = [&](const char *what, const uint64_t have, const uint64_t want) {
        if (have != want) {
            err = std::string("scene validation: sky ") + what + " holds " + std::to_string(have) + " elements, its dimensions need " + std::to_string(want);
            return false;
        }
        return true;
    };
    if (k.transmittance_lut_w <= 0 || k.transmittance_lut_h <= 0 || k.multiscatter_lut_res < 0 || !pow2(k.weather_res) || !pow2(k.noise3d_res) ||
        !pow2(k.curl_res) || !pow2(k.moon_w) || !pow2(k.moon_h) || !pow2(k.cirrus_res) || k.weather_res > 4096 || k.noise3d_res > 512 ||
        k.curl_res > 4096 || k.moon_w > 8192 || k.moon_h > 8192 || k.cirrus_res > 4096 || k.transmittance_lut_w > 4096 || k.transmittance_lut_h > 4096 ||
        k.multiscatter_lut_res > 1024) {
        err = "scene validation: sky texture / table dimensions out of range (wrapped textures need power-of-two sizes)";
        return false;
    }
    if (!sized("transmittance table", d.sky_transmittance_lut_count, uint64_t(4) * uint64_t(k.transmittance_lut_w) * uint64_t(k.transmittance_lut_h)) ||
        !sized("multiple-scattering table", d.sky_multiscatter_lut_count, uint64_t(4) * uint64_t(k.multiscatter_lut_res) * uint64_t(k.multiscatter_lut_res)) ||
        !sized("weather map", d.sky_weather_tex_count, uint64_t(3) * uint64_t(k.weather_res) * uint64_t(k.weather_res)) ||
        !sized("3-d noise", d.sky_noise3d_tex_count, uint64_t(k.noise3d_res) * uint64_t(k.noise3d_res) * uint64_t(k.noise3d_res)) ||
        !sized("curl noise", d.sky_curl_tex_count, uint64_t(3) * uint64_t(k.curl_res) * uint64_t(k.curl_res)) ||
        !sized("moon map", d.sky_moon_tex_count, uint64_t(3) * uint64_t(k.moon_w) * uint64_t(k.moon_h)) ||
        !sized("cirrus map", d.sky_cirrus_tex_count, uint64_t(2) * uint64_t(k.cirrus_res) * uint64_t(k.cirrus_res))) {
        return false;
    }
    for (uint32_t i = 0; i < d.sky_dir_lights_count; ++i) {
        if (d.sky_dir_lights[i] >= d.lights_count || (d.lights[d.sky_dir_lights[i]].flags & 7u) != 1u /* LIGHT_TYPE_DIR */) {
            err = "scene validation: sky_dir_lights[" + std::to_string(i) + "] does not name a directional light";
            return false;
        }
    }
    return true;
}

inline bool validate_lights(const rayhip_scene_desc &d, std::string &err) {
    constexpr uint32_t NONE = 0xffffffffu, LIGHT_LEAF_BIT = 1u << 31, LIGHT_TYPE_TRI = 5;
    const uint32_t n_real_tris = d.vtx_indices_count / 3;
    for (uint32_t k = 0; k < d.li_indices_count; ++k) {
        const uint32_t i = d.li_indices[k];
        if (i >= d.lights_count) {
            return fail(err, "li_indices", k, i, d.lights_count);
        }
        const rayhip_light &l = d.lights[i];
        if ((l.flags & 7u) == LIGHT_TYPE_TRI) {
            uint32_t tri, mi;
            memcpy(&tri, &l.params[0], 4), memcpy(&mi, &l.params[1], 4);
            if (tri >= n_real_tris) {
                return fail(err, "triangle light -> triangle", i, tri, n_real_tris);
            }
            if (mi >= d.mesh_instances_count) {
                return fail(err, "triangle light -> mesh instance", i, mi, d.mesh_instances_count);
            }
            for (int c = 0; c < 3; ++c) {
                if (d.vtx_indices[size_t(tri) * 3 + c] >= d.vertices_count) {
                    return fail(err, "triangle light vertex", i, d.vtx_indices[size_t(tri) * 3 + c], d.vertices_count);
                }
            }
        }
    }
    for (uint32_t n = 0; n < d.light_cwnodes_count; ++n) {
        const rayhip_light_cwbvh_node &node = d.light_cwnodes[n];
        for (int j = 0; j < 8; ++j) {
            const uint32_t ch = node.child[j];
            if (ch == 0x7fffffffu) {
                continue; // empty slot (Core.cpp:1134-1138): zero flux, never picked
            }
            if (ch & LIGHT_LEAF_BIT) {
                if ((ch & ~LIGHT_LEAF_BIT) >= d.lights_count) {
                    return fail(err, "light tree leaf", n, ch & ~LIGHT_LEAF_BIT, d.lights_count);
                }
            } else if (ch >= d.light_cwnodes_count) {
                return fail(err, "light tree link", n, ch, d.light_cwnodes_count);
            }
        }
    }
    if (d.env.light_index != NONE && d.env.light_index >= d.lights_count) {
        return fail(err, "env.light_index", 0, d.env.light_index, d.lights_count);
    }
    return true;
}

} // namespace rayhip_validate
