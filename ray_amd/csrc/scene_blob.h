// scene_blob.h -- a flat, relocatable serialisation of (rayhip_scene_desc, rayhip_camera, filter table).
//
// Why it exists: the scene arrays are produced by the reference's host-side scene code, which is only
// available where /root/reference is.  A blob lets a finalized scene travel (tests/golden/*.rayscene) and be
// uploaded through the C ABI with no reference code present.  It is also how the Python layer hands big
// procedural scenes to librayhip without describing 14 arrays through ctypes.
//
// Layout: Header, then `count` Section records, then the payloads (16-byte aligned).  All little-endian POD.
#pragma once

#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rayhip.h"

namespace rayhip_blob {

static const char MAGIC[8] = {'R', 'A', 'Y', 'H', 'I', 'P', 'S', '1'};

struct Header {
    char magic[8];
    uint32_t section_count;
    uint32_t _pad;
};
struct Section {
    char name[24];
    uint64_t offset; // from blob start
    uint64_t size;   // bytes
};

// scalar tail of rayhip_scene_desc that is not an array
struct Scalars {
    uint32_t tex_table[8];
    rayhip_environment env;
    uint32_t tlas_root, visible_lights_count, blocker_lights_count, texture_flags; // (the last one: padding, i.e. 0, in blobs of round 1)
    float bbox_min[3], bbox_max[3];
};

inline void add_section(std::vector<Section> &secs, std::vector<uint8_t> &payload, const char *name, const void *data,
                        size_t size) {
    Section s = {};
    strncpy(s.name, name, sizeof(s.name) - 1);
    while (payload.size() % 16) {
        payload.push_back(0);
    }
    s.offset = payload.size();
    s.size = size;
    if (size) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        payload.insert(payload.end(), p, p + size);
    }
    secs.push_back(s);
}

// `tonemap_lut`: the dims^3 table of cam.view_transform (nullptr for the Standard transform)
inline std::vector<uint8_t> serialize(const rayhip_scene_desc &d, const rayhip_camera &cam, const float *filter_table,
                                      int filter_table_count, const uint32_t *tonemap_lut = nullptr, int tonemap_lut_dims = 0) {
    std::vector<Section> secs;
    std::vector<uint8_t> payload;
#define ARR(field) add_section(secs, payload, #field, d.field, size_t(d.field##_count) * sizeof(*d.field));
    ARR(nodes)
    ARR(tris)
    ARR(tri_indices)
    ARR(tri_materials)
    ARR(materials)
    ARR(vertices)
    ARR(vtx_indices)
    ARR(mesh_instances)
    ARR(lights)
    ARR(li_indices)
    ARR(light_cwnodes)
    ARR(textures)
    ARR(texels)
    ARR(env_qtree)
    if (d.sky_count != 0) { // the physical sky (rayhip_sky): present only when the environment is one
        ARR(sky)
        ARR(sky_transmittance_lut)
        ARR(sky_multiscatter_lut)
        ARR(sky_dir_lights)
        ARR(sky_weather_tex)
        ARR(sky_noise3d_tex)
        ARR(sky_curl_tex)
        ARR(sky_moon_tex)
        ARR(sky_cirrus_tex)
    }
#undef ARR
    Scalars sc = {};
    memcpy(sc.tex_table, d.tex_table, sizeof(sc.tex_table));
    sc.env = d.env;
    sc.tlas_root = d.tlas_root, sc.visible_lights_count = d.visible_lights_count;
    sc.blocker_lights_count = d.blocker_lights_count;
    sc.texture_flags = d.texture_flags;
    memcpy(sc.bbox_min, d.bbox_min, 12), memcpy(sc.bbox_max, d.bbox_max, 12);
    add_section(secs, payload, "scalars", &sc, sizeof(sc));
    add_section(secs, payload, "camera", &cam, sizeof(cam));
    add_section(secs, payload, "filter_table", filter_table, size_t(filter_table_count) * sizeof(float));
    if (tonemap_lut && tonemap_lut_dims > 0) {
        add_section(secs, payload, "tonemap_lut", tonemap_lut,
                    size_t(tonemap_lut_dims) * size_t(tonemap_lut_dims) * size_t(tonemap_lut_dims) * sizeof(uint32_t));
    }

    Header h = {};
    memcpy(h.magic, MAGIC, 8);
    h.section_count = uint32_t(secs.size());
    size_t base = sizeof(Header) + secs.size() * sizeof(Section);
    base = (base + 15) & ~size_t(15);
    for (Section &s : secs) {
        s.offset += base;
    }
    std::vector<uint8_t> out(base + payload.size(), 0);
    memcpy(out.data(), &h, sizeof(h));
    memcpy(out.data() + sizeof(h), secs.data(), secs.size() * sizeof(Section));
    if (!payload.empty()) {
        memcpy(out.data() + base, payload.data(), payload.size());
    }
    return out;
}

// Pointers in `d` alias `blob` (which must stay alive and be at least 16-byte aligned).
struct Extras {
    const uint32_t *tonemap_lut = nullptr; // table of cam.view_transform, if the blob carries one
    int tonemap_lut_dims = 0;
};
inline bool deserialize(const void *blob, size_t size, rayhip_scene_desc &d, rayhip_camera &cam, const float **filter_table,
                        int *filter_table_count, std::string &err, Extras *extras = nullptr) {
    const uint8_t *b = static_cast<const uint8_t *>(blob);
    if (size < sizeof(Header)) {
        err = "scene blob too small";
        return false;
    }
    Header h;
    memcpy(&h, b, sizeof(h));
    if (memcmp(h.magic, MAGIC, 8) != 0) {
        err = "bad scene blob magic";
        return false;
    }
    if (sizeof(Header) + size_t(h.section_count) * sizeof(Section) > size) {
        err = "truncated scene blob";
        return false;
    }
    d = {};
    d.struct_size = uint32_t(sizeof(rayhip_scene_desc));
    bool have_scalars = false, have_cam = false;
    for (uint32_t i = 0; i < h.section_count; ++i) {
        Section s;
        memcpy(&s, b + sizeof(Header) + size_t(i) * sizeof(Section), sizeof(s));
        if (s.offset > size || s.size > size - s.offset) { // (written so that offset + size cannot wrap)
            err = "scene blob section out of range";
            return false;
        }
        if (s.offset % 16 != 0) { // sections are reinterpreted as arrays of 16-byte aligned structs
            err = "scene blob section is not 16-byte aligned";
            return false;
        }
        const uint8_t *p = b + s.offset;
        const std::string name(s.name, strnlen(s.name, sizeof(s.name)));
#define ARR(field, type)                                                                                               \
    if (name == #field) {                                                                                              \
        if (s.size % sizeof(type) != 0) {                                                                             \
            err = "scene blob section " #field " is not a whole number of elements";                                  \
            return false;                                                                                              \
        }                                                                                                              \
        d.field = reinterpret_cast<const type *>(p);                                                                  \
        d.field##_count = uint32_t(s.size / sizeof(type));                                                            \
        continue;                                                                                                      \
    }
        ARR(nodes, rayhip_bvh2_node)
        ARR(tris, rayhip_tri_accel)
        ARR(tri_indices, uint32_t)
        ARR(tri_materials, rayhip_tri_mat_data)
        ARR(materials, rayhip_material)
        ARR(vertices, rayhip_vertex)
        ARR(vtx_indices, uint32_t)
        ARR(mesh_instances, rayhip_mesh_instance)
        ARR(lights, rayhip_light)
        ARR(li_indices, uint32_t)
        ARR(light_cwnodes, rayhip_light_cwbvh_node)
        ARR(textures, rayhip_texture)
        ARR(texels, uint32_t)
        ARR(env_qtree, float)
        ARR(sky, rayhip_sky)
        ARR(sky_transmittance_lut, float)
        ARR(sky_multiscatter_lut, float)
        ARR(sky_dir_lights, uint32_t)
        ARR(sky_weather_tex, uint8_t)
        ARR(sky_noise3d_tex, uint8_t)
        ARR(sky_curl_tex, uint8_t)
        ARR(sky_moon_tex, uint8_t)
        ARR(sky_cirrus_tex, uint8_t)
#undef ARR
        if (name == "tonemap_lut") {
            int dims = 1;
            while (size_t(dims) * size_t(dims) * size_t(dims) * sizeof(uint32_t) < s.size) {
                ++dims;
            }
            if (size_t(dims) * size_t(dims) * size_t(dims) * sizeof(uint32_t) != s.size) {
                err = "tonemap_lut section is not a cube";
                return false;
            }
            if (extras) {
                extras->tonemap_lut = reinterpret_cast<const uint32_t *>(p), extras->tonemap_lut_dims = dims;
            }
            continue;
        }
        if (name == "scalars" && s.size == sizeof(Scalars)) {
            Scalars sc;
            memcpy(&sc, p, sizeof(sc));
            memcpy(d.tex_table, sc.tex_table, sizeof(sc.tex_table));
            d.env = sc.env;
            d.tlas_root = sc.tlas_root, d.visible_lights_count = sc.visible_lights_count;
            d.blocker_lights_count = sc.blocker_lights_count;
            d.texture_flags = sc.texture_flags;
            memcpy(d.bbox_min, sc.bbox_min, 12), memcpy(d.bbox_max, sc.bbox_max, 12);
            have_scalars = true;
        } else if (name == "camera" && s.size == sizeof(rayhip_camera)) {
            memcpy(&cam, p, sizeof(cam));
            have_cam = true;
        } else if (name == "filter_table") {
            *filter_table = reinterpret_cast<const float *>(p);
            *filter_table_count = int(s.size / sizeof(float));
        }
    }
    if (!have_scalars || !have_cam) {
        err = "scene blob lacks scalars/camera section";
        return false;
    }
    return true;
}

} // namespace rayhip_blob
