// hostsim.cpp -- TEST INFRASTRUCTURE.  Compiles the kernel bodies of ray_amd/csrc/rt_*.h with g++ and runs the
// RenderScene stage schedule as plain loops on the CPU, behind the same C signatures as include/rayhip.h
// (prefix hostsim_ instead of rayhip_).
//
// Purpose: the GPU box is a scarce resource and the integrand is chaotic, so the restatement of the
// reference's arithmetic is first proven on the CPU, where -- built without fma like the reference
// (-msse2 -mno-avx, glibc libm) -- it must match RendererRef BIT FOR BIT (tests/test_hostsim_parity.py).
// What is left for the GPU tests is then only what differs on the device: the device libm, wave-level
// compaction, the LDS stack and memory layout.
//
// This file is never linked into librayhip.so and nothing under ray_amd/ refers to it: the product has no
// CPU path (rayhip_* fail loudly without a GPU).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../ray_amd/csrc/bvh4_build.h"
#include "../../ray_amd/csrc/bvh8_build.h"
#include "../../ray_amd/csrc/bvh_layout.h"
#include "../../ray_amd/csrc/rt_arealights.h"
#include "../../ray_amd/csrc/rt_denoise.h"
#include "../../ray_amd/csrc/rt_params.h"
#include "../../ray_amd/csrc/rt_pixel.h"
#include "../../ray_amd/csrc/scene_blob.h"
#include "../../ray_amd/csrc/scene_rebuild.h"
#include "../../ray_amd/csrc/scene_update.h"
#include "../../ray_amd/csrc/scene_validate.h"

using namespace rt;

namespace {
thread_local std::string g_err;

struct HostScene {
    std::vector<rayhip_bvh2_node> nodes;
    std::vector<rayhip_tri_accel> tris;
    std::vector<uint32_t> tri_indices;
    std::vector<rayhip_tri_mat_data> tri_materials;
    std::vector<rayhip_material> materials;
    std::vector<rayhip_vertex> vertices;
    std::vector<uint32_t> vtx_indices;
    std::vector<rayhip_mesh_instance> mesh_instances;
    std::vector<rayhip_light> lights;
    std::vector<uint32_t> li_indices;
    std::vector<rayhip_light_cwbvh_node> light_cwnodes;
    std::vector<float4> light_children, light_tri_geom, tri_verts, tri_bitangents;
    std::vector<float> env_qtree;
    std::vector<Bvh4Node> nodes4;
    std::vector<Bvh8Node> nodes8;
    std::vector<uint32_t> blas_root4; // roots in the wide form in use (nodes4 or nodes8)
    std::vector<rayhip_texture> textures;
    std::vector<uint32_t> texels;
    // the physical sky (rayhip_sky): a copy of everything rt_sky.h reads
    rayhip_sky sky;
    std::vector<float> sky_transmittance_lut, sky_multiscatter_lut;
    std::vector<uint32_t> sky_dir_lights;
    std::vector<uint8_t> sky_weather, sky_noise3d, sky_curl, sky_moon, sky_cirrus;
    SkyView sky_view(const rayhip_scene_desc &d) {
        SkyView v = {};
        if (!(d.env.sky_map_spread_angle > 0.0f) || d.sky_count == 0) {
            return v;
        }
        sky = *d.sky;
        sky_transmittance_lut.assign(d.sky_transmittance_lut, d.sky_transmittance_lut + d.sky_transmittance_lut_count);
        sky_multiscatter_lut.assign(d.sky_multiscatter_lut, d.sky_multiscatter_lut + d.sky_multiscatter_lut_count);
        sky_dir_lights.assign(d.sky_dir_lights, d.sky_dir_lights + d.sky_dir_lights_count);
        sky_weather.assign(d.sky_weather_tex, d.sky_weather_tex + d.sky_weather_tex_count);
        sky_noise3d.assign(d.sky_noise3d_tex, d.sky_noise3d_tex + d.sky_noise3d_tex_count);
        sky_curl.assign(d.sky_curl_tex, d.sky_curl_tex + d.sky_curl_tex_count);
        sky_moon.assign(d.sky_moon_tex, d.sky_moon_tex + d.sky_moon_tex_count);
        sky_cirrus.assign(d.sky_cirrus_tex, d.sky_cirrus_tex + d.sky_cirrus_tex_count);
        v.desc = &sky, v.transmittance_lut = sky_transmittance_lut.data(), v.multiscatter_lut = sky_multiscatter_lut.data();
        v.dir_lights = sky_dir_lights.data(), v.dir_lights_count = uint32_t(sky_dir_lights.size());
        v.weather = sky_weather.data(), v.noise3d = sky_noise3d.data(), v.curl = sky_curl.data(), v.moon = sky_moon.data(), v.cirrus = sky_cirrus.data();
        return v;
    }
};
} // namespace

struct hostsim_ctx {
    int w = 0, h = 0;
    std::vector<uint32_t> pmj;
    std::vector<float> filter_table;
    HostScene hs;
    SceneView sc = {};
    float bbox_min[3] = {}, bbox_max[3] = {};
    std::vector<float4> temp, full, half, raw, final_, base_color, depth_normals;
    std::vector<uint16_t> required_samples;
    rayhip_trav_counters counters[2] = {};
    Shard shard = {64, 1, 0};
    bool layout_applied = false;
    int wide = 0; // walk the 4-wide BLAS (HOSTSIM_BVH4=1) or the 8-wide one (HOSTSIM_BVH8=1)
    rayhip_update::MeshRefs mesh_refs; // for hostsim_scene_update_instances_blob (scene_update.h)
    uint32_t nodes_used = 0;           // node slots of the last full upload; top levels built later go behind them
    bool have_scene = false;
    std::vector<uint32_t> tonemap_lut;
    int lut_transform = 0, lut_dims = 0;
};

// test hook: did the last scene upload go through the HBM layout pass?
extern "C" __attribute__((visibility("default"))) int hostsim_layout_applied(hostsim_ctx *c);

#define HS_API extern "C" __attribute__((visibility("default")))

HS_API const char *hostsim_last_error(void) { return g_err.c_str(); }
HS_API int hostsim_device_count(void) { return 1; }

HS_API int hostsim_ctx_create(int, hostsim_ctx **out) {
    *out = new hostsim_ctx();
    return 0;
}
HS_API void hostsim_ctx_destroy(hostsim_ctx *c) { delete c; }
int hostsim_layout_applied(hostsim_ctx *c) { return c->layout_applied ? 1 : 0; }
HS_API int hostsim_ctx_device_name(hostsim_ctx *, char *buf, int cap) {
    snprintf(buf, size_t(cap), "hostsim (CPU, test only)");
    return 0;
}
HS_API int hostsim_upload_static(hostsim_ctx *c, const uint32_t *pmj, uint32_t count) {
    c->pmj.assign(pmj, pmj + count);
    c->sc.pmj = c->pmj.data();
    return 0;
}
HS_API int hostsim_resize(hostsim_ctx *c, int w, int h) {
    if (c->w != w || c->h != h) {
        const size_t n = size_t(w) * size_t(h);
        const float4 z = {0, 0, 0, 0};
        c->temp.assign(n, z), c->full.assign(n, z), c->half.assign(n, z), c->raw.assign(n, z), c->final_.assign(n, z);
        c->base_color.assign(n, z), c->depth_normals.assign(n, z);
        c->required_samples.assign(n, 0xffff);
        c->w = w, c->h = h;
    }
    return 0;
}
HS_API int hostsim_clear(hostsim_ctx *c, const float rgba[4]) { // RendererCPU.h:297-301
    const float4 v = {rgba[0], rgba[1], rgba[2], rgba[3]};
    std::fill(c->full.begin(), c->full.end(), v);
    std::fill(c->half.begin(), c->half.end(), v);
    std::fill(c->required_samples.begin(), c->required_samples.end(), uint16_t(0xffff));
    return 0;
}
HS_API int hostsim_scene_upload(hostsim_ctx *c, const rayhip_scene_desc *d_in) {
    if (d_in->struct_size != sizeof(rayhip_scene_desc)) { // (the library's own check: rayhip_scene_upload)
        g_err = "rayhip_scene_desc::struct_size does not match this build's struct";
        return 1;
    }
    const rayhip_layout::AlignedDesc aligned0(*d_in);
    if (!rayhip_validate::validate(aligned0.d, g_err)) {
        return 1;
    }
    // HOSTSIM_LBVH=<leaf_max>: both levels of the acceleration structure rebuilt by the linear builder (lbvh.h, the host loop
    // over the element functions the device kernels call) instead of the trees the scene came with
    rayhip_rebuild::Rebuilt rebuilt;
    rayhip_scene_desc d_rebuilt = aligned0.d;
    // HOSTSIM_REFINE=<leaf_max>: the scene's own trees with every larger leaf replaced by a subtree of the same builder -- what
    // rayhip_scene_upload does by default (leaf_max 2).  Off by default here: the plain host build walks the reference's trees
    // as they are, which also pins the order in which exactly-tied triangles are met
    {
        int refine = 0, rebuild = 0;
        if (const char *e = getenv("HOSTSIM_REFINE")) {
            refine = atoi(e);
        }
        if (const char *e = getenv("HOSTSIM_LBVH")) {
            rebuild = atoi(e);
        }
        if (rebuild > 0 || refine > 0) {
            const bool sah = getenv("HOSTSIM_REFINE_SAH") != nullptr && atoi(getenv("HOSTSIM_REFINE_SAH")) != 0; // (scene_rebuild.h: SmallSahBuilder)
            rebuilt = rebuild > 0 ? rayhip_rebuild::rebuild_host(aligned0.d, uint32_t(rebuild))
                                  : (sah ? rayhip_rebuild::refine_host_sah(aligned0.d, uint32_t(refine)) : rayhip_rebuild::refine_host(aligned0.d, uint32_t(refine)));
            if (!rebuilt.ok) {
                g_err = "scene rebuild failed: " + rebuilt.why;
                return 1;
            }
            d_rebuilt.nodes = rebuilt.nodes.data(), d_rebuilt.nodes_count = uint32_t(rebuilt.nodes.size());
            d_rebuilt.tris = rebuilt.tris.data(), d_rebuilt.tris_count = uint32_t(rebuilt.tris.size());
            d_rebuilt.tri_indices = rebuilt.tri_indices.data(), d_rebuilt.tri_indices_count = uint32_t(rebuilt.tri_indices.size());
            d_rebuilt.mesh_instances = rebuilt.mesh_instances.data();
            d_rebuilt.tlas_root = rebuilt.tlas_root;
            if (!rayhip_validate::validate(d_rebuilt, g_err)) {
                g_err = "rebuilt scene: " + g_err;
                return 1;
            }
        }
    }
    const rayhip_layout::AlignedDesc aligned(d_rebuilt);
    const rayhip_scene_desc *d = &aligned.d;
    HostScene &s = c->hs;
#define CP(field) s.field.assign(d->field, d->field + d->field##_count)
    CP(nodes);
    CP(tris);
    CP(tri_indices);
    CP(tri_materials);
    CP(materials);
    CP(vertices);
    CP(vtx_indices);
    CP(mesh_instances);
    CP(lights);
    CP(li_indices);
    CP(light_cwnodes);
    s.light_children.resize(size_t(d->light_cwnodes_count) * LIGHT_CHILDREN_STRIDE);
    for (uint32_t n = 0; n < d->light_cwnodes_count; ++n) {
        fill_light_children(d->light_cwnodes[n], &s.light_children[size_t(n) * LIGHT_CHILDREN_STRIDE]);
    }
    s.tri_verts.resize(size_t(d->vtx_indices_count / 3) * TRI_VERTS_STRIDE);
    s.tri_bitangents.resize(size_t(d->vtx_indices_count / 3) * TRI_BITANGENTS_STRIDE);
    for (uint32_t t = 0; t < d->vtx_indices_count / 3; ++t) {
        fill_tri_verts(d->vertices, d->vertices_count, d->vtx_indices, t, d->tri_materials, d->tri_materials_count, &s.tri_verts[size_t(t) * TRI_VERTS_STRIDE],
                       &s.tri_bitangents[size_t(t) * TRI_BITANGENTS_STRIDE]);
    }
    s.light_tri_geom.assign(size_t(d->lights_count) * 4, mkfloat4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t k = 0; k < d->li_indices_count; ++k) { // (sparse pool: only the slots li_indices[] names hold lights)
        const uint32_t i = d->li_indices[k];
        fill_light_tri_geom(d->lights[i], d->mesh_instances, d->vtx_indices, d->vertices, &s.light_tri_geom[size_t(i) * 4]);
    }
    CP(textures);
    CP(texels);
    CP(env_qtree);
#undef CP
    // the same HBM layout pass librayhip applies at upload (results must not depend on it); HOSTSIM_NO_LAYOUT=1 skips
    uint32_t tlas_root = d->tlas_root;
    {
        const char *e = getenv("HOSTSIM_NO_LAYOUT");
        if (!(e && e[0] == '1')) {
            rayhip_layout::Result lay = rayhip_layout::optimize(*d);
            if (lay.applied) {
                s.nodes.swap(lay.nodes), s.tris.swap(lay.tris), s.tri_indices.swap(lay.tri_indices);
                s.mesh_instances.swap(lay.mesh_instances);
                tlas_root = lay.tlas_root;
            }
            c->layout_applied = lay.applied;
            if (!lay.applied && getenv("HOSTSIM_VERBOSE")) {
                fprintf(stderr, "hostsim: layout pass skipped: %s\n", lay.why_not);
            }
        }
    }
    // wide quantised BLAS: 8-wide (rt_bvh8.h, HOSTSIM_BVH8=1: re-orders the triangle records and re-bases the BVH2's leaf
    // words with them) or 4-wide (rt_bvh4.h, HOSTSIM_BVH4=1)
    c->wide = 0;
    s.nodes4.clear(), s.nodes8.clear(), s.blas_root4.clear();
    {
        const char *e8 = getenv("HOSTSIM_BVH8"), *e4 = getenv("HOSTSIM_BVH4");
        if (e8 && e8[0] == '1') {
            rayhip_bvh8::CostModel cm;
            if (const char *e = getenv("HOSTSIM_BVH8_CPRIM")) {
                cm.c_prim = float(atof(e));
            }
            if (const char *e = getenv("HOSTSIM_BVH8_PMAX")) {
                cm.p_max = uint32_t(atoi(e));
            }
            rayhip_bvh8::Result b8 = rayhip_bvh8::build(s.nodes.data(), uint32_t(s.nodes.size()), s.mesh_instances.data(),
                                                        uint32_t(s.mesh_instances.size()), tlas_root, s.tris.data(), s.tri_indices.data(),
                                                        uint32_t(s.tris.size()), cm);
            if (b8.ok) {
                s.nodes8.swap(b8.nodes), s.blas_root4.swap(b8.blas_root8);
                s.tris.swap(b8.tris), s.tri_indices.swap(b8.tri_indices);
                c->wide = 8;
                if (getenv("HOSTSIM_VERBOSE")) {
                    size_t inner = 0, leaves = 0;
                    for (const Bvh8Node &n : s.nodes8) {
                        inner += size_t(__builtin_popcount(n.exps_imask >> 24));
                        for (int k = 0; k < 8; ++k) {
                            leaves += ((n.meta[k >> 2] >> (8 * (k & 3))) & 0xffu) != 0u;
                        }
                    }
                    fprintf(stderr, "hostsim: BVH8: %zu nodes, %.2f inner + %.2f leaf children per node, %zu triangle records\n", s.nodes8.size(),
                            double(inner) / double(s.nodes8.size()), double(leaves) / double(s.nodes8.size()), s.tris.size());
                }
            } else if (getenv("HOSTSIM_VERBOSE")) {
                fprintf(stderr, "hostsim: BVH8 build failed: %s\n", b8.why_not);
            }
        }
        if (c->wide == 0) {
            rayhip_bvh4::Result b4 = rayhip_bvh4::build(s.nodes.data(), uint32_t(s.nodes.size()), s.mesh_instances.data(),
                                                        uint32_t(s.mesh_instances.size()), tlas_root);
            if (b4.ok) {
                s.nodes4.swap(b4.nodes), s.blas_root4.swap(b4.blas_root4);
                c->wide = (e4 && e4[0] == '1') ? 4 : 0;
            } else if (getenv("HOSTSIM_VERBOSE")) {
                fprintf(stderr, "hostsim: BVH4 build failed\n");
            }
        }
    }
    rayhip_update::collect_mesh_refs(s.nodes.data(), uint32_t(s.nodes.size()), s.mesh_instances.data(), uint32_t(s.mesh_instances.size()), tlas_root,
                                     s.blas_root4.empty() ? nullptr : s.blas_root4.data(), c->mesh_refs);
    c->nodes_used = uint32_t(s.nodes.size());
    c->have_scene = true;
    SceneView &v = c->sc;
    v.nodes = s.nodes.data(), v.tris = s.tris.data(), v.tri_pitch = 3, v.all_solid = 0, v.tri_indices = s.tri_indices.data();
    v.nodes4 = s.nodes4.empty() ? nullptr : s.nodes4.data(), v.blas_root4 = s.blas_root4.empty() ? nullptr : s.blas_root4.data();
    v.nodes8 = s.nodes8.empty() ? nullptr : s.nodes8.data();
    v.tri_materials = s.tri_materials.data(), v.materials = s.materials.data(), v.vertices = s.vertices.data();
    v.vtx_indices = s.vtx_indices.data(), v.mesh_instances = s.mesh_instances.data(), v.lights = s.lights.data();
    v.light_children = s.light_children.data();
    v.light_tri_geom = s.light_tri_geom.data();
    v.tri_verts = s.tri_verts.data();
    v.tri_bitangents = s.tri_bitangents.data();
    v.env_qtree = reinterpret_cast<const float4 *>(s.env_qtree.data());
    for (int lod = 0, off = 0; lod < 16; ++lod) {
        v.env_qtree_offset[lod] = uint32_t(off);
        if (lod < d->env.qtree_levels) {
            off += 1 << (2 * (d->env.qtree_levels - 1 - lod));
        }
    }
    v.li_indices = s.li_indices.data(), v.light_cwnodes = s.light_cwnodes.data(), v.textures = s.textures.data();
    v.texels = s.texels.data();
    memcpy(v.tex_table, d->tex_table, sizeof(v.tex_table));
    v.tex_flags = d->texture_flags;
    v.li_indices_count = d->li_indices_count;
    v.light_cwnodes_count = d->light_cwnodes_count;
    v.visible_lights_count = d->visible_lights_count;
    v.blocker_lights_count = d->blocker_lights_count;
    v.tlas_root = tlas_root;
    v.env = d->env;
    v.sky = s.sky_view(*d);
    memcpy(c->bbox_min, d->bbox_min, 12), memcpy(c->bbox_max, d->bbox_max, 12);
    return 0;
}
// rayhip_scene_update_instances on the host build: the same planning (scene_update.h), the same builder functions as host
// loops (lbvh.h: build_host), the top level appended behind the uploaded nodes.  Returns 0 / 1 / 2 like the library.
HS_API int hostsim_scene_update_instances(hostsim_ctx *c, const rayhip_scene_desc *d_in) {
    if (!c->have_scene) {
        g_err = "hostsim_scene_update_instances before hostsim_scene_upload";
        return 2;
    }
    const rayhip_layout::AlignedDesc aligned(*d_in);
    const rayhip_scene_desc *d = &aligned.d;
    HostScene &s = c->hs;
    if (d->vertices_count != s.vertices.size() || d->vtx_indices_count != s.vtx_indices.size() || d->tri_materials_count != s.tri_materials.size() ||
        d->materials_count != s.materials.size()) {
        g_err = "geometry arrays changed size since the last upload";
        return 2;
    }
    rayhip_update::Plan up;
    if (const int rc = rayhip_update::plan(*d, c->mesh_refs, up, g_err)) {
        return rc;
    }
    {
        rayhip_scene_desc lights_only = *d;
        lights_only.mesh_instances = up.instances.data();
        if (!rayhip_validate::validate_lights(lights_only, g_err)) {
            return 1;
        }
    }
    uint32_t tlas_root = 0xffffffffu;
    s.nodes.resize(c->nodes_used);
    if (!up.live.empty()) {
        const std::vector<uint32_t> group(up.live.size(), 0);
        rayhip_lbvh::Output tlas = rayhip_lbvh::build_host(rayhip_update::top_level_input(up, group));
        tlas_root = rayhip_update::relocate_top_level(tlas, up, c->nodes_used);
        s.nodes.insert(s.nodes.end(), tlas.nodes.begin(), tlas.nodes.end());
    }
    s.mesh_instances = up.instances;
    if (!s.blas_root4.empty()) {
        s.blas_root4 = up.root4;
    }
    s.lights.assign(d->lights, d->lights + d->lights_count);
    s.li_indices.assign(d->li_indices, d->li_indices + d->li_indices_count);
    s.light_cwnodes.assign(d->light_cwnodes, d->light_cwnodes + d->light_cwnodes_count);
    s.light_children.resize(size_t(d->light_cwnodes_count) * LIGHT_CHILDREN_STRIDE);
    for (uint32_t n = 0; n < d->light_cwnodes_count; ++n) {
        fill_light_children(d->light_cwnodes[n], &s.light_children[size_t(n) * LIGHT_CHILDREN_STRIDE]);
    }
    s.light_tri_geom.assign(size_t(d->lights_count) * 4, mkfloat4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t k = 0; k < d->li_indices_count; ++k) {
        const uint32_t i = d->li_indices[k];
        fill_light_tri_geom(d->lights[i], d->mesh_instances, d->vtx_indices, d->vertices, &s.light_tri_geom[size_t(i) * 4]);
    }
    s.env_qtree.assign(d->env_qtree, d->env_qtree + d->env_qtree_count);
    SceneView &v = c->sc;
    v.nodes = s.nodes.data(), v.mesh_instances = s.mesh_instances.data(), v.lights = s.lights.data();
    v.blas_root4 = s.blas_root4.empty() ? nullptr : s.blas_root4.data();
    v.light_children = s.light_children.data(), v.light_tri_geom = s.light_tri_geom.data();
    v.li_indices = s.li_indices.data(), v.light_cwnodes = s.light_cwnodes.data();
    v.env_qtree = reinterpret_cast<const float4 *>(s.env_qtree.data());
    for (int lod = 0, off = 0; lod < 16; ++lod) {
        v.env_qtree_offset[lod] = uint32_t(off);
        if (lod < d->env.qtree_levels) {
            off += 1 << (2 * (d->env.qtree_levels - 1 - lod));
        }
    }
    v.li_indices_count = d->li_indices_count, v.light_cwnodes_count = d->light_cwnodes_count;
    v.visible_lights_count = d->visible_lights_count, v.blocker_lights_count = d->blocker_lights_count;
    v.tlas_root = tlas_root;
    v.env = d->env;
    v.sky = s.sky_view(*d); // (the suns are lights: they may have moved)
    memcpy(c->bbox_min, d->bbox_min, 12), memcpy(c->bbox_max, d->bbox_max, 12);
    return 0;
}
HS_API int hostsim_scene_update_instances_blob(hostsim_ctx *c, const void *blob, size_t size, rayhip_camera *out_cam) {
    rayhip_scene_desc d;
    const float *ft = nullptr;
    int ftn = 0;
    if (!rayhip_blob::deserialize(blob, size, d, *out_cam, &ft, &ftn, g_err)) {
        return 1;
    }
    return hostsim_scene_update_instances(c, &d);
}

HS_API int hostsim_scene_bvh_width(hostsim_ctx *c) { return !c->have_scene ? 0 : c->wide ? c->wide : 2; }

HS_API int hostsim_set_filter_table(hostsim_ctx *c, const float *t, int count) {
    c->filter_table.assign(t, t + count);
    return 0;
}

HS_API int hostsim_set_tonemap_lut(hostsim_ctx *c, int view_transform, const uint32_t *lut, int dims) {
    c->tonemap_lut.assign(lut, lut + size_t(dims) * dims * dims);
    c->lut_transform = view_transform, c->lut_dims = dims;
    return 0;
}

// the baked sky map: the host build of sky_bake_texel (rt_sky.h) over the uploaded scene's sky, and -- to compare it with -- the texels of the
// environment map the scene came with (baked by the reference's CalcSkyEnvTexture when the scene was finalized)
HS_API int hostsim_bake_sky(hostsim_ctx *c, int w, int h, uint32_t *out_rgbe8) {
    if (!c->have_scene || c->sc.sky.desc == nullptr) {
        g_err = "hostsim_bake_sky: the uploaded scene has no physical sky";
        return 1;
    }
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            out_rgbe8[size_t(y) * size_t(w) + size_t(x)] = sky_bake_texel(c->sc.sky, c->sc.lights, x, y, w, h);
        }
    }
    return 0;
}
HS_API int hostsim_env_map_texels(hostsim_ctx *c, uint32_t *out, size_t capacity, int out_wh[2]) {
    if (!c->have_scene || c->sc.env.env_map == 0xffffffffu) {
        g_err = "hostsim_env_map_texels: the uploaded scene has no environment map";
        return 1;
    }
    const uint32_t handle = c->sc.env.env_map;
    const rayhip_texture &t = c->sc.textures[c->sc.tex_table[handle >> 28] + (handle & 0x00ffffffu)];
    out_wh[0] = int(t.width[0]), out_wh[1] = int(t.height[0]);
    const size_t n = size_t(t.width[0]) * size_t(t.height[0]);
    if (n > capacity) {
        g_err = "hostsim_env_map_texels: buffer too small";
        return 1;
    }
    memcpy(out, c->sc.texels + t.offset[0], n * sizeof(uint32_t));
    return 0;
}

HS_API int hostsim_scene_upload_blob(hostsim_ctx *c, const void *blob, size_t size, rayhip_camera *out_cam) {
    rayhip_scene_desc d;
    const float *ft = nullptr;
    int ftn = 0;
    std::string err;
    rayhip_blob::Extras extras;
    if (!rayhip_blob::deserialize(blob, size, d, *out_cam, &ft, &ftn, err, &extras)) {
        g_err = err;
        return 1;
    }
    if (extras.tonemap_lut && out_cam->view_transform != 0) {
        hostsim_set_tonemap_lut(c, out_cam->view_transform, extras.tonemap_lut, extras.tonemap_lut_dims);
    }
    if (hostsim_scene_upload(c, &d)) {
        return 1;
    }
    if (ft) {
        hostsim_set_filter_table(c, ft, ftn);
    }
    return 0;
}

template <class Stack>
static void trace_closest(const hostsim_ctx *c, const TraceParams &tp, Ray &r, Hit &h, Stack &st, TravCount *cnt) {
    if (c->wide == 8) {
        intersect_scene_closest<8>(c->sc, tp, r, h, st, cnt);
    } else if (c->wide == 4) {
        intersect_scene_closest<4>(c->sc, tp, r, h, st, cnt);
    } else {
        intersect_scene_closest<0>(c->sc, tp, r, h, st, cnt);
    }
}
template <class Stack>
static f3 trace_shadow(const hostsim_ctx *c, const TraceParams &tp, const ShadowRay &r, Stack &st, TravCount *cnt) {
    return c->wide == 8   ? intersect_scene_shadow<8>(c->sc, tp, r, st, cnt)
           : c->wide == 4 ? intersect_scene_shadow<4>(c->sc, tp, r, st, cnt)
                          : intersect_scene_shadow<0>(c->sc, tp, r, st, cnt);
}

static void add_counters(rayhip_trav_counters &dst, const TravCount &tc) {
    dst.rays += 1, dst.nodes += tc.nodes, dst.tris += tc.tris, dst.instances += tc.instances, dst.nodes4 += tc.nodes4;
    dst.max_stack = tc.max_stack > dst.max_stack ? tc.max_stack : dst.max_stack;
}

HS_API int hostsim_render(hostsim_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration, uint32_t flags,
                          rayhip_stats *) {
    const bool count = (flags & (RAYHIP_FLAG_COUNT_TRAVERSAL | RAYHIP_FLAG_COUNT_WIDE)) != 0;
    const int w = c->w;
    const RayGenParams rg = make_raygen_params(*cam, c->w, c->h, rect, iteration, c->shard);
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    const float mix_factor = 1.0f / float(iteration);

    std::vector<Ray> rays, next_rays;
    std::vector<Hit> hits;
    std::vector<ShadowRay> shadow;
    // K1: primary rays (CoreRef.cpp:1471-1553), skipping pixels that need no more samples
    for (int y = rect[1]; y < rect[1] + rect[3]; ++y) {
        for (int x = rect[0]; x < rect[0] + rect[2]; ++x) {
            if (!pixel_owned(c->shard, w, x, y) || c->required_samples[size_t(y) * w + x] < iteration) {
                continue;
            }
            Ray r;
            Hit h;
            generate_primary_ray(rg, c->sc.pmj, c->filter_table.data(), x, y, r, h);
            rays.push_back(r), hits.push_back(h);
        }
    }
    ArrayStack st;
    // K2: primary trace (RendererCPU.h:455)
    if (c->sc.tlas_root != 0xffffffff) {
        for (size_t i = 0; i < rays.size(); ++i) {
            TravCount tc = {};
            trace_closest(c, tp, rays[i], hits[i], st, count ? &tc : nullptr);
            if (count) {
                add_counters(c->counters[0], tc);
            }
        }
    }
    for (int bounce = 0; bounce <= int(cam->pass_settings.max_total_depth); ++bounce) {
        if (bounce > 0) {
            if (rays.empty()) {
                break;
            }
            // K2: secondary trace, hits reset to default (RendererCPU.h:533-541).  (No sort: order-free.)
            hits.assign(rays.size(), make_hit());
            for (size_t i = 0; i < rays.size(); ++i) {
                TravCount tc = {};
                trace_closest(c, tp, rays[i], hits[i], st, count ? &tc : nullptr);
                if (count) {
                    add_counters(c->counters[0], tc);
                }
                // K4 (TraceRays(..., trace_lights = true), CoreRef.cpp:4847-4849)
                if (c->sc.visible_lights_count != 0) {
                    LightStack lst;
                    intersect_area_lights(c->sc, rays[i].o, rays[i].d, rays[i].depth, hits[i], lst);
                }
            }
        }
        // K5: shade
        const ShadeParams sp = make_shade_params(*cam, iteration, bounce);
        next_rays.clear(), shadow.clear();
        std::vector<size_t> sky_rays; // ShadeSkyPrimary / ShadeSkySecondary run over these after the surface shading (RendererCPU.h:484-486, 555-557)
        for (size_t i = 0; i < rays.size(); ++i) {
            Ray nr;
            ShadowRay sr;
            const ShadeResult res = shade_surface(c->sc, sp, hits[i], rays[i], nr, sr);
            if (res.defer_sky) {
                sky_rays.push_back(i);
            }
            if (bounce == 0) {
                write_primary_pixel(res, rays[i].xy, w, mix_factor, c->temp.data(), c->base_color.data(),
                                    c->depth_normals.data());
            } else {
                add_secondary_pixel(res, rays[i].xy, w, c->temp.data());
            }
            if (res.emit_secondary) {
                next_rays.push_back(nr);
            }
            if (res.emit_shadow) {
                shadow.push_back(sr);
            }
        }
        for (const size_t i : sky_rays) {
            add_sky_pixel(shade_sky_ray(c->sc, rays[i], hits[i], sp.iteration, int(sp.ps.max_total_depth), sp.limits[0]), rays[i].xy, w, c->temp.data());
        }
        // K3: shadow rays
        const float limit = shadow_clamp_limit(*cam, bounce);
        for (size_t i = 0; i < shadow.size(); ++i) {
            TravCount tc = {};
            f3 rc = trace_shadow(c, tp, shadow[i], st, count ? &tc : nullptr);
            if (count) {
                add_counters(c->counters[1], tc);
            }
            if (c->sc.blocker_lights_count != 0) { // CoreRef.cpp:4868-4870
                LightStack lst;
                rc *= intersect_area_lights_shadow(c->sc, shadow[i], lst);
            }
            add_shadow_pixel(rc, limit, shadow[i].xy, w, c->temp.data());
        }
        rays.swap(next_rays);
    }
    // K10+K11
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        g_err = "view transform needs its look-up table";
        return 1;
    }
    AccumParams ap = make_accum_params(*cam, w, rect, iteration, c->shard);
    ap.lut = c->tonemap_lut.data(), ap.lut_dims = c->lut_dims;
    for (int y = rect[1]; y < rect[1] + rect[3]; ++y) {
        for (int x = rect[0]; x < rect[0] + rect[2]; ++x) {
            if (!pixel_owned(c->shard, w, x, y)) {
                continue;
            }
            accumulate_pixel(ap, x, y, c->temp.data() + (size_t(y) * w + x), c->temp.data() + (size_t(y) * w + x), c->full.data(), c->half.data(), c->raw.data(), c->final_.data(),
                             c->required_samples.data());
        }
    }
    return 0;
}

HS_API int hostsim_readback(hostsim_ctx *c, int which, float *dst, int pitch_px) {
    const std::vector<float4> *src = nullptr;
    switch (which) {
    case RAYHIP_BUF_FINAL:
        src = &c->final_;
        break;
    case RAYHIP_BUF_RAW:
        src = &c->raw;
        break;
    case RAYHIP_BUF_BASE_COLOR:
        src = &c->base_color;
        break;
    case RAYHIP_BUF_DEPTH_NORMALS:
        src = &c->depth_normals;
        break;
    default:
        g_err = "bad buffer id";
        return 1;
    }
    for (int y = 0; y < c->h; ++y) {
        memcpy(dst + size_t(y) * pitch_px * 4, src->data() + size_t(y) * c->w, size_t(c->w) * 16);
    }
    return 0;
}
HS_API int hostsim_sync(hostsim_ctx *) { return 0; }
// the reference has no batching: iterations one by one (what rayhip_render_batch must reproduce bit for bit)
HS_API int hostsim_render_batch(hostsim_ctx *c, const rayhip_camera *cam, const int rect[4], int first_iteration, int count,
                                uint32_t flags, rayhip_stats *st) {
    for (int i = 0; i < count; ++i) {
        hostsim_render(c, cam, rect, first_iteration + i, flags, st);
    }
    return 0;
}

HS_API int hostsim_max_batch(hostsim_ctx *) { return 64; }
HS_API int hostsim_reserve_batch(hostsim_ctx *, int) { return 0; }
// RendererBase::DenoiseImage(region) with the host build of rt_denoise.h (the variance estimate lives in c->temp, as in
// the reference)
HS_API int hostsim_denoise_nlm(hostsim_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration) {
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        g_err = "view transform needs its look-up table";
        return 1;
    }
    DenoiseParams p;
    p.w = c->w, p.h = c->h;
    for (int i = 0; i < 4; ++i) {
        p.rect[i] = rect[i];
    }
    p.ext_w = rect[2] + 2 * NLM_EXT_RADIUS, p.ext_h = rect[3] + 2 * NLM_EXT_RADIUS;
    p.iteration = iteration;
    AccumParams tone = make_accum_params(*cam, c->w, rect, iteration, c->shard);
    tone.lut = c->tonemap_lut.data(), tone.lut_dims = c->lut_dims;
    p.variance_threshold = tone.variance_threshold;
    const float4 z = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    std::vector<float4> tm(size_t(p.ext_w) * p.ext_h, z), var_h(tm.size(), z), var(tm.size(), z);
    for (int y = 0; y < p.ext_h; ++y) {
        for (int x = 0; x < p.ext_w; ++x) {
            nlm_prepare_h(p, x, y, c->full.data(), c->temp.data(), tm.data(), var_h.data());
        }
    }
    for (int y = 4; y < p.ext_h - 4; ++y) {
        for (int x = 4; x < p.ext_w - 4; ++x) {
            nlm_prepare_v(p, x, y, var_h.data(), var.data());
        }
    }
    for (int y = 0; y < rect[3]; ++y) {
        for (int x = 0; x < rect[2]; ++x) {
            const f4 nlm = nlm_filter_pixel(p, x, y, c->base_color.data(), c->depth_normals.data(), [&](const int ex, const int ey, const int which) {
                return ld4((which ? var : tm)[size_t(ey) * p.ext_w + ex]);
            });
            const int idx = (rect[1] + y) * c->w + (rect[0] + x);
            nlm_finish_pixel(p, tone, idx, ld4(var[size_t(NLM_EXT_RADIUS + y) * p.ext_w + (NLM_EXT_RADIUS + x)]), nlm, c->raw.data(),
                             c->final_.data(), c->required_samples.data());
        }
    }
    return 0;
}

HS_API int hostsim_set_shard(hostsim_ctx *c, int tile, int shard_count, int shard_index) {
    c->shard = Shard{tile, shard_count, shard_index};
    return 0;
}

// the exchange of a tile-sharded render with the caller's transport (twins of rayhip_owned_bytes / _export_owned / _import_owned /
// _finish_import over the same slot <-> pixel mapping, rt_base.h: shard_slot_pixel); images: running mean, base colour,
// depth-normals (this build keeps no separate variance image)
namespace {
int hs_selected(uint32_t what, std::vector<float4> *out[3], hostsim_ctx *c) {
    if (what == 0) {
        what = 7u;
    }
    int n = 0;
    std::vector<float4> *all[3] = {&c->full, &c->base_color, &c->depth_normals};
    for (int k = 0; k < 3; ++k) {
        if (what & (1u << k)) {
            out[n++] = all[k];
        }
    }
    return n;
}
} // namespace
HS_API size_t hostsim_owned_bytes(hostsim_ctx *c, uint32_t what, int nranks, int rank) {
    std::vector<float4> *sel[3];
    const int n_sel = hs_selected(what, sel, c);
    const ShardTiles st = shard_tiles(c->w, c->h, c->shard.tile);
    return size_t(shard_owned_tiles(st.total, nranks, rank)) * size_t(c->shard.tile) * size_t(c->shard.tile) * 16u * size_t(n_sel);
}
HS_API int hostsim_export_owned(hostsim_ctx *c, uint32_t what, void *dst, size_t capacity) {
    std::vector<float4> *sel[3];
    const int n_sel = hs_selected(what, sel, c);
    if (hostsim_owned_bytes(c, what, c->shard.count, c->shard.index) > capacity) {
        g_err = "hostsim_export_owned: destination too small";
        return 1;
    }
    const ShardTiles st = shard_tiles(c->w, c->h, c->shard.tile);
    const int n = shard_owned_tiles(st.total, c->shard.count, c->shard.index) * c->shard.tile * c->shard.tile;
    float4 *out = static_cast<float4 *>(dst);
    for (int k = 0; k < n_sel; ++k) {
        for (int i = 0; i < n; ++i) {
            int x, y;
            out[size_t(k) * n + i] = shard_slot_pixel(c->shard, c->w, c->h, i, x, y) ? (*sel[k])[size_t(y) * c->w + x] : float4{0, 0, 0, 0};
        }
    }
    return 0;
}
HS_API int hostsim_import_owned(hostsim_ctx *c, uint32_t what, int from_rank, const void *src, size_t bytes) {
    std::vector<float4> *sel[3];
    const int n_sel = hs_selected(what, sel, c);
    if (from_rank < 0 || from_rank >= c->shard.count || bytes < hostsim_owned_bytes(c, what, c->shard.count, from_rank)) {
        g_err = "hostsim_import_owned: bad rank or short buffer";
        return 1;
    }
    if (from_rank == c->shard.index) {
        return 0;
    }
    const Shard sender = {c->shard.tile, c->shard.count, from_rank};
    const ShardTiles st = shard_tiles(c->w, c->h, c->shard.tile);
    const int n = shard_owned_tiles(st.total, c->shard.count, from_rank) * c->shard.tile * c->shard.tile;
    const float4 *in = static_cast<const float4 *>(src);
    for (int k = 0; k < n_sel; ++k) {
        for (int i = 0; i < n; ++i) {
            int x, y;
            if (shard_slot_pixel(sender, c->w, c->h, i, x, y)) {
                (*sel[k])[size_t(y) * c->w + x] = in[size_t(k) * n + i];
            }
        }
    }
    return 0;
}
HS_API int hostsim_finish_import(hostsim_ctx *c, const rayhip_camera *cam) {
    const int rect[4] = {0, 0, c->w, c->h};
    AccumParams ap = make_accum_params(*cam, c->w, rect, 1);
    ap.lut = c->tonemap_lut.empty() ? nullptr : c->tonemap_lut.data(), ap.lut_dims = c->lut_dims;
    for (size_t i = 0; i < c->full.size(); ++i) {
        c->raw[i] = c->full[i];
        const f4 t = tonemap(ap, f4{c->full[i].x, c->full[i].y, c->full[i].z, c->full[i].w});
        c->final_[i] = float4{t.x, t.y, t.z, t.w};
    }
    return 0;
}

HS_API int hostsim_get_trav_counters(hostsim_ctx *c, rayhip_trav_counters out[2], int reset) {
    out[0] = c->counters[0], out[1] = c->counters[1];
    if (reset) {
        c->counters[0] = c->counters[1] = rayhip_trav_counters{};
    }
    return 0;
}

// ---- kernel-level hooks ---------------------------------------------------------------------------------
static Ray from_abi(const rayhip_ray &a) {
    Ray r;
    r.o = mk3(a.o), r.d = mk3(a.d), r.pdf = a.pdf, r.c = mk3(a.c);
    memcpy(r.ior, a.ior, 16);
    r.cone_width = a.cone_width, r.cone_spread = a.cone_spread, r.xy = a.xy, r.depth = a.depth;
    return r;
}
static rayhip_ray to_abi(const Ray &r) {
    rayhip_ray a;
    a.o[0] = r.o.x, a.o[1] = r.o.y, a.o[2] = r.o.z;
    a.d[0] = r.d.x, a.d[1] = r.d.y, a.d[2] = r.d.z;
    a.pdf = r.pdf;
    a.c[0] = r.c.x, a.c[1] = r.c.y, a.c[2] = r.c.z;
    memcpy(a.ior, r.ior, 16);
    a.cone_width = r.cone_width, a.cone_spread = r.cone_spread, a.xy = r.xy, a.depth = r.depth;
    return a;
}

HS_API int hostsim_k_generate_primary_rays(hostsim_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration,
                                           rayhip_ray *out_rays, rayhip_hit *out_hits, int *out_count) {
    const RayGenParams rg = make_raygen_params(*cam, c->w, c->h, rect, iteration, c->shard);
    int n = 0;
    for (int y = rect[1]; y < rect[1] + rect[3]; ++y) {
        for (int x = rect[0]; x < rect[0] + rect[2]; ++x) {
            if (c->required_samples[size_t(y) * c->w + x] < iteration) {
                continue;
            }
            Ray r;
            Hit h;
            generate_primary_ray(rg, c->sc.pmj, c->filter_table.data(), x, y, r, h);
            out_rays[n] = to_abi(r);
            out_hits[n] = rayhip_hit{h.obj_index, h.prim_index, h.t, h.u, h.v};
            ++n;
        }
    }
    *out_count = n;
    return 0;
}

HS_API int hostsim_k_intersect_closest(hostsim_ctx *c, const rayhip_camera *cam, rayhip_ray *rays, rayhip_hit *hits,
                                       int count, int iteration, uint32_t flags, rayhip_trav_counters *out_counters) {
    const int wide = (flags & RAYHIP_FLAG_COUNT_TRAVERSAL) == 0 ? c->wide : 0;
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    ArrayStack st;
    rayhip_trav_counters acc = {};
    for (int i = 0; i < count; ++i) {
        Ray r = from_abi(rays[i]);
        Hit h = {hits[i].obj_index, hits[i].prim_index, hits[i].t, hits[i].u, hits[i].v};
        TravCount tc = {};
        if (wide == 8) {
            intersect_scene_closest<8>(c->sc, tp, r, h, st, &tc);
        } else if (wide == 4) {
            intersect_scene_closest<4>(c->sc, tp, r, h, st, &tc);
        } else {
            intersect_scene_closest<0>(c->sc, tp, r, h, st, &tc);
        }
        add_counters(acc, tc);
        rays[i] = to_abi(r);
        hits[i] = rayhip_hit{h.obj_index, h.prim_index, h.t, h.u, h.v};
    }
    if (out_counters) {
        *out_counters = acc;
    }
    return 0;
}

HS_API int hostsim_k_intersect_shadow(hostsim_ctx *c, const rayhip_camera *cam, const rayhip_shadow_ray *rays, int count,
                                      int iteration, float *out_rc, rayhip_trav_counters *out_counters) {
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    ArrayStack st;
    rayhip_trav_counters acc = {};
    for (int i = 0; i < count; ++i) {
        ShadowRay r;
        r.o = mk3(rays[i].o), r.depth = rays[i].depth, r.d = mk3(rays[i].d), r.dist = rays[i].dist;
        r.c = mk3(rays[i].c), r.xy = rays[i].xy;
        TravCount tc = {};
        const f3 rc = trace_shadow(c, tp, r, st, &tc);
        add_counters(acc, tc);
        out_rc[4 * i + 0] = rc.x, out_rc[4 * i + 1] = rc.y, out_rc[4 * i + 2] = rc.z, out_rc[4 * i + 3] = 0.0f;
    }
    if (out_counters) {
        *out_counters = acc;
    }
    return 0;
}

HS_API int hostsim_k_shade(hostsim_ctx *c, const rayhip_camera *cam, int bounce, int iteration, const rayhip_ray *rays,
                           const rayhip_hit *hits, int count, float *inout_color, rayhip_ray *out_secondary, int *out_secondary_count,
                           rayhip_shadow_ray *out_shadow, int *out_shadow_count) {
    const ShadeParams sp = make_shade_params(*cam, iteration, bounce);
    float4 *color = reinterpret_cast<float4 *>(inout_color);
    std::vector<float4> base(size_t(c->w) * c->h), dn(size_t(c->w) * c->h);
    int n_sec = 0, n_sh = 0;
    for (int i = 0; i < count; ++i) {
        const Ray r = from_abi(rays[i]);
        const Hit h = {hits[i].obj_index, hits[i].prim_index, hits[i].t, hits[i].u, hits[i].v};
        Ray nr;
        ShadowRay sr;
        const ShadeResult res = shade_surface(c->sc, sp, h, r, nr, sr);
        if (bounce == 0) {
            write_primary_pixel(res, r.xy, c->w, 1.0f / float(iteration), color, base.data(), dn.data());
        } else {
            add_secondary_pixel(res, r.xy, c->w, color);
        }
        if (res.emit_secondary) {
            out_secondary[n_sec++] = to_abi(nr);
        }
        if (res.emit_shadow) {
            rayhip_shadow_ray &o = out_shadow[n_sh++];
            o.o[0] = sr.o.x, o.o[1] = sr.o.y, o.o[2] = sr.o.z, o.depth = sr.depth;
            o.d[0] = sr.d.x, o.d[1] = sr.d.y, o.d[2] = sr.d.z, o.dist = sr.dist;
            o.c[0] = sr.c.x, o.c[1] = sr.c.y, o.c[2] = sr.c.z, o.xy = sr.xy;
        }
    }
    *out_secondary_count = n_sec, *out_shadow_count = n_sh;
    return 0;
}

HS_API int hostsim_k_scrambled_rand(hostsim_ctx *c, const uint32_t *dims, const uint32_t *seeds, const int32_t *samples,
                                    int count, float *out_xy) {
    for (int i = 0; i < count; ++i) {
        const f2 r = get_scrambled_2d_rand(dims[i], seeds[i], samples[i], c->sc.pmj);
        out_xy[2 * i + 0] = r.x, out_xy[2 * i + 1] = r.y;
    }
    return 0;
}
