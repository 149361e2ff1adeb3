// scene_export.h -- gathers the flat arrays of a finalized Ray::Cpu::Scene into a rayhip_scene_desc.
//
// This is the HIP backend's counterpart of the scene_data_t gathering every CPU RenderScene does
// (reference internal/RendererCPU.h:390-413) and of SceneVK's buffer uploads (internal/SceneGPU.h:62-104).
// The reference's host-side scene code (AddMesh -> PreprocessMesh -> ConvertToBVH2, RebuildTLAS,
// RebuildLightTree: SceneCPU.cpp:342-546,928-1015,1214-1521) is reused unchanged, as SURVEY.md section 2 /
// section 8(a18) prescribes; SceneHIP derives from Cpu::Scene and calls this after Finalize.
//
// Compiles only next to the reference sources (includes internal/SceneCPU.h).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "internal/SceneCPU.h"

#include "../../include/rayhip.h"

namespace Ray {
// the five textures the reference compiles into AtmosphereRef.cpp (internal/precomputed/*.inl: `extern const` in namespace Ray)
extern const int WEATHER_TEX_RES, NOISE_3D_RES, CURL_TEX_RES, MOON_TEX_W, MOON_TEX_H, CIRRUS_TEX_RES;
extern const uint8_t __weather_tex[], __3d_noise_tex[], __curl_tex[], __moon_tex[], __cirrus_tex[];

namespace Hip {

static_assert(sizeof(rayhip_tri_accel) == sizeof(tri_accel_t), "layout");
static_assert(sizeof(rayhip_bvh2_node) == sizeof(bvh2_node_t), "layout");
static_assert(sizeof(rayhip_vertex) == sizeof(vertex_t), "layout");
static_assert(sizeof(rayhip_mesh_instance) == sizeof(mesh_instance_t), "layout");
static_assert(sizeof(rayhip_tri_mat_data) == sizeof(tri_mat_data_t), "layout");
static_assert(sizeof(rayhip_material) == sizeof(material_t), "layout");
static_assert(offsetof(rayhip_material, ior) == offsetof(material_t, ior), "layout");
static_assert(offsetof(rayhip_material, normal_map_strength_unorm) == offsetof(material_t, normal_map_strength_unorm), "layout");
static_assert(sizeof(rayhip_light) == sizeof(light_t), "layout");
static_assert(offsetof(rayhip_light, params) == offsetof(light_t, sph), "layout");
static_assert(sizeof(rayhip_light_cwbvh_node) == sizeof(light_cwbvh_node_t), "layout");
static_assert(offsetof(rayhip_light_cwbvh_node, flux) == offsetof(light_cwbvh_node_t, flux), "layout");
static_assert(sizeof(rayhip_camera) == sizeof(camera_t), "layout");
static_assert(offsetof(rayhip_camera, pass_settings) == offsetof(camera_t, pass_settings), "layout");
static_assert(offsetof(rayhip_camera, origin) == offsetof(camera_t, origin), "layout");
static_assert(sizeof(rayhip_pass_settings) == sizeof(pass_settings_t), "layout");
static_assert(sizeof(rayhip_stats) == sizeof(RendererBase::stats_t), "layout");
static_assert(sizeof(rayhip_atmosphere) == sizeof(atmosphere_params_t), "layout");
static_assert(offsetof(rayhip_atmosphere, moon_dir) == offsetof(atmosphere_params_t, moon_dir), "layout");
static_assert(offsetof(rayhip_atmosphere, ground_albedo) == offsetof(atmosphere_params_t, ground_albedo), "layout");
static_assert(offsetof(rayhip_atmosphere, cirrus_clouds_height) == offsetof(atmosphere_params_t, cirrus_clouds_height), "layout");

// ---- the block-compressed storages as they are kept ----------------------------------------------------
// TexStorageBCn<N> (TextureStorageCPU.h:364-617) keeps its images -- 4x4 blocks per mip -- in a private member and offers
// texel-wise Get only.  The explicit instantiation below hands out a pointer to that member (explicit instantiations may name
// private members, [temp.spec]/6); a maintainer integrating this backend would add a `const ImgData &image(int) const`
// accessor instead.
template <int N> struct BcImagesTag {
    friend auto bc_images_member(BcImagesTag); // defined by the instantiation of BcImagesGrant below
};
template <class Tag, auto Member> struct BcImagesGrant {
    friend auto bc_images_member(Tag) { return Member; }
};
template struct BcImagesGrant<BcImagesTag<1>, &Cpu::TexStorageBCn<1>::images_>;
template struct BcImagesGrant<BcImagesTag<2>, &Cpu::TexStorageBCn<2>::images_>;
template struct BcImagesGrant<BcImagesTag<3>, &Cpu::TexStorageBCn<3>::images_>;
template struct BcImagesGrant<BcImagesTag<4>, &Cpu::TexStorageBCn<4>::images_>;

// Owns the converted texture pool; every other pointer in `desc` aliases the scene's own storage and stays
// valid until the next scene mutation.
struct FlatScene {
    rayhip_scene_desc desc = {};
    std::vector<rayhip_texture> textures;
    std::vector<uint32_t> texels;
    std::vector<float> env_qtree;
    rayhip_sky sky = {};
};

// Access to Cpu::Scene / SceneCommon protected members through a derived class (legal: the member pointers
// are formed inside a member of the derived class and have type `T Cpu::Scene::*`).
class SceneAccess : public Cpu::Scene {
    SceneAccess() = delete;

    template <typename S, int N> static void export_storage(const S &st, FlatScene &out) {
        for (int i = 0; i < st.img_count(); ++i) {
            rayhip_texture t = {};
            int prev_res[2] = {-1, -1};
            for (int lod = 0; lod < NUM_MIP_LEVELS; ++lod) {
                int res[2];
                st.GetIRes(i, lod, res);
                t.width[lod] = uint32_t(res[0]), t.height[lod] = uint32_t(res[1]);
                if (lod && res[0] == prev_res[0] && res[1] == prev_res[1]) {
                    // the storage aliases missing mip levels to the last real one (TextureStorageCPU.cpp:244-249)
                    t.offset[lod] = t.offset[lod - 1];
                    continue;
                }
                t.offset[lod] = uint32_t(out.texels.size());
                for (int y = 0; y < res[1]; ++y) {
                    for (int x = 0; x < res[0]; ++x) {
                        const auto c = st.Get(i, x, y, lod);
                        // same channel replication as TexStorageSwizzled::Fetch (TextureStorageCPU.h:293-305)
                        uint32_t v[4];
                        for (int k = 0; k < N; ++k) {
                            v[k] = c.v[k];
                        }
                        for (int k = N; k < 4; ++k) {
                            v[k] = v[N - 1];
                        }
                        out.texels.push_back(v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24));
                    }
                }
                prev_res[0] = res[0], prev_res[1] = res[1];
            }
            out.textures.push_back(t);
        }
    }

    // a block-compressed storage as it is: the 4x4 blocks of every mip, copied word for word (RAYHIP_TEX_RAW_BC)
    template <int N> static void export_blocks(const Cpu::TexStorageBCn<N> &st, FlatScene &out) {
        const auto &images = st.*bc_images_member(BcImagesTag<N>{});
        const int block_bytes = (N == 4 || N == 2) ? 16 : 8;
        for (int i = 0; i < st.img_count(); ++i) {
            const auto &img = images[size_t(i)];
            rayhip_texture t = {};
            for (int lod = 0; lod < NUM_MIP_LEVELS; ++lod) {
                t.width[lod] = uint32_t(img.res[lod][0]), t.height[lod] = uint32_t(img.res[lod][1]);
                if (lod && img.lod_offsets[lod] == img.lod_offsets[lod - 1]) {
                    t.offset[lod] = t.offset[lod - 1]; // a missing level aliases the last real one (TextureStorageCPU.cpp:349-353)
                    continue;
                }
                const size_t bytes = size_t(img.res_in_tiles[lod][0]) * size_t(img.res_in_tiles[lod][1]) * size_t(block_bytes);
                t.offset[lod] = uint32_t(out.texels.size());
                out.texels.resize(out.texels.size() + bytes / 4);
                memcpy(&out.texels[t.offset[lod]], &img.pixels[size_t(img.lod_offsets[lod])], bytes);
            }
            out.textures.push_back(t);
        }
    }

  public:
    static const camera_t &CurrentCamera(const Cpu::Scene &_s) {
        const auto &s = static_cast<const SceneAccess &>(_s);
        return s.cams_[s.current_cam_._index];
    }
    static std::shared_timed_mutex &Mutex(const Cpu::Scene &_s) { return static_cast<const SceneAccess &>(_s).mtx_; }

    // Caller must hold at least a shared lock on the scene.
    // `with_textures` = false: everything but the (decoded) texture pool -- what rayhip_scene_update_instances reads
    // `raw_blocks`: the block-compressed storages are handed over as blocks and decoded per fetch on the device (a quarter to
    // an eighth of the footprint); false = decoded here, texel by texel, with the reference's own TexStorageBCn::Get
    // (RAY_HIP_DECODE_BC=1 selects the second form: A/B measurements and tests)
    static bool RawBlocksByDefault() {
        const char *e = getenv("RAY_HIP_DECODE_BC");
        return !(e && e[0] == '1');
    }
    // the physical sky: what the analytic evaluation of narrow rays (rayhip_sky; AtmosphereRef.cpp: ShadeSky) and the bake of the sky map need
    static void ExportSkyMembers(const SceneAccess &s, FlatScene &out, rayhip_scene_desc &d) {
        const environment_t &e = s.env_;
        d.sky = nullptr, d.sky_count = 0;
        if (e.sky_map_spread_angle > 0.0f) {
            rayhip_sky &k = out.sky;
            k = {};
            memcpy(&k.atmosphere, &e.atmosphere, sizeof(k.atmosphere));
            k.transmittance_lut_w = SKY_TRANSMITTANCE_LUT_W, k.transmittance_lut_h = SKY_TRANSMITTANCE_LUT_H;
            k.multiscatter_lut_res = s.sky_multiscatter_lut_.empty() ? 0 : SKY_MULTISCATTER_LUT_RES;
            k.weather_res = WEATHER_TEX_RES, k.noise3d_res = NOISE_3D_RES, k.curl_res = CURL_TEX_RES;
            k.moon_w = MOON_TEX_W, k.moon_h = MOON_TEX_H, k.cirrus_res = CIRRUS_TEX_RES;
            d.sky = &k, d.sky_count = 1;
            d.sky_transmittance_lut = s.sky_transmittance_lut_.data(), d.sky_transmittance_lut_count = uint32_t(s.sky_transmittance_lut_.size());
            d.sky_multiscatter_lut = s.sky_multiscatter_lut_.data(), d.sky_multiscatter_lut_count = uint32_t(s.sky_multiscatter_lut_.size());
            d.sky_dir_lights = s.dir_lights_.data(), d.sky_dir_lights_count = uint32_t(s.dir_lights_.size());
            d.sky_weather_tex = __weather_tex, d.sky_weather_tex_count = uint32_t(3 * WEATHER_TEX_RES * WEATHER_TEX_RES);
            d.sky_noise3d_tex = __3d_noise_tex, d.sky_noise3d_tex_count = uint32_t(NOISE_3D_RES * NOISE_3D_RES * NOISE_3D_RES);
            d.sky_curl_tex = __curl_tex, d.sky_curl_tex_count = uint32_t(3 * CURL_TEX_RES * CURL_TEX_RES);
            d.sky_moon_tex = __moon_tex, d.sky_moon_tex_count = uint32_t(3 * MOON_TEX_W * MOON_TEX_H);
            d.sky_cirrus_tex = __cirrus_tex, d.sky_cirrus_tex_count = uint32_t(2 * CIRRUS_TEX_RES * CIRRUS_TEX_RES);
        }
    }
    // ... for rayhip_bake_sky: the sky members + the light array, nothing else (called from SceneHIP::Finalize with the scene locked)
    static void ExportSkyForBake(const Cpu::Scene &_s, FlatScene &out) {
        const auto &s = static_cast<const SceneAccess &>(_s);
        rayhip_scene_desc &d = out.desc;
        d = {};
        d.struct_size = uint32_t(sizeof(rayhip_scene_desc));
        d.lights = reinterpret_cast<const rayhip_light *>(s.lights_.data());
        d.lights_count = s.lights_.capacity();
        d.env.sky_map_spread_angle = s.env_.sky_map_spread_angle;
        ExportSkyMembers(s, out, d);
    }

    static void Export(const Cpu::Scene &_s, FlatScene &out, const bool with_textures = true, const bool raw_blocks = RawBlocksByDefault()) {
        // NOTE: the cast never touches SceneAccess-specific state (there is none); it only names the members
        const auto &s = static_cast<const SceneAccess &>(_s);
        if (s.use_wide_bvh_) {
            throw std::runtime_error("SceneHIP needs the 2-wide BVH (create the scene with use_wide_bvh=false)");
        }
        rayhip_scene_desc &d = out.desc;
        d = {};
        d.struct_size = uint32_t(sizeof(rayhip_scene_desc));
#define SPARSE(field, member, type)                                                                                    \
    d.field = reinterpret_cast<const type *>(s.member.data());                                                        \
    d.field##_count = s.member.capacity();
        SPARSE(nodes, nodes_, rayhip_bvh2_node)
        SPARSE(tris, tris_, rayhip_tri_accel)
        SPARSE(tri_indices, tri_indices_, uint32_t)
        SPARSE(tri_materials, tri_materials_, rayhip_tri_mat_data)
        SPARSE(materials, materials_, rayhip_material)
        SPARSE(vertices, vertices_, rayhip_vertex)
        SPARSE(vtx_indices, vtx_indices_, uint32_t)
        SPARSE(mesh_instances, mesh_instances_, rayhip_mesh_instance)
        SPARSE(lights, lights_, rayhip_light)
#undef SPARSE
        d.li_indices = s.li_indices_.data();
        d.li_indices_count = uint32_t(s.li_indices_.size());
        d.light_cwnodes = reinterpret_cast<const rayhip_light_cwbvh_node *>(s.light_cwnodes_.data());
        d.light_cwnodes_count = uint32_t(s.light_cwnodes_.size());

        if (with_textures) {
            out.textures.clear(), out.texels.clear();
            d.tex_table[0] = uint32_t(out.textures.size());
            export_storage<Cpu::TexStorageRGBA, 4>(s.tex_storage_rgba_, out);
            d.tex_table[1] = uint32_t(out.textures.size());
            export_storage<Cpu::TexStorageRGB, 3>(s.tex_storage_rgb_, out);
            d.tex_table[2] = uint32_t(out.textures.size());
            export_storage<Cpu::TexStorageRG, 2>(s.tex_storage_rg_, out);
            d.tex_table[3] = uint32_t(out.textures.size());
            export_storage<Cpu::TexStorageR, 1>(s.tex_storage_r_, out);
            // Block-compressed storages (settings_t::use_tex_compression, the reference's default; eTextureFormat::BC1..BC5
            // inputs): kept as blocks (RAYHIP_TEX_RAW_BC; the device decodes the texel it fetches, rt_texture.h), or --
            // raw_blocks = false -- decoded here with the reference's own TexStorageBCn::Get (TextureStorageCPU.h:384-544).
            // Either way the values a CPU fetch returns, hence the same images as the reference renders from the compressed data.
            d.tex_table[4] = uint32_t(out.textures.size());
            raw_blocks ? export_blocks<3>(s.tex_storage_bc1_, out) : export_storage<Cpu::TexStorageBCn<3>, 3>(s.tex_storage_bc1_, out);
            d.tex_table[5] = uint32_t(out.textures.size());
            raw_blocks ? export_blocks<4>(s.tex_storage_bc3_, out) : export_storage<Cpu::TexStorageBCn<4>, 4>(s.tex_storage_bc3_, out);
            d.tex_table[6] = uint32_t(out.textures.size());
            raw_blocks ? export_blocks<1>(s.tex_storage_bc4_, out) : export_storage<Cpu::TexStorageBCn<1>, 1>(s.tex_storage_bc4_, out);
            d.tex_table[7] = uint32_t(out.textures.size());
            raw_blocks ? export_blocks<2>(s.tex_storage_bc5_, out) : export_storage<Cpu::TexStorageBCn<2>, 2>(s.tex_storage_bc5_, out);
            d.texture_flags = raw_blocks ? RAYHIP_TEX_RAW_BC : 0u;
        } else {
            out.textures.clear(), out.texels.clear();
        }
        d.textures = out.textures.data();
        d.textures_count = uint32_t(out.textures.size());
        d.texels = out.texels.data();
        d.texels_count = uint32_t(out.texels.size());

        const environment_t &e = s.env_;
        out.env_qtree.clear();
        for (int lod = 0; lod < e.qtree_levels; ++lod) {
            const size_t quads = size_t(1) << (2 * (e.qtree_levels - 1 - lod));
            if (s.env_map_qtree_.mips[lod].size() < quads) {
                throw std::runtime_error("SceneHIP: unexpected env-map quadtree layout");
            }
            const float *src = reinterpret_cast<const float *>(s.env_map_qtree_.mips[lod].data());
            out.env_qtree.insert(out.env_qtree.end(), src, src + quads * 4);
        }
        d.env_qtree = out.env_qtree.data();
        d.env_qtree_count = uint32_t(out.env_qtree.size());
        memcpy(d.env.env_col, e.env_col, 12);
        d.env.env_map = e.env_map;
        memcpy(d.env.back_col, e.back_col, 12);
        d.env.back_map = e.back_map;
        d.env.env_map_rotation = e.env_map_rotation;
        d.env.back_map_rotation = e.back_map_rotation;
        d.env.light_index = e.light_index;
        d.env.sky_map_spread_angle = e.sky_map_spread_angle;
        d.env.qtree_levels = e.qtree_levels;

        ExportSkyMembers(s, out, d);

        d.tlas_root = s.tlas_root_;
        d.visible_lights_count = s.visible_lights_count_;
        d.blocker_lights_count = s.blocker_lights_count_;
        s.GetBounds(d.bbox_min, d.bbox_max);
    }
};

} // namespace Hip
} // namespace Ray
