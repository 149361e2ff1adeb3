// RendererHIP.cpp -- Ray::RendererBase / Ray::SceneBase implementation on top of the librayhip C ABI.
//
// This is the reference-side half of the drop-in: it would live in the reference tree as
// internal/RendererHIP.{h,cpp} + internal/SceneHIP.h next to RendererVK.cpp / SceneVK.h, and like them it
// contains no kernel code -- only the RendererBase plumbing (RendererBase.h:138-252).  It includes no HIP
// header: everything device-side happens behind include/rayhip.h.
//
// SceneHIP reuses the reference's host-side scene code as is ("host side reused as-is by
// SceneHIP"): it IS a Cpu::Scene built with the 2-wide BVH (use_wide_bvh = false, like the Reference backend,
// RendererCPU.h:368-371) whose flat arrays are uploaded after every mutation.
#include "RendererHIP.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "Log.h"
#include "internal/CDFUtils.h"
#include "internal/Core.h"
#include "internal/SceneCPU.h"
#include "internal/UNetFilter.h"

#include "../../include/rayhip.h"
#include "../csrc/scene_blob.h"
#include "scene_export.h"

namespace Ray {
// the reference's precomputed view-transform tables (internal/TonemapRef.cpp:4-26), indexed by eViewTransform
extern const int LUT_DIMS;
extern const uint32_t *transform_luts[];

namespace Hip {

// Versions come from ONE process-wide counter: every Scene instance and every mutation gets an id no other (scene, state)
// pair ever had, so a renderer's "is what I uploaded still current?" test cannot be fooled by a new scene that the
// allocator placed at the address of a deleted one with the same number of mutator calls behind it.
inline uint64_t next_scene_version() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1);
}

class Scene final : public Cpu::Scene {
    // `version_`: any change of an uploaded array.  `geometry_version_`: changes of what rayhip_scene_update_instances
    // cannot replace (meshes, materials, textures); when only the first one moved, the renderer updates the instances,
    // lights and environment on the device and rebuilds the top-level tree there instead of uploading the scene again.
    std::atomic<uint64_t> version_{next_scene_version()};
    std::atomic<uint64_t> geometry_version_{version_.load()};

  public:
    Scene(ILog *log, const bool use_tex_compression)
        : Cpu::Scene(log, false /* use_wide_bvh */, use_tex_compression /* decoded at export, scene_export.h */, false) {}

    uint64_t version() const { return version_.load(); }
    uint64_t geometry_version() const { return geometry_version_.load(); }

// every mutator that changes an uploaded array bumps the version (cameras are passed per RenderScene call)
#define BUMP(ret, name, params, args)                                                                                  \
    ret name params override {                                                                                        \
        version_ = next_scene_version();                                                                               \
        return Cpu::Scene::name args;                                                                                  \
    }
#define BUMP_GEOMETRY(ret, name, params, args)                                                                         \
    ret name params override {                                                                                        \
        geometry_version_ = version_ = next_scene_version();                                                           \
        return Cpu::Scene::name args;                                                                                  \
    }
    BUMP(void, SetEnvironment, (const environment_desc_t &env), (env))
    BUMP_GEOMETRY(TextureHandle, AddTexture, (const tex_desc_t &t), (t))
    BUMP_GEOMETRY(void, RemoveTexture, (const TextureHandle t), (t))
    BUMP_GEOMETRY(MaterialHandle, AddMaterial, (const shading_node_desc_t &m), (m))
    BUMP_GEOMETRY(MaterialHandle, AddMaterial, (const principled_mat_desc_t &m), (m))
    BUMP_GEOMETRY(void, RemoveMaterial, (const MaterialHandle m), (m))
    BUMP_GEOMETRY(MeshHandle, AddMesh, (const mesh_desc_t &m), (m))
    BUMP_GEOMETRY(void, RemoveMesh, (MeshHandle m), (m))
    BUMP(LightHandle, AddLight, (const directional_light_desc_t &l), (l))
    BUMP(LightHandle, AddLight, (const sphere_light_desc_t &l), (l))
    BUMP(LightHandle, AddLight, (const spot_light_desc_t &l), (l))
    BUMP(LightHandle, AddLight, (const rect_light_desc_t &l, const float *xform), (l, xform))
    BUMP(LightHandle, AddLight, (const disk_light_desc_t &l, const float *xform), (l, xform))
    BUMP(LightHandle, AddLight, (const line_light_desc_t &l, const float *xform), (l, xform))
    BUMP(void, RemoveLight, (LightHandle l), (l))
    BUMP(MeshInstanceHandle, AddMeshInstance, (const mesh_instance_desc_t &mi), (mi))
    BUMP(void, SetMeshInstanceTransform, (MeshInstanceHandle mi, const float *xform), (mi, xform))
    BUMP(void, RemoveMeshInstance, (MeshInstanceHandle mi), (mi))
#undef BUMP_GEOMETRY
#undef BUMP

    // The sky environment map is baked ON THE DEVICE when the scene belongs to a renderer (round 5): `sky_baker_` is rayhip_bake_sky on the
    // renderer's root context.  What follows is Cpu::Scene::Finalize (SceneCPU.cpp:882-926) with PrepareSkyEnvMap_nolock (:1017-1056) spelt
    // out and its CalcSkyEnvTexture call (the host loop over IntegrateScattering) replaced -- as the reference's own GPU scene replaces it by a
    // compute pass (SceneGPU.h:1697-1768).  A scene without a renderer (Hip::CreateScene: tools that build blobs on a machine without a GPU), or
    // RAY_HIP_SKY_BAKE_ON_HOST=1, keeps the reference's host loop.
    std::function<bool(const rayhip_scene_desc &, int, int, uint32_t *)> sky_baker_;
    const char *sky_baked_on_ = "none";

  public:
    void set_sky_baker(std::function<bool(const rayhip_scene_desc &, int, int, uint32_t *)> f) { sky_baker_ = std::move(f); }
    const char *sky_baked_on() const { return sky_baked_on_; }

    void Finalize(const std::function<void(int, int, ParallelForFunction &&)> &parallel_for) override {
        version_ = next_scene_version();
        const char *on_host = getenv("RAY_HIP_SKY_BAKE_ON_HOST");
        const bool physical_sky = env_.env_map != InvalidTextureHandle._index &&
                                  (env_.env_map == PhysicalSkyTexture._index || env_.env_map == physical_sky_texture_._index);
        if (!sky_baker_ || (on_host && on_host[0] == '1') || !physical_sky) {
            sky_baked_on_ = physical_sky ? "host" : "none";
            Cpu::Scene::Finalize(parallel_for);
            return;
        }
        std::unique_lock<std::shared_timed_mutex> lock(mtx_);
        // The bake comes FIRST, before anything of the scene is changed for it: the reference's host bake cannot fail at this point, the device's
        // can (out of memory, a lost device), and a throw must leave the scene as it was -- not with the old sky texture freed and still named by
        // physical_sky_texture_, the environment light erased and env_ half rewritten.
        const int res[] = {env_.envmap_resolution, env_.envmap_resolution / 2};
        std::vector<color_rgba8_t> rgbe_pixels(size_t(res[0]) * size_t(res[1]));
        static_assert(sizeof(color_rgba8_t) == sizeof(uint32_t), "RGBE texels are four bytes");
        {
            dir_lights_.clear();
            for (auto it = lights_.cbegin(); it != lights_.cend(); ++it) {
                if (it->type == LIGHT_TYPE_DIR) {
                    dir_lights_.push_back(it.index());
                }
            }
            const float spread_before = env_.sky_map_spread_angle;
            env_.sky_map_spread_angle = 2 * PI / float(env_.envmap_resolution); // (the exported description says "a physical sky" through it)
            FlatScene sky_only;
            SceneAccess::ExportSkyForBake(*this, sky_only);
            if (!sky_baker_(sky_only.desc, res[0], res[1], reinterpret_cast<uint32_t *>(rgbe_pixels.data()))) {
                env_.sky_map_spread_angle = spread_before;
                throw std::runtime_error(std::string("SceneHIP::Finalize: the sky bake failed on the device: ") + rayhip_last_error());
            }
            sky_baked_on_ = "device";
        }
        if (env_map_light_ != InvalidLightHandle) {
            lights_.Erase(env_map_light_._block);
        }
        env_map_qtree_ = {};
        env_.qtree_levels = 0;
        env_.light_index = 0xffffffff;
        env_.env_map_rotation = 0.0f;
        if (env_.back_map == env_.env_map) {
            env_.back_map_rotation = 0.0f;
        }
        { // PrepareSkyEnvMap_nolock, with the texels that are already there
            if (physical_sky_texture_ != InvalidTextureHandle) {
                tex_storages_[physical_sky_texture_._index >> 24]->Free(physical_sky_texture_._index & 0x00ffffff);
                physical_sky_texture_ = InvalidTextureHandle;
            }
            const int index = tex_storage_rgba_.Allocate(rgbe_pixels, res, false);
            physical_sky_texture_._index = (uint32_t(0) << 28) | uint32_t(index);
            env_.env_map = physical_sky_texture_._index;
            if (env_.back_map == PhysicalSkyTexture._index) {
                env_.back_map = physical_sky_texture_._index;
            }
        }
        if (env_.importance_sample && env_.env_col[0] > 0.0f && env_.env_col[1] > 0.0f && env_.env_col[2] > 0.0f) {
            if (env_.env_map != InvalidTextureHandle._index) {
                PrepareEnvMapQTree_nolock(parallel_for);
            }
            light_t l = {}; // the environment as a light source
            l.type = LIGHT_TYPE_ENV;
            l.visible = 1;
            l.cast_shadow = 1;
            l.col[0] = l.col[1] = l.col[2] = 1.0f;
            l.ray_visibility |= RAY_TYPE_DIFFUSE_BIT;
            l.ray_visibility |= RAY_TYPE_SPECULAR_BIT;
            l.ray_visibility |= RAY_TYPE_REFR_BIT;
            const std::pair<uint32_t, uint32_t> li = lights_.push(l);
            env_map_light_ = LightHandle{li.first, li.second};
            env_.light_index = env_map_light_._index;
        }
        RebuildTLAS_nolock();
        RebuildLightTree_nolock();
    }
};

// One device, or several (RAY_HIP_DEVICES=0,1,2,3 | all; settings_t::preferred_device takes the same list): the scene is
// replicated, rank r renders the 64 x 64 tiles r, r + N, ... of every RenderScene (rayhip_comm_bind deals them), and the
// moment somebody looks at pixels the root gathers the other ranks' tiles (rayhip_comm_reduce_framebuffers: peer copies
// over xGMI) -- the whole of the reference API stays as it is, a host that asks for four devices gets one image.
class Renderer final : public RendererBase {
    ILog *log_;
    rayhip_ctx *ctx_ = nullptr;        // the root: the context pixels are read from
    std::vector<rayhip_ctx *> ctxs_;   // all ranks, ctxs_[0] == ctx_
    rayhip_comm *comm_ = nullptr;      // set when ctxs_.size() > 1
    mutable bool assembled_ = true;    // the root holds every rank's tiles of what was rendered so far
    std::string device_name_;
    int w_ = 0, h_ = 0;

    const Scene *uploaded_scene_ = nullptr;
    uint64_t uploaded_version_ = 0, uploaded_geometry_version_ = 0;

    ePixelFilter filter_table_filter_ = ePixelFilter(-1);
    float filter_table_width_ = 0.0f;

    bool collect_stats_ = true;
    std::mutex mtx_;

    // host mirrors handed out by get_*_pixels_ref (valid until the next mutating call, like RendererVK.cpp:1698-1757)
    // Deferred iterations: consecutive RenderScene calls on an unchanged scene / camera / rect are only counted and
    // then rendered together by rayhip_render_batch (up to max_batch_ iterations per wavefront pass) the moment
    // anything could observe them -- bit-identical to rendering them one by one, much fuller launches.
    mutable int pending_count_ = 0, pending_first_ = 0;
    mutable rayhip_camera pending_cam_ = {};
    mutable int pending_rect_[4] = {};
    bool have_cam_ = false; // pending_cam_ holds the camera of the last RenderScene
    int lut_transform_ = 0; // view transform whose look-up table is on the device
    bool use_tex_compression_ = false; // settings_t::use_tex_compression, handed to the scenes this renderer creates
    int max_batch_ = 64; // further limited by rayhip_max_batch() (frame size)
    bool unet_ready_ = false;

    void Flush() const {
        if (pending_count_ > 0) {
            const int n = pending_count_;
            pending_count_ = 0;
            for (rayhip_ctx *c : ctxs_) { // (a pass is only enqueued: the devices work side by side)
                // stage times are the root's (GetStats / ResetStats read ctx_ only): a rank that is never asked for its times must not
                // record events either -- its pending marks and event pool would grow with every pass
                const uint32_t flags = (collect_stats_ && c == ctx_) ? RAYHIP_FLAG_TIME_STAGES : 0u;
                check(rayhip_render_batch(c, &pending_cam_, pending_rect_, pending_first_, n, flags, nullptr), "rayhip_render_batch");
            }
            assembled_ = comm_ == nullptr;
        }
    }
    // several devices: bring the other ranks' tiles (radiance, aux images, variance estimate) to the root
    void Assemble() const {
        Flush();
        if (!assembled_ && comm_) {
            check(rayhip_comm_reduce_framebuffers(comm_, 0, RAYHIP_REDUCE_ALL, &pending_cam_), "rayhip_comm_reduce_framebuffers");
            assembled_ = true;
        }
    }
    template <class F> void ForAll(F &&f, const char *what) const {
        for (rayhip_ctx *c : ctxs_) {
            check(f(c), what);
        }
    }

    mutable std::vector<color_rgba_t> host_[4];
    mutable bool host_dirty_[4] = {true, true, true, true};

    void check(const int status, const char *what) const {
        if (status != 0) {
            log_->Error("RendererHIP: %s failed: %s", what, rayhip_last_error());
        }
    }

    color_data_rgba_t fetch(const int which) const {
        Assemble();
        if (host_dirty_[which]) {
            host_[which].resize(size_t(w_) * h_);
            check(rayhip_readback(ctx_, which, &host_[which][0].v[0], w_), "rayhip_readback");
            host_dirty_[which] = false;
        }
        return {host_[which].data(), w_};
    }

    void UpdateFilterTable(ePixelFilter filter, float filter_width) {
        // same construction as Cpu::Renderer::UpdateFilterTable, RendererCPU.h:1234-1258
        float (*filter_func)(float v, float width) = nullptr;
        switch (filter) {
        case ePixelFilter::Box:
            filter_func = filter_box;
            filter_width = 1.0f;
            break;
        case ePixelFilter::Gaussian:
            filter_func = filter_gaussian;
            filter_width *= 3.0f;
            break;
        case ePixelFilter::BlackmanHarris:
            filter_func = filter_blackman_harris;
            filter_width *= 2.0f;
            break;
        default:
            log_->Error("RendererHIP: unknown pixel filter");
            return;
        }
        const std::vector<float> table = Ray::CDFInverted(FILTER_TABLE_SIZE, 0.0f, filter_width * 0.5f,
                                                          std::bind(filter_func, std::placeholders::_1, filter_width), true);
        ForAll([&](rayhip_ctx *c) { return rayhip_set_filter_table(c, table.data(), int(table.size())); }, "rayhip_set_filter_table");
    }

    // "0,2,3" | "all" -> device ordinals
    static std::vector<int> ParseDevices(const std::string &spec, const int have) {
        std::vector<int> out;
        if (spec == "all") {
            for (int d = 0; d < have; ++d) {
                out.push_back(d);
            }
            return out;
        }
        size_t pos = 0;
        while (pos < spec.size()) {
            const size_t end = spec.find(',', pos);
            const std::string item = spec.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
            if (!item.empty()) {
                out.push_back(atoi(item.c_str()));
            }
            if (end == std::string::npos) {
                break;
            }
            pos = end + 1;
        }
        return out;
    }

  public:
    Renderer(const settings_t &s, ILog *log) : log_(log), use_tex_compression_(s.use_tex_compression) {
        if (rayhip_device_count() <= 0) {
            throw std::runtime_error("no HIP device found");
        }
        std::vector<int> devices = {0};
        if (const char *e = getenv("RAY_HIP_DEVICE")) {
            devices = {atoi(e)};
        }
        if (const char *e = getenv("RAY_HIP_DEVICES")) {
            devices = ParseDevices(e, rayhip_device_count());
        }
        if (!s.preferred_device.empty()) {
            // settings_t::preferred_device (RendererBase.h:54): device ordinals for this backend, "1" or "0,1,2,3" or "all"
            devices = ParseDevices(std::string(s.preferred_device), rayhip_device_count());
        }
        if (devices.empty()) {
            throw std::runtime_error("RendererHIP: empty device list");
        }
        auto destroy_all = [&]() {
            if (comm_) {
                rayhip_comm_destroy(comm_);
            }
            for (rayhip_ctx *c : ctxs_) {
                rayhip_ctx_destroy(c);
            }
        };
        for (const int d : devices) {
            rayhip_ctx *c = nullptr;
            if (rayhip_ctx_create(d, &c) != 0) {
                const std::string err = rayhip_last_error();
                destroy_all();
                throw std::runtime_error("rayhip_ctx_create: " + err);
            }
            ctxs_.push_back(c);
        }
        ctx_ = ctxs_[0];
        if (ctxs_.size() > 1) {
            bool ok = rayhip_comm_create(int(devices.size()), devices.data(), &comm_) == 0;
            for (size_t r = 0; ok && r < ctxs_.size(); ++r) {
                ok = rayhip_comm_bind(comm_, int(r), ctxs_[r]) == 0;
            }
            if (!ok) {
                const std::string err = rayhip_last_error();
                destroy_all();
                throw std::runtime_error("rayhip_comm_create: " + err);
            }
        }
        char name[256] = {};
        rayhip_ctx_device_name(ctx_, name, sizeof(name));
        device_name_ = name;
        if (ctxs_.size() > 1) {
            device_name_ += " x" + std::to_string(ctxs_.size());
        }
        collect_stats_ = getenv("RAY_HIP_NO_STATS") == nullptr;
        if (const char *e = getenv("RAY_HIP_BATCH")) { // 1 = render every iteration in its own pass
            max_batch_ = atoi(e) > 0 ? atoi(e) : 1;
        }

        log->Info("============================================================================");
        log->Info("Device       is %s", device_name_.c_str());
        log->Info("Wavefront    is 64 lanes, traversal stack 24 words/lane in LDS (+ spill to HBM up to %i)", 2 * MAX_STACK_SIZE);
        log->Info("Devices      %i (the frame's 64x64 tiles are dealt round-robin)", int(ctxs_.size()));
        log->Info("============================================================================");

        // PMJ02 table upload, RendererVK.cpp:299-311
        for (rayhip_ctx *c : ctxs_) {
            if (rayhip_upload_static(c, __pmj02_samples, uint32_t(__pmj02_dims_count) * 2u * uint32_t(__pmj02_sample_count)) != 0) {
                const std::string err = rayhip_last_error();
                destroy_all();
                throw std::runtime_error("rayhip_upload_static: " + err);
            }
        }
        Resize(s.w, s.h);
    }
    ~Renderer() override {
        pending_count_ = 0; // nobody can look at them any more
        if (comm_) {
            rayhip_comm_destroy(comm_);
        }
        for (rayhip_ctx *c : ctxs_) {
            rayhip_ctx_destroy(c);
        }
    }

    eRendererType type() const override { return RendererTypeHIP; }
    ILog *log() const override { return log_; }
    std::string_view device_name() const override { return device_name_; }
    std::pair<int, int> size() const override { return std::pair{w_, h_}; }

    color_data_rgba_t get_pixels_ref() const override { return fetch(RAYHIP_BUF_FINAL); }
    color_data_rgba_t get_raw_pixels_ref() const override { return fetch(RAYHIP_BUF_RAW); }
    color_data_rgba_t get_aux_pixels_ref(const eAUXBuffer buf) const override {
        if (buf == eAUXBuffer::BaseColor) {
            return fetch(RAYHIP_BUF_BASE_COLOR);
        } else if (buf == eAUXBuffer::DepthNormals) {
            return fetch(RAYHIP_BUF_DEPTH_NORMALS);
        }
        return {};
    }
    const shl1_data_t *get_sh_data_ref() const override { return nullptr; }

    void Resize(const int w, const int h) override {
        Flush();
        if (w_ != w || h_ != h) {
            ForAll([&](rayhip_ctx *c) { return rayhip_resize(c, w, h); }, "rayhip_resize");
            w_ = w, h_ = h;
            for (bool &d : host_dirty_) {
                d = true;
            }
        }
    }
    void Clear(const color_rgba_t &c) override {
        Flush();
        ForAll([&](rayhip_ctx *cx) { return rayhip_clear(cx, c.v); }, "rayhip_clear");
        assembled_ = true;
        for (bool &d : host_dirty_) {
            d = true;
        }
    }

    SceneBase *CreateScene() override {
        Scene *s = new Scene(log_, use_tex_compression_);
        rayhip_ctx *ctx = ctx_; // (a scene must not outlive the renderer that made it: the reference's GPU scenes hold their renderer's context too)
        s->set_sky_baker([ctx](const rayhip_scene_desc &d, int w, int h, uint32_t *out) { return rayhip_bake_sky(ctx, &d, w, h, out) == 0; });
        return s;
    }

    void RenderScene(const SceneBase &scene, RegionContext &region) override {
        const auto *s = dynamic_cast<const Scene *>(&scene);
        if (!s) {
            log_->Error("RendererHIP: scene was not created by this backend");
            return;
        }
        std::shared_lock<std::shared_timed_mutex> scene_lock(SceneAccess::Mutex(*s));

        if (uploaded_scene_ != s || uploaded_version_ != s->version()) {
            Flush(); // pending iterations belong to the scene that is on the device now
            try {
                FlatScene flat;
                int rc = 2;
                if (uploaded_scene_ == s && uploaded_geometry_version_ == s->geometry_version()) {
                    // instances / lights / environment only: the top level is rebuilt on the device
                    SceneAccess::Export(*s, flat, false /* with_textures */);
                    for (rayhip_ctx *c : ctxs_) { // (the same decision on every device: they hold the same scene)
                        rc = rayhip_scene_update_instances(c, &flat.desc);
                        if (rc != 0) {
                            break;
                        }
                    }
                    if (rc == 1) {
                        // the device state may be half replaced: nothing usable until a full upload succeeds
                        log_->Error("RendererHIP: rayhip_scene_update_instances failed: %s", rayhip_last_error());
                        uploaded_scene_ = nullptr, uploaded_version_ = uploaded_geometry_version_ = 0;
                        return;
                    }
                }
                if (rc == 2) {
                    SceneAccess::Export(*s, flat);
                    bool ok = true;
                    for (rayhip_ctx *c : ctxs_) {
                        ok = ok && rayhip_scene_upload(c, &flat.desc) == 0;
                    }
                    if (!ok) {
                        // nothing usable is on the device: forget what was there, so that the next RenderScene tries again
                        log_->Error("RendererHIP: rayhip_scene_upload failed: %s", rayhip_last_error());
                        uploaded_scene_ = nullptr, uploaded_version_ = uploaded_geometry_version_ = 0;
                        return;
                    }
                }
            } catch (std::exception &e) {
                log_->Error("RendererHIP: %s", e.what());
                uploaded_scene_ = nullptr, uploaded_version_ = uploaded_geometry_version_ = 0;
                return;
            }
            uploaded_scene_ = s;
            uploaded_version_ = s->version();
            uploaded_geometry_version_ = s->geometry_version();
        }

        const camera_t &cam = SceneAccess::CurrentCamera(*s);
        if (cam.filter != filter_table_filter_ || cam.filter_width != filter_table_width_) {
            Flush();
            UpdateFilterTable(cam.filter, cam.filter_width);
            filter_table_filter_ = cam.filter;
            filter_table_width_ = cam.filter_width;
        }

        if (cam.view_transform != eViewTransform::Standard && int(cam.view_transform) != lut_transform_) {
            // what RendererVK::RenderScene does with its 3-D texture (RendererVK.cpp:404-415)
            Flush();
            try {
                ForAll([&](rayhip_ctx *c) { return rayhip_set_tonemap_lut(c, int(cam.view_transform), transform_luts[int(cam.view_transform)], LUT_DIMS); },
                       "rayhip_set_tonemap_lut");
            } catch (std::exception &e) {
                log_->Error("RendererHIP: %s", e.what());
                return;
            }
            lut_transform_ = int(cam.view_transform);
        }

        ++region.iteration; // RendererCPU.h:384

        rayhip_camera rc;
        memcpy(&rc, &cam, sizeof(rc));
        const rect_t &rect = region.rect();
        const int r[4] = {rect.x, rect.y, rect.w, rect.h};
        // stage times come from HIP events recorded on the stream and are resolved lazily in GetStats -- the
        // counterpart of the timestamp queries RendererVK reads back one frame later (RendererVK.cpp:452-487)
        const bool extends = pending_count_ > 0 && region.iteration == pending_first_ + pending_count_ &&
                             memcmp(&rc, &pending_cam_, sizeof(rc)) == 0 && memcmp(r, pending_rect_, sizeof(r)) == 0;
        have_cam_ = true;
        if (!extends) {
            Flush();
            pending_cam_ = rc;
            memcpy(pending_rect_, r, sizeof(r));
            pending_first_ = region.iteration;
        }
        ++pending_count_;
        if (pending_count_ >= std::min(max_batch_, std::max(1, rayhip_max_batch(ctx_)))) {
            Flush();
        }
        for (bool &d : host_dirty_) {
            d = true;
        }
    }

    // post-processing / caching stages are outside the hot path (OUT OF SCOPE)
    // NLM denoiser (RendererCPU.h:661-783): filters what the iterations so far accumulated, with the camera (tonemap, adaptive
    // sampling threshold) of the last RenderScene, as the reference does through tonemap_params_ / variance_threshold_
    void DenoiseImage(const RegionContext &region) override {
        Assemble(); // the filter reads every pixel's radiance, guides and variance estimate: on the root, after the gather
        if (!have_cam_) {
            log_->Error("RendererHIP: DenoiseImage before the first RenderScene");
            return;
        }
        const rect_t &r = region.rect();
        const int rect[4] = {r.x, r.y, r.w, r.h};
        check(rayhip_denoise_nlm(ctx_, &pending_cam_, rect, region.iteration), "rayhip_denoise_nlm");
        for (bool &d : host_dirty_) {
            d = true;
        }
    }
    // UNet denoiser (RendererCPU.h:790-1007): the sixteen convolution passes run on the matrix cores of the root device
    // (rayhip_denoise_unet), on what the iterations so far accumulated -- every rank's tiles gathered first
    void DenoiseImage(const int pass, const RegionContext &region) override {
        Assemble();
        if (!have_cam_) {
            log_->Error("RendererHIP: DenoiseImage before the first RenderScene");
            return;
        }
        if (!unet_ready_) {
            log_->Error("RendererHIP: DenoiseImage(pass, region) needs InitUNetFilter first");
            return;
        }
        const rect_t &r = region.rect();
        const int rect[4] = {r.x, r.y, r.w, r.h};
        check(rayhip_denoise_unet(ctx_, &pending_cam_, rect, pass), "rayhip_denoise_unet");
        for (bool &d : host_dirty_) {
            d = true;
        }
    }
    void UpdateSpatialCache(const SceneBase &, RegionContext &) override {}
    void ResolveSpatialCache(const SceneBase &, const std::function<void(int, int, ParallelForFunction &&)> &) override {}
    void ResetSpatialCache(const SceneBase &, const std::function<void(int, int, ParallelForFunction &&)> &) override {}

    void GetStats(stats_t &st) override {
        std::lock_guard<std::mutex> _(mtx_);
        Flush();
        rayhip_stats rs = {};
        check(rayhip_get_stage_times(ctx_, &rs, 0), "rayhip_get_stage_times");
        memcpy(&st, &rs, sizeof(st));
    }
    void ResetStats() override {
        std::lock_guard<std::mutex> _(mtx_);
        Flush();
        rayhip_stats rs = {};
        check(rayhip_get_stage_times(ctx_, &rs, 1), "rayhip_get_stage_times");
    }

    // the network's weights come from the reference's own tables, laid out by its own SetupUNetWeights (UNetFilter.cpp:296-570,
    // what Cpu::Renderer::InitUNetFilter does at RendererCPU.h:1261-1266); librayhip re-packs them for its kernels.  The passes
    // do not alias memory here, so no pass depends on another one's storage being free.
    unet_filter_properties_t InitUNetFilter(bool, const std::function<void(int, int, ParallelForFunction &&)> &) override {
        unet_weight_offsets_t offsets;
        std::vector<float> weights(size_t(SetupUNetWeights<float>(8, nullptr, nullptr)));
        SetupUNetWeights(8, &offsets, weights.data());
        static_assert(sizeof(offsets) == 32 * sizeof(int32_t), "unet_weight_offsets_t is 32 ints");
        check(rayhip_unet_init(ctx_, weights.data(), int(weights.size()), reinterpret_cast<const int32_t *>(&offsets), 8), "rayhip_unet_init");
        // half precision where the device has the matrix hardware for it -- every gfx950 has -- as the reference's GPU backends choose
        // (RendererVK.cpp:254-263, 1834-1844: use_fp16_ / use_coop_matrix_ select the fp16 shader set); RAY_HIP_UNET_F32=1 keeps the exact f32 form
        const char *f32 = getenv("RAY_HIP_UNET_F32");
        check(rayhip_unet_set_precision(ctx_, (f32 && f32[0] == '1') ? 0 : 1), "rayhip_unet_set_precision");
        unet_ready_ = true;
        unet_filter_properties_t props;
        props.pass_count = UNetFilterPasses;
        for (int i = 0; i < UNetFilterPasses; ++i) {
            std::fill(&props.alias_dependencies[i][0], &props.alias_dependencies[i][0] + 4, -1);
        }
        return props;
    }
};

RendererBase *CreateRenderer(const settings_t &s, ILog *log) { return new Renderer(s, log); }

SceneBase *CreateScene(ILog *log, const bool use_tex_compression) { return new Scene(log, use_tex_compression); }

const char *SkyBakedOn(const SceneBase &scene) {
    const auto *s = dynamic_cast<const Scene *>(&scene);
    return s ? s->sky_baked_on() : "none";
}

std::vector<uint8_t> ExportSceneBlob(const SceneBase &scene) {
    const auto *s = dynamic_cast<const Cpu::Scene *>(&scene);
    if (!s) {
        throw std::runtime_error("ExportSceneBlob: not a CPU-side scene");
    }
    std::shared_lock<std::shared_timed_mutex> scene_lock(SceneAccess::Mutex(*s));
    FlatScene flat;
    SceneAccess::Export(*s, flat);
    const camera_t &cam = SceneAccess::CurrentCamera(*s);
    rayhip_camera rc;
    memcpy(&rc, &cam, sizeof(rc));
    // pixel-filter table, same construction as Cpu::Renderer::UpdateFilterTable (RendererCPU.h:1234-1258)
    float (*filter_func)(float v, float width) = nullptr;
    float filter_width = cam.filter_width;
    switch (cam.filter) {
    case ePixelFilter::Box:
        filter_func = filter_box;
        filter_width = 1.0f;
        break;
    case ePixelFilter::Gaussian:
        filter_func = filter_gaussian;
        filter_width *= 3.0f;
        break;
    case ePixelFilter::BlackmanHarris:
        filter_func = filter_blackman_harris;
        filter_width *= 2.0f;
        break;
    default:
        throw std::runtime_error("unknown pixel filter");
    }
    const std::vector<float> table = Ray::CDFInverted(FILTER_TABLE_SIZE, 0.0f, filter_width * 0.5f,
                                                      std::bind(filter_func, std::placeholders::_1, filter_width), true);
    const bool lut = rc.view_transform != 0;
    return rayhip_blob::serialize(flat.desc, rc, table.data(), int(table.size()), lut ? transform_luts[rc.view_transform] : nullptr,
                                  lut ? LUT_DIMS : 0);
}

} // namespace Hip
} // namespace Ray
