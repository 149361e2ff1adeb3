// rayhip.hip -- implementation of the librayhip C ABI (include/rayhip.h): device memory, stage schedule, readback.
//
// Stage schedule of one rayhip_render == one RenderScene iteration, restating the control flow of
// reference internal/RendererCPU.h:373-659 with the GPU-side bookkeeping of internal/RendererVK.cpp:368-791
// (everything on one stream, ray counts never leave HBM).
//
// There is NO host fallback in this library: without a gfx950 device every entry point fails.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rayhip.h"
#include "kernels.hip.h" // first: it configures the profiling macros the rt_*.h headers expand
#include "shade_launch.h"
#include "bvh4_build.h"
#include "bvh4_build.hip.h"
#include "bvh8_build.h"
#include "bvh_layout.h"
#include "unet.h"
#include "lbvh.hip.h"
#include "scene_blob.h"
#include "scene_rebuild.h"
#include "scene_update.h"
#include "scene_validate.h"
#include "sort.h"

using namespace rt;

namespace {
thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[1024];
    va_list vl;
    va_start(vl, fmt);
    vsnprintf(buf, sizeof(buf), fmt, vl);
    va_end(vl);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        const hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess) {                                                                                        \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);                    \
        }                                                                                                              \
    } while (0)

constexpr int MAX_BOUNCE_SLOTS = 130; // max_total_depth is a uint8 but bounded by MAX_BOUNCES = 128 (Constants.inl:5)

// the stream fresh allocations are touched on: the current context's (set by use_device)
thread_local hipStream_t g_touch_stream = nullptr;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        if (n <= bytes && p) {
            return 0;
        }
        release();
        if (n == 0) {
            n = 16;
        }
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        // touch it now: the first write to fresh device memory is several times slower than the following ones, and the
        // wavefront-state buffers would otherwise pay that inside the first large pass (measured: ray generation 31 ms
        // instead of 0.8 ms in a 20-iteration pass that followed a 5-iteration warm-up)
        // On the context's own stream, so that it is ordered before every later use without stalling other contexts /
        // streams of the device; the wait keeps growth inside a pass out of the stage timers.
        if (g_touch_stream) {
            HIP_TRY(hipMemsetAsync(p, 0, n, g_touch_stream));
            HIP_TRY(hipStreamSynchronize(g_touch_stream));
        } else { // (no context yet: the null stream does not order against non-blocking streams, so wait for the device)
            HIP_TRY(hipMemset(p, 0, n));
            HIP_TRY(hipDeviceSynchronize());
        }
        return 0;
    }
    void release() {
        if (p) {
            (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
    }
    void swap(DevBuf &o) {
        std::swap(p, o.p);
        std::swap(bytes, o.bytes);
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};
} // namespace

struct rayhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t tri_pitch = 3; // 16-byte rows per record of `tris` as uploaded (SceneView::tri_pitch)
    uint32_t all_solid = 0; // SceneView::all_solid
    hipDeviceProp_t props = {};
    int grid_waves = 0; // resident-ish grid for the wave-per-block kernels
    bool small_scene = false; // BLAS nodes + triangles fit one XCD's L2: traversal kernels with the smaller register footprint
    int refill_waves = 0; // largest grid of the persistent closest-hit kernel (blocks); 0 = kernel switched off
    int refill_resident = 0; // ... and the number of its blocks the device holds at once
    int sort_key_mode = 0; // RAYHIP_SORT_KEY (tuning, rt_sort.h)
    // RAYHIP_PRIMARY_WAVES / RAYHIP_SHADOW_WAVES: register footprint of the plain K2 (primary rays) / of K3.  K3 runs at 5 waves per
    // SIMD (96 VGPRs, 32 spilled registers per ray instead of 51 at 6 waves: same time, a third less scratch traffic)
    int tune_primary_waves = 0, tune_shadow_waves = 5;
    bool refill_secondary_only = false; // RAYHIP_REFILL=2: primary rays (coherent, every lane busy to the end) keep the plain kernel
    bool refill_primary_whole = false;  // RAYHIP_REFILL=3: ... or take the flat kernel with whole-chunk refills (no spill stores in the walk)
    bool refill_pool = false;           // RAYHIP_REFILL=4: 3 + the secondary bounces of 4-wide scenes through the pooled kernel (k_trace_closest_pool)
    int pool_waves = 0, pool_resident = 0; // its grid
    bool pool_scene = false;               // ... and whether the scene in place suits it (refresh_scene_view)

    DevBuf pmj, filter_table;
    // scene
    DevBuf nodes, tris, tri_indices, tri_materials, materials, vertices, vtx_indices, mesh_instances, lights, li_indices,
        light_cwnodes, light_children, light_tri_geom, tri_verts, tri_bitangents, textures, texels, nodes4, nodes8, blas_root4, env_qtree;
    // UNet denoiser (unet.h): per pass the repacked weights + bias, and the fifteen activation tensors of the current frame size
    struct UNetPass {
        DevBuf weights, bias;
        int n_tiles = 0;
    };
    UNetPass unet_pass[16];
    DevBuf unet_tensor[15];
    int unet_w = 0, unet_h = 0; // frame size the tensors were sized for
    bool unet_ready = false;
    SceneView sc = {};
    float bbox_min[3] = {}, bbox_max[3] = {};
    bool have_scene = false;
    Shard shard = {64, 1, 0};

    // frame
    int w = 0, h = 0;
    DevBuf px_temp, px_full, px_half, px_raw, px_final, px_base, px_dn, px_req, px_aux_base, px_aux_dn;
    size_t slots_cap = 0; // wavefront-state slots allocated
    DevBuf px_variance, nlm_tm, nlm_var_h, nlm_var; // DenoiseImage: variance estimate [h][w]; [ext_h][ext_w] intermediates
    DevBuf tonemap_lut;   // table of view transform `lut_transform` (rayhip_set_tonemap_lut)
    DevBuf shard_stage;   // [4][h][w] float4: this rank's owned pixels of full / base colour / depth-normals / variance, zero elsewhere
                          // (what the multi-GPU frame reduce sums; rayhip_comm_reduce_framebuffers, rayhip_export_shard_device)
    // what rayhip_scene_update_instances needs of the last full upload: per mesh (key: mesh_instance_t::mesh_index) the roots
    // of its bottom-level trees as uploaded; node slots reserved behind the uploaded nodes for top-level trees built later
    // on the device
    rayhip_update::MeshRefs mesh_refs;
    uint32_t nodes_used = 0, nodes_reserved = 0;
    uint32_t tlas_half = 0; // which half of the reserved node slots the next rebuilt top level goes to (the live one sits in the other)
    int wide = 0; // the wide BLAS form the kernels walk: 4 (rt_bvh4.h, default), 8 (rt_bvh8.h) or 0 (the reference's BVH2)
    uint32_t tex_table[8] = {}, textures_count = 0, tex_flags = 0;
    struct { uint32_t vertices, vtx_indices, tri_materials, materials; } geometry = {};
    bool adaptive_dirty = false; // a pass ran with variance_threshold != 0 since the last Clear / Resize: required_samples may
                                 // lie below the next iteration, so passes are not batched (rayhip_render_batch)
    int lut_transform = 0, lut_dims = 0;
    PixelBuffers px = {};

    // wavefront state, sized w*h
    DevBuf ray_planes[2][5], hit_planes[2], shadow_planes[3], deferred_planes[2], point_planes[7], nee_index;
    PointSoA points = {};
    // how the shade stage is cut into launches (kernels.hip.h): bit 0 = the light pick as its own kernel, bit 1 = next-event
    // estimation and continuation as two scatter launches.  RAYHIP_SHADE_SPLIT overrides (A/B measurements).
    int shade_split = 5; // shade_launch.h: bit 0 pick as its own kernel, bit 1 NEE / continuation as two launches, bit 2 NEE over the compacted queue of points that got a light
    RaySoA rays[2] = {};
    HitSoA hits = {};
    ShadowSoA shadow = {};
    DeferredSoA deferred = {};
    DevBuf counters;      // uint32 [MAX_BOUNCE_SLOTS][5: rays, shadow rays, deferred emitters, shade points, points with a light][QUEUE_MAX_STRIPES * QUEUE_COUNTER_STRIDE]
    DevBuf trav_counters; // u64 [2][TRAV_COUNTER_WORDS]
    DevBuf stack_spill;   // per-wave overflow slabs of the traversal stack
    DevBuf sort_keys[2], sort_idx[2], sort_temp;
    SortGrid sort_grid = {};

    // timing: events are recorded without synchronising; intervals are resolved lazily (resolve_timing)
    struct Mark {
        size_t ev;
        int stage; // index into rayhip_stats, or -1
        int trav;  // 0 = closest kernel, 1 = shadow kernel, -1 = none
        bool first; // first mark of a render call (no interval ends here)
    };
    std::vector<hipEvent_t> events;
    size_t events_used = 0;
    std::vector<Mark> pending;
    double trav_ms[2] = {0.0, 0.0};
    unsigned long long trav_launches[2] = {0, 0};
    double stage_us[11] = {};

    static constexpr size_t QUEUE_WORDS = size_t(QUEUE_MAX_STRIPES) * QUEUE_COUNTER_STRIDE;
    static constexpr int QUEUES_PER_BOUNCE = 5;
    uint32_t *ray_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b) * QUEUE_WORDS; }
    uint32_t *shadow_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 1) * QUEUE_WORDS; }
    uint32_t *deferred_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 2) * QUEUE_WORDS; }
    uint32_t *point_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 3) * QUEUE_WORDS; }
    uint32_t *nee_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 4) * QUEUE_WORDS; }
    // queue geometry for a frame of `items` pixels split over `stripes` stripes
    static RayQueue make_queue(uint32_t *counts, size_t items, uint32_t stripes) {
        const size_t chunks = (items + WAVE - 1) / WAVE;
        return RayQueue{counts, stripes, uint32_t((chunks + stripes - 1) / stripes)};
    }
    RayQueue ray_queue(int b, size_t items, uint32_t stripes) const { return make_queue(ray_count(b), items, stripes); }
    RayQueue shadow_queue(int b, size_t items, uint32_t stripes) const { return make_queue(shadow_count(b), items, stripes); }
    RayQueue deferred_queue(int b, size_t items, uint32_t stripes) const { return make_queue(deferred_count(b), items, stripes); }
    RayQueue point_queue(int b, size_t items, uint32_t stripes) const { return make_queue(point_count(b), items, stripes); }
    RayQueue nee_queue(int b, size_t items, uint32_t stripes) const { return make_queue(nee_count(b), items, stripes); }
    int clear_queues(int bounces, hipStream_t s) const {
        return hipMemsetAsync(counters.p, 0, size_t(QUEUES_PER_BOUNCE * bounces) * QUEUE_WORDS * sizeof(uint32_t), s) == hipSuccess ? 0 : 1;
    }
};

namespace {

int upload(rayhip_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    if (b.alloc(bytes)) {
        return 1;
    }
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    }
    return 0;
}

int use_device(rayhip_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    g_touch_stream = c->stream;
    return 0;
}

// wavefront-state slots a w x h rect needs: ray generation deals whole 8x8 pixel tiles (k_raygen)
size_t tile_slots(int w, int h) { return size_t((w + 7) / 8) * size_t((h + 7) / 8) * 64u; }

// Most iterations one pass can carry (Layering, rt_base.h): layers are stacked `cols` wide and `rows` high in a virtual
// frame whose coordinates must fit the two 16-bit halves of ray_data_t::xy.
constexpr int MAX_LAYERS = 512;
int max_layers_for(int w, int h) {
    if (w <= 0 || h <= 0) {
        return 0;
    }
    // (the pixel helpers index the virtual frame with 32-bit ints: cols * rows * w * h must stay below 2^31; make_layering
    // rounds the layer count up to whole columns, hence the margin of one column)
    const size_t npix = size_t(w) * size_t(h), max_rows = size_t(65535 / h);
    const size_t by_index = ((size_t(1) << 31) - 1) / npix;
    const size_t by_index_cols = by_index > max_rows ? (by_index / max_rows) * max_rows : by_index;
    return int(std::max<size_t>(1, std::min<size_t>({size_t(MAX_LAYERS), size_t(65535 / w) * max_rows, by_index_cols})));
}
// the virtual frame of a pass of `layers` iterations: as few columns as the row limit allows
Layering make_layering(int w, int h, int layers) {
    const int max_rows = std::max(1, 65535 / h);
    const int cols = (layers + max_rows - 1) / max_rows;
    return Layering{h, layers, w, std::max(1, cols)};
}
int layer_rows(const Layering &L) { return (L.count + L.cols - 1) / L.cols; }
// slots a pass of `layers` iterations over a rect needs under the context's shard (k_raygen's tiling)
size_t pass_slots(const rayhip_ctx *c, int frame_w, int frame_h, int rect_w, int rect_h, int layers) {
    return size_t(make_raygen_tiling(frame_w, frame_h, rect_w, rect_h, c->shard).tiles) * 64u * size_t(layers);
}

int alloc_frame(rayhip_ctx *c, int w, int h, int layers) {
    const size_t npix = size_t(w) * size_t(h);
    const Layering L = make_layering(w, h, layers);
    const size_t vpix = npix * size_t(L.cols) * size_t(layer_rows(L)); // the virtual frame (>= npix * layers)
    if (c->px_temp.alloc(vpix * 16) ||
        (layers > 1 && (c->px_aux_base.alloc(vpix * 16) || c->px_aux_dn.alloc(vpix * 16))) || c->px_full.alloc(npix * 16) || c->px_half.alloc(npix * 16) || c->px_raw.alloc(npix * 16) ||
        c->px_final.alloc(npix * 16) || c->px_base.alloc(npix * 16) || c->px_dn.alloc(npix * 16) || c->px_req.alloc(npix * 2)) {
        return 1;
    }
    c->px.temp = c->px_temp.as<float4>(), c->px.full = c->px_full.as<float4>(), c->px.half = c->px_half.as<float4>();
    c->px.raw = c->px_raw.as<float4>(), c->px.final_ = c->px_final.as<float4>();
    c->px.base_color = c->px_base.as<float4>(), c->px.depth_normals = c->px_dn.as<float4>();
    c->px.required_samples = c->px_req.as<uint16_t>();
    c->px.aux_base_layers = c->px_aux_base.as<float4>(), c->px.aux_dn_layers = c->px_aux_dn.as<float4>();
    if (c->px_variance.alloc(npix * 16)) {
        return 1;
    }
    c->px.variance = c->px_variance.as<float4>();

    // wavefront-state slots: one per pixel this context renders (its shard's share when the frame is tile-sharded, but
    // never less than one full frame: the kernel-level hooks and single-iteration passes of any shard fit) + the
    // rounding of the striped queues (each stripe holds whole chunks)
    const size_t n = std::max(tile_slots(w, h), pass_slots(c, w, h, w, h, layers)) + size_t(WAVE) * QUEUE_MAX_STRIPES;
    // (slots_cap is raised only after every plane below exists: a failed hipMalloc must not make pass_fits() lie)
    const size_t old_cap = c->slots_cap;
    c->slots_cap = 0;
    for (int k = 0; k < 2; ++k) {
        for (int pl = 0; pl < 5; ++pl) {
            if (c->ray_planes[k][pl].alloc(n * (pl == 4 ? 8 : 16))) {
                return 1;
            }
        }
        c->rays[k].o_pdf = c->ray_planes[k][0].as<float4>(), c->rays[k].d_cw = c->ray_planes[k][1].as<float4>();
        c->rays[k].c_cs = c->ray_planes[k][2].as<float4>(), c->rays[k].ior = c->ray_planes[k][3].as<float4>();
        c->rays[k].xy_depth = c->ray_planes[k][4].as<uint2>();
    }
    if (c->hit_planes[0].alloc(n * 16) || c->hit_planes[1].alloc(n * 4)) {
        return 1;
    }
    c->hits.oi_pi_t_u = c->hit_planes[0].as<float4>(), c->hits.v = c->hit_planes[1].as<float>();
    for (int pl = 0; pl < 3; ++pl) {
        if (c->shadow_planes[pl].alloc(n * 16)) {
            return 1;
        }
    }
    c->shadow.o_depth = c->shadow_planes[0].as<float4>(), c->shadow.d_dist = c->shadow_planes[1].as<float4>();
    c->shadow.c_xy = c->shadow_planes[2].as<float4>();
    if (c->deferred_planes[0].alloc(n * 16) || c->deferred_planes[1].alloc(n * 16)) {
        return 1;
    }
    c->deferred.a = c->deferred_planes[0].as<float4>(), c->deferred.b = c->deferred_planes[1].as<float4>();
    for (int pl = 0; pl < 7; ++pl) {
        if (c->point_planes[pl].alloc(n * 16)) {
            return 1;
        }
    }
    c->points.p_slot = c->point_planes[0].as<float4>(), c->points.n_gx = c->point_planes[1].as<float4>();
    c->points.b_gy = c->point_planes[2].as<float4>(), c->points.base_gz = c->point_planes[3].as<float4>();
    c->points.scalars = c->point_planes[4].as<float4>(), c->points.misc = c->point_planes[5].as<float4>();
    c->points.light = c->point_planes[6].as<float4>();
    if (c->nee_index.alloc(n * 4)) {
        return 1;
    }
    c->points.nee_index = c->nee_index.as<uint32_t>();
    // the ray sort only runs on single-iteration passes
    const size_t n_sort = tile_slots(w, h) + size_t(WAVE) * QUEUE_MAX_STRIPES;
    size_t temp_bytes = 0;
    if (sort_pairs_temp_bytes(n_sort, SORT_KEY_BITS, &temp_bytes) != hipSuccess) {
        return fail("rocPRIM temp-size query failed");
    }
    if (c->sort_keys[0].alloc(n_sort * 4) || c->sort_keys[1].alloc(n_sort * 4) || c->sort_idx[0].alloc(n_sort * 4) ||
        c->sort_idx[1].alloc(n_sort * 4) || c->sort_temp.alloc(temp_bytes)) {
        return 1;
    }
    c->slots_cap = std::max(old_cap, n); // (DevBuf never shrinks)
    return 0;
}

int grid_for(const rayhip_ctx *c, size_t items, int block) {
    const size_t need = (items + size_t(block) - 1) / size_t(block);
    const size_t cap = size_t(c->props.multiProcessorCount) * 8u * (256u / unsigned(block) > 0 ? 256u / unsigned(block) : 1u);
    size_t g = need < cap ? need : cap;
    return int(g ? g : 1);
}

// HIP-event stopwatch over the context stream.  Marks are only recorded here (no synchronisation, so the stage
// schedule keeps streaming); resolve_timing() turns them into per-stage and per-kernel times later.
struct StageTimer {
    rayhip_ctx *c;
    bool on;
    bool first = true;
    StageTimer(rayhip_ctx *ctx, bool enabled) : c(ctx), on(enabled) {}
    // the mark labels the interval that STARTS at it
    int mark(int stage, int trav) {
        if (!on) {
            return 0;
        }
        if (c->events_used == c->events.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            c->events.push_back(e);
        }
        HIP_TRY(hipEventRecord(c->events[c->events_used], c->stream));
        c->pending.push_back({c->events_used, stage, trav, first});
        first = false;
        ++c->events_used;
        return 0;
    }
};

int resolve_timing(rayhip_ctx *c) {
    if (c->pending.empty()) {
        return 0;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k + 1 < c->pending.size(); ++k) {
        const rayhip_ctx::Mark &a = c->pending[k], &b = c->pending[k + 1];
        if (b.first) {
            continue;
        }
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->events[a.ev], c->events[b.ev]));
        if (a.stage >= 0) {
            c->stage_us[a.stage] += double(ms) * 1000.0;
        }
        if (a.trav >= 0) {
            c->trav_ms[a.trav] += double(ms);
            c->trav_launches[a.trav] += 1;
        }
    }
    c->pending.clear();
    c->events_used = 0;
    return 0;
}

enum { ST_GEN = 0, ST_PTRACE, ST_PSHADE, ST_PSHADOW, ST_SORT, ST_STRACE, ST_SSHADE, ST_SSHADOW };

void rays_to_soa(const rayhip_ray *in, int n, std::vector<float4> pl[4], std::vector<uint2> &xd) {
    for (int k = 0; k < 4; ++k) {
        pl[k].resize(size_t(n));
    }
    xd.resize(size_t(n));
    for (int i = 0; i < n; ++i) {
        const rayhip_ray &r = in[i];
        pl[0][i] = make_float4(r.o[0], r.o[1], r.o[2], r.pdf);
        pl[1][i] = make_float4(r.d[0], r.d[1], r.d[2], r.cone_width);
        pl[2][i] = make_float4(r.c[0], r.c[1], r.c[2], r.cone_spread);
        pl[3][i] = make_float4(r.ior[0], r.ior[1], r.ior[2], r.ior[3]);
        xd[i] = make_uint2(r.xy, r.depth);
    }
}
} // namespace

extern "C" {

const char *rayhip_last_error(void) { return g_err.c_str(); }

int rayhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

int rayhip_ctx_create(int device, rayhip_ctx **out_ctx) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        return fail("no HIP device available (librayhip has no CPU path)");
    }
    if (device < 0 || device >= n) {
        return fail("device %d out of range (have %d)", device, n);
    }
    rayhip_ctx *c = new rayhip_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->props, device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail("failed to initialise HIP device %d", device);
    }
    g_touch_stream = c->stream;
    // persistent grid of the wave-per-block kernels: as many blocks as are resident (LDS stack + VGPR budget)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_closest<false, 8>, WAVE, 0) != hipSuccess || per_cu <= 0) {
        per_cu = 8;
    }
    // 16x more blocks than are resident: each block then owns 1/16 of the chunks a resident wave would, and the
    // hardware dispatcher hands the next block to whichever CU drains first.  Measured on the Bistro-class scene,
    // 32-iteration passes: x1 295, x2 306, x4 326, x8 338, x16 342, x32 342 Msamples/s.  RAYHIP_GRID_MULT overrides.
    int grid_mult = 16;
    if (const char *e = getenv("RAYHIP_GRID_MULT")) {
        const int m = atoi(e);
        if (m >= 1 && m <= 64) {
            grid_mult = m;
        }
    }
    c->grid_waves = c->props.multiProcessorCount * per_cu * grid_mult;
    if (const char *e = getenv("RAYHIP_PRIMARY_WAVES")) {
        c->tune_primary_waves = atoi(e);
    }
    if (const char *e = getenv("RAYHIP_SHADOW_WAVES")) {
        c->tune_shadow_waves = atoi(e);
    }
    if (const char *e = getenv("RAYHIP_SORT_KEY")) {
        c->sort_key_mode = std::max(0, std::min(3, atoi(e)));
    }
    if (const char *e = getenv("RAYHIP_SHADE_SPLIT")) {
        c->shade_split = atoi(e) & 7;
    }
    // The persistent ray-refill form of the closest-hit kernel (kernels.hip.h): lanes whose ray is finished fetch the next one
    // instead of idling until the longest walk of their wavefront ends.  RAYHIP_REFILL: 2 = for the secondary
    // bounces (incoherent rays, 45 % of the lane slots of the plain kernel belong to finished rays: K2 2.25 -> 2.05 ms per
    // iteration on the Bistro-class scene), the coherent primary rays keep the plain kernel (refill: 0.52 vs 0.35 ms);
    // 1 = every bounce; 0 = off.  The grid is RAYHIP_REFILL_MULT (default 16) blocks per resident wave slot: with exactly
    // one block per slot the launch ends on its slowest wavefront (1x: 1.84, 4x: 1.79, 16x: 1.74 ms; 64x the same).
    // Scenes that fit L2 gain too (03_principled 2048^2: 965 -> 994 Msamples/s, Cornell 1024^2: 1056 -> 1064).
    {
        // 3 (default): the secondary bounces refill lane by lane; the coherent primary rays run the same flat kernel but take their
        // chunks whole (the plain kernel's schedule without its 60 spilled registers per ray: 2.13 vs 2.10 ms per iteration for
        // K2, 7.8 GB fewer scratch writes per primary launch)
        const int mode = getenv("RAYHIP_REFILL") != nullptr ? atoi(getenv("RAYHIP_REFILL")) : 3;
        if (mode != 0) {
            int per_cu_refill = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_refill, k_trace_closest_refill<8>, WAVE, 0) != hipSuccess || per_cu_refill <= 0) {
                per_cu_refill = per_cu;
            }
            int refill_mult = 16;
            if (const char *e = getenv("RAYHIP_REFILL_MULT")) {
                refill_mult = std::max(1, std::min(64, atoi(e)));
            }
            c->refill_resident = c->props.multiProcessorCount * per_cu_refill;
            c->refill_waves = std::min(c->grid_waves, c->refill_resident * refill_mult);
            c->refill_secondary_only = mode == 2 || mode == 3 || mode == 4;
            c->refill_primary_whole = mode == 3 || mode == 4;
            if (mode == 4) {
                int per_cu_pool = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_pool, k_trace_closest_pool<>, WAVE, 0) != hipSuccess || per_cu_pool <= 0) {
                    per_cu_pool = per_cu_refill;
                }
                c->refill_pool = true;
                c->pool_resident = c->props.multiProcessorCount * per_cu_pool;
                c->pool_waves = std::min(c->grid_waves, c->pool_resident * refill_mult);
            }
        }
    }
    // (sized for the shallowest LDS stack any kernel keeps: the pooled closest-hit kernel trades stack entries for its pool)
    if (c->stack_spill.alloc(size_t(c->grid_waves) * std::max(STACK_SPILL_DEPTH * WAVE, POOL_SLAB_WORDS) * sizeof(uint32_t))) {
        delete c;
        return 1;
    }
    if (c->counters.alloc(sizeof(uint32_t) * rayhip_ctx::QUEUES_PER_BOUNCE * MAX_BOUNCE_SLOTS * rayhip_ctx::QUEUE_WORDS) || c->trav_counters.alloc(sizeof(unsigned long long) * 2 * TRAV_COUNTER_WORDS)) {
        delete c;
        return 1;
    }
    (void)hipMemsetAsync(c->trav_counters.p, 0, sizeof(unsigned long long) * 2 * TRAV_COUNTER_WORDS, c->stream);
    // the stage stopwatch's events exist up front: creating them lazily put ~30 ms of runtime initialisation into the first
    // timed pass of a process
    for (int k = 0; k < 256; ++k) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            break;
        }
        c->events.push_back(e);
    }
    *out_ctx = c;
    return 0;
}

void rayhip_ctx_destroy(rayhip_ctx *c) {
    if (!c) {
        return;
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->events) {
        (void)hipEventDestroy(e);
    }
    DevBuf *all[] = {&c->pmj, &c->filter_table, &c->nodes, &c->tris, &c->tri_indices, &c->tri_materials, &c->materials,
                     &c->vertices, &c->vtx_indices, &c->mesh_instances, &c->lights, &c->li_indices, &c->light_cwnodes, &c->light_children, &c->light_tri_geom, &c->tri_verts, &c->tri_bitangents, &c->nodes4, &c->nodes8, &c->blas_root4, &c->env_qtree,
                     &c->textures, &c->texels, &c->px_temp, &c->px_full, &c->px_half, &c->px_raw, &c->px_final, &c->px_base,
                     &c->px_dn, &c->px_req, &c->px_aux_base, &c->px_aux_dn, &c->px_variance, &c->nlm_tm, &c->nlm_var_h, &c->nlm_var,
                     &c->tonemap_lut, &c->hit_planes[0], &c->hit_planes[1], &c->shadow_planes[0], &c->shadow_planes[1],
                     &c->shadow_planes[2], &c->deferred_planes[0], &c->deferred_planes[1], &c->counters, &c->trav_counters, &c->stack_spill,
                     &c->sort_keys[0], &c->sort_keys[1], &c->sort_idx[0], &c->sort_idx[1], &c->sort_temp, &c->shard_stage};
    for (DevBuf *b : all) {
        b->release();
    }
    for (int k = 0; k < 2; ++k) {
        for (int pl = 0; pl < 5; ++pl) {
            c->ray_planes[k][pl].release();
        }
    }
    for (DevBuf &b : c->point_planes) {
        b.release();
    }
    c->nee_index.release();
    for (auto &up : c->unet_pass) { // (the UNet's weights and its fifteen tensors -- 1.3 GB at 1080p -- were leaked)
        up.weights.release();
        up.bias.release();
    }
    for (DevBuf &b : c->unet_tensor) {
        b.release();
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int rayhip_ctx_device_name(rayhip_ctx *c, char *buf, int cap) {
    snprintf(buf, size_t(cap), "%s (%s, %d CUs)", c->props.name, c->props.gcnArchName, c->props.multiProcessorCount);
    return 0;
}

int rayhip_upload_static(rayhip_ctx *c, const uint32_t *pmj02_samples, uint32_t count) {
    if (use_device(c)) {
        return 1;
    }
    if (count != uint32_t(RAND_DIMS_COUNT) * 2u * uint32_t(RAND_SAMPLES_COUNT)) {
        return fail("PMJ02 table must hold %u entries, got %u", RAND_DIMS_COUNT * 2 * RAND_SAMPLES_COUNT, count);
    }
    if (upload(c, c->pmj, pmj02_samples, size_t(count) * 4)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sc.pmj = c->pmj.as<uint32_t>();
    return 0;
}

int rayhip_resize(rayhip_ctx *c, int w, int h) {
    if (use_device(c)) {
        return 1;
    }
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) {
        return fail("bad frame size %dx%d (pixel coordinates are 16-bit)", w, h);
    }
    if (size_t(w) * size_t(h) > (size_t(1) << 30)) {
        return fail("frame of %dx%d pixels is too large (pixel indices are 32-bit)", w, h);
    }
    if (c->w == w && c->h == h) {
        return 0;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (alloc_frame(c, w, h, 1)) {
        return 1;
    }
    c->w = w, c->h = h;
    const size_t n = size_t(w) * size_t(h);
    // Resize zero-fills every buffer and arms required_samples (RendererCPU.h:266-295)
    float4 *bufs[] = {c->px.temp, c->px.full, c->px.half, c->px.raw, c->px.final_, c->px.base_color, c->px.depth_normals, c->px.variance};
    for (float4 *b : bufs) {
        HIP_TRY(hipMemsetAsync(b, 0, n * 16, c->stream));
    }
    k_fill_u16<<<grid_for(c, n, 256), 256, 0, c->stream>>>(c->px.required_samples, uint16_t(0xffff), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->adaptive_dirty = false;
    return 0;
}

int rayhip_clear(rayhip_ctx *c, const float rgba[4]) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w) {
        return fail("rayhip_clear before rayhip_resize");
    }
    const size_t n = size_t(c->w) * size_t(c->h);
    const float4 v = make_float4(rgba[0], rgba[1], rgba[2], rgba[3]);
    // RendererCPU.h:297-301: full, half <- c ; required_samples <- 0xffff
    k_fill_f4<<<grid_for(c, n, 256), 256, 0, c->stream>>>(c->px.full, v, n);
    k_fill_f4<<<grid_for(c, n, 256), 256, 0, c->stream>>>(c->px.half, v, n);
    k_fill_u16<<<grid_for(c, n, 256), 256, 0, c->stream>>>(c->px.required_samples, uint16_t(0xffff), n);
    HIP_TRY(hipGetLastError());
    c->adaptive_dirty = false;
    return 0;
}

// lights, their index list, the light tree (+ its per-node importance table) and the world-space corners of the triangle
// lights: everything an instance / light change replaces besides the top-level tree
static int upload_lights(rayhip_ctx *c, const rayhip_scene_desc *d) {
    if (upload(c, c->lights, d->lights, size_t(d->lights_count) * sizeof(*d->lights)) ||
        upload(c, c->li_indices, d->li_indices, size_t(d->li_indices_count) * sizeof(uint32_t)) ||
        upload(c, c->light_cwnodes, d->light_cwnodes, size_t(d->light_cwnodes_count) * sizeof(*d->light_cwnodes))) {
        return 1;
    }
    // node-only half of the light-tree importance, evaluated once per scene (shade_lights.h: decode_light_child)
    std::vector<float4> lc(size_t(d->light_cwnodes_count) * LIGHT_CHILDREN_STRIDE);
    for (uint32_t n = 0; n < d->light_cwnodes_count; ++n) {
        fill_light_children(d->light_cwnodes[n], &lc[size_t(n) * LIGHT_CHILDREN_STRIDE]);
    }
    if (upload(c, c->light_children, lc.data(), lc.size() * sizeof(float4))) {
        return 1;
    }
    // world-space corners of the TRI lights (shade_lights.h: fill_light_tri_geom)
    // (the light array is a sparse pool: only the slots li_indices[] names hold lights)
    std::vector<float4> tg(size_t(d->lights_count) * 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t k = 0; k < d->li_indices_count; ++k) {
        const uint32_t i = d->li_indices[k];
        if (i >= d->lights_count) {
            return fail("li_indices[%u] = %u is outside the light array", k, i);
        }
        const rayhip_light &l = d->lights[i];
        if (light_type(l) == LIGHT_TYPE_TRI) {
            const uint32_t tri = float_as_uint(l.params[0]), mi = float_as_uint(l.params[1]);
            if (mi >= d->mesh_instances_count || size_t(tri) * 3 + 2 >= d->vtx_indices_count) {
                return fail("triangle light %u refers to triangle %u of instance %u: out of range", i, tri, mi);
            }
        }
        fill_light_tri_geom(l, d->mesh_instances, d->vtx_indices, d->vertices, &tg[size_t(i) * 4]);
    }
    if (upload(c, c->light_tri_geom, tg.data(), tg.size() * sizeof(float4))) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream)); // `lc`, `tg` go out of scope
    return 0;
}


// the kernels' view of what is on the device (SceneView), after a full upload or an instance update
// distinct instances the top level of a (validated) scene holds: a leaf word of the top level stands for one instance (the reference's
// one-leaf tree is a root whose two links are the same leaf word, the second one behind a point box at the origin: Core.cpp:1191-1213)
static uint32_t count_top_level_instances(const rayhip_bvh2_node *nodes, const uint32_t nodes_count, const uint32_t root) {
    if (root == 0xffffffffu) {
        return 0;
    }
    std::vector<uint32_t> seen, todo(1, root);
    while (!todo.empty()) {
        const uint32_t w = todo.back();
        todo.pop_back();
        if ((w & BVH2_PRIM_COUNT_BITS) != 0) {
            const uint32_t mi = w & BVH2_PRIM_INDEX_BITS;
            if (std::find(seen.begin(), seen.end(), mi) == seen.end()) {
                if (seen.size() >= 2) {
                    return 3; // (more than one is all the caller asks)
                }
                seen.push_back(mi);
            }
        } else if (w < nodes_count) {
            todo.push_back(nodes[w].left_child), todo.push_back(nodes[w].right_child);
        }
    }
    return uint32_t(seen.size());
}

static void refresh_scene_view(rayhip_ctx *c, const rayhip_scene_desc *d, const uint32_t tlas_root, const rayhip_lbvh::Box &root_box,
                               const uint32_t live_instances) {
    SceneView &v = c->sc;
    // the pooled closest-hit kernel hands prepared rays from lane to lane; a ray can change lanes only while nothing is pending at the top
    // level, which is every ray of a scene with ONE instance (RAYHIP_POOL_ANY=1: the pooled kernel for any scene -- tests of its other path)
    c->pool_scene = (live_instances == 1 || getenv("RAYHIP_POOL_ANY") != nullptr) && d->mesh_instances_count < (1u << 24);
    v.nodes = c->nodes.as<rayhip_bvh2_node>(), v.tris = c->tris.as<rayhip_tri_accel>(), v.tri_pitch = c->tri_pitch, v.all_solid = getenv("RAYHIP_NO_ALL_SOLID") ? 0u : c->all_solid;
    v.tri_indices = c->tri_indices.as<uint32_t>(), v.tri_materials = c->tri_materials.as<rayhip_tri_mat_data>();
    v.materials = c->materials.as<rayhip_material>(), v.vertices = c->vertices.as<rayhip_vertex>();
    v.vtx_indices = c->vtx_indices.as<uint32_t>(), v.mesh_instances = c->mesh_instances.as<rayhip_mesh_instance>();
    v.lights = c->lights.as<rayhip_light>(), v.li_indices = c->li_indices.as<uint32_t>();
    v.light_children = c->light_children.as<float4>();
    v.light_tri_geom = c->light_tri_geom.as<float4>();
    v.tri_verts = c->tri_verts.as<float4>();
    v.tri_bitangents = c->tri_bitangents.as<float4>();
    v.env_qtree = c->env_qtree.as<float4>();
    for (int lod = 0, off = 0; lod < 16; ++lod) {
        v.env_qtree_offset[lod] = uint32_t(off);
        if (lod < d->env.qtree_levels) {
            off += 1 << (2 * (d->env.qtree_levels - 1 - lod));
        }
    }
    v.nodes4 = c->wide == 4 ? c->nodes4.as<Bvh4Node>() : nullptr;
    v.nodes8 = c->wide == 8 ? c->nodes8.as<Bvh8Node>() : nullptr;
    v.blas_root4 = c->wide ? c->blas_root4.as<uint32_t>() : nullptr;
    v.light_cwnodes = c->light_cwnodes.as<rayhip_light_cwbvh_node>(), v.textures = c->textures.as<rayhip_texture>();
    v.texels = c->texels.as<uint32_t>();
    memcpy(v.tex_table, c->tex_table, sizeof(v.tex_table));
    v.tex_flags = c->tex_flags;
    v.li_indices_count = d->li_indices_count;
    v.light_cwnodes_count = d->light_cwnodes_count;
    v.visible_lights_count = d->visible_lights_count;
    v.blocker_lights_count = d->blocker_lights_count;
    v.tlas_root = tlas_root;
    v.env = d->env;
    memcpy(c->bbox_min, d->bbox_min, 12), memcpy(c->bbox_max, d->bbox_max, 12);
    // ray-sort grid: true bounds of the TLAS root (Scene::GetBounds takes fminf for the max corner, SceneCPU.cpp:1553)
    for (int i = 0; i < 3; ++i) {
        const bool have = root_box.lo[i] <= root_box.hi[i];
        const float mn = have ? root_box.lo[i] : d->bbox_min[i], mx = have ? root_box.hi[i] : d->bbox_max[i];
        const float ext = mx - mn;
        c->sort_grid.root_min[i] = mn;
        c->sort_grid.inv_cell[i] = (ext > 0.0f && ext < 1e30f) ? 256.0f / ext : 0.0f;
    }
}


#define UPLOAD_TRACE(msg)                                                                                              \
    if (getenv("RAYHIP_TRACE_UPLOAD")) {                                                                               \
        fprintf(stderr, "rayhip_scene_upload: %8.1f ms  %s\n",                                                         \
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - upload_t0).count(), msg);  \
    }

int rayhip_scene_upload(rayhip_ctx *c, const rayhip_scene_desc *d_in) {
    if (use_device(c)) {
        return 1;
    }
    const auto upload_t0 = std::chrono::steady_clock::now();
    (void)upload_t0;
    UPLOAD_TRACE("begin")
    const rayhip_layout::AlignedDesc aligned(*d_in); // see bvh_layout.h
    const rayhip_scene_desc *d = &aligned.d;
    if (d->env.qtree_levels < 0 || d->env.qtree_levels > 16) {
        return fail("bad env-map quadtree depth %d", d->env.qtree_levels);
    }
    if (d->env.sky_map_spread_angle > 0.0f) {
        // With PhysicalSkyTexture the reference evaluates narrow rays analytically (ShadeSky*, AtmosphereRef.cpp: SURVEY
        // section 2, out of scope) and only wide ones through the baked map; rendering all of them from the map would be
        // a silently different image.
        return fail("the physical sky (environment_t::sky_map_spread_angle > 0) is not supported by the HIP backend");
    }
    {
        size_t quads = 0;
        for (int lod = 0; lod < d->env.qtree_levels; ++lod) {
            quads += size_t(1) << (2 * (d->env.qtree_levels - 1 - lod));
        }
        if (size_t(d->env_qtree_count) != quads * 4) {
            return fail("env_qtree holds %u floats, %d levels need %zu", d->env_qtree_count, d->env.qtree_levels, quads * 4);
        }
    }
    bool all_sides_solid = false; // (over the triangles the trees reach: the pools are sparse, an unused slot is all zeros)
    { // every index a kernel would follow without a bound of its own (scene_validate.h)
        std::string why;
        if (!rayhip_validate::validate(*d, why, &all_sides_solid)) {
            return fail("%s", why.c_str());
        }
    }
    UPLOAD_TRACE("validated")
    // Leaf refinement (scene_rebuild.h): the scene's trees are kept, every leaf with more than `leaf_max` triangles is replaced
    // by a subtree of the linear builder.  RAYHIP_REFINE_LEAVES=<leaf_max> (0 = leave the trees as they are; default 2).
    // RAYHIP_REBUILD_BVH=<leaf_max>: both levels rebuilt from the triangles and instance transforms instead (lbvh.h).
    rayhip_rebuild::Rebuilt rebuilt;
    rayhip_scene_desc d_rebuilt = *d;
    {
        int refine = 2, rebuild = 0;
        if (const char *e = getenv("RAYHIP_REFINE_LEAVES")) {
            refine = std::max(0, std::min(8, atoi(e)));
        }
        if (const char *e = getenv("RAYHIP_REBUILD_BVH")) {
            rebuild = std::max(0, std::min(8, atoi(e)));
        }
        if (rebuild > 0 || refine > 0) {
            // the builder itself runs on the device (lbvh.hip.h); RAYHIP_BVH_BUILD_ON_HOST=1 runs the same element functions
            // as host loops instead (A/B and debugging: the two produce identical arrays)
            const bool on_host = getenv("RAYHIP_BVH_BUILD_ON_HOST") != nullptr && atoi(getenv("RAYHIP_BVH_BUILD_ON_HOST")) != 0;
            auto build = [&](const rayhip_lbvh::Input &in, rayhip_lbvh::Output &out, std::string &why) {
                if (on_host) {
                    out = rayhip_lbvh::build_host(in);
                    return true;
                }
                return rayhip_lbvh::build_device(c->stream, in, out, why);
            };
            rebuilt = rebuild > 0 ? rayhip_rebuild::rebuild_with(*d, uint32_t(rebuild), build) : rayhip_rebuild::refine_with(*d, uint32_t(refine), build);
            UPLOAD_TRACE(rebuild > 0 ? "both levels rebuilt" : "leaves refined")
            if (!rebuilt.ok) {
                return fail("acceleration-structure %s failed: %s", rebuild > 0 ? "rebuild" : "refinement", rebuilt.why.c_str());
            }
            d_rebuilt.nodes = rebuilt.nodes.data(), d_rebuilt.nodes_count = uint32_t(rebuilt.nodes.size());
            d_rebuilt.tris = rebuilt.tris.data(), d_rebuilt.tris_count = uint32_t(rebuilt.tris.size());
            d_rebuilt.tri_indices = rebuilt.tri_indices.data(), d_rebuilt.tri_indices_count = uint32_t(rebuilt.tri_indices.size());
            d_rebuilt.mesh_instances = rebuilt.mesh_instances.data();
            d_rebuilt.tlas_root = rebuilt.tlas_root;
            d = &d_rebuilt;
            std::string why;
            if (!rayhip_validate::validate(*d, why)) {
                return fail("rebuilt scene: %s", why.c_str());
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
#define UP(field)                                                                                                      \
    if (upload(c, c->field, d->field, size_t(d->field##_count) * sizeof(*d->field))) {                                 \
        return 1;                                                                                                      \
    }
    // HBM layout pass (bvh_layout.h): depth-first node order with sibling pairs in one 128-byte line, triangles in
    // leaf-visit order.  Off by default since round 3 (RAYHIP_LAYOUT=1 switches it on): the kernels walk the 4-wide collapse,
    // whose node order is the collapse's own, and the triangle records come out of the leaf refinement grouped leaf by leaf
    // in the order of a depth-first walk already -- measured, Bistro-class scene: K2 2.11 ms per iteration with the pass,
    // 2.13 without, Sponza-class 1.70 / 1.70 (profiles/r03/experiments/variants_layout_*.txt) -- for 181 ms of host time
    // per upload.  Without leaf refinement (RAYHIP_REFINE_LEAVES=0) the pass still runs: the reference builder's order is poor.
    rayhip_layout::Result lay;
    {
        const char *e = getenv("RAYHIP_LAYOUT"), *off = getenv("RAYHIP_NO_LAYOUT");
        const bool want_layout = e ? e[0] == '1' : !rebuilt.ok;
        if (want_layout && !(off && off[0] == '1')) {
            lay = rayhip_layout::optimize(*d);
        }
    }
    UPLOAD_TRACE(lay.applied ? "layout applied" : lay.why_not)
    uint32_t tlas_root = d->tlas_root;
    { // room behind the nodes for top-level trees rebuilt on the device later (rayhip_scene_update_instances)
        const size_t n_now = lay.applied ? lay.nodes.size() : size_t(d->nodes_count);
        c->nodes_used = uint32_t(n_now);
        c->nodes_reserved = uint32_t(std::max<size_t>(8192, 8 * size_t(d->mesh_instances_count)));
        c->tlas_half = 0;
        if (c->nodes.alloc((n_now + c->nodes_reserved) * sizeof(rayhip_bvh2_node))) {
            return 1;
        }
    }
    if (lay.applied) {
        tlas_root = lay.tlas_root;
    }
    // Wide quantised BLAS trees over the node order just decided.  RAYHIP_BVH_WIDTH: 8 (default; rt_bvh8.h -- it also decides the
    // order of the triangle records and re-bases the BVH2's leaf words onto it), 4 (rt_bvh4.h: round 2's form), 2 keeps the
    // kernels on the reference's BVH2 (RAYHIP_NO_BVH4=1 says the same; A/B measurements)
    int wide = 0;
    std::vector<uint32_t> blas_root4;
    {
        // Default 4: measured on the MI355X (profiles/r03/experiments/variants_bvh8.txt) the 8-wide walk performs 29 % fewer node
        // visits and 17 % more triangle tests per ray and takes the same time -- 2.17 vs 2.11 ms per iteration on the Bistro-class
        // scene, within 1 % on the other workloads: the kernel is bound by random cache-line fetches per second, and an 80-byte
        // node costs two 64-byte sectors.  The narrower form needs no dynamic-programming collapse at upload (0.5 s) either.
        int want = 4;
        if (const char *e = getenv("RAYHIP_BVH_WIDTH")) {
            want = atoi(e);
        }
        if (const char *e = getenv("RAYHIP_NO_BVH4")) {
            want = e[0] == '1' ? 2 : want;
        }
        std::vector<rayhip_bvh2_node> nodes2_own; // a copy the 8-wide build may re-base (the caller's arrays are const)
        rayhip_bvh2_node *n2 = nullptr;
        if (lay.applied) {
            n2 = lay.nodes.data();
        } else {
            nodes2_own.assign(d->nodes, d->nodes + d->nodes_count);
            n2 = nodes2_own.data();
        }
        const uint32_t n2_count = lay.applied ? uint32_t(lay.nodes.size()) : d->nodes_count;
        const rayhip_mesh_instance *mis = lay.applied ? lay.mesh_instances.data() : d->mesh_instances;
        const rayhip_tri_accel *tris_in = lay.applied ? lay.tris.data() : d->tris;
        const uint32_t *tri_indices_in = lay.applied ? lay.tri_indices.data() : d->tri_indices;
        size_t n_tris = lay.applied ? lay.tris.size() : size_t(d->tris_count);
        rayhip_bvh8::Result b8;
        if (want == 8) {
            b8 = rayhip_bvh8::build(n2, n2_count, mis, d->mesh_instances_count, tlas_root, tris_in, tri_indices_in, uint32_t(n_tris));
            UPLOAD_TRACE(b8.ok ? "bvh8 built" : b8.why_not)
        }
        if (b8.ok && !b8.nodes.empty()) {
            tris_in = b8.tris.data(), tri_indices_in = b8.tri_indices.data(), n_tris = b8.tris.size();
        }
        if (upload(c, c->nodes, n2, size_t(n2_count) * sizeof(rayhip_bvh2_node)) ||
            upload(c, c->tris, tris_in, n_tris * sizeof(rayhip_tri_accel)) ||
            upload(c, c->tri_indices, tri_indices_in, n_tris * sizeof(uint32_t)) ||
            upload(c, c->mesh_instances, mis, size_t(d->mesh_instances_count) * sizeof(rayhip_mesh_instance))) {
            return 1;
        }
        // the walks' triangle table: the reference's 48-byte array as it is; RAYHIP_TRI_PITCH=64 re-pitches it so that every record lies in
        // its own 64-byte sector (half of the 48-byte records straddle two) -- measured neutral (K2 2.14 against 2.12 ms,
        // profiles/r03/experiments/variants_tripitch.txt: the kernel is bound by instruction issue, not by sectors), so it stays an option
        c->tri_pitch = 3;
        if (n_tris && getenv("RAYHIP_TRI_PITCH") && atoi(getenv("RAYHIP_TRI_PITCH")) == 64) {
            DevBuf padded;
            if (padded.alloc(n_tris * 64)) {
                return 1;
            }
            k_pad_tris<<<unsigned((n_tris * 4 + 255) / 256), 256, 0, c->stream>>>(c->tris.as<float4>(), padded.as<float4>(), n_tris);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->tris.swap(padded);
            padded.release();
            c->tri_pitch = 4;
        }
        size_t wide_bytes = 0;
        if (b8.ok && !b8.nodes.empty()) {
            if (upload(c, c->nodes8, b8.nodes.data(), b8.nodes.size() * sizeof(Bvh8Node)) ||
                upload(c, c->blas_root4, b8.blas_root8.data(), b8.blas_root8.size() * sizeof(uint32_t))) {
                return 1;
            }
            wide = 8, blas_root4 = b8.blas_root8, wide_bytes = b8.nodes.size() * sizeof(Bvh8Node);
        } else if (want == 4 || want == 8) {
            // the collapse runs on the device over the nodes just uploaded (bvh4_build.hip.h); RAYHIP_BVH_BUILD_ON_HOST=1: the host
            // driver over the same element functions (A/B: the same tree in another node order)
            const bool on_host = getenv("RAYHIP_BVH_BUILD_ON_HOST") != nullptr && atoi(getenv("RAYHIP_BVH_BUILD_ON_HOST")) != 0;
            if (on_host) {
                rayhip_bvh4::Result b4 = rayhip_bvh4::build(n2, n2_count, mis, d->mesh_instances_count, tlas_root);
                if (b4.ok && !b4.nodes.empty()) {
                    if (upload(c, c->nodes4, b4.nodes.data(), b4.nodes.size() * sizeof(Bvh4Node)) ||
                        upload(c, c->blas_root4, b4.blas_root4.data(), b4.blas_root4.size() * sizeof(uint32_t))) {
                        return 1;
                    }
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    wide = 4, blas_root4 = b4.blas_root4, wide_bytes = b4.nodes.size() * sizeof(Bvh4Node);
                }
            } else {
                std::vector<uint32_t> roots;
                uint32_t n_wide = 0;
                std::string why;
                if (rayhip_bvh4::collect_roots(n2, n2_count, mis, d->mesh_instances_count, tlas_root, roots, blas_root4) && !roots.empty()) {
                    if (c->nodes4.alloc(size_t(n2_count) * sizeof(Bvh4Node))) {
                        return 1;
                    }
                    if (!rayhip_bvh4::build_device(c->stream, c->nodes.as<rayhip_bvh2_node>(), n2_count, roots, c->nodes4.as<Bvh4Node>(), n_wide, why)) {
                        return fail("4-wide collapse failed: %s", why.c_str());
                    }
                    if (upload(c, c->blas_root4, blas_root4.data(), blas_root4.size() * sizeof(uint32_t))) {
                        return 1;
                    }
                    wide = 4, wide_bytes = size_t(n_wide) * sizeof(Bvh4Node);
                } else {
                    blas_root4.clear();
                }
            }
            UPLOAD_TRACE(wide == 4 ? "bvh4 built" : "no wide BLAS")
        }
        HIP_TRY(hipStreamSynchronize(c->stream)); // the builders' arrays go out of scope
        // "small": the BLAS working set (nodes + triangle records) fits one XCD's 4 MB L2 with room to spare
        c->small_scene = wide != 0 && getenv("RAYHIP_NO_SMALL") == nullptr && wide_bytes + n_tris * sizeof(rayhip_tri_accel) <= (size_t(2) << 20);
        // the meshes in use, for rayhip_scene_update_instances: the roots of their trees as uploaded
        rayhip_update::collect_mesh_refs(n2, n2_count, mis, d->mesh_instances_count, tlas_root, wide ? blas_root4.data() : nullptr, c->mesh_refs);
    }
    UPLOAD_TRACE("bvh uploaded")
    UP(tri_materials)
    // is there a triangle side that is not plainly solid?  (the closest-hit kernels skip the per-hit material fetch when not; round 4: judged
    // over the reachable triangles -- the headline scene's pool has unused slots, and the flag had never been set for it)
    c->all_solid = all_sides_solid ? 1u : 0u;
    UPLOAD_TRACE(all_sides_solid ? "every reachable triangle side is solid" : "some triangle sides are not solid")
    UP(materials)
    UP(vertices)
    UP(vtx_indices)
    { // vertices gathered per triangle (shade_point.h: fill_tri_verts), on the device from the arrays just uploaded
        const uint32_t n_tris = d->vtx_indices_count / 3;
        if (c->tri_verts.alloc(size_t(n_tris) * TRI_VERTS_STRIDE * sizeof(float4)) ||
            c->tri_bitangents.alloc(size_t(n_tris) * TRI_BITANGENTS_STRIDE * sizeof(float4))) {
            return 1;
        }
        if (n_tris) {
            k_fill_tri_verts<<<(n_tris + 255) / 256, 256, 0, c->stream>>>(c->vertices.as<rayhip_vertex>(), d->vertices_count,
                                                                          c->vtx_indices.as<uint32_t>(), n_tris, c->tri_verts.as<float4>(),
                                                                          c->tri_bitangents.as<float4>());
            HIP_TRY(hipGetLastError());
        }
    }
    UPLOAD_TRACE("tri_verts done")
    if (upload_lights(c, d)) {
        return 1;
    }
    UPLOAD_TRACE("lights done")
    UP(textures)
    UP(texels)
    UP(env_qtree)
#undef UP
    HIP_TRY(hipStreamSynchronize(c->stream)); // host arrays may go away after this call
    c->wide = wide;
    memcpy(c->tex_table, d->tex_table, sizeof(c->tex_table));
    c->textures_count = d->textures_count;
    c->tex_flags = d->texture_flags;
    c->geometry = {d->vertices_count, d->vtx_indices_count, d->tri_materials_count, d->materials_count};
    {
        rayhip_lbvh::Box root_box = rayhip_lbvh::empty_box();
        if (d->tlas_root != 0xffffffffu && d->tlas_root < d->nodes_count) {
            root_box = rayhip_rebuild::node_box(d->nodes[d->tlas_root]);
        }
        refresh_scene_view(c, d, tlas_root, root_box, count_top_level_instances(d->nodes, d->nodes_count, d->tlas_root));
    }
    c->have_scene = true;
    UPLOAD_TRACE("done")
    return 0;
}

// ---- instance / light / environment update without a new upload of the geometry -----------------------------------------
// What SceneBase::SetMeshInstanceTransform / AddMeshInstance / RemoveMeshInstance / AddLight / RemoveLight / SetEnvironment /
// Finalize change (SceneCPU.cpp:1004-1094, 1103-1162 RebuildTLAS, 1411-1521 RebuildLightTree): the instance array, the
// top-level tree, the light arrays and the environment.  The top level is rebuilt ON THE DEVICE by the linear builder
// (lbvh.hip.h) over the instance boxes; the host's own top-level tree in `d` (node numbering of the host arrays, which the
// device does not share after the layout pass) only tells which instance slots are alive and their world-space boxes.
// Returns 0, 1 = error, 2 = the scene needs rayhip_scene_upload (an instance of a mesh that is not on the device, geometry
// arrays of another size, no room for the tree).
int rayhip_scene_bvh_width(rayhip_ctx *c) { return !c || !c->have_scene ? 0 : c->wide ? c->wide : 2; }

int rayhip_closest_hit_form(rayhip_ctx *c) {
    if (!c || !c->have_scene || !c->wide || !c->refill_waves) {
        return 0;
    }
    return (c->wide == 4 && c->refill_pool && c->pool_scene) ? 2 : 1;
}

int rayhip_scene_update_instances(rayhip_ctx *c, const rayhip_scene_desc *d) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->have_scene) {
        (void)fail("rayhip_scene_update_instances before rayhip_scene_upload");
        return 2;
    }
    const auto upload_t0 = std::chrono::steady_clock::now();
    (void)upload_t0;
    if (d->vertices_count != c->geometry.vertices || d->vtx_indices_count != c->geometry.vtx_indices ||
        d->tri_materials_count != c->geometry.tri_materials || d->materials_count != c->geometry.materials) {
        (void)fail("geometry arrays changed size since the last upload");
        return 2;
    }
    if (d->env.qtree_levels < 0 || d->env.qtree_levels > 16) {
        return fail("bad env-map quadtree depth %d", d->env.qtree_levels);
    }
    if (d->env.sky_map_spread_angle > 0.0f) {
        return fail("the physical sky (environment_t::sky_map_spread_angle > 0) is not supported by the HIP backend");
    }
    {
        size_t quads = 0;
        for (int lod = 0; lod < d->env.qtree_levels; ++lod) {
            quads += size_t(1) << (2 * (d->env.qtree_levels - 1 - lod));
        }
        if (size_t(d->env_qtree_count) != quads * 4) {
            return fail("env_qtree holds %u floats, %d levels need %zu", d->env_qtree_count, d->env.qtree_levels, quads * 4);
        }
        for (const uint32_t handle : {d->env.env_map, d->env.back_map}) {
            if (handle != 0xffffffffu &&
                ((handle >> 28) >= 8u || uint64_t(c->tex_table[handle >> 28]) + (handle & 0x00ffffffu) >= c->textures_count)) {
                return fail("environment map handle outside the texture table on the device");
            }
        }
    }
    rayhip_update::Plan up;
    {
        std::string why;
        const int rc = rayhip_update::plan(*d, c->mesh_refs, up, why);
        if (rc) {
            (void)fail("%s", why.c_str());
            return rc;
        }
    }
    const std::vector<uint32_t> &live = up.live;
    std::vector<rayhip_mesh_instance> &mis = up.instances;
    std::vector<uint32_t> &root4 = up.root4;
    {
        rayhip_scene_desc lights_only = *d;
        lights_only.mesh_instances = mis.data();
        std::string why;
        if (!rayhip_validate::validate_lights(lights_only, why)) {
            return fail("%s", why.c_str());
        }
    }
    uint32_t tlas_root = 0xffffffffu;
    rayhip_lbvh::Box root_box = rayhip_lbvh::empty_box();
    if (!live.empty()) {
        const std::vector<uint32_t> group(live.size(), 0);
        const rayhip_lbvh::Input ti = rayhip_update::top_level_input(up, group);
        rayhip_lbvh::Output tlas;
        std::string why;
        if (!rayhip_lbvh::build_device(c->stream, ti, tlas, why)) {
            return fail("top-level build failed: %s", why.c_str());
        }
        // two halves, used in turn: the tree the scene view still points at is never overwritten, so a failure further
        // down (rc 1) leaves a context that renders the previous top level
        const uint32_t half = c->nodes_reserved / 2;
        if (tlas.nodes.size() > half || tlas.group_root.empty() || tlas.group_root[0] == 0xffffffffu) {
            (void)fail("no room for a top-level tree of %zu nodes", tlas.nodes.size());
            return 2;
        }
        const uint32_t base = c->nodes_used + c->tlas_half * half;
        c->tlas_half ^= 1u;
        tlas_root = rayhip_update::relocate_top_level(tlas, up, base);
        root_box = tlas.bounds;
        UPLOAD_TRACE("top level built")
        // pending passes read the old tree: the caller flushed (RendererHIP) or synchronises through the stream order here
        HIP_TRY(hipMemcpyAsync(c->nodes.as<rayhip_bvh2_node>() + base, tlas.nodes.data(), tlas.nodes.size() * sizeof(rayhip_bvh2_node),
                               hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream)); // `tlas` goes out of scope
    }
    if (upload(c, c->mesh_instances, mis.data(), mis.size() * sizeof(rayhip_mesh_instance)) ||
        (c->wide && upload(c, c->blas_root4, root4.data(), root4.size() * sizeof(uint32_t)))) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    {
        rayhip_scene_desc with_roots = *d; // triangle lights are placed by their instance's transform only
        if (upload_lights(c, &with_roots) ||
            upload(c, c->env_qtree, d->env_qtree, size_t(d->env_qtree_count) * sizeof(float))) {
            return 1;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    refresh_scene_view(c, d, tlas_root, root_box, uint32_t(live.size()));
    UPLOAD_TRACE("instances updated")
    return 0;
}

int rayhip_set_filter_table(rayhip_ctx *c, const float *table, int count) {
    if (use_device(c)) {
        return 1;
    }
    if (count != FILTER_TABLE_SIZE) {
        return fail("filter table must have %d entries", FILTER_TABLE_SIZE);
    }
    if (upload(c, c->filter_table, table, size_t(count) * 4)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_set_tonemap_lut(rayhip_ctx *c, int view_transform, const uint32_t *lut, int dims) {
    if (use_device(c)) {
        return 1;
    }
    if (view_transform <= 0 || !lut || dims < 2 || dims > 256) {
        return fail("bad tonemap table (view transform %d, dims %d)", view_transform, dims);
    }
    const size_t bytes = size_t(dims) * size_t(dims) * size_t(dims) * sizeof(uint32_t);
    HIP_TRY(hipStreamSynchronize(c->stream)); // a pass in flight may still read the old table
    if (c->tonemap_lut.alloc(bytes)) {
        return 1;
    }
    HIP_TRY(hipMemcpy(c->tonemap_lut.p, lut, bytes, hipMemcpyHostToDevice));
    c->lut_transform = view_transform, c->lut_dims = dims;
    return 0;
}

int rayhip_scene_upload_blob(rayhip_ctx *c, const void *blob, size_t size, rayhip_camera *out_cam) {
    rayhip_scene_desc d;
    const float *ft = nullptr;
    int ftn = 0;
    std::string err;
    rayhip_blob::Extras extras;
    if (!rayhip_blob::deserialize(blob, size, d, *out_cam, &ft, &ftn, err, &extras)) {
        return fail("%s", err.c_str());
    }
    if (extras.tonemap_lut && out_cam->view_transform != 0 &&
        rayhip_set_tonemap_lut(c, out_cam->view_transform, extras.tonemap_lut, extras.tonemap_lut_dims)) {
        return 1;
    }
    if (rayhip_scene_upload(c, &d)) {
        return 1;
    }
    if (ft && rayhip_set_filter_table(c, ft, ftn)) {
        return 1;
    }
    return 0;
}

int rayhip_scene_update_instances_blob(rayhip_ctx *c, const void *blob, size_t size, rayhip_camera *out_cam) {
    rayhip_scene_desc d;
    const float *ft = nullptr;
    int ftn = 0;
    std::string err;
    rayhip_blob::Extras extras;
    if (!rayhip_blob::deserialize(blob, size, d, *out_cam, &ft, &ftn, err, &extras)) {
        return fail("%s", err.c_str());
    }
    return rayhip_scene_update_instances(c, &d);
}

// do the allocated per-iteration pixel buffers and wavefront state hold a pass of `n` iterations over `rect`?
static bool pass_fits(const rayhip_ctx *c, const int rect[4], int n) {
    const Layering L = make_layering(c->w, c->h, n);
    const size_t vbytes = size_t(c->w) * size_t(c->h) * size_t(L.cols) * size_t(layer_rows(L)) * 16u;
    return vbytes <= c->px_temp.bytes && (n <= 1 || (vbytes <= c->px_aux_base.bytes && vbytes <= c->px_aux_dn.bytes)) &&
           pass_slots(c, c->w, c->h, rect[2], rect[3], n) + size_t(WAVE) * QUEUE_MAX_STRIPES <= c->slots_cap;
}
// grow the per-iteration pixel buffers / the wavefront state if a pass of `n` iterations over `rect` needs more
static bool rect_inside(const rayhip_ctx *c, const int rect[4]) {
    return rect[0] >= 0 && rect[1] >= 0 && rect[2] > 0 && rect[3] > 0 && rect[0] <= c->w - rect[2] && rect[1] <= c->h - rect[3];
}
static int ensure_pass(rayhip_ctx *c, const int rect[4], int n) {
    if (!rect_inside(c, rect)) { // before anything is (re)allocated for it
        return fail("rect outside the frame");
    }
    if (!pass_fits(c, rect, n)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (alloc_frame(c, c->w, c->h, n)) {
            return 1;
        }
    }
    return 0;
}

// K5 of one bounce: shade the rays of queue `bounce` in ray buffer `cur` -> secondary rays into queue bounce + 1 of the
// other ray buffer, shadow rays into shadow queue `bounce`, radiance into the per-iteration pixel buffer.  One place for
// rayhip_render and the kernel-level hook rayhip_k_shade.
static void launch_shade(rayhip_ctx *c, const rayhip_camera &cam, int iteration, int bounce, int cur, size_t nslots, uint32_t stripes,
                         int gtrace, int vw, float mix_factor, const Layering &layers) {
    ShadeLaunch a;
    a.sc = c->sc;
    a.sp = make_shade_params(cam, iteration, bounce);
    a.rays_in = c->rays[cur], a.rays_out = c->rays[cur ^ 1];
    a.hits = c->hits, a.shadow = c->shadow, a.deferred = c->deferred, a.points = c->points;
    a.in = c->ray_queue(bounce, nslots, stripes), a.pts = c->point_queue(bounce, nslots, stripes);
    a.out_rays = c->ray_queue(bounce + 1, nslots, stripes), a.out_shadow = c->shadow_queue(bounce, nslots, stripes);
    a.out_deferred = c->deferred_queue(bounce, nslots, stripes), a.nee = c->nee_queue(bounce, nslots, stripes);
    a.px = c->px, a.layers = layers, a.vw = vw, a.mix_factor = mix_factor;
    a.bounce = bounce, a.grid = gtrace, a.split = c->shade_split, a.stream = c->stream;
    shade::launch(a);
}

// One wavefront pass over `count` consecutive iterations of the rect (count == 1: the plain case; > 1: layered, see
// Layering in rt_base.h).  The caller has checked that a batch is admissible.
static int render_pass(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration, int count_iterations,
                       uint32_t flags, rayhip_stats *stats) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w || !c->have_scene || !c->pmj.p || !c->filter_table.p) {
        return fail("rayhip_render needs resize + upload_static + scene_upload + set_filter_table first");
    }
    if (iteration < 1) {
        return fail("iteration is 1-based");
    }
    if (cam->type != 0 /* eCamType::Persp */) {
        return fail("only perspective cameras are supported");
    }
    if (cam->view_transform != 0 /* eViewTransform::Standard */ && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    if (!rect_inside(c, rect)) {
        return fail("rect outside the frame");
    }
    const int max_depth = cam->pass_settings.max_total_depth;
    if (max_depth + 2 > MAX_BOUNCE_SLOTS) {
        return fail("max_total_depth too large");
    }
    const bool count_wide = (flags & RAYHIP_FLAG_COUNT_WIDE) != 0;
    const bool count = !count_wide && (flags & RAYHIP_FLAG_COUNT_TRAVERSAL) != 0;
    const bool sort_rays = (flags & RAYHIP_FLAG_SORT_RAYS) != 0;
    if (cam->pass_settings.variance_threshold != 0.0f) {
        c->adaptive_dirty = true;
    }
    hipStream_t s = c->stream;
    const size_t npix = size_t(rect[2]) * size_t(rect[3]);
    const Layering layers = make_layering(c->w, c->h, count_iterations);
    const int vw = virtual_width(layers); // row pitch of the per-iteration pixel buffers
    // ray slots: the 8x8 tiles the ray generator walks (this rank's share under a shard), one set per iteration in flight
    const RayGenTiling tiling = make_raygen_tiling(c->w, c->h, rect[2], rect[3], c->shard);
    const size_t nslots = size_t(tiling.tiles) * 64u * size_t(count_iterations);
    if (!pass_fits(c, rect, count_iterations)) {
        return fail("internal: pass of %d iterations exceeds the allocated wavefront state", count_iterations);
    }
    const int gw = c->grid_waves;
    const int gtrace = int(std::min<size_t>(size_t(gw), nslots / WAVE));
    unsigned long long *tc = c->trav_counters.as<unsigned long long>();
    uint32_t *spill = c->stack_spill.as<uint32_t>();
    const TraceParams tp_ = make_trace_params(*cam, c->sc.tlas_root, iteration);

    // striped queues unless a stage needs one dense ray array (the sort)
    const uint32_t stripes = sort_rays ? 1u : QUEUE_MAX_STRIPES;
    // K2 launcher (instrumented variant on request)
    auto launch_closest = [&](const RaySoA &r, const RayQueue &q, int init_hits) {
        const int wide = c->wide;
#define K2_ARGS c->sc, tp_, r, c->hits, q, init_hits, spill, tc, layers
        if (count_wide && wide == 8) {
            k_trace_closest<true, 8><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (count_wide && wide == 4) {
            k_trace_closest<true, 4><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (count) {
            k_trace_closest<true, 0><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide && c->refill_waves && c->refill_primary_whole && !init_hits) {
            // primary rays (coherent): the flat kernel, chunks taken whole (RAYHIP_REFILL=3)
            if (wide == 8) {
                k_trace_closest_refill<8, WAVE><<<gtrace, WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers);
            } else {
                k_trace_closest_refill<4, WAVE><<<gtrace, WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers);
            }
        } else if (wide == 4 && c->refill_pool && c->pool_scene && init_hits) {
            // secondary bounces, pooled kernel (grid: as for the refill kernel below)
            const int want = int(std::min<size_t>(size_t(c->pool_waves), std::max<size_t>(size_t(c->pool_resident), nslots / WAVE / 8)));
            k_trace_closest_pool<><<<std::min(gtrace, want), WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers);
        } else if (wide && c->refill_waves && !(c->refill_secondary_only && !init_hits)) {
            // blocks: enough to even out the end of the launch (16 per wave slot on a full-size pass), but never so many that a
            // block gets fewer than ~8 chunks of 64 rays -- below that the kernel degenerates into the plain one with extra
            // set-up per block (a rank of 8 at 20 spp: 5.2 M rays per pass; 16 blocks per slot 6.6 ms, 4: 5.97, plain 5.98)
            const int want = int(std::min<size_t>(size_t(c->refill_waves), std::max<size_t>(size_t(c->refill_resident), nslots / WAVE / 8)));
            if (wide == 8) {
                k_trace_closest_refill<8><<<std::min(gtrace, want), WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers);
            } else {
                k_trace_closest_refill<4><<<std::min(gtrace, want), WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers);
            }
        } else if (wide == 8 && (c->small_scene || c->tune_primary_waves == 5)) {
            k_trace_closest<false, 8, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide == 8) {
            k_trace_closest<false, 8><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide == 4 && (c->small_scene || c->tune_primary_waves == 5)) {
            k_trace_closest<false, 4, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide == 4) {
            k_trace_closest<false, 4><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else {
            k_trace_closest<false, 0><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        }
#undef K2_ARGS
    };

    StageTimer tm(c, stats != nullptr || (flags & RAYHIP_FLAG_TIME_STAGES) != 0);

    if (c->clear_queues(max_depth + 2, s)) {
        return fail("queue counter clear failed");
    }

    const RayGenParams rg = make_raygen_params(*cam, c->w, c->h, rect, iteration, c->shard);
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    const float mix_factor = 1.0f / float(iteration);

    const bool trace_launch = getenv("RAYHIP_TRACE_LAUNCH") != nullptr; // (diagnostics: host time of the first calls of a pass)
    const auto h0 = std::chrono::steady_clock::now();
    if (tm.mark(ST_GEN, -1)) {
        return 1;
    }
    const auto h1 = std::chrono::steady_clock::now();
    k_raygen<<<grid_for(c, nslots, 256), 256, 0, s>>>(rg, c->sc.pmj, c->filter_table.as<float>(), c->px.required_samples,
                                                      c->rays[0], c->hits, c->ray_queue(0, nslots, stripes), layers, tiling);
    const auto h2 = std::chrono::steady_clock::now();
    if (tm.mark(ST_PTRACE, 0)) {
        return 1;
    }
    if (trace_launch) {
        const auto h3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "rayhip pass start (host): event record %.0f us, k_raygen launch %.0f us, event record %.0f us\n", us(h0, h1), us(h1, h2), us(h2, h3));
    }
    if (c->sc.tlas_root != 0xffffffffu) {
        launch_closest(c->rays[0], c->ray_queue(0, nslots, stripes), 0);
    }
    int cur = 0;
    for (int bounce = 0; bounce <= max_depth; ++bounce) {
        if (bounce > 0) {
            if (sort_rays) {
                // K6-K8 (RendererVK.cpp:641-652): key -> radix sort of (key, index) -> gather into the idle ray buffer
                if (tm.mark(ST_SORT, -1)) {
                    return 1;
                }
                k_ray_keys<<<grid_for(c, npix, 256), 256, 0, s>>>(c->rays[cur], c->ray_count(bounce), uint32_t(npix), c->sort_grid,
                                                                  c->sort_keys[0].as<uint32_t>(), c->sort_idx[0].as<uint32_t>(), c->sort_key_mode);
                HIP_TRY(sort_pairs(c->sort_temp.p, c->sort_temp.bytes, c->sort_keys[0].as<uint32_t>(),
                                   c->sort_keys[1].as<uint32_t>(), c->sort_idx[0].as<uint32_t>(), c->sort_idx[1].as<uint32_t>(),
                                   npix, ray_sort_key_bits(c->sort_key_mode), s));
                k_reorder_rays<<<grid_for(c, npix, 256), 256, 0, s>>>(c->rays[cur], c->rays[cur ^ 1], c->sort_idx[1].as<uint32_t>(),
                                                                      c->ray_count(bounce));
                cur ^= 1;
            }
            if (tm.mark(ST_STRACE, 0)) {
                return 1;
            }
            launch_closest(c->rays[cur], c->ray_queue(bounce, nslots, stripes), 1);
            if (c->sc.visible_lights_count != 0) {
                k_intersect_area_lights<<<gtrace, WAVE, 0, s>>>(c->sc, c->rays[cur], c->hits, c->ray_queue(bounce, nslots, stripes));
            }
        }
        if (tm.mark(bounce == 0 ? ST_PSHADE : ST_SSHADE, -1)) {
            return 1;
        }
        launch_shade(c, *cam, iteration, bounce, cur, nslots, stripes, gtrace, vw, mix_factor, layers);
        if (tm.mark(bounce == 0 ? ST_PSHADOW : ST_SSHADOW, 1)) {
            return 1;
        }
        const float limit = shadow_clamp_limit(*cam, bounce);
        if (c->sc.blocker_lights_count != 0) {
            k_shadow_blockers<<<gtrace, WAVE, 0, s>>>(c->sc, c->shadow, c->shadow_queue(bounce, nslots, stripes));
        }
        {
            const int wide = c->wide;
#define K3_ARGS c->sc, tp, c->shadow, c->shadow_queue(bounce, nslots, stripes), limit, vw, c->px.temp, nullptr, spill, tc + TRAV_COUNTER_WORDS, layers
            if (count_wide && wide == 8) {
                k_trace_shadow<true, 8><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (count_wide && wide == 4) {
                k_trace_shadow<true, 4><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (count) {
                k_trace_shadow<true, 0><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 8 && (c->small_scene || c->tune_shadow_waves == 5)) {
                k_trace_shadow<false, 8, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 8) {
                k_trace_shadow<false, 8><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 4 && (c->small_scene || c->tune_shadow_waves == 5)) {
                k_trace_shadow<false, 4, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 4) {
                k_trace_shadow<false, 4><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else {
                k_trace_shadow<false, 0><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            }
#undef K3_ARGS
        }
        cur ^= 1;
    }
    if (tm.mark(-1, -1)) {
        return 1;
    }
    AccumParams ap = make_accum_params(*cam, c->w, rect, iteration, c->shard);
    ap.lut = c->tonemap_lut.as<uint32_t>(), ap.lut_dims = c->lut_dims;
    for (int base = 0; base < count_iterations; base += MAX_BATCH) { // the layers are folded in iteration order
        const int n = std::min(MAX_BATCH, count_iterations - base);
        AccumLayers per_layer = {};
        for (int k = 0; k < n; ++k) {
            const AccumParams al = make_accum_params(*cam, c->w, rect, iteration + base + k, c->shard);
            per_layer.l[k] = AccumLayer{al.iteration, al.mix_factor, al.half_mix_factor, al.is_class_a, al.variance_threshold};
        }
        k_accumulate<<<grid_for(c, npix, 256), 256, 0, s>>>(ap, c->px, layers, per_layer, base, n);
    }
    HIP_TRY(hipGetLastError());
    if (tm.mark(-1, -1)) {
        return 1;
    }
    if (stats) {
        // synchronous mode: resolve now and hand this call's stage times to the caller
        double before[11];
        memcpy(before, c->stage_us, sizeof(before));
        if (resolve_timing(c)) {
            return 1;
        }
        unsigned long long *slots = reinterpret_cast<unsigned long long *>(stats);
        for (int i = 0; i < 11; ++i) {
            slots[i] += (unsigned long long)(c->stage_us[i] - before[i]);
        }
    }
    return 0;
}

int rayhip_max_batch(rayhip_ctx *c) {
    if (!c || !c->h) {
        return 0;
    }
    return max_layers_for(c->w, c->h);
}

int rayhip_reserve_batch(rayhip_ctx *c, int count) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w) {
        return fail("rayhip_reserve_batch before rayhip_resize");
    }
    const int rect[4] = {0, 0, c->w, c->h};
    return ensure_pass(c, rect, std::max(1, std::min(count, rayhip_max_batch(c))));
}

int rayhip_render_batch(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int first_iteration, int count,
                        uint32_t flags, rayhip_stats *stats) {
    if (use_device(c)) {
        return 1;
    }
    if (count < 1) {
        return fail("batch of %d iterations", count);
    }
    if (!c->w) {
        return fail("rayhip_render needs resize + upload_static + scene_upload + set_filter_table first");
    }
    // A batch is exact only while adaptive sampling is inert (the reference re-queues every pixel every iteration when
    // variance_threshold == 0); the ray sort works on one dense ray array; pixel rows are 16-bit.
    // ... and only while no pixel can have required_samples < first_iteration: once an adaptive pass has run, pixels may be
    // parked; a one-by-one run would wake them up again after the first iteration with threshold 0, a batch would not
    // (k_raygen decides liveness once per pass).  adaptive_dirty is cleared by Clear / Resize.
    int max_layers = rayhip_max_batch(c);
    if (cam->pass_settings.variance_threshold != 0.0f || c->adaptive_dirty || (flags & RAYHIP_FLAG_SORT_RAYS) != 0) {
        max_layers = 1;
    }
    int done = 0;
    while (done < count) {
        const int n = std::min(count - done, max_layers);
        if (ensure_pass(c, rect, n)) {
            return 1;
        }
        if (render_pass(c, cam, rect, first_iteration + done, n, flags, stats)) {
            return 1;
        }
        done += n;
    }
    return 0;
}

int rayhip_render(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration, uint32_t flags,
                  rayhip_stats *stats) {
    if (use_device(c)) {
        return 1;
    }
    if (c->w && ensure_pass(c, rect, 1)) {
        return 1;
    }
    return render_pass(c, cam, rect, iteration, 1, flags, stats);
}

int rayhip_denoise_nlm(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w) {
        return fail("rayhip_denoise_nlm before rayhip_resize");
    }
    if (rect[0] < 0 || rect[1] < 0 || rect[2] <= 0 || rect[3] <= 0 || rect[0] + rect[2] > c->w || rect[1] + rect[3] > c->h) {
        return fail("rect outside the frame");
    }
    if (iteration < 1) {
        return fail("iteration is 1-based (the RegionContext::iteration of the last RenderScene)");
    }
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    DenoiseParams p;
    p.w = c->w, p.h = c->h;
    for (int i = 0; i < 4; ++i) {
        p.rect[i] = rect[i];
    }
    p.ext_w = rect[2] + 2 * NLM_EXT_RADIUS, p.ext_h = rect[3] + 2 * NLM_EXT_RADIUS;
    p.iteration = iteration;
    AccumParams tone = make_accum_params(*cam, c->w, rect, iteration, c->shard);
    tone.lut = c->tonemap_lut.as<uint32_t>(), tone.lut_dims = c->lut_dims;
    p.variance_threshold = tone.variance_threshold; // what the last RenderScene left in variance_threshold_ (RendererCPU.h:583-604)
    const size_t n_ext = size_t(p.ext_w) * size_t(p.ext_h);
    if (c->nlm_tm.alloc(n_ext * 16) || c->nlm_var_h.alloc(n_ext * 16) || c->nlm_var.alloc(n_ext * 16)) {
        return 1;
    }
    hipStream_t s = c->stream;
    k_nlm_prepare_h<<<grid_for(c, n_ext, 256), 256, 0, s>>>(p, c->px, c->nlm_tm.as<float4>(), c->nlm_var_h.as<float4>());
    k_nlm_prepare_v<<<grid_for(c, n_ext, 256), 256, 0, s>>>(p, c->nlm_var_h.as<float4>(), c->nlm_var.as<float4>());
    const size_t tiles = size_t((rect[2] + 15) / 16) * size_t((rect[3] + 15) / 16);
    k_nlm_filter<<<int(std::min<size_t>(tiles, size_t(c->props.multiProcessorCount) * 32u)), 256, 0, s>>>(
        p, tone, c->px, c->nlm_tm.as<float4>(), c->nlm_var.as<float4>());
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- UNet denoiser: InitUNetFilter / DenoiseImage(pass, region) (RendererCPU.h:1261-1310, 790-1007) ------------------------------
namespace {
// the sixteen passes: inputs (tensor ids, -1 = none; `up`: the first input is read through the nearest-neighbour upsample;
// `img`: the renderer's three images as nine more channels), output, resolution divider of the pass, pooling
struct UNetPassDesc {
    int a, a_ch, up, b, b_ch, img, cout, out, div, pool;
};
constexpr UNetPassDesc UNET_PASSES[16] = {
    {-1, 0, 0, -1, 0, 1, 32, 0, 1, 0},     // enc_conv0     images -> encConv0
    {0, 32, 0, -1, 0, 0, 32, 1, 1, 1},     // enc_conv1     -> pool1 (1/2)
    {1, 32, 0, -1, 0, 0, 48, 2, 2, 1},     // enc_conv2     -> pool2 (1/4)
    {2, 48, 0, -1, 0, 0, 64, 3, 4, 1},     // enc_conv3     -> pool3 (1/8)
    {3, 64, 0, -1, 0, 0, 80, 4, 8, 1},     // enc_conv4     -> pool4 (1/16)
    {4, 80, 0, -1, 0, 0, 96, 5, 16, 0},    // enc_conv5a
    {5, 96, 0, -1, 0, 0, 96, 6, 16, 0},    // enc_conv5b    -> upsample4 (still 1/16: the consumer upsamples)
    {6, 96, 1, 3, 64, 0, 112, 7, 8, 0},    // dec_conv4a    up(upsample4) ++ pool3
    {7, 112, 0, -1, 0, 0, 112, 8, 8, 0},   // dec_conv4b    -> upsample3
    {8, 112, 1, 2, 48, 0, 96, 9, 4, 0},    // dec_conv3a    up(upsample3) ++ pool2
    {9, 96, 0, -1, 0, 0, 96, 10, 4, 0},    // dec_conv3b    -> upsample2
    {10, 96, 1, 1, 32, 0, 64, 11, 2, 0},   // dec_conv2a    up(upsample2) ++ pool1
    {11, 64, 0, -1, 0, 0, 64, 12, 2, 0},   // dec_conv2b    -> upsample1
    {12, 64, 1, -1, 0, 1, 64, 13, 1, 0},   // dec_conv1a    up(upsample1) ++ images
    {13, 64, 0, -1, 0, 0, 32, 14, 1, 0},   // dec_conv1b
    {14, 32, 0, -1, 0, 0, 3, -1, 1, 0},    // dec_conv0     -> the filtered image
};
// resolution divider and channels of the fifteen tensors (UNetFilter.cpp:118-134)
constexpr int UNET_TENSOR_DIV[15] = {1, 2, 4, 8, 16, 16, 16, 8, 8, 4, 4, 2, 2, 1, 1};
constexpr int UNET_TENSOR_CH[15] = {32, 32, 48, 64, 80, 96, 96, 112, 112, 96, 96, 64, 64, 64, 32};
int round_up16(const int v) { return 16 * ((v + 15) / 16); }
} // namespace

// `weights` / `offsets`: what the reference's SetupUNetWeights<float>(alignment, &offsets, weights) produces (UNetFilter.cpp:296-570:
// per output channel three rows of round_up(3 * in_channels, alignment) floats, [ky][kx * in_channels + c]; a concatenating
// convolution keeps the rows of its two inputs one after the other) -- `offsets` is unet_weight_offsets_t as 32 ints.
int rayhip_unet_init(rayhip_ctx *c, const float *weights, int weights_count, const int32_t offsets[32], int alignment) {
    if (use_device(c)) {
        return 1;
    }
    if (!weights || !offsets || alignment < 1) {
        return fail("rayhip_unet_init: bad arguments");
    }
    using rt::unet::CHUNK;
    auto rt_of = [&](const int ch) { return alignment * ((3 * ch + alignment - 1) / alignment); };
    for (int pass = 0; pass < 16; ++pass) {
        const UNetPassDesc &d = UNET_PASSES[pass];
        const int c1 = d.a >= 0 ? d.a_ch : 0, c2 = d.b >= 0 ? d.b_ch : (d.img ? 9 : 0);
        const int n_tiles = (d.cout + 15) / 16, wp = rt::unet::weight_pitch(n_tiles);
        const int chunks = c1 / CHUNK + (d.b >= 0 ? d.b_ch / CHUNK : 0) + (d.img ? 1 : 0);
        // rows of one output channel in the reference blob: the first input's, then the second's (images count as one input)
        const int rt1 = c1 ? rt_of(c1) : 0, rt2 = c2 ? rt_of(c2) : 0;
        const int per_out = 3 * (rt1 + rt2);
        const int64_t w_off = offsets[2 * pass], b_off = offsets[2 * pass + 1];
        if (w_off < 0 || b_off < 0 || w_off + int64_t(per_out) * d.cout > weights_count || b_off + d.cout > weights_count) {
            return fail("rayhip_unet_init: pass %d reads outside the weight blob", pass);
        }
        std::vector<float> w(size_t(chunks) * 9 * CHUNK * wp, 0.0f), b(size_t(n_tiles) * 16, 0.0f);
        for (int n = 0; n < d.cout; ++n) {
            b[size_t(n)] = weights[b_off + n];
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                for (int cin = 0; cin < c1 + c2; ++cin) {
                    float v;
                    int chunk, cc;
                    if (cin < c1) {
                        v = weights[w_off + int64_t(n) * per_out + ky * rt1 + kx * c1 + cin];
                        chunk = cin / CHUNK, cc = cin % CHUNK;
                    } else {
                        const int k = cin - c1;
                        v = weights[w_off + int64_t(n) * per_out + 3 * rt1 + ky * rt2 + kx * c2 + k];
                        chunk = c1 / CHUNK + k / CHUNK, cc = k % CHUNK;
                    }
                    w[((size_t(chunk) * 9 + tap) * CHUNK + cc) * wp + n] = v;
                }
            }
        }
        if (upload(c, c->unet_pass[pass].weights, w.data(), w.size() * sizeof(float)) ||
            upload(c, c->unet_pass[pass].bias, b.data(), b.size() * sizeof(float))) {
            return 1;
        }
        c->unet_pass[pass].n_tiles = n_tiles;
        HIP_TRY(hipStreamSynchronize(c->stream)); // (w, b go out of scope)
    }
    c->unet_ready = true;
    return 0;
}

namespace {
// the activation tensors of the current frame size: one-pixel border, zero (only interiors are ever written)
int unet_tensors(rayhip_ctx *c) {
    if (c->unet_w == c->w && c->unet_h == c->h) {
        return 0;
    }
    const int wr = round_up16(c->w), hr = round_up16(c->h);
    c->unet_w = c->unet_h = 0; // (a failure part-way leaves tensors of two frame sizes: none of them counts as sized)
    for (int t = 0; t < 15; ++t) {
        const size_t n = size_t(wr / UNET_TENSOR_DIV[t] + 2) * size_t(hr / UNET_TENSOR_DIV[t] + 2) * size_t(UNET_TENSOR_CH[t]);
        c->unet_tensor[t].release(); // a fresh, zeroed allocation: the borders must be zero
        if (c->unet_tensor[t].alloc(n * sizeof(float))) {
            return 1;
        }
    }
    c->unet_w = c->w, c->unet_h = c->h;
    return 0;
}
float *unet_interior(rayhip_ctx *c, const int t) {
    const int wr = round_up16(c->w);
    return c->unet_tensor[t].as<float>() + size_t(wr / UNET_TENSOR_DIV[t] + 3) * size_t(UNET_TENSOR_CH[t]);
}
} // namespace

int rayhip_denoise_unet(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int pass) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->unet_ready) {
        return fail("rayhip_denoise_unet before rayhip_unet_init (RendererBase::InitUNetFilter)");
    }
    if (!c->w) {
        return fail("rayhip_denoise_unet before rayhip_resize");
    }
    if (pass < -1 || pass > 15) {
        return fail("the UNet filter has passes 0 .. 15 (-1: all of them)");
    }
    if (rect[0] < 0 || rect[1] < 0 || rect[2] <= 0 || rect[3] <= 0 || rect[0] + rect[2] > c->w || rect[1] + rect[3] > c->h) {
        return fail("rect outside the frame");
    }
    if ((rect[0] % 16) != 0 || (rect[1] % 16) != 0) {
        return fail("the UNet filter works on regions whose corner is a multiple of 16 pixels (the network pools four times)");
    }
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    if (unet_tensors(c)) {
        return 1;
    }
    const int wr = round_up16(c->w), hr = round_up16(c->h);
    for (int p = (pass < 0 ? 0 : pass); p <= (pass < 0 ? 15 : pass); ++p) {
        const UNetPassDesc &d = UNET_PASSES[p];
        // the region of this pass in its own resolution (RendererCPU.h:797-802 and the head of every case)
        int rx = rect[0], ry = rect[1], rw = rect[2], rh = rect[3];
        if (p < 15) {
            rw = round_up16(rw), rh = round_up16(rh);
        }
        rx /= d.div, ry /= d.div, rw = (rw + d.div - 1) / d.div, rh = (rh + d.div - 1) / d.div;
        rt::unet::ConvParams cp = {};
        if (d.a >= 0) {
            cp.a = unet_interior(c, d.a), cp.a_stride = wr / UNET_TENSOR_DIV[d.a] + 2, cp.a_ch = d.a_ch, cp.a_up = d.up;
        }
        if (d.b >= 0) {
            cp.b = unet_interior(c, d.b), cp.b_stride = wr / UNET_TENSOR_DIV[d.b] + 2, cp.b_ch = d.b_ch;
        }
        if (d.img) {
            cp.img_full = c->px.full, cp.img_base = c->px.base_color, cp.img_dn = c->px.depth_normals;
            cp.img_w = c->w, cp.img_h = c->h;
        }
        cp.weights = c->unet_pass[p].weights.as<float>(), cp.bias = c->unet_pass[p].bias.as<float>();
        cp.x0 = rx, cp.y0 = ry, cp.w = rw, cp.h = rh;
        cp.in_w = wr / d.div, cp.in_h = hr / d.div;
        cp.pool = d.pool;
        if (d.out >= 0) {
            cp.out = unet_interior(c, d.out), cp.out_stride = wr / UNET_TENSOR_DIV[d.out] + 2, cp.out_ch = d.cout;
        } else {
            cp.out = reinterpret_cast<float *>(c->px.raw), cp.out_stride = c->w, cp.out_ch = 3, cp.final_image = 1;
        }
        HIP_TRY(rt::unet::launch_conv(cp, c->unet_pass[p].n_tiles, c->stream));
        if (p == 15) {
            AccumParams tone = make_accum_params(*cam, c->w, rect, 1, c->shard);
            tone.lut = c->tonemap_lut.as<uint32_t>(), tone.lut_dims = c->lut_dims;
            k_tonemap_raw_rect<<<grid_for(c, size_t(rect[2]) * size_t(rect[3]), 256), 256, 0, c->stream>>>(tone, c->px);
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

// test hook: one activation tensor (0 .. 14, the order of unet_filter_tensors_t) with its border, NHWC; dims = {rows, columns, channels}
int rayhip_unet_read_tensor(rayhip_ctx *c, int which, float *dst, size_t capacity_floats, int out_dims[3]) {
    if (use_device(c)) {
        return 1;
    }
    if (which < 0 || which > 14 || c->unet_w != c->w || !c->unet_tensor[which].p) {
        return fail("rayhip_unet_read_tensor: no such tensor (run rayhip_denoise_unet first)");
    }
    const int wr = round_up16(c->w), hr = round_up16(c->h);
    out_dims[0] = hr / UNET_TENSOR_DIV[which] + 2, out_dims[1] = wr / UNET_TENSOR_DIV[which] + 2, out_dims[2] = UNET_TENSOR_CH[which];
    const size_t n = size_t(out_dims[0]) * out_dims[1] * out_dims[2];
    if (n > capacity_floats) {
        return fail("rayhip_unet_read_tensor: %zu floats needed", n);
    }
    HIP_TRY(hipMemcpyAsync(dst, c->unet_tensor[which].p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_set_shard(rayhip_ctx *c, int tile, int shard_count, int shard_index) {
    if (tile <= 0 || shard_count <= 0 || shard_index < 0 || shard_index >= shard_count) {
        return fail("bad shard (tile %d, %d of %d)", tile, shard_index, shard_count);
    }
    c->shard = Shard{tile, shard_count, shard_index};
    return 0;
}

static float4 *pick_buffer(rayhip_ctx *c, int which) {
    switch (which) {
    case RAYHIP_BUF_FINAL:
        return c->px.final_;
    case RAYHIP_BUF_RAW:
        return c->px.raw;
    case RAYHIP_BUF_BASE_COLOR:
        return c->px.base_color;
    case RAYHIP_BUF_DEPTH_NORMALS:
        return c->px.depth_normals;
    case RAYHIP_BUF_VARIANCE:
        return c->px.variance;
    default:
        return nullptr;
    }
}

int rayhip_readback(rayhip_ctx *c, int which, float *dst_rgba, int pitch_px) {
    if (use_device(c)) {
        return 1;
    }
    float4 *src = pick_buffer(c, which);
    if (!src) {
        return fail("bad buffer id %d", which);
    }
    if (pitch_px == c->w) { // one linear copy (a 2-D copy of 1080 rows into pageable memory is staged row by row)
        HIP_TRY(hipMemcpyAsync(dst_rgba, src, size_t(c->w) * size_t(c->h) * 16, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(dst_rgba, size_t(pitch_px) * 16, src, size_t(c->w) * 16, size_t(c->w) * 16, size_t(c->h),
                                 hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_readback_device(rayhip_ctx *c, int which, void *dst_device_rgba, int pitch_px) {
    if (use_device(c)) {
        return 1;
    }
    float4 *src = pick_buffer(c, which);
    if (!src) {
        return fail("bad buffer id %d", which);
    }
    HIP_TRY(hipMemcpy2DAsync(dst_device_rgba, size_t(pitch_px) * 16, src, size_t(c->w) * 16, size_t(c->w) * 16, size_t(c->h),
                             hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_set_raw_device(rayhip_ctx *c, const void *src_device_rgba, int pitch_px, const rayhip_camera *cam) {
    if (use_device(c)) {
        return 1;
    }
    HIP_TRY(hipMemcpy2DAsync(c->px.full, size_t(c->w) * 16, src_device_rgba, size_t(pitch_px) * 16, size_t(c->w) * 16,
                             size_t(c->h), hipMemcpyDeviceToDevice, c->stream));
    const int rect[4] = {0, 0, c->w, c->h};
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    AccumParams ap = make_accum_params(*cam, c->w, rect, 1);
    ap.lut = c->tonemap_lut.as<uint32_t>(), ap.lut_dims = c->lut_dims;
    k_retonemap<<<grid_for(c, size_t(c->w) * c->h, 256), 256, 0, c->stream>>>(ap, c->px, c->h);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_sync(rayhip_ctx *c) {
    if (use_device(c)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_get_trav_counters(rayhip_ctx *c, rayhip_trav_counters out[2], int reset) {
    if (use_device(c)) {
        return 1;
    }
    unsigned long long h[2 * TRAV_COUNTER_WORDS];
    HIP_TRY(hipMemcpyAsync(h, c->trav_counters.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k) {
        const unsigned long long *w = h + TRAV_COUNTER_WORDS * k;
        out[k].rays = w[0], out[k].nodes = w[1], out[k].tris = w[2], out[k].instances = w[3], out[k].max_stack = w[4], out[k].nodes4 = w[5];
    }
    if (reset) {
        HIP_TRY(hipMemsetAsync(c->trav_counters.p, 0, sizeof(h), c->stream));
    }
    return 0;
}

int rayhip_get_stage_times(rayhip_ctx *c, rayhip_stats *out, int reset) {
    if (use_device(c) || resolve_timing(c)) {
        return 1;
    }
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(out);
    for (int i = 0; i < 11; ++i) {
        slots[i] = (unsigned long long)(c->stage_us[i]);
        if (reset) {
            c->stage_us[i] = 0.0;
        }
    }
    return 0;
}

int rayhip_get_trav_timing(rayhip_ctx *c, double out_ms[2], unsigned long long out_launches[2], int reset) {
    if (use_device(c) || resolve_timing(c)) {
        return 1;
    }
    for (int k = 0; k < 2; ++k) {
        out_ms[k] = c->trav_ms[k];
        out_launches[k] = c->trav_launches[k];
        if (reset) {
            c->trav_ms[k] = 0.0, c->trav_launches[k] = 0;
        }
    }
    return 0;
}

// ---- kernel-level hooks ---------------------------------------------------------------------------------------

int rayhip_k_generate_primary_rays(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration,
                                   rayhip_ray *out_rays, rayhip_hit *out_hits, int *out_count) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w || !c->pmj.p || !c->filter_table.p) {
        return fail("k_generate_primary_rays needs resize + upload_static + set_filter_table first");
    }
    hipStream_t s = c->stream;
    const RayGenTiling tiling = make_raygen_tiling(c->w, c->h, rect[2], rect[3], c->shard);
    const size_t nslots = size_t(tiling.tiles) * 64u;
    if (c->clear_queues(1, s)) {
        return fail("queue counter clear failed");
    }
    const RayGenParams rg = make_raygen_params(*cam, c->w, c->h, rect, iteration, c->shard);
    // kernel-level hooks use one dense stripe so that the host sees a plain array
    k_raygen<<<grid_for(c, nslots, 256), 256, 0, s>>>(rg, c->sc.pmj, c->filter_table.as<float>(), c->px.required_samples,
                                                      c->rays[0], c->hits, c->ray_queue(0, nslots, 1), single_layer(c->w, c->h), tiling);
    HIP_TRY(hipGetLastError());
    uint32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, c->ray_count(0), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<float4> pl[4], hp(n);
    std::vector<uint2> xd(n);
    std::vector<float> hv(n);
    for (int k = 0; k < 4; ++k) {
        pl[k].resize(n);
        HIP_TRY(hipMemcpyAsync(pl[k].data(), c->ray_planes[0][k].p, size_t(n) * 16, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(xd.data(), c->ray_planes[0][4].p, size_t(n) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hp.data(), c->hit_planes[0].p, size_t(n) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hv.data(), c->hit_planes[1].p, size_t(n) * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < n; ++i) {
        rayhip_ray &r = out_rays[i];
        r.o[0] = pl[0][i].x, r.o[1] = pl[0][i].y, r.o[2] = pl[0][i].z, r.pdf = pl[0][i].w;
        r.d[0] = pl[1][i].x, r.d[1] = pl[1][i].y, r.d[2] = pl[1][i].z, r.cone_width = pl[1][i].w;
        r.c[0] = pl[2][i].x, r.c[1] = pl[2][i].y, r.c[2] = pl[2][i].z, r.cone_spread = pl[2][i].w;
        r.ior[0] = pl[3][i].x, r.ior[1] = pl[3][i].y, r.ior[2] = pl[3][i].z, r.ior[3] = pl[3][i].w;
        r.xy = xd[i].x, r.depth = xd[i].y;
        rayhip_hit &h = out_hits[i];
        memcpy(&h.obj_index, &hp[i].x, 4), memcpy(&h.prim_index, &hp[i].y, 4);
        h.t = hp[i].z, h.u = hp[i].w, h.v = hv[i];
    }
    *out_count = int(n);
    return 0;
}

int rayhip_k_intersect_closest(rayhip_ctx *c, const rayhip_camera *cam, rayhip_ray *rays, rayhip_hit *hits, int count,
                               int iteration, uint32_t flags, rayhip_trav_counters *out_counters) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->have_scene || !c->pmj.p) {
        return fail("k_intersect_closest needs a scene and the PMJ table");
    }
    if (size_t(count) > size_t(c->w) * size_t(c->h)) {
        return fail("ray count exceeds the wavefront buffers (w*h)");
    }
    hipStream_t s = c->stream;
    std::vector<float4> pl[4], hp;
    hp.resize(size_t(count));
    std::vector<uint2> xd;
    std::vector<float> hv{};
    hv.resize(size_t(count));
    rays_to_soa(rays, count, pl, xd);
    for (int i = 0; i < count; ++i) {
        float oi, pi;
        memcpy(&oi, &hits[i].obj_index, 4), memcpy(&pi, &hits[i].prim_index, 4);
        hp[i] = make_float4(oi, pi, hits[i].t, hits[i].u);
        hv[i] = hits[i].v;
    }
    for (int k = 0; k < 4; ++k) {
        HIP_TRY(hipMemcpyAsync(c->ray_planes[0][k].p, pl[k].data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(c->ray_planes[0][4].p, xd.data(), size_t(count) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[0].p, hp.data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[1].p, hv.data(), size_t(count) * 4, hipMemcpyHostToDevice, s));
    const uint32_t n = uint32_t(count);
    HIP_TRY(hipMemcpyAsync(c->ray_count(0), &n, 4, hipMemcpyHostToDevice, s));
    unsigned long long *tc = c->trav_counters.as<unsigned long long>();
    unsigned long long before[TRAV_COUNTER_WORDS], after[TRAV_COUNTER_WORDS];
    HIP_TRY(hipMemcpyAsync(before, tc, sizeof(before), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(tc, 0, sizeof(before), s));
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    const int g = int(std::min<size_t>(size_t(c->grid_waves), (size_t(count) + WAVE - 1) / WAVE));
    const RayQueue q = c->ray_queue(0, size_t(count), 1);
    {
        const int gg = g ? g : 1;
#define KK_ARGS c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), tc, single_layer(c->w, c->h)
        if ((flags & RAYHIP_FLAG_COUNT_WIDE) && c->wide == 8) { // the product walk with counters
            k_trace_closest<true, 8><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if ((flags & RAYHIP_FLAG_COUNT_WIDE) && c->wide == 4) {
            k_trace_closest<true, 4><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if (flags & RAYHIP_FLAG_COUNT_TRAVERSAL) { // instrumented walk of the reference's BVH2
            k_trace_closest<true, 0><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if (c->wide == 4 && c->refill_pool && c->pool_scene) { // what rayhip_render launches (the pooled form also takes preset hits)
            k_trace_closest_pool<><<<std::max(1, std::min(g, c->pool_waves)), WAVE, 0, s>>>(c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), single_layer(c->w, c->h));
        } else if (c->wide == 8 && c->refill_waves) {
            k_trace_closest_refill<8><<<std::max(1, std::min(g, c->refill_waves)), WAVE, 0, s>>>(c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), single_layer(c->w, c->h));
        } else if (c->wide == 4 && c->refill_waves) {
            k_trace_closest_refill<4><<<std::max(1, std::min(g, c->refill_waves)), WAVE, 0, s>>>(c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), single_layer(c->w, c->h));
        } else if (c->wide == 8) {
            k_trace_closest<false, 8><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if (c->wide == 4) {
            k_trace_closest<false, 4><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else {
            k_trace_closest<false, 0><<<gg, WAVE, 0, s>>>(KK_ARGS);
        }
#undef KK_ARGS
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpyAsync(after, tc, sizeof(after), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(tc, before, sizeof(before), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (out_counters) {
        out_counters->rays = after[0], out_counters->nodes = after[1];
        out_counters->tris = after[2], out_counters->instances = after[3];
        out_counters->max_stack = after[4], out_counters->nodes4 = after[5];
    }
    HIP_TRY(hipMemcpyAsync(pl[2].data(), c->ray_planes[0][2].p, size_t(count) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(xd.data(), c->ray_planes[0][4].p, size_t(count) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hp.data(), c->hit_planes[0].p, size_t(count) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hv.data(), c->hit_planes[1].p, size_t(count) * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < count; ++i) {
        rays[i].c[0] = pl[2][i].x, rays[i].c[1] = pl[2][i].y, rays[i].c[2] = pl[2][i].z;
        rays[i].depth = xd[i].y;
        memcpy(&hits[i].obj_index, &hp[i].x, 4), memcpy(&hits[i].prim_index, &hp[i].y, 4);
        hits[i].t = hp[i].z, hits[i].u = hp[i].w, hits[i].v = hv[i];
    }
    return 0;
}

int rayhip_k_intersect_shadow(rayhip_ctx *c, const rayhip_camera *cam, const rayhip_shadow_ray *rays, int count,
                              int iteration, float *out_rc, rayhip_trav_counters *out_counters) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->have_scene || !c->pmj.p) {
        return fail("k_intersect_shadow needs a scene and the PMJ table");
    }
    if (size_t(count) > size_t(c->w) * size_t(c->h)) {
        return fail("ray count exceeds the wavefront buffers (w*h)");
    }
    hipStream_t s = c->stream;
    std::vector<float4> pl[3];
    for (int k = 0; k < 3; ++k) {
        pl[k].resize(size_t(count));
    }
    for (int i = 0; i < count; ++i) {
        float depth_f, xy_f;
        memcpy(&depth_f, &rays[i].depth, 4), memcpy(&xy_f, &rays[i].xy, 4);
        pl[0][i] = make_float4(rays[i].o[0], rays[i].o[1], rays[i].o[2], depth_f);
        pl[1][i] = make_float4(rays[i].d[0], rays[i].d[1], rays[i].d[2], rays[i].dist);
        pl[2][i] = make_float4(rays[i].c[0], rays[i].c[1], rays[i].c[2], xy_f);
    }
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(hipMemcpyAsync(c->shadow_planes[k].p, pl[k].data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    }
    const uint32_t n = uint32_t(count);
    HIP_TRY(hipMemcpyAsync(c->shadow_count(0), &n, 4, hipMemcpyHostToDevice, s));
    unsigned long long *tc = c->trav_counters.as<unsigned long long>() + TRAV_COUNTER_WORDS;
    unsigned long long before[TRAV_COUNTER_WORDS], after[TRAV_COUNTER_WORDS];
    HIP_TRY(hipMemcpyAsync(before, tc, sizeof(before), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(tc, 0, sizeof(before), s));
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    const int g = int(std::min<size_t>(size_t(c->grid_waves), (size_t(count) + WAVE - 1) / WAVE));
    // results land in the (otherwise idle) hit plane
    k_trace_shadow<true, 0><<<g ? g : 1, WAVE, 0, s>>>(c->sc, tp, c->shadow, c->shadow_queue(0, size_t(count), 1), FLT_MAX, c->w, c->px.temp,
                                                    c->hit_planes[0].as<float4>(), c->stack_spill.as<uint32_t>(), tc, single_layer(c->w, c->h));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpyAsync(after, tc, sizeof(after), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(tc, before, sizeof(before), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (out_counters) {
        out_counters->rays = after[0], out_counters->nodes = after[1];
        out_counters->tris = after[2], out_counters->instances = after[3];
        out_counters->max_stack = after[4], out_counters->nodes4 = after[5];
    }
    HIP_TRY(hipMemcpyAsync(out_rc, c->hit_planes[0].p, size_t(count) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int rayhip_k_shade(rayhip_ctx *c, const rayhip_camera *cam, int bounce, int iteration, const rayhip_ray *rays, const rayhip_hit *hits,
                   int count, float *inout_color, rayhip_ray *out_secondary, int *out_secondary_count, rayhip_shadow_ray *out_shadow,
                   int *out_shadow_count) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->have_scene || !c->pmj.p || !c->w) {
        return fail("k_shade needs resize + upload_static + scene_upload first");
    }
    if (count < 0 || size_t(count) > size_t(c->w) * size_t(c->h)) {
        return fail("ray count exceeds the wavefront buffers (w*h)");
    }
    if (bounce < 0 || bounce + 2 > MAX_BOUNCE_SLOTS || iteration < 1) {
        return fail("bad bounce / iteration");
    }
    hipStream_t s = c->stream;
    const size_t npix = size_t(c->w) * size_t(c->h);
    std::vector<float4> pl[4], hp(static_cast<size_t>(count));
    std::vector<uint2> xd;
    std::vector<float> hv(static_cast<size_t>(count));
    rays_to_soa(rays, count, pl, xd);
    for (int i = 0; i < count; ++i) {
        float oi, pi;
        memcpy(&oi, &hits[i].obj_index, 4), memcpy(&pi, &hits[i].prim_index, 4);
        hp[i] = make_float4(oi, pi, hits[i].t, hits[i].u);
        hv[i] = hits[i].v;
    }
    for (int k = 0; k < 4; ++k) {
        HIP_TRY(hipMemcpyAsync(c->ray_planes[0][k].p, pl[k].data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(c->ray_planes[0][4].p, xd.data(), size_t(count) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[0].p, hp.data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[1].p, hv.data(), size_t(count) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->px.temp, inout_color, npix * 16, hipMemcpyHostToDevice, s));
    if (c->clear_queues(bounce + 2, s)) {
        return fail("queue counter clear failed");
    }
    const uint32_t n = uint32_t(count);
    HIP_TRY(hipMemcpyAsync(c->ray_count(bounce), &n, 4, hipMemcpyHostToDevice, s));
    const int g = std::max(1, int(std::min<size_t>(size_t(c->grid_waves), (size_t(count) + WAVE - 1) / WAVE)));
    // one dense stripe, so the host sees plain arrays; the kernels are the ones rayhip_render launches
    launch_shade(c, *cam, iteration, bounce, 0, size_t(count), 1, g, c->w, 1.0f / float(iteration), single_layer(c->w, c->h));
    HIP_TRY(hipGetLastError());
    uint32_t n_sec = 0, n_sh = 0;
    HIP_TRY(hipMemcpyAsync(&n_sec, c->ray_count(bounce + 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&n_sh, c->shadow_count(bounce), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(inout_color, c->px.temp, npix * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 4; ++k) {
        pl[k].resize(n_sec);
        HIP_TRY(hipMemcpyAsync(pl[k].data(), c->ray_planes[1][k].p, size_t(n_sec) * 16, hipMemcpyDeviceToHost, s));
    }
    xd.resize(n_sec);
    HIP_TRY(hipMemcpyAsync(xd.data(), c->ray_planes[1][4].p, size_t(n_sec) * 8, hipMemcpyDeviceToHost, s));
    std::vector<float4> sp_[3];
    for (int k = 0; k < 3; ++k) {
        sp_[k].resize(n_sh);
        HIP_TRY(hipMemcpyAsync(sp_[k].data(), c->shadow_planes[k].p, size_t(n_sh) * 16, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < n_sec; ++i) {
        rayhip_ray &r = out_secondary[i];
        r.o[0] = pl[0][i].x, r.o[1] = pl[0][i].y, r.o[2] = pl[0][i].z, r.pdf = pl[0][i].w;
        r.d[0] = pl[1][i].x, r.d[1] = pl[1][i].y, r.d[2] = pl[1][i].z, r.cone_width = pl[1][i].w;
        r.c[0] = pl[2][i].x, r.c[1] = pl[2][i].y, r.c[2] = pl[2][i].z, r.cone_spread = pl[2][i].w;
        r.ior[0] = pl[3][i].x, r.ior[1] = pl[3][i].y, r.ior[2] = pl[3][i].z, r.ior[3] = pl[3][i].w;
        r.xy = xd[i].x, r.depth = xd[i].y;
    }
    for (uint32_t i = 0; i < n_sh; ++i) {
        rayhip_shadow_ray &r = out_shadow[i];
        r.o[0] = sp_[0][i].x, r.o[1] = sp_[0][i].y, r.o[2] = sp_[0][i].z, memcpy(&r.depth, &sp_[0][i].w, 4);
        r.d[0] = sp_[1][i].x, r.d[1] = sp_[1][i].y, r.d[2] = sp_[1][i].z, r.dist = sp_[1][i].w;
        r.c[0] = sp_[2][i].x, r.c[1] = sp_[2][i].y, r.c[2] = sp_[2][i].z, memcpy(&r.xy, &sp_[2][i].w, 4);
    }
    *out_secondary_count = int(n_sec), *out_shadow_count = int(n_sh);
    return 0;
}

int rayhip_k_scrambled_rand(rayhip_ctx *c, const uint32_t *dims, const uint32_t *seeds, const int32_t *samples, int count,
                            float *out_xy) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->pmj.p) {
        return fail("k_scrambled_rand needs the PMJ table");
    }
    hipStream_t s = c->stream;
    DevBuf d, sd, sm, o;
    if (d.alloc(size_t(count) * 4) || sd.alloc(size_t(count) * 4) || sm.alloc(size_t(count) * 4) || o.alloc(size_t(count) * 8)) {
        return 1;
    }
    HIP_TRY(hipMemcpyAsync(d.p, dims, size_t(count) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(sd.p, seeds, size_t(count) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(sm.p, samples, size_t(count) * 4, hipMemcpyHostToDevice, s));
    k_scrambled_rand<<<(count + 255) / 256, 256, 0, c->stream>>>(d.as<uint32_t>(), sd.as<uint32_t>(), sm.as<int32_t>(), count,
                                                                 c->sc.pmj, o.as<float2>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyAsync(out_xy, o.p, size_t(count) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    d.release(), sd.release(), sm.release(), o.release();
    return 0;
}

#if defined(RT_PROFILE_SHADE) || defined(RT_PROFILE_TRACE)
// tuning build only (tools/variants.py): cycles per shade-kernel section, see RT_PROF in kernels.hip.h
__attribute__((visibility("default"))) int rayhip_tuning_read_profile(rayhip_ctx *c, unsigned long long out[32], int reset) {
    if (use_device(c)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_prof_acc), 32 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[32] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rt::g_prof_acc), z, sizeof(z)));
    }
    return 0;
}
#endif

} // extern "C"

#include "comm.hip.h"
