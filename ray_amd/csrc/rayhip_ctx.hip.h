// rayhip_ctx.hip.h -- part of librayhip's host side (one translation unit: included by rayhip.hip, in this order, after the kernels):
// the context: device buffers, the rayhip_ctx struct, error / timing helpers, create / destroy / resize / clear.
#pragma once

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
    bool plain_ior = false; // no material of the scene refracts: ShadeParams::plain_ior for the passes that start at the camera
    hipDeviceProp_t props = {};
    int grid_waves = 0; // resident-ish grid for the wave-per-block kernels
    bool small_scene = false; // BLAS nodes + triangles fit one XCD's L2: traversal kernels with the smaller register footprint
    int refill_waves = 0; // largest grid of the persistent closest-hit kernel (blocks); 0 = kernel switched off
    int refill_resident = 0; // ... and the number of its blocks the device holds at once
    int refill_resident4 = 0; // ... of its 4-wide form (the grid of a launch whose chunks are handed out dynamically)
    int sort_key_mode = 0; // RAYHIP_SORT_KEY (tuning, rt_sort.h)
    // RAYHIP_PRIMARY_WAVES / RAYHIP_SHADOW_WAVES: register footprint of the plain K2 (primary rays) / of K3.  K3 runs at 5 waves per
    // SIMD (96 VGPRs, 32 spilled registers per ray instead of 51 at 6 waves: same time, a third less scratch traffic)
    int tune_primary_waves = 0, tune_shadow_waves = 5;
    // layered passes: a primary wavefront holds S samples of each of 64 / S pixels of an 8x8 tile (RayGenTiling).  Measured on the headline
    // scene, 64-layer passes (profiles/r04/experiments/raygen_samples_per_wave.txt): primary K2 268 / 250 / 237 / 223 us per sample for
    // S = 1 / 4 / 16 / 64 -- the S rays of a pixel walk the same nodes -- and the primary shade 287 / 294 / 397 / 545: its three 16-byte
    // pixel writes per ray stay in runs of four pixels at S = 4 and scatter over the layers beyond.  4 is the default; RAYHIP_RAYGEN_SAMPLES
    int raygen_samples_per_wave = 4;
    bool shadow_refill = true; // K3 as the flat persistent kernel (4-wide walk); RAYHIP_SHADOW_REFILL=0: the nested form
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
        DevBuf weights_h; // the f16 form (unet.h: ConvParamsH): [chunk of 32][tap][out channel][32], rows swizzled
        int n_tiles = 0;
    };
    DevBuf unet_tensor_h[15], unet_images_h; // activation tensors of the f16 form
    int unet_half = 0;                       // rayhip_unet_set_precision: 0 = the exact f32 form, 1 = f16 tensors / weights, f32 accumulate
    int unet_h_w = 0, unet_h_h = 0;          // frame size the f16 tensors were sized for
    UNetPass unet_pass[16];
    DevBuf unet_tensor[15];
    DevBuf sky_desc, sky_transmittance_lut, sky_multiscatter_lut, sky_dir_lights, sky_weather, sky_noise3d, sky_curl, sky_moon, sky_cirrus; // the physical sky (rt_sky.h)
    SkyView sky_view = {};
    DevBuf sky_index;  // ray slots of the paths that ended in the physical sky, per stripe (k_surface -> k_shade_sky)
    DevBuf unet_images; // the renderer's three images as one 16-channel tensor (unet_kernels.hip: k_image_inputs)
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
    // round 6 (shade_split bit 4): the picks of the light pick that runs before the surface stage, by ray slot and stamped with the launch's tag;
    // the ray's part of a next-event record (shade_launch.h)
    DevBuf pick_plane, record_ray_planes[4];
    RaySoA record_rays = {};
    bool notex_kernels = true; // RAYHIP_NOTEX_KERNELS=0: scenes without textures take the general k_surface_scatter too
    bool pick_lds = true; // RAYHIP_PICK_LDS=0: the light pick reads every row of the light table from memory (shade_kernels.hip)
    uint32_t shade_tag = 0;
    uint32_t next_shade_tag() {
        if (++shade_tag == 0u) { // (2^32 launches later: the plane may hold every old tag -- clear it and start again)
            (void)hipMemsetAsync(pick_plane.p, 0, pick_plane.bytes, stream);
            shade_tag = 1u;
        }
        return shade_tag;
    }
    // how the shade stage is cut into launches (kernels.hip.h): bit 0 = the light pick as its own kernel, bit 1 = next-event
    // estimation and continuation as two scatter launches.  RAYHIP_SHADE_SPLIT overrides (A/B measurements).
    int shade_split = 29; // shade_launch.h: bit 4 (round 6, the default) pick first -> surface + continuation fused -> next-event estimation over dense records; without it: bit 0 pick as its own kernel, bit 1 NEE / continuation as two launches, bit 2 NEE over the compacted queue of points that got a light, bit 3 the light pick with lane refill
    RaySoA rays[2] = {};
    HitSoA hits = {};
    ShadowSoA shadow = {};
    DeferredSoA deferred = {};
    DevBuf counters;      // uint32 [MAX_BOUNCE_SLOTS][QUEUES_PER_BOUNCE = 6: rays, shadow rays, deferred emitters, shade points, points with a light, sky paths][QUEUE_MAX_STRIPES * QUEUE_COUNTER_STRIDE]
    DevBuf trav_counters; // u64 [2][TRAV_COUNTER_WORDS]
    // Dynamic chunk hand-out (wavefront.hip.h: ChunkWalk, round 5) of the persistent kernels with lane refill (K2's secondary bounces, K3, the
    // light pick): one counter (a 256-byte line) per such launch of a pass, WORK_PER_BOUNCE per bounce, cleared with the pass's queue counters
    // and dealt out in launch order.  OPT-IN (RAYHIP_DYNAMIC=1): measured neutral -- K2's secondary bounces 87.3 against 87.4 ms per frame, a rank of
    // 8 12.5 against 12.6 ms (profiles/r05/experiments/dynamic_chunks.txt): what a launch loses at its end is the latency of its longest ray,
    // not the balance between blocks.  Bit-identical, kept for the next kernel that is short of balance.
    static constexpr uint32_t WORK_PER_BOUNCE = 4;
    DevBuf work_counters;
    uint32_t work_cursor = 0, work_cleared = 0;
    bool dynamic_chunks = false;
    int dyn_mult = 1;        // blocks per resident wave slot of a dynamic launch (RAYHIP_DYN_MULT)
    int shadow_resident = 0; // blocks of k_trace_shadow_refill the device holds at once
    uint32_t *next_work() {
        if (!dynamic_chunks || work_cursor >= work_cleared) {
            return nullptr; // (a launch without a counter walks statically)
        }
        return work_counters.as<uint32_t>() + size_t(work_cursor++) * WORK_COUNTER_STRIDE;
    }
    int clear_work(int bounces, hipStream_t s) {
        work_cursor = 0, work_cleared = 0;
        if (!dynamic_chunks) {
            return 0;
        }
        const uint32_t n = uint32_t(bounces) * WORK_PER_BOUNCE;
        if (hipMemsetAsync(work_counters.p, 0, size_t(n) * WORK_COUNTER_STRIDE * sizeof(uint32_t), s) != hipSuccess) {
            return 1;
        }
        work_cleared = n;
        return 0;
    }
    // Queue census (round 5): how full every queue of the last pass was, as a fraction of the pass's ray slots -- what the next pass sizes its
    // launches from (rays of bounce b, shadow rays, shade points, lit points ... of the same scene and camera vary by a few per cent from pass
    // to pass; a persistent kernel is correct at ANY grid size, so a stale or missing census only costs time).  The reference sizes every
    // dispatch from device counters as well (internal/shaders/prepare_indir_args.comp.glsl:17-77, recorded up front at
    // internal/RendererVK.cpp:641-712); HIP has no indirect dispatch, hence one pass of delay.  k_queue_totals writes the sums straight into
    // page-locked host memory at the end of a pass; the host looks at them when the next pass starts, if the event says they are there.
    // RAYHIP_CENSUS=0: every launch at the pass's full grid, as before.
    uint32_t *census_host = nullptr, *census_dev = nullptr; // [MAX_BOUNCE_SLOTS * QUEUES_PER_BOUNCE], one allocation seen from both sides
    hipEvent_t census_event = nullptr;
    bool census_on = true, census_pending = false, census_valid = false;
    int census_bounces = 0;    // bounces the pending / valid census covers
    size_t census_slots = 0;   // ray slots of the pass the pending census belongs to
    std::vector<float> census; // [bounce][queue]: fill / slots of the last census read
    int chunks_per_block = 8;  // live chunks a block of a streaming kernel should find (RAYHIP_CHUNKS_PER_BLOCK)
    // live chunks queue `q` (0 rays, 1 shadow rays, 2 deferred emitters, 3 shade points, 4 lit points, 5 sky paths) of bounce `b` is expected to
    // hold in a pass of `slots` ray slots; 0 = unknown
    uint32_t expect_chunks(int b, int q, size_t slots, uint32_t stripes) const {
        if (!census_valid || b >= census_bounces) {
            return 0u;
        }
        const double rays = double(census[size_t(b) * QUEUES_PER_BOUNCE + size_t(q)]) * double(slots) * 1.25;
        return uint32_t(std::min<double>(double(slots / WAVE + stripes), rays / WAVE + double(stripes))) + 1u;
    }
    // Which form of the shade stage a pass takes (round 6).  The form without the point queue (shade_split bit 4) lists a next-event record per LIT
    // point -- 148 bytes written and read again -- where the three-kernel form reads the point it stored anyway: with a fifth of the points lit
    // (the atria, the street) it wins 6-10 % of the frame, with every point lit (the Cornell boxes) it loses 8-9 %
    // (profiles/r06/experiments/shade_forms_by_workload.txt).  Both are bit-identical, so the choice is made per pass from the census of the pass
    // before: lit points / rays over all bounces; unknown (first pass of a scene): the new form.  RAYHIP_SHADE_SPLIT pins a form.
    bool shade_form_auto = true;
    int shade_split_for_pass() const {
        if (!shade_form_auto || (shade_split & 16) == 0 || !census_valid) {
            return shade_split;
        }
        double lit = 0.0, rays = 0.0;
        for (int b = 0; b < census_bounces; ++b) {
            lit += double(census[size_t(b) * QUEUES_PER_BOUNCE + 4u]), rays += double(census[size_t(b) * QUEUES_PER_BOUNCE + 0u]);
        }
        return (rays > 0.0 && lit >= 0.5 * rays) ? (shade_split & ~16) : shade_split;
    }
    DevBuf stack_spill;   // per-wave overflow slabs of the traversal stack
    // Round 5: the shadow rays of bounce b (K3: reads the shadow-ray planes, adds to the per-iteration pixel buffer) and the closest-hit launch of
    // bounce b + 1 (K2: reads the ray planes, writes the hit planes) touch disjoint state, so K3 goes to a second, low-priority stream and fills
    // the wave slots K2 leaves idle while its last, longest rays finish (a launch ends on the dependent fetch chain of ONE ray: 0.3-0.4 ms at any
    // launch size -- profiles/r05/timeline_rank0_of_8.txt).  The shade stage of bounce b + 1 waits for both (the order of the pixel additions is
    // the reference's: ShadeSecondary += after the shadow += of the bounce before).  RAYHIP_OVERLAP_SHADOW=0: one stream, as before.
    hipStream_t stream2 = nullptr;
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    DevBuf stack_spill2; // K3's own overflow slabs while it runs next to K2
    bool overlap_shadow = true;
    struct Interval { // a timed launch on the second stream (resolve_timing)
        size_t ev0, ev1;
        int stage, trav;
    };
    std::vector<Interval> pending2;
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
    static constexpr int QUEUES_PER_BOUNCE = 6;
    uint32_t *ray_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b) * QUEUE_WORDS; }
    uint32_t *shadow_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 1) * QUEUE_WORDS; }
    uint32_t *deferred_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 2) * QUEUE_WORDS; }
    uint32_t *point_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 3) * QUEUE_WORDS; }
    uint32_t *nee_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 4) * QUEUE_WORDS; }
    uint32_t *sky_count(int b) const { return counters.as<uint32_t>() + size_t(QUEUES_PER_BOUNCE * b + 5) * QUEUE_WORDS; }
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
    RayQueue sky_queue(int b, size_t items, uint32_t stripes) const { return make_queue(sky_count(b), items, stripes); }
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
    if (c->pick_plane.alloc(n * 16)) {
        return 1;
    }
    for (int pl = 0; pl < 4; ++pl) {
        if (c->record_ray_planes[pl].alloc(n * (pl == 3 ? 8 : 16))) {
            return 1;
        }
    }
    c->record_rays.o_pdf = nullptr, c->record_rays.d_cw = c->record_ray_planes[0].as<float4>(), c->record_rays.c_cs = c->record_ray_planes[1].as<float4>();
    c->record_rays.ior = c->record_ray_planes[2].as<float4>(), c->record_rays.xy_depth = c->record_ray_planes[3].as<uint2>();
    if (c->sky_index.alloc(n * 4)) {
        return 1;
    }
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
    // a plain timestamp on stream `s` (second-stream launches: rayhip_ctx::pending2); returns the event's index or -1
    long stamp(hipStream_t s) {
        if (c->events_used == c->events.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) {
                return -1;
            }
            c->events.push_back(e);
        }
        if (hipEventRecord(c->events[c->events_used], s) != hipSuccess) {
            return -1;
        }
        return long(c->events_used++);
    }
};

int resolve_timing(rayhip_ctx *c) {
    if (c->pending.empty()) {
        return 0;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (const rayhip_ctx::Interval &iv : c->pending2) { // launches that ran on the second stream, next to the main one: their own time
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->events[iv.ev0], c->events[iv.ev1]));
        if (iv.stage >= 0) {
            c->stage_us[iv.stage] += double(ms) * 1000.0;
        }
        if (iv.trav >= 0) {
            c->trav_ms[iv.trav] += double(ms);
            c->trav_launches[iv.trav] += 1;
        }
    }
    c->pending2.clear();
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

int rayhip_abi_version(void) { return RAYHIP_ABI_VERSION; }

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
    if (const char *e = getenv("RAYHIP_OVERLAP_SHADOW")) {
        c->overlap_shadow = atoi(e) != 0;
    }
    if (c->overlap_shadow) {
        int least = 0, greatest = 0; // (numerically: greatest priority <= least priority)
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, least) != hipSuccess ||
            hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->join_event, hipEventDisableTiming) != hipSuccess) {
            c->overlap_shadow = false;
        }
    }
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
    if (const char *e = getenv("RAYHIP_RAYGEN_SAMPLES")) {
        const int v = atoi(e);
        c->raygen_samples_per_wave = (v == 4 || v == 16 || v == 64) ? v : 1;
    }
    if (const char *e = getenv("RAYHIP_SHADOW_REFILL")) {
        c->shadow_refill = atoi(e) != 0;
    }
    if (const char *e = getenv("RAYHIP_SORT_KEY")) {
        c->sort_key_mode = std::max(0, std::min(3, atoi(e)));
    }
    if (const char *e = getenv("RAYHIP_SHADE_SPLIT")) {
        c->shade_split = atoi(e) & 31;
        c->shade_form_auto = false;
    }
    if (const char *e = getenv("RAYHIP_NOTEX_KERNELS")) {
        c->notex_kernels = atoi(e) != 0;
    }
    if (const char *e = getenv("RAYHIP_PICK_LDS")) {
        c->pick_lds = atoi(e) != 0;
    }
    // The persistent ray-refill form of the closest-hit kernel (kernels_closest_refill.hip.h): lanes whose ray is finished fetch the next one
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
            {
                int per_cu4 = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, k_trace_closest_refill<4>, WAVE, 0) != hipSuccess || per_cu4 <= 0) {
                    per_cu4 = per_cu_refill;
                }
                c->refill_resident4 = c->props.multiProcessorCount * per_cu4;
            }
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
    if (c->stack_spill.alloc(size_t(c->grid_waves) * std::max(STACK_SPILL_DEPTH * WAVE, POOL_SLAB_WORDS) * sizeof(uint32_t)) ||
        (c->overlap_shadow && c->stack_spill2.alloc(size_t(c->grid_waves) * size_t(STACK_SPILL_DEPTH * WAVE) * sizeof(uint32_t)))) {
        delete c;
        return 1;
    }
    if (c->counters.alloc(sizeof(uint32_t) * rayhip_ctx::QUEUES_PER_BOUNCE * MAX_BOUNCE_SLOTS * rayhip_ctx::QUEUE_WORDS) || c->trav_counters.alloc(sizeof(unsigned long long) * 2 * TRAV_COUNTER_WORDS) ||
        c->work_counters.alloc(sizeof(uint32_t) * size_t(MAX_BOUNCE_SLOTS) * rayhip_ctx::WORK_PER_BOUNCE * WORK_COUNTER_STRIDE)) {
        delete c;
        return 1;
    }
    if (const char *e = getenv("RAYHIP_DYNAMIC")) {
        c->dynamic_chunks = atoi(e) != 0;
    }
    if (const char *e = getenv("RAYHIP_DYN_MULT")) {
        c->dyn_mult = std::max(1, std::min(16, atoi(e)));
    }
    if (const char *e = getenv("RAYHIP_CENSUS")) {
        c->census_on = atoi(e) != 0;
    }
    if (const char *e = getenv("RAYHIP_CHUNKS_PER_BLOCK")) {
        c->chunks_per_block = std::max(1, std::min(1024, atoi(e)));
    }
    if (c->census_on) {
        void *hp = nullptr, *dp = nullptr;
        const size_t bytes = sizeof(uint32_t) * size_t(MAX_BOUNCE_SLOTS) * rayhip_ctx::QUEUES_PER_BOUNCE;
        if (hipHostMalloc(&hp, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || // (coherent: the device's stores are visible to the host once the event has fired)
             hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess ||
            hipEventCreateWithFlags(&c->census_event, hipEventDisableTiming) != hipSuccess) {
            if (hp) {
                (void)hipHostFree(hp);
            }
            c->census_on = false; // (a tuning aid: without it the launches keep their full grids)
        } else {
            c->census_host = static_cast<uint32_t *>(hp), c->census_dev = static_cast<uint32_t *>(dp);
            memset(hp, 0, bytes);
            c->census.assign(size_t(MAX_BOUNCE_SLOTS) * rayhip_ctx::QUEUES_PER_BOUNCE, 0.0f);
        }
    }
    {
        int per_cu_shadow = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_shadow, k_trace_shadow_refill<>, WAVE, 0) != hipSuccess || per_cu_shadow <= 0) {
            per_cu_shadow = per_cu;
        }
        c->shadow_resident = c->props.multiProcessorCount * per_cu_shadow;
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
    if (c->stream2) {
        (void)hipStreamSynchronize(c->stream2);
        (void)hipStreamDestroy(c->stream2);
    }
    for (hipEvent_t e : {c->fork_event, c->join_event}) {
        if (e) {
            (void)hipEventDestroy(e);
        }
    }
    c->stack_spill2.release();
    for (hipEvent_t e : c->events) {
        (void)hipEventDestroy(e);
    }
    if (c->census_event) {
        (void)hipEventDestroy(c->census_event);
    }
    if (c->census_host) {
        (void)hipHostFree(c->census_host);
    }
    DevBuf *all[] = {&c->pmj, &c->filter_table, &c->nodes, &c->tris, &c->tri_indices, &c->tri_materials, &c->materials,
                     &c->vertices, &c->vtx_indices, &c->mesh_instances, &c->lights, &c->li_indices, &c->light_cwnodes, &c->light_children, &c->light_tri_geom, &c->tri_verts, &c->tri_bitangents, &c->nodes4, &c->nodes8, &c->blas_root4, &c->env_qtree,
                     &c->textures, &c->texels, &c->px_temp, &c->px_full, &c->px_half, &c->px_raw, &c->px_final, &c->px_base,
                     &c->px_dn, &c->px_req, &c->px_aux_base, &c->px_aux_dn, &c->px_variance, &c->nlm_tm, &c->nlm_var_h, &c->nlm_var,
                     &c->tonemap_lut, &c->hit_planes[0], &c->hit_planes[1], &c->shadow_planes[0], &c->shadow_planes[1],
                     &c->shadow_planes[2], &c->deferred_planes[0], &c->deferred_planes[1], &c->counters, &c->trav_counters, &c->stack_spill,
                     &c->sort_keys[0], &c->sort_keys[1], &c->sort_idx[0], &c->sort_idx[1], &c->sort_temp, &c->shard_stage, &c->work_counters};
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
    c->pick_plane.release();
    for (DevBuf &b : c->record_ray_planes) {
        b.release();
    }
    for (auto &up : c->unet_pass) { // (the UNet's weights and its fifteen tensors -- 1.3 GB at 1080p -- were leaked)
        up.weights.release();
        up.weights_h.release();
        up.bias.release();
    }
    for (DevBuf &b : c->unet_tensor) {
        b.release();
    }
    for (DevBuf &b : c->unet_tensor_h) {
        b.release();
    }
    c->unet_images.release();
    c->unet_images_h.release();
    for (DevBuf *b : {&c->sky_desc, &c->sky_transmittance_lut, &c->sky_multiscatter_lut, &c->sky_dir_lights, &c->sky_weather, &c->sky_noise3d, &c->sky_curl, &c->sky_moon,
                      &c->sky_cirrus, &c->sky_index}) {
        b->release();
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
