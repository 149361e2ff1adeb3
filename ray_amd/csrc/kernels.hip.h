// kernels.hip.h -- the __global__ kernels of librayhip (gfx950 / CDNA4, wave64).
//
// Kernel map (reference GLSL kernel -> here; SURVEY.md section 2 kernel table):
//   K1  primary_ray_gen.comp.glsl            -> k_raygen
//   K2  intersect_scene.comp.glsl            -> k_trace_closest<COUNT, WIDE> (the roofline kernel; product form WIDE: the
//                                               4-wide quantised BLAS of rt_bvh4.h, majority-scheduled; COUNT: the
//                                               reference's BVH2 with visit counters), k_trace_closest_refill (the secondary bounces)
//                                               [kernels_closest_refill.hip.h; kernels_closest_pool.hip.h: the opt-in pooled form]
//   K3  intersect_scene_shadow.comp.glsl     -> k_trace_shadow, k_trace_shadow_refill (the passes' form) [kernels_shadow.hip.h]
//   K4  intersect_area_lights.comp.glsl      -> k_intersect_area_lights (+ k_shadow_blockers for the shadow-ray form)
//   K5  shade.comp.glsl (PRIMARY/SECONDARY)  -> shade_kernels.hip: k_surface / k_light_pick / k_scatter / k_shade_emissive
//   K9  prepare_indir_args.comp.glsl         -> (gone) ray counts stay in HBM; every stage is launched with a
//                                               fixed grid and grid-strides over the count it reads there, so the
//                                               bounce loop never returns to the host (RendererVK.cpp:641-712 records
//                                               all bounces up front for the same reason)
//   K10 mix_incremental + K11 postprocess    -> k_accumulate (fused: one pass over the rect)
//
// Conventions
//   * one wavefront (64 lanes) per workgroup in the traversal kernels: the traversal stack is a per-wavefront
//     LDS array laid out depth-major, stack[depth][lane] -> bank == lane for every lane at ANY mix of depths,
//     i.e. conflict-free by construction (MI355X_MICROARCH.md LDS table: ds_read/write_b32 conflict only
//     inside a 32-lane half).  RT_LDS_STACK_DEPTH (24) entries x 64 lanes x 4 B = 6 KiB per wave live in LDS, deeper
//     entries spill to a per-wave slab in HBM (the reference's shader keeps 48 in shared memory,
//     shaders/intersect_scene.comp.glsl:87).
//   * ray compaction between stages uses one atomic per wavefront: ballot + mbcnt prefix (wave_alloc) instead of
//     the reference's per-thread atomicAdd (shaders/shade.comp.glsl:2432,2452), and the wavefronts are spread over
//     64 counters (RayQueue below) because same-address atomics serialise at ~11 ns each on MI355X.
//   * all per-ray state is SoA float4 planes (rt_types.h): 16 B per lane per access, 1 KiB per wave instruction.
#pragma once

#include <hip/hip_runtime.h>

#ifdef RT_PROFILE_TRACE
// Tuning build: the same wave-time attribution inside k_trace_closest (sections 16..27 of g_prof_acc)
namespace rt {
__device__ unsigned long long g_prof_acc[32];
__shared__ unsigned long long s_prof_last;
__shared__ unsigned long long s_prof_acc[32];
__device__ __forceinline__ void prof_mark(const int k) {
    const unsigned long long mask = __ballot(1);
    if (int(__lane_id()) == __ffsll((long long)mask) - 1) {
        const unsigned long long t = __builtin_readcyclecounter();
        s_prof_acc[k] += t - s_prof_last;
        s_prof_last = t;
    }
}
__device__ __forceinline__ void prof_wait(float4 &a, float4 &b, float4 &c, float4 &d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x));
}
} // namespace rt
#define RT_PROF_T(k) ::rt::prof_mark(k);
#define RT_PROF_LANES(k)                                                                                                \
    {                                                                                                                   \
        const unsigned long long m_ = __ballot(1);                                                                      \
        if (int(__lane_id()) == __ffsll((long long)m_) - 1) {                                                           \
            ::rt::s_prof_acc[k] += (unsigned long long)__popcll(m_);                                                    \
            ::rt::s_prof_acc[(k) + 1] += 1ull;                                                                          \
        }                                                                                                               \
    }
#define RT_PROF_WAIT(a, b, c, d) ::rt::prof_wait(const_cast<float4 &>(a), const_cast<float4 &>(b), const_cast<float4 &>(c), const_cast<float4 &>(d));
#endif
#include "rt_arealights.h"
#include "rt_denoise.h"
#include "rt_params.h"
#include "rt_pixel.h"
#include "rt_sort.h"
#include "wavefront.hip.h"

namespace rt {

// Build-time knobs (defaults are what ships; tools/variants.py sweeps them):
//   RT_LDS_STACK_DEPTH   traversal-stack entries per lane kept in LDS (TLAS + BLAS levels share them).  LDS per
//                        wave = depth * 256 B, so 48 entries (the reference's MAX_STACK_SIZE, Constants.inl:4) cap a
//                        CU at 13 traversal waves, 32 at 20, 24 at 26, 16 at 40.  Entries beyond the LDS part
//                        spill to a per-wave slab in HBM (same depth-major layout), so any depth up to
//                        RT_STACK_TOTAL_DEPTH stays correct; SAH trees of the Bistro-class scene use <= 17.
//   RT_TRACE_MIN_WAVES   __launch_bounds__ occupancy hint (waves per SIMD) for the traversal kernels
#ifndef RT_LDS_STACK_DEPTH
#define RT_LDS_STACK_DEPTH 24
#endif
#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 6 // 80 VGPRs.  Sweep with the final kernels, 32-iteration passes: 4 waves 289, 5 waves 319, 6 waves 328 Msamples/s
#endif
constexpr int LDS_STACK_DEPTH = RT_LDS_STACK_DEPTH;
constexpr int STACK_TOTAL_DEPTH = 2 * MAX_STACK_SIZE; // TLAS + BLAS, 48 each in the reference
constexpr int STACK_SPILL_DEPTH = STACK_TOTAL_DEPTH - LDS_STACK_DEPTH;

// Depth-major per-wavefront stack: entries [0, DEPTH) live in LDS, deeper ones in a per-wave HBM slab (DEPTH = LDS_STACK_DEPTH for
// every kernel but the pooled closest-hit kernel, which trades stack entries for its pool of prepared rays).
template <int DEPTH>
struct LdsStackT {
    uint32_t *lane_base;  // &lds[0][lane]
    uint32_t *spill_base; // &slab[wave][0][lane]
    uint32_t size;
    __device__ __forceinline__ void push(uint32_t v) {
        if (size < uint32_t(DEPTH)) {
            lane_base[size * WAVE] = v;
        } else if (size < uint32_t(STACK_TOTAL_DEPTH)) {
            spill_base[(size - DEPTH) * WAVE] = v;
        }
        ++size;
    }
    __device__ __forceinline__ uint32_t pop() {
        --size;
        if (size < uint32_t(DEPTH)) {
            return lane_base[size * WAVE];
        }
        return size < uint32_t(STACK_TOTAL_DEPTH) ? spill_base[(size - DEPTH) * WAVE] : 0x1fffffffu;
    }
    // slot access for the 4-wide walk (rt_bvh4.h): three consecutive slots are one address + immediate offsets
    __device__ __forceinline__ bool fast_range(const uint32_t idx_end) const { return idx_end <= uint32_t(DEPTH); }
    __device__ __forceinline__ void write3_fast(const uint32_t idx, const uint32_t a, const uint32_t b, const uint32_t c) {
        uint32_t *p = lane_base + idx * WAVE;
        p[0] = a, p[WAVE] = b, p[2 * WAVE] = c;
    }
    __device__ __forceinline__ void write_at(const uint32_t idx, const uint32_t v) {
        if (idx < uint32_t(DEPTH)) {
            lane_base[idx * WAVE] = v;
        } else if (idx < uint32_t(STACK_TOTAL_DEPTH)) {
            spill_base[(idx - DEPTH) * WAVE] = v;
        }
    }
    __device__ __forceinline__ uint32_t read_at(const uint32_t idx) const {
        if (idx < uint32_t(DEPTH)) {
            return lane_base[idx * WAVE];
        }
        return idx < uint32_t(STACK_TOTAL_DEPTH) ? spill_base[(idx - DEPTH) * WAVE] : 0x1fffffffu;
    }
    // two-word entries of the 8-wide walk (rt_bvh8.h): slots idx and idx + 1 (one address, two immediate offsets -> a single
    // ds_write2st64_b32 / ds_read2st64_b32 while both are in the LDS part)
    __device__ __forceinline__ void write2_at(const uint32_t idx, const uint32_t a, const uint32_t b) {
        if (idx + 2 <= uint32_t(DEPTH)) {
            uint32_t *p = lane_base + idx * WAVE;
            p[0] = a, p[WAVE] = b;
        } else {
            write_at(idx, a), write_at(idx + 1, b);
        }
    }
    __device__ __forceinline__ void read2_at(const uint32_t idx, uint32_t &a, uint32_t &b) const {
        if (idx + 2 <= uint32_t(DEPTH)) {
            const uint32_t *p = lane_base + idx * WAVE;
            a = p[0], b = p[WAVE];
        } else if (idx + 2 <= uint32_t(STACK_TOTAL_DEPTH)) {
            a = read_at(idx), b = read_at(idx + 1);
        } else {
            a = 0xffffffffu, b = 0u; // beyond every stack: the walk's own sentinel (as read_at hands out the BVH2 one)
        }
    }
};
using LdsStack = LdsStackT<LDS_STACK_DEPTH>;

// (wave_alloc, RayQueue, the SoA / pixel-buffer structs: wavefront.hip.h; the shade kernels K5: shade_kernels.hip)

// which 8x8 pixel tiles the ray generator walks (see k_raygen)
struct RayGenTiling {
    uint32_t tiles;         // 8x8 tiles per layer
    uint32_t tiles_x;       // unsharded: tiles per row of the rect
    uint32_t owned_only;    // 1: walk only the shard tiles this rank owns
    uint32_t sub;           // 8x8 tiles per shard-tile side (shard.tile / 8)
    uint32_t shard_tiles_x; // shard tiles per frame row
    uint32_t samples_per_wave; // 1: a wavefront = one 8x8 tile of one layer; S = 4 / 16 / 64: a block of 64 / S pixels of the tile in S consecutive layers
};
// host and device: the tiling of a rect under a shard (owned_only needs shard tiles that are whole 8x8 tiles)
__host__ __device__ inline RayGenTiling make_raygen_tiling(const int frame_w, const int frame_h, const int rect_w, const int rect_h, const Shard sh) {
    RayGenTiling t;
    t.tiles_x = uint32_t(rect_w + 7) / 8u;
    t.tiles = t.tiles_x * (uint32_t(rect_h + 7) / 8u);
    t.owned_only = 0, t.sub = 1, t.shard_tiles_x = 1, t.samples_per_wave = 1;
    if (sh.count > 1 && sh.tile % 8 == 0) {
        const uint32_t stx = uint32_t(frame_w + sh.tile - 1) / uint32_t(sh.tile), sty = uint32_t(frame_h + sh.tile - 1) / uint32_t(sh.tile);
        const uint32_t total = stx * sty;
        const uint32_t owned = total > uint32_t(sh.index) ? (total - uint32_t(sh.index) + uint32_t(sh.count) - 1u) / uint32_t(sh.count) : 0u;
        t.owned_only = 1, t.sub = uint32_t(sh.tile) / 8u, t.shard_tiles_x = stx;
        t.tiles = owned * t.sub * t.sub;
    }
    return t;
}

// ---- K1 ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_raygen(const RayGenParams p, const uint32_t *__restrict__ pmj,
                                               const float *__restrict__ filter_table,
                                               const uint16_t *__restrict__ required_samples, const RaySoA rays,
                                               const HitSoA hits, const RayQueue out, const Layering layers,
                                               const RayGenTiling tiling) {
    // one wavefront = one 8x8 pixel tile: primary rays of a wave start coherent in both directions, so they share
    // BVH nodes in K2 and materials in the primary shade (a 64x1 strip only is coherent along x)
    //
    // Which 8x8 tiles a pass walks (RayGenTiling, computed on the host):
    //   * unsharded: the tiles of the rect, row-major from the rect's corner;
    //   * tile-sharded (rayhip_set_shard): only the 8x8 tiles inside the shard tiles this rank owns, frame-aligned --
    //     owned shard tile j is frame tile shard.index + j * shard.count -- so that the chunk numbering, the stripes and
    //     the wavefront-state buffers are as dense as the rank's share of the frame, not as the frame.
    const uint32_t tiles = tiling.tiles, n_chunks = tiles * uint32_t(layers.count), waves_per_block = blockDim.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    // pixel chunk pc -> stripe pc % stripes: every stripe gets at most ceil(n_chunks / stripes) chunks
    for (uint32_t pc = blockIdx.x * waves_per_block + threadIdx.x / WAVE; pc < n_chunks; pc += gridDim.x * waves_per_block) {
        uint32_t layer = pc / tiles;
        const uint32_t tile = pc % tiles; // wave-uniform
        uint32_t in_x = lane & 7u, in_y = lane >> 3; // the lane's pixel inside the 8x8 tile
        if (tiling.samples_per_wave > 1) {
            // S = 4, 16 or 64 samples of a pixel in one wavefront: chunk `layer` of a group of S stands for one (8 / sqrt S)-pixel-wide
            // block of the tile in all S layers of the group (the S rays of a pixel walk the same nodes, hit the same triangle, read
            // the same texels)
            const uint32_t S = tiling.samples_per_wave, side = S == 4 ? 4u : (S == 16 ? 2u : 1u), per_row = 8u / side;
            const uint32_t block = layer & (S - 1u);
            in_x = (lane % side) + side * (block % per_row), in_y = ((lane / side) % side) + side * (block / per_row);
            layer = (layer & ~(S - 1u)) + lane / (side * side);
        }
        int x, y;
        bool in_rect;
        if (tiling.owned_only) {
            const uint32_t sub_n = tiling.sub * tiling.sub, owned = tile / sub_n, sub = tile % sub_n;
            const uint32_t t = uint32_t(p.shard.index) + owned * uint32_t(p.shard.count); // frame shard-tile ordinal
            x = int((t % tiling.shard_tiles_x) * uint32_t(p.shard.tile) + (sub % tiling.sub) * 8u + in_x);
            y = int((t / tiling.shard_tiles_x) * uint32_t(p.shard.tile) + (sub / tiling.sub) * 8u + in_y);
            in_rect = x >= p.rect[0] && y >= p.rect[1] && x < p.rect[0] + p.rect[2] && y < p.rect[1] + p.rect[3];
            x = in_rect ? x : p.rect[0], y = in_rect ? y : p.rect[1];
        } else {
            const int lx = int((tile % tiling.tiles_x) * 8u + in_x), ly = int((tile / tiling.tiles_x) * 8u + in_y);
            in_rect = lx < p.rect[2] && ly < p.rect[3];
            x = p.rect[0] + (in_rect ? lx : 0), y = p.rect[1] + (in_rect ? ly : 0);
        }
        // (a batch is only formed when adaptive sampling is inert, so the check of the first iteration holds for all)
        const bool live = in_rect && pixel_owned(p.shard, p.w, x, y) && !(required_samples[y * p.w + x] < p.iteration);
        const uint32_t slot = out.alloc(pc % out.stripes, live);
        if (live) {
            Ray r;
            Hit h;
            if (layer == 0) {
                generate_primary_ray(p, pmj, filter_table, x, y, r, h);
            } else {
                RayGenParams pl = p;
                pl.iteration = p.iteration + int(layer);
                pl.rand_seed = layer_rand_seed(pl.iteration);
                generate_primary_ray(pl, pmj, filter_table, x, y, r, h);
                r.xy += layer_offset_xy(layers, layer);
            }
            store_ray(rays, slot, r, p.skip_ior == 0);
            store_hit(hits, slot, h);
        }
    }
}

// instrumented builds: fold one ray's visit counts into the per-kernel totals (TRAV_COUNTER_WORDS u64 per kernel)
constexpr int TRAV_COUNTER_WORDS = 6;
__device__ __forceinline__ void flush_trav_count(unsigned long long *__restrict__ counters, const TravCount &tc) {
    atomicAdd(&counters[0], 1ull);
    atomicAdd(&counters[1], (unsigned long long)tc.nodes);
    atomicAdd(&counters[2], (unsigned long long)tc.tris);
    atomicAdd(&counters[3], (unsigned long long)tc.instances);
    atomicMax(&counters[4], (unsigned long long)tc.max_stack);
    atomicAdd(&counters[5], (unsigned long long)tc.nodes4);
}

// ---- K2 ---------------------------------------------------------------------------------------------------
// One ray per lane; grid-stride over the device-resident ray count.
// COUNT: with visit counters; WIDE: 0 = the reference's BVH2, 4 / 8 = the quantised wide BLAS (rt_bvh4.h / rt_bvh8.h).
// <true, 0> = instrumented walk of the reference's BVH2 (counters = the reference algorithm's bytes), <true, 4 | 8> = the
// product walk with counters (its OWN algorithmic bytes)
// MINW: occupancy hint.  The default (6 waves/SIMD, 80 VGPRs, some spills outside the hot loop) is best when node and
// triangle fetches miss the caches; a scene that fits L2 (RT_TRACE_SMALL_WAVES = 5: 96 VGPRs, fewer spills) has little
// latency to hide and runs 15 % faster in K2 with the smaller footprint (Cornell: 0.42 -> 0.35 ms per iteration).
#ifndef RT_TRACE_SMALL_WAVES
#define RT_TRACE_SMALL_WAVES 5
#endif
template <bool COUNT, int WIDE, int MINW = RT_TRACE_MIN_WAVES>
__global__ void __launch_bounds__(WAVE, MINW) k_trace_closest(const SceneView sc, const TraceParams tp, const RaySoA rays,
                                                       const HitSoA hits, const RayQueue queue,
                                                       const int init_hits, uint32_t *__restrict__ stack_spill,
                                                       unsigned long long *__restrict__ counters, const Layering layers) {
    __shared__ uint32_t lds_stack[LDS_STACK_DEPTH * WAVE];
    const uint32_t lane = threadIdx.x;
#ifdef RT_PROFILE_TRACE
    if (WIDE) {
        if (threadIdx.x < 32) {
            s_prof_acc[threadIdx.x] = 0;
        }
        if (threadIdx.x == 0) {
            s_prof_last = __builtin_readcyclecounter();
        }
    }
#endif
    const uint32_t n_live_chunks = queue.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!queue.chunk(c, stripe, slot0, n_live) || lane >= n_live) {
            continue;
        }
        RT_PROF_T(24)
        const uint32_t i = slot0 + lane;
        // Only what the walk needs is held in registers: origin, direction, the ray type.  Throughput, pixel and depth
        // counters stay in the ray planes and are fetched by `tail` if -- and only if -- the ray meets a non-solid surface
        // (they used to be spilled to scratch around the walk: ~170 B of scratch traffic per ray).
        Ray r;
        load_ray_od(rays, i, r);
        r.depth = rays.xy_depth[i].y;
        r.c = {0.0f, 0.0f, 0.0f}, r.cone_spread = 0.0f, r.xy = 0;
        Hit h = init_hits ? make_hit() : load_hit(hits, i);

        LdsStack st;
        st.lane_base = &lds_stack[lane];
        st.spill_base = stack_spill + size_t(blockIdx.x) * size_t(STACK_SPILL_DEPTH * WAVE) + lane;
        st.size = 0;
        TravCount tc = {0, 0, 0, 0, 0};
        RT_PROF_T(25)
        struct Tail {
            const RaySoA &rays;
            const Layering &layers;
            uint32_t i;
            bool fetched;
            uint32_t xy_virtual;
            __device__ void operator()(Ray &r, TraceParams &tp) {
                const float4 c = rays.c_cs[i];
                const uint2 xd = rays.xy_depth[i];
                r.c = {c.x, c.y, c.z}, r.cone_spread = c.w;
                r.depth = xd.y;
                // a later iteration of a batched pass: its own sample index / seed, random numbers keyed by the real pixel
                xy_virtual = xd.x;
                const uint32_t layer = xy_layer(xy_virtual, layers);
                r.xy = xy_real(xy_virtual, layers, layer);
                if (layer != 0) {
                    tp.iteration += int(layer);
                    tp.rand_seed = layer_rand_seed(tp.iteration);
                }
                fetched = true;
            }
        } tail = {rays, layers, i, false, 0u};
        intersect_scene_closest<WIDE>(sc, tp, r, h, st, COUNT ? &tc : nullptr, tail);
        RT_PROF_T(26)

        store_hit(hits, i, h);
        if (tail.fetched) { // (a ray that only met solid surfaces has nothing to write back)
            rays.c_cs[i] = mkfloat4(r.c.x, r.c.y, r.c.z, r.cone_spread);
            uint2 xd;
            xd.x = tail.xy_virtual, xd.y = r.depth;
            rays.xy_depth[i] = xd;
        }
        if (COUNT) {
            flush_trav_count(counters, tc);
        }
    }
#ifdef RT_PROFILE_TRACE
    if (WIDE) {
        RT_PROF_T(27)
        if (threadIdx.x < 32 && s_prof_acc[threadIdx.x] != 0) {
            atomicAdd(&g_prof_acc[threadIdx.x], s_prof_acc[threadIdx.x]);
        }
    }
#endif
}

#include "kernels_closest_refill.hip.h"
#include "kernels_closest_pool.hip.h"
#include "kernels_shadow.hip.h"

// ---- K4 ---------------------------------------------------------------------------------------------------
// Implicit hits of visible analytic lights for secondary rays; runs right after K2 on the same queue
// (CoreRef.cpp:4847-4849).  Only launched when the scene has visible lights.
__global__ void __launch_bounds__(WAVE) k_intersect_area_lights(const SceneView sc, const RaySoA rays, const HitSoA hits,
                                                               const RayQueue queue) {
    __shared__ uint32_t lds_light_stack[LIGHT_STACK_LDS_WORDS]; // 36 KiB: (index, distance, factor) x 48 entries x 64 lanes, depth-major
    const uint32_t lane = threadIdx.x;
    LightStackT<LightStackLds> st;
    st.s.lane_base = &lds_light_stack[lane];
    const uint32_t n_live_chunks = queue.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!queue.chunk(c, stripe, slot0, n_live) || lane >= n_live) {
            continue;
        }
        const uint32_t i = slot0 + lane;
        Ray r;
        load_ray_od(rays, i, r);
        const uint32_t depth = rays.xy_depth[i].y;
        Hit h = load_hit(hits, i);
        const Hit before = h;
        intersect_area_lights(sc, r.o, r.d, depth, h, st);
        if (h.obj_index != before.obj_index || h.t != before.t || h.u != before.u || h.v != before.v) {
            store_hit(hits, i, h);
        }
    }
}

// Shadow-ray form (CoreRef.cpp:4868-4870: rc *= IntersectAreaLights(sh_r)).  The factor is exactly 0 or 1, so it is
// applied to the ray's throughput BEFORE K3 instead of to K3's result: a blocked ray carries c = 0 through the
// any-hit walk and adds 0 to its pixel, bit for bit what the reference adds.
__global__ void __launch_bounds__(WAVE) k_shadow_blockers(const SceneView sc, const ShadowSoA shadow, const RayQueue queue) {
    __shared__ uint32_t lds_light_stack[LIGHT_STACK_LDS_WORDS];
    const uint32_t lane = threadIdx.x;
    LightStackT<LightStackLds> st;
    st.s.lane_base = &lds_light_stack[lane];
    const uint32_t n_live_chunks = queue.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!queue.chunk(c, stripe, slot0, n_live) || lane >= n_live) {
            continue;
        }
        const uint32_t i = slot0 + lane;
        const ShadowRay r = load_shadow(shadow, i);
        if (intersect_area_lights_shadow(sc, r, st) == 0.0f) {
            float4 cx = shadow.c_xy[i];
            cx.x = cx.y = cx.z = 0.0f;
            shadow.c_xy[i] = cx;
        }
    }
}

// ---- K6 / K8: ray sort ----------------------------------------------------------------------------------------
// K6: one sort key per live secondary ray (slots >= the live count get the DEAD key so a fixed-size sort works)
__global__ void __launch_bounds__(256) k_ray_keys(const RaySoA rays, const uint32_t *__restrict__ ray_count, const uint32_t cap,
                                                 const SortGrid grid, uint32_t *__restrict__ keys, uint32_t *__restrict__ idx, const int mode) {
    const uint32_t n = *ray_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
        uint32_t key = SORT_KEY_DEAD >> (32 - ray_sort_key_bits(mode));
        if (i < n) {
            const float4 o = rays.o_pdf[i], d = rays.d_cw[i];
            key = ray_sort_key(grid, f3{o.x, o.y, o.z}, f3{d.x, d.y, d.z}, mode);
        }
        keys[i] = key;
        idx[i] = i;
    }
}
// K8: gather the rays into sorted order (reference shaders/sort_reorder_rays.comp.glsl:29-36)
__global__ void __launch_bounds__(256) k_reorder_rays(const RaySoA src, const RaySoA dst, const uint32_t *__restrict__ idx,
                                                     const uint32_t *__restrict__ ray_count) {
    const uint32_t n = *ray_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t j = idx[i];
        dst.o_pdf[i] = src.o_pdf[j];
        dst.d_cw[i] = src.d_cw[j];
        dst.c_cs[i] = src.c_cs[j];
        dst.ior[i] = src.ior[j];
        dst.xy_depth[i] = src.xy_depth[j];
    }
}

// ---- K10 + K11 ----------------------------------------------------------------------------------------------
// Batched passes: `per_layer` describes layers [layer_base, layer_base + layer_count) (kernel arguments hold at most
// MAX_BATCH of them; longer passes fold their layers in several launches, in order).
__global__ void __launch_bounds__(256) k_accumulate(const AccumParams p, const PixelBuffers px, const Layering layers,
                                                   const AccumLayers per_layer, const int layer_base, const int layer_count) {
    const int n = p.rect[2] * p.rect[3];
    const size_t pitch = size_t(virtual_width(layers));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int x = p.rect[0] + i % p.rect[2], y = p.rect[1] + i / p.rect[2];
        if (!pixel_owned(p.shard, p.w, x, y)) {
            continue; // another rank's pixel: stays zero here, filled in by the frame reduce
        }
        if (layers.count <= 1) {
            accumulate_pixel(p, x, y, px.temp + (y * p.w + x), px.variance + (y * p.w + x), px.full, px.half, px.raw, px.final_, px.required_samples);
            continue;
        }
        // batched pass: fold the layers into the running means in iteration order (what one call per iteration does).  The pixel's state
        // -- both means, the mark of the adaptive sampling, the two feature means -- stays in registers across the layers and is written
        // once: a call per layer wrote 114 bytes per pixel and layer (15 GB per 64-layer 1080p pass, three times what the pass reads)
        const int idx = y * p.w + x;
        AccumPixel st = load_accum_pixel(idx, px.full, px.half, px.required_samples);
        const float4 b0 = px.base_color[idx], d0 = px.depth_normals[idx];
        f4 base = {b0.x, b0.y, b0.z, b0.w}, dn = {d0.x, d0.y, d0.z, d0.w};
        AccumParams pl = p;
        for (int k = 0; k < layer_count; ++k) {
            pl.iteration = per_layer.l[k].iteration;
            pl.mix_factor = per_layer.l[k].mix_factor, pl.half_mix_factor = per_layer.l[k].half_mix_factor;
            pl.is_class_a = per_layer.l[k].is_class_a, pl.variance_threshold = per_layer.l[k].variance_threshold;
            // this pixel on layer layer_base + k of the virtual frame
            const uint32_t off = layer_offset_xy(layers, uint32_t(layer_base + k));
            const size_t vidx = (size_t(off & 0xffffu) + size_t(y)) * pitch + size_t(off >> 16) + size_t(x);
            // (a one-by-one run samples the pixel in this iteration iff the previous accumulate left it queued)
            if (!(st.required < pl.iteration)) { // blend_aux_pixel (rt_pixel.h) on the register copies
                const float4 nb = px.aux_base_layers[vidx], nd = px.aux_dn_layers[vidx];
                base += (f4{nb.x, nb.y, nb.z, nb.w} - base) * pl.mix_factor;
                dn += (f4{nd.x, nd.y, nd.z, nd.w} - dn) * pl.mix_factor;
            }
            accumulate_step(pl, px.temp[vidx], st);
        }
        px.base_color[idx] = mkfloat4(base.x, base.y, base.z, base.w);
        px.depth_normals[idx] = mkfloat4(dn.x, dn.y, dn.z, dn.w);
        store_accum_pixel(pl, idx, st, px.variance + idx, px.full, px.half, px.raw, px.final_, px.required_samples);
    }
}

// tonemap-only pass used after the multi-GPU frame reduce (raw/full already hold the combined image)
__global__ void __launch_bounds__(256) k_retonemap(const AccumParams p, const PixelBuffers px, const int h) {
    const int n = p.w * h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 ff = px.full[i];
        px.raw[i] = ff;
        const f4 c = tonemap(p, f4{ff.x, ff.y, ff.z, ff.w});
        px.final_[i] = mkfloat4(c.x, c.y, c.z, c.w);
    }
}

// FINAL <- Tonemap(RAW) over a rect: the tail of DenoiseImage(pass 15) (RendererCPU.h:984-995)
__global__ void __launch_bounds__(256) k_tonemap_raw_rect(const AccumParams p, const PixelBuffers px) {
    const int n = p.rect[2] * p.rect[3];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int idx = (p.rect[1] + i / p.rect[2]) * p.w + p.rect[0] + i % p.rect[2];
        const float4 r = px.raw[idx];
        const f4 c = tonemap(p, f4{r.x, r.y, r.z, r.w});
        px.final_[idx] = mkfloat4(c.x, c.y, c.z, c.w);
    }
}

// ---- multi-GPU frame exchange (rayhip_comm_reduce_framebuffers, rayhip_export_owned / rayhip_import_owned) -------------------
// The shards of a frame are DISJOINT sets of tiles, so "reduce" is a gather: every rank packs the tiles it owns densely
// (owned tile j of rank r = frame tile r + j * N, a full tile x tile slot each, ragged edge tiles padded with zeros) and the
// root scatters what it receives into its images.  A rank's operand is 1/N of the frame; nothing another rank's buffers
// hold on pixels they do not own ever travels.
// (slot <-> pixel mapping: rt_base.h, shared with the host build)
__global__ void __launch_bounds__(256) k_pack_owned_dense(const float4 *__restrict__ src, float4 *__restrict__ dst, const int w, const int h,
                                                         const Shard shard, const int owned) {
    const int n = owned * shard.tile * shard.tile;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int x, y;
        dst[i] = shard_slot_pixel(shard, w, h, i, x, y) ? src[y * w + x] : mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}
// `shard` = the SENDER's (tile, count, index)
__global__ void __launch_bounds__(256) k_unpack_owned_dense(const float4 *__restrict__ src, float4 *__restrict__ dst, const int w, const int h,
                                                           const Shard shard, const int owned) {
    const int n = owned * shard.tile * shard.tile;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int x, y;
        if (shard_slot_pixel(shard, w, h, i, x, y)) {
            dst[y * w + x] = src[i];
        }
    }
}
// full-frame form (owned pixels, zero elsewhere): rayhip_export_shard_device, for hosts that sum frames themselves
__global__ void __launch_bounds__(256) k_pack_owned(const float4 *__restrict__ src, float4 *__restrict__ dst, const int w, const int h,
                                                   const Shard shard) {
    const int n = w * h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        dst[i] = pixel_owned(shard, w, i % w, i / w) ? src[i] : mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}
// triangle records re-pitched from 48 to 64 bytes (rt_isect.h: TriTable): thread = one 16-byte row of the padded table
__global__ void __launch_bounds__(256) k_pad_tris(const float4 *__restrict__ src, float4 *__restrict__ dst, const size_t n_tris) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n_tris * 4) {
        const size_t t = i >> 2, r = i & 3;
        dst[i] = r < 3 ? src[t * 3 + r] : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

__global__ void __launch_bounds__(256) k_copy_f4(const float4 *__restrict__ src, float4 *__restrict__ dst, const size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        dst[i] = src[i];
    }
}

// ---- DenoiseImage (NLM): rt_denoise.h ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nlm_prepare_h(const DenoiseParams p, const PixelBuffers px, float4 *__restrict__ tm,
                                                      float4 *__restrict__ var_h) {
    const int n = p.ext_w * p.ext_h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        nlm_prepare_h(p, i % p.ext_w, i / p.ext_w, px.full, px.variance, tm, var_h);
    }
}
__global__ void __launch_bounds__(256) k_nlm_prepare_v(const DenoiseParams p, const float4 *__restrict__ var_h, float4 *__restrict__ var) {
    const int iw = p.ext_w - 8, ih = p.ext_h - 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < iw * ih; i += gridDim.x * blockDim.x) {
        nlm_prepare_v(p, 4 + i % iw, 4 + i / iw, var_h, var);
    }
}
// Stage 3.  One 256-thread block per 16x16 pixel tile.  The reference evaluates, for every pixel i and window position
// j, nine pair distances d(i + o, j + o) over the 3x3 patch offsets o -- 441 per pixel, each with four IEEE divisions.
// d(i + o, j + o) only depends on the pixel i + o and on the window offset j - i, so for one window offset the block
// computes the distance image ONCE over its tile plus a one-pixel rim (18x18 values in LDS) and every pixel sums its
// nine neighbours of that image, in the reference's order (q outer, p inner): the same additions of the same values,
// hence the same result, with 9x fewer divisions.  The tile's part of the two inputs (24x24 with the 3 + 1 pixel rim) is
// staged through LDS once.  Measured, 1080p: 2.25 ms for the per-pixel form -> see DESIGN.md.
__global__ void __launch_bounds__(256) k_nlm_filter(const DenoiseParams p, const AccumParams tone, const PixelBuffers px,
                                                   const float4 *__restrict__ tm, const float4 *__restrict__ var) {
    constexpr int T = 16, R = 4, S = T + 2 * R; // tile, rim (window radius 3 + patch radius 1), staged side
    constexpr int DS = T + 2;                   // side of the distance image
    __shared__ float4 s_tm[S * S], s_var[S * S], s_d[DS * DS];
    const int tiles_x = (p.rect[2] + T - 1) / T, tiles_y = (p.rect[3] + T - 1) / T;
    const int lx = int(threadIdx.x % T), ly = int(threadIdx.x / T);
    for (int t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {
        const int tx0 = (t % tiles_x) * T, ty0 = (t / tiles_x) * T; // region coordinates of the tile's corner
        __syncthreads();                                            // (the previous tile's readers are done)
        for (int i = int(threadIdx.x); i < S * S; i += 256) {
            // staged position (i % S, i / S) <-> extended-region pixel (tx0 + 8 - 4 + ., ty0 + 8 - 4 + .); positions past the
            // extended region only feed pixels past the region, which are not written: clamp to stay inside the buffers
            const int ex = min(tx0 + NLM_EXT_RADIUS - R + i % S, p.ext_w - 1), ey = min(ty0 + NLM_EXT_RADIUS - R + i / S, p.ext_h - 1);
            s_tm[i] = tm[ey * p.ext_w + ex], s_var[i] = var[ey * p.ext_w + ex];
        }
        __syncthreads();
        const int x = tx0 + lx, y = ty0 + ly;
        const bool valid = x < p.rect[2] && y < p.rect[3];
        const int ix = NLM_EXT_RADIUS + (valid ? x : 0), iy = NLM_EXT_RADIUS + (valid ? y : 0);
        const f4 f0_i = nlm_feature(p, px.base_color, ix, iy), f1_i = nlm_feature(p, px.depth_normals, ix, iy);
        f4 sum_output = {0.0f, 0.0f, 0.0f, 0.0f};
        float sum_weight = 0.0f;
        for (int k = -3; k <= 3; ++k) {
            for (int l = -3; l <= 3; ++l) {
                for (int i = int(threadIdx.x); i < DS * DS; i += 256) { // distance image of window offset (l, k)
                    const int u = i % DS - 1 + R, v = i / DS - 1 + R;    // staged coordinates of pixel i + o
                    s_d[i] = st4(nlm_pair_distance(ld4(s_tm[v * S + u]), ld4(s_tm[(v + k) * S + (u + l)]), ld4(s_var[v * S + u]),
                                                   ld4(s_var[(v + k) * S + (u + l)])));
                }
                __syncthreads();
                f4 color_distance = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int q = -1; q <= 1; ++q) {
                    for (int pp = -1; pp <= 1; ++pp) {
                        color_distance += ld4(s_d[(ly + 1 + q) * DS + (lx + 1 + pp)]);
                    }
                }
                const float weight = nlm_weight(color_distance, f0_i, nlm_feature(p, px.base_color, ix + l, iy + k), f1_i,
                                                nlm_feature(p, px.depth_normals, ix + l, iy + k));
                sum_output += ld4(s_tm[(ly + R + k) * S + (lx + R + l)]) * weight;
                sum_weight += weight;
                __syncthreads(); // s_d is rewritten for the next offset
            }
        }
        if (valid) {
            if (sum_weight != 0.0f) {
                sum_output = sum_output / sum_weight;
            }
            const int idx = (p.rect[1] + y) * p.w + (p.rect[0] + x);
            nlm_finish_pixel(p, tone, idx, ld4(s_var[(ly + R) * S + (lx + R)]), sum_output, px.raw, px.final_, px.required_samples);
        }
    }
}

// per-triangle vertex table (shade_point.h: fill_tri_verts) from the vertex and index arrays already in HBM: one thread per
// triangle, the same function the host build runs (IEEE operations: the same bits)
__global__ void __launch_bounds__(256) k_fill_tri_verts(const rayhip_vertex *__restrict__ vertices, const uint32_t vertices_count,
                                                       const uint32_t *__restrict__ vtx_indices, const uint32_t n_tris,
                                                       const rayhip_tri_mat_data *__restrict__ tri_materials, const uint32_t tri_materials_count,
                                                       float4 *__restrict__ tri_verts, float4 *__restrict__ tri_bitangents) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tris) {
        float4 rows[TRI_VERTS_STRIDE], brows[TRI_BITANGENTS_STRIDE];
        fill_tri_verts(vertices, vertices_count, vtx_indices, t, tri_materials, tri_materials_count, rows, brows);
        for (int k = 0; k < TRI_VERTS_STRIDE; ++k) {
            tri_verts[size_t(t) * TRI_VERTS_STRIDE + k] = rows[k];
        }
        for (int k = 0; k < TRI_BITANGENTS_STRIDE; ++k) {
            tri_bitangents[size_t(t) * TRI_BITANGENTS_STRIDE + k] = brows[k];
        }
    }
}

// The baked sky map on the device (rt_sky.h: sky_bake_texel; the reference's GPU scene does the same in a compute pass, SceneGPU.h:1697-1768): one
// thread per texel -- a texel is a view-ray integral through air, the cloud layer (48 steps with a 24-step shadow march each) and cirrus per light.
__global__ void __launch_bounds__(64) k_bake_sky(const SkyView sky, const rayhip_light *__restrict__ lights, const int w, const int h, uint32_t *__restrict__ out) {
    const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < w * h) {
        out[i] = sky_bake_texel(sky, lights, i % w, i / w, w, h);
    }
}

// Fill of every queue of a pass (sum over the stripe counters), one wavefront per queue: what the next pass sizes its launches from
// (rayhip_render.hip.h: queue census).  `out` is host memory the device writes directly.
__global__ void __launch_bounds__(WAVE) k_queue_totals(const uint32_t *__restrict__ counters, uint32_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x;
    uint32_t n = lane < QUEUE_MAX_STRIPES ? counters[size_t(blockIdx.x) * QUEUE_MAX_STRIPES * QUEUE_COUNTER_STRIDE + lane * QUEUE_COUNTER_STRIDE] : 0u;
    for (int m = 32; m >= 1; m >>= 1) {
        n += uint32_t(__shfl_xor(int(n), m));
    }
    if (lane == 0) {
        out[blockIdx.x] = n;
    }
}

__global__ void k_fill_u16(uint16_t *p, const uint16_t v, const size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        p[i] = v;
    }
}
__global__ void k_fill_f4(float4 *p, const float4 v, const size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        p[i] = v;
    }
}

// test hook: get_scrambled_2d_rand on the device
__global__ void k_scrambled_rand(const uint32_t *dims, const uint32_t *seeds, const int32_t *samples, const int n,
                                 const uint32_t *__restrict__ pmj, float2 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const f2 r = get_scrambled_2d_rand(dims[i], seeds[i], samples[i], pmj);
        out[i] = make_float2(r.x, r.y);
    }
}

} // namespace rt
