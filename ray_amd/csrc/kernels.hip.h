// kernels.hip.h -- the __global__ kernels of librayhip (gfx950 / CDNA4, wave64).
//
// Kernel map (reference GLSL kernel -> here; SURVEY.md section 2 kernel table):
//   K1  primary_ray_gen.comp.glsl            -> k_raygen
//   K2  intersect_scene.comp.glsl            -> k_trace_closest<COUNT, WIDE> (the roofline kernel; product form WIDE: the
//                                               4-wide quantised BLAS of rt_bvh4.h, majority-scheduled; COUNT: the
//                                               reference's BVH2 with visit counters), k_trace_closest_refill (the secondary bounces)
//   K3  intersect_scene_shadow.comp.glsl     -> k_trace_shadow
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

// ---- K2, persistent form (the secondary bounces by default, rayhip.hip: RAYHIP_REFILL): wavefronts with ray refill ------------
// With one ray per lane traced to completion (k_trace_closest above) a wavefront is as slow as its longest ray: measured
// on the Bistro-class scene inside the majority-scheduled BLAS loop, 45 % of the lane slots belong to rays that are
// already finished.  Here a wavefront is persistent and a lane that finishes its ray takes the next one from the queue
// while the other lanes keep walking (Aila & Laine 2009), on top of the majority scheduling of rt_bvh4.h.
//
// Per ray this performs exactly the node visits, instance entries and triangle tests of intersect_scene_closest<true>
// (same functions, same order), so the hits are the same bits (test_gpu_parity.py::test_refill_kernel_is_bit_identical);
// only the interleaving between lanes differs.  The nested loops of rt_traverse.h become:
//   BLAS part     the majority loop (node visit vs leaf) over the lanes that are inside an instance, left as soon as
//                 RT_REFILL_MIN lanes wait outside
//   service, D    finish rays whose TLAS walk is over (index indirection, transparency round of IntersectScene, store)
//                 and hand the idle lanes their next rays
//   service, C    TLAS steps (BVH2 node / instance entry) until every lane is inside an instance or through
// One stack per lane for both levels: the top lives in a register (`tos`, rt_bvh4.h); entering an instance saves the
// TLAS `tos` below a sentinel, so the pop that ends the BLAS walk restores it.
//
// RESULTS (MI355X, Bistro-class 1080p).  Round 1, 32-iteration passes, one block per resident wave slot, every bounce:
// lanes busy in a node visit 48 % -> 65 %, in a triangle test 21 % -> 40 %, wave-level node visits -24 %, triangle tests
// -46 % -- and 2.92 ms instead of 2.87: the per-LANE work is unchanged, and that is what the memory pipeline (texture
// addresser + L2 + random 64-byte HBM reads) sees.  Round 2, after the cheaper node test and the leaf refinement made
// instruction issue the larger term (64-iteration passes): the coherent primary rays lose (0.52 vs 0.35 ms: every lane of
// a primary wavefront is busy to the end anyway, the refill only adds its service loop), the secondary bounces gain, and
// a grid of 16 blocks per wave slot instead of 1 removes the tail of the launch: 1.90 -> 1.74 ms per iteration for the
// secondary bounces (K2 2.25 -> 2.05 ms), bit-identical frames.  That is the default now (rayhip.hip: RAYHIP_REFILL=2).
// RT_REFILL_MIN (lanes waiting before the wavefront leaves the BLAS loop to serve them): 16: 2.15, 24: 2.06, 32: 2.07,
// 40: 2.04 ms.
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 40
#endif
#ifndef RT_REFILL_MIN_WAVES
#define RT_REFILL_MIN_WAVES 6 // 80 VGPRs, 40 bytes of scratch.  Round 2: 5 (96 VGPRs; 6 spilled in the loop and lost).  With round 3's shorter node
                              // test 6 wins: K2 1.92 against 2.02 ms per iteration; 7 (72 VGPRs, 76 bytes of scratch) 2.26
#endif
// MIN_WAIT: lanes that must be waiting before the wavefront leaves the BLAS loop to serve them.  RT_REFILL_MIN for the incoherent
// secondary bounces; WAVE for coherent primary rays -- the wavefront then finishes its 64 rays together and takes the next
// chunk whole, i.e. the schedule of the plain kernel, with this kernel's flat register footprint (96 VGPRs, no spill stores
// inside the walk, where the plain kernel's nested walks spill 60 VGPRs at 80: 7.8 GB of scratch writes per primary launch)
// Tried on top of this schedule and dropped (round 3, numbers in DESIGN.md section 3a): putting a leaf aside and going on with the next
// node ("postponed leaves": busier lanes, but a stale distance limit -- 6 % more node visits, slower); requesting the next node
// into an LDS sink the moment it is known (global_load_lds as a prefetch: much slower); other vote weights (flat).
// the vote between a node step and a leaf step: a node step while  n_node * DEN >= n_leaf * NUM  (1 / 1: plain majority)
#ifndef RT_REFILL_VOTE_NUM
#define RT_REFILL_VOTE_NUM 1
#define RT_REFILL_VOTE_DEN 1
#endif
template <int WIDE, int MIN_WAIT = RT_REFILL_MIN>
__global__ void __launch_bounds__(WAVE, RT_REFILL_MIN_WAVES) k_trace_closest_refill(const SceneView sc, const TraceParams tp, const RaySoA rays,
                                                               const HitSoA hits, const RayQueue queue, const int init_hits,
                                                               uint32_t *__restrict__ stack_spill, const Layering layers) {
    __shared__ uint32_t lds_stack[LDS_STACK_DEPTH * WAVE];
    const uint32_t lane = threadIdx.x;
#ifdef RT_PROFILE_TRACE
    if (threadIdx.x < 32) {
        s_prof_acc[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) {
        s_prof_last = __builtin_readcyclecounter();
    }
#endif
    LdsStack st;
    st.lane_base = &lds_stack[lane];
    st.spill_base = stack_spill + size_t(blockIdx.x) * size_t(STACK_SPILL_DEPTH * WAVE) + lane;
    st.size = 0;

    enum : uint32_t { IDLE = 0, TLAS = 1, BLAS = 2 };
#ifdef RT_PROFILE_TRACE
    uint32_t st_a = 0, st_b = 0, st_iter = 0, st_serv = 0, st_serv_lanes = 0, st_tlas = 0; // (uniform)
#endif
    // lane state.  4-wide: `cur` / `tos` are node words at both levels.  8-wide (rt_bvh8.h): inside an instance `cur` / `tos` are the
    // child_base of the current / topmost group and `cur_bits` / `tos_bits` their pending masks, (`tri_base`, `l0`, `l1`) the hit leaf
    // children of the node visited last; at the top level they are BVH2 node words as before.
    uint32_t lvl = IDLE, slot = 0, cur = BVH4_SENTINEL, tos = BVH4_SENTINEL, size = 0, mi_index = 0, ray_flags = 0;
    uint32_t cur_bits = 0, tos_bits = 0, tri_base = 0, l0 = 0, l1 = 0, oct_inv = 0;
    bool res = false;
    f3 ro = {0.0f, 0.0f, 0.0f}, rd = {0.0f, 0.0f, 1.0f}; // world-space origin of the current transparency segment, direction
    f3 o = ro, d = rd, inv_d = rd;                        // object-space ray of the instance being walked
    Hit h = make_hit();
    float t_val = 0.0f;
    // wavefront state (uniform): the chunk being handed out, the next chunk index of this wavefront
    uint32_t pool_slot = 0, pool_left = 0;
    ChunkWalk walk(queue.live_chunks());

    auto begin_round = [&]() { // IntersectScene loop head + walk prologue at TLAS level
        t_val = h.t;
        res = false;
        size = 0;
        st.write_at(size++, BVH4_SENTINEL);
        tos = BVH4_SENTINEL;
        cur = tp.root_index;
        lvl = TLAS;
    };
    auto pop = [&]() {
        cur = tos;
        tos = st.read_at(--size);
    };
    // the pop that ends a BLAS walk hands back the sentinel and restores the TLAS `tos`: continue the TLAS walk
    auto leave_blas = [&]() {
        if (WIDE == 8) {
            if (lvl == BLAS && cur == BVH8_SENTINEL) { // (a real group never stays current once it is exhausted: pop8)
                lvl = TLAS;
                cur = st.read_at(--size); // the top-level `tos` saved at the entry of the instance
                tos = st.read_at(--size);
            }
        } else if (lvl == BLAS && cur == BVH4_SENTINEL) {
            lvl = TLAS;
            pop();
        }
    };
    auto pop8 = [&]() {
        cur = tos, cur_bits = tos_bits;
        size -= 2;
        st.read2_at(size, tos, tos_bits);
    };

    uint32_t n_dead = 0; // idle lanes that can no longer be refilled (uniform)
    for (;;) {
        // ---- BLAS part: the majority-scheduled walk of rt_bvh4.h over the lanes that are inside an instance; left as soon
        // as RT_REFILL_MIN lanes wait outside for the service part below
        for (;;) {
            const bool in_blas = (lvl == BLAS);
            // (the sentinel never stays in `cur`: leave_blas)
            const bool at_leaf = in_blas && (WIDE == 8 ? (l0 | l1) != 0u : (cur & BVH2_PRIM_COUNT_BITS) != 0);
            const bool at_node = in_blas && !at_leaf;
            const int n_node = __popcll(__ballot(at_node)), n_leaf = __popcll(__ballot(at_leaf));
            const int n_out = WAVE - n_node - n_leaf - int(n_dead);
#ifdef RT_PROFILE_TRACE
            st_a += n_node, st_b += n_leaf, st_iter += 1;
#endif
            // (the counts come from ballots: uniform, the branches are scalar)
            if (n_node + n_leaf == 0 || n_out >= MIN_WAIT) {
                break;
            }
            if (n_node * RT_REFILL_VOTE_DEN >= n_leaf * RT_REFILL_VOTE_NUM) {
                if (at_node) {
                    if (WIDE == 8) {
                        const uint32_t node = bvh8_take_child(cur, cur_bits, oct_inv);
                        if ((cur_bits >> 8) != 0u) { // siblings remain: the group goes onto the stack
                            st.write2_at(size, tos, tos_bits);
                            size += 2;
                            tos = cur, tos_bits = cur_bits;
                        }
                        Bvh8Visit v;
                        bvh8_test_node(sc.nodes8, node, o, inv_d, h.t, oct_inv, v);
                        cur = v.child_base, cur_bits = v.bits, tri_base = v.tri_base, l0 = v.leaf[0], l1 = v.leaf[1];
                        if ((cur_bits >> 8) == 0u && (l0 | l1) == 0u) {
                            pop8();
                        }
                    } else {
                        bvh4_visit(sc.nodes4, o, inv_d, h.t, st, cur, tos, size);
                    }
                    leave_blas();
                }
                RT_PROF_T(19)
            } else {
                if (at_leaf) {
                    const uint32_t word = WIDE == 8 ? bvh8_take_leaf(tri_base, l0, l1) : cur;
                    const int tri_start = int(word & BVH2_PRIM_INDEX_BITS), tri_end = int(tri_start + ((word & BVH2_PRIM_COUNT_BITS) >> 29) + 1);
                    const bool hit = intersect_tris_closest(o, d, tri_table(sc), tri_start, tri_end, int(mi_index), h);
                    res |= hit;
                    if (WIDE == 8) {
                        if ((l0 | l1) == 0u && (cur_bits >> 8) == 0u) {
                            pop8();
                        }
                    } else {
                        pop();
                    }
                    leave_blas();
                }
                RT_PROF_T(26)
            }
        }

        // ---- service part, D: finish rays whose TLAS walk is over, hand idle lanes their next rays
#ifdef RT_PROFILE_TRACE
        st_serv += 1, st_serv_lanes += uint32_t(__popcll(__ballot(lvl != BLAS))) - n_dead;
#endif
        {
            const bool in_fin = (lvl == TLAS) && (cur == BVH4_SENTINEL);
            if (in_fin) {
                // end of Traverse_TLAS_WithStack_ClosestHit: primitive index indirection (runs on misses too)
                if (h.prim_index < 0) {
                    h.prim_index = -int(sc.tri_indices[-h.prim_index - 1]) - 1;
                } else {
                    h.prim_index = int(sc.tri_indices[h.prim_index]);
                }
                bool again = false;
                if (res && !hit_side_is_solid(sc, h)) { // tail of the IntersectScene round (rare: the hit is not on a solid surface): what does it mean
                    const float4 cc = rays.c_cs[slot];
                    const uint2 xd = rays.xy_depth[slot];
                    Ray r;
                    r.c = {cc.x, cc.y, cc.z};
                    r.cone_spread = cc.w;
                    r.xy = xd.x, r.depth = xd.y;
                    const uint32_t xy_virtual = r.xy, layer = xy_layer(xy_virtual, layers);
                    TraceParams tpl = tp;
                    if (layer != 0) { // a later iteration of the batch: its own sample index / seed, keyed by the real pixel
                        tpl.iteration = tp.iteration + int(layer);
                        tpl.rand_seed = layer_rand_seed(tpl.iteration);
                        r.xy = xy_real(xy_virtual, layers, layer);
                    }
                    const uint32_t rand_hash = hash_combine(hash(r.xy), tpl.rand_seed);
                    uint32_t rand_dim = RAND_DIM_BASE_COUNT + get_total_depth(r.depth) * RAND_DIM_BOUNCE_COUNT;
                    const uint32_t depth_in = r.depth;
                    const f3 c_in = r.c;
                    again = closest_resolve_transparency(sc, tpl, r, h, t_val, rd, ro, rand_dim, rand_hash);
                    if (r.depth != depth_in || r.c.x != c_in.x || r.c.y != c_in.y || r.c.z != c_in.z) {
                        rays.c_cs[slot] = mkfloat4(r.c.x, r.c.y, r.c.z, r.cone_spread);
                        uint2 xo;
                        xo.x = xy_virtual, xo.y = r.depth;
                        rays.xy_depth[slot] = xo;
                    }
                }
                if (again) {
                    begin_round();
                } else {
                    const float4 o0 = rays.o_pdf[slot];
                    h.t += length(f3{o0.x, o0.y, o0.z} - ro);
                    store_hit(hits, slot, h);
                    lvl = IDLE;
                }
            }
            for (;;) {
                const unsigned long long idle_mask = __ballot(lvl == IDLE);
                if (idle_mask == 0ull) {
                    break;
                }
                if (pool_left == 0) {
                    int found = 0;
                    uint32_t next_chunk;
                    while (!found && walk.next(next_chunk)) { // (uniform)
                        uint32_t stripe, slot0, n_live;
                        found = __builtin_amdgcn_readfirstlane(int(queue.chunk(next_chunk, stripe, slot0, n_live)));
                        if (found) {
                            pool_slot = uint32_t(__builtin_amdgcn_readfirstlane(int(slot0)));
                            pool_left = uint32_t(__builtin_amdgcn_readfirstlane(int(n_live)));
                        }
                    }
                    if (!found) {
                        break;
                    }
                }
                const uint32_t rank = uint32_t(__popcll(idle_mask & ((1ull << lane) - 1ull)));
                const uint32_t n_take = min(uint32_t(__popcll(idle_mask)), pool_left);
                if (lvl == IDLE && rank < n_take) {
                    slot = pool_slot + rank;
                    const float4 a = rays.o_pdf[slot], b = rays.d_cw[slot];
                    const uint2 xd = rays.xy_depth[slot];
                    ro = {a.x, a.y, a.z};
                    rd = {b.x, b.y, b.z};
                    ray_flags = (1u << get_ray_type(xd.y));
                    h = init_hits ? make_hit() : load_hit(hits, slot);
                    begin_round();
                }
                pool_slot += n_take, pool_left -= n_take;
            }
            RT_PROF_T(25)
        }
        // whoever is idle now stays idle
        n_dead = uint32_t(__builtin_amdgcn_readfirstlane(__popcll(__ballot(lvl == IDLE))));
        if (__builtin_amdgcn_readfirstlane(int(n_dead == uint32_t(WAVE)))) {
            break; // nothing left in this wavefront and nothing left to fetch
        }

        // ---- service part, C: TLAS steps until every lane is inside an instance or through with its TLAS walk
        for (;;) {
            const bool in_c = (lvl == TLAS) && (cur != BVH4_SENTINEL);
            if (__builtin_amdgcn_readfirstlane(int(__ballot(in_c) == 0ull))) {
                break;
            }
#ifdef RT_PROFILE_TRACE
            st_tlas += 1;
#endif
            if (in_c) {
                if ((cur & BVH2_PRIM_COUNT_BITS) == 0) { // TLAS node (reference BVH2): near child first, far child pushed
                    const f3 inv = safe_invert(rd);
                    const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + cur);
                    const float4 d0 = np[0], d1 = np[1], d2 = np[2], links = np[3];
                    const uint32_t left_child = float_as_uint(links.x), right_child = float_as_uint(links.y);
                    const float ch0_min[3] = {d0.x, d0.z, d2.x}, ch0_max[3] = {d0.y, d0.w, d2.y};
                    const float ch1_min[3] = {d1.x, d1.z, d2.z}, ch1_max[3] = {d1.y, d1.w, d2.w};
                    float ch0_dist, ch1_dist;
                    const bool ch0_res = bbox_test(ro, inv, h.t, ch0_min, ch0_max, ch0_dist);
                    const bool ch1_res = bbox_test(ro, inv, h.t, ch1_min, ch1_max, ch1_dist);
                    if (!ch0_res && !ch1_res) {
                        pop();
                    } else if (ch0_res && ch1_res) {
                        const bool swap = ch1_dist < ch0_dist;
                        st.write_at(size++, tos);
                        tos = swap ? left_child : right_child;
                        cur = swap ? right_child : left_child;
                    } else {
                        cur = ch0_res ? left_child : right_child;
                    }
                } else { // TLAS leaf: one mesh instance
                    const uint32_t mi = (cur & BVH2_PRIM_INDEX_BITS);
                    const rayhip_mesh_instance &inst = sc.mesh_instances[mi];
                    if ((inst.ray_visibility & ray_flags) != 0) {
                        mi_index = mi;
                        o = transform_point(ro, inst.inv_xform);
                        d = transform_direction(rd, inst.inv_xform);
                        inv_d = safe_invert(d);
                        st.write_at(size++, tos); // the TLAS walk resumes from here
                        if (WIDE == 8) {
                            oct_inv = bvh8_oct_inv(inv_d);
                            st.write2_at(size, BVH8_SENTINEL, 0u); // (second sentinel: the read-ahead of a pop stays inside this level)
                            size += 2;
                            tos = BVH8_SENTINEL, tos_bits = 0u;
                            cur = sc.blas_root4[mi], cur_bits = (1u << (8u + oct_inv)) | 1u; // a virtual group holding the root in slot 0
                            l0 = l1 = 0u;
                        } else {
                            tos = BVH4_SENTINEL;
                            cur = sc.blas_root4[mi];
                        }
                        lvl = BLAS;
                        leave_blas(); // (a BLAS whose root is the sentinel: nothing to walk)
                    } else {
                        pop();
                    }
                }
            }
            RT_PROF_T(23)
        }
    }
#ifdef RT_PROFILE_TRACE
    RT_PROF_T(27)
    if (threadIdx.x < 32 && s_prof_acc[threadIdx.x] != 0) {
        atomicAdd(&g_prof_acc[threadIdx.x], s_prof_acc[threadIdx.x]);
    }
    if (lane == 0) {
        atomicAdd(&g_prof_acc[6], (unsigned long long)st_a), atomicAdd(&g_prof_acc[7], (unsigned long long)st_b);
        atomicAdd(&g_prof_acc[8], (unsigned long long)st_iter);
        atomicAdd(&g_prof_acc[9], (unsigned long long)st_serv), atomicAdd(&g_prof_acc[10], (unsigned long long)st_serv_lanes);
        atomicAdd(&g_prof_acc[11], (unsigned long long)st_tlas);
    }
#endif
}

// ---- K2, pooled form (round 4; the secondary bounces of one-instance scenes, rayhip.hip: RAYHIP_REFILL=4) --------------------------------
// What round 3's counters said about the refill kernel above: it issues vector instructions at ~90 % of what the ALU can take for its
// instruction mix -- with 35 of 64 lanes.  A third of a wavefront's lanes stand OUTSIDE the walk: a finished lane waits until
// RT_REFILL_MIN (40) lanes wait, because the service part (finish a ray, fetch the next one, top-level walk, instance transform: ~300
// instructions and four dependent memory round trips) costs the same whether it serves 4 lanes or 40.
//
// Here the expensive half of the service part always runs with FULL wavefronts, and a finished lane is back in the walk after a short
// swap.  Every wavefront keeps a POOL of prepared rays in LDS -- rays that are already through the top level and the instance
// transform (slot, object-space origin / direction / reciprocal direction, instance | ray type, BLAS root: 12 words, SoA, 64 entries):
//   * batch prepare (pool empty, a lane wants a ray): all 64 lanes take one ray of the next chunk each -- whatever ray they carry in
//     their registers stays there -- walk the top level with it up to its first visible instance, transform it and write it to the
//     pool; the scratch entries of that top-level walk live ABOVE the lane's own stack top;
//   * swap (a third kind of step inside the walk loop, taken when RT_POOL_SWAP_MIN lanes are through with their rays): those lanes
//     store their hit and read the next pool entry -- LDS reads and two stores, no dependent global load: the index indirection that
//     ends a ray (tri_indices[prim]) is requested the moment a leaf reports a hit and has long arrived;
//   * the walk itself, the slow finish of a ray (a hit on a non-solid surface: transparency round) and the top-level steps of rays
//     that need them are the refill kernel's, statement for statement -- per ray the same visits in the same order, so the same
//     hits bit for bit (test_gpu_parity.py::test_refill_kernel_is_bit_identical runs all forms).
// A pool entry stands for a ray whose top-level walk has nothing pending when it enters its first instance (the stack holds the two
// sentinels only): true for every ray of a single-instance scene.  A ray that enters its first instance with top-level nodes pending
// cannot hand that stack to another lane; its entry is marked UNPREPARED and the lane that takes it walks the top level itself, as in
// the refill kernel (the pooled form is therefore chosen per scene: rayhip.hip).  Rays that miss the top level altogether are
// finished by the batch (the miss record is stored, no pool entry).
// LDS per wavefront: RT_POOL_STACK_DEPTH x 256 B of stack + 3072 B of pool (deeper stacks spill to the HBM slab as in every traversal
// kernel; the headline scene's deepest walk uses 17 entries).
#ifndef RT_POOL_SWAP_MIN
#define RT_POOL_SWAP_MIN 8
#endif
#ifndef RT_POOL_SLOW_MIN
#define RT_POOL_SLOW_MIN 32 // lanes waiting before the wavefront leaves the walk for the full service part when a swap cannot serve them
#endif
#ifndef RT_POOL_STACK_DEPTH
#define RT_POOL_STACK_DEPTH 14
#endif
#ifndef RT_POOL_MIN_WAVES
#define RT_POOL_MIN_WAVES 6
#endif
#ifndef RT_POOL_PREFETCH_INDEX
#define RT_POOL_PREFETCH_INDEX 1
#endif
constexpr int POOL_STACK_DEPTH = RT_POOL_STACK_DEPTH;
constexpr int POOL_FIELDS = 12;                   // slot | o.xyz | d.xyz | 1/d.xyz | instance + (ray type << 24) | BLAS root
constexpr uint32_t POOL_UNPREPARED = 0xfffffffeu; // in the root field: the taker walks the top level itself (never a node word)
// per-wave slab in HBM: the spill part of the stack + 8 x 64 words of slow-path lane state (see `slow` in the kernel)
constexpr int POOL_SLAB_WORDS = (STACK_TOTAL_DEPTH - POOL_STACK_DEPTH + 8) * WAVE;
template <int MIN_WAIT = RT_POOL_SWAP_MIN>
__global__ void __launch_bounds__(WAVE, RT_POOL_MIN_WAVES) k_trace_closest_pool(const SceneView sc, const TraceParams tp, const RaySoA rays,
                                                                                const HitSoA hits, const RayQueue queue, const int init_hits,
                                                                                uint32_t *__restrict__ stack_spill, const Layering layers) {
    __shared__ uint32_t lds_stack[POOL_STACK_DEPTH * WAVE];
    __shared__ uint32_t lds_pool[POOL_FIELDS * WAVE];
    const uint32_t lane = threadIdx.x;
#ifdef RT_PROFILE_TRACE
    if (threadIdx.x < 32) {
        s_prof_acc[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) {
        s_prof_last = __builtin_readcyclecounter();
    }
    uint32_t st_a = 0, st_b = 0, st_iter = 0, st_serv = 0, st_serv_lanes = 0, st_tlas = 0, st_prep = 0, st_prep_lanes = 0, st_swap = 0, st_swap_lanes = 0; // (uniform)
#endif
    LdsStackT<POOL_STACK_DEPTH> st;
    st.lane_base = &lds_stack[lane];
    st.spill_base = stack_spill + size_t(blockIdx.x) * size_t(POOL_SLAB_WORDS) + lane;
    st.size = 0;

    enum : uint32_t { IDLE = 0, TLAS = 1, BLAS = 2, FIN = 3 }; // FIN: through, index resolved, the hit is on a non-solid surface -> the slow finish
    // lane state (as in the refill kernel); `world`: ro / rd hold the ray's world-space origin and direction (a ray taken from the
    // pool arrives in object space and fetches them only if it meets a non-solid surface).  RT_POOL_PREFETCH_INDEX: once a leaf has
    // reported a hit (`res`), h.prim_index is tri_indices[] of the closest hit -- requested by the leaf step that found it, into the
    // register the record keeps anyway -- and `back` says that the hit was on the back face (the sign the raw index carries)
    uint32_t lvl = IDLE, slot = 0, cur = BVH4_SENTINEL, tos = BVH4_SENTINEL, size = 0, mi_index = 0;
    bool res = false, world = false, back = false;
    f3 o = {0.0f, 0.0f, 0.0f}, d = {0.0f, 0.0f, 1.0f}, inv_d = d;
    Hit h = make_hit();
    // what only the slow paths need -- the world-space origin of the current transparency segment, the direction, the hit distance the
    // round started with -- lives in this lane's column of a per-wave slab behind the stack spill area, not in registers (the walk loop
    // has 80 of them): slow[k * WAVE], k = 0..2 ro, 3..5 rd, 6 t_val; valid while `world` is set
    float *const slow = reinterpret_cast<float *>(stack_spill + size_t(blockIdx.x) * size_t(POOL_SLAB_WORDS) + size_t((STACK_TOTAL_DEPTH - POOL_STACK_DEPTH) * WAVE)) + lane;
    auto slow_ro = [&]() { return f3{slow[0 * WAVE], slow[1 * WAVE], slow[2 * WAVE]}; };
    auto slow_rd = [&]() { return f3{slow[3 * WAVE], slow[4 * WAVE], slow[5 * WAVE]}; };
    // wavefront state (uniform)
    uint32_t pool_head = 0, pool_n = 0;
    bool exhausted = false;
    ChunkWalk walk(queue.live_chunks());

    auto begin_round = [&]() { // IntersectScene loop head + walk prologue at TLAS level (a lane that walks the top level is a `world` lane)
        slow[6 * WAVE] = h.t;
        res = false;
        size = 0;
        st.write_at(size++, BVH4_SENTINEL);
        tos = BVH4_SENTINEL;
        cur = tp.root_index;
        lvl = TLAS;
    };
    auto pop = [&]() {
        cur = tos;
        tos = st.read_at(--size);
    };
    auto leave_blas = [&]() {
        if (lvl == BLAS && cur == BVH4_SENTINEL) {
            lvl = TLAS;
            pop();
        }
    };
    // end of Traverse_TLAS_WithStack_ClosestHit: primitive index indirection (runs on misses too, on whatever index the record holds)
    auto resolve_prim_index = [&]() {
        if (RT_POOL_PREFETCH_INDEX && res) {
            h.prim_index = back ? -h.prim_index - 1 : h.prim_index;
        } else if (h.prim_index < 0) {
            h.prim_index = -int(sc.tri_indices[-h.prim_index - 1]) - 1;
        } else {
            h.prim_index = int(sc.tri_indices[h.prim_index]);
        }
    };
    auto store_final_hit = [&]() {
        if (world) { // (a ray that never left object space has ro == its origin: the reference's  t += length(o - ro)  adds 0)
            const float4 o0 = rays.o_pdf[slot];
            h.t += length(f3{o0.x, o0.y, o0.z} - slow_ro());
        }
        store_hit(hits, slot, h);
        lvl = IDLE;
    };
    // idle lanes take the next entries of the pool (uniform: idle_mask, pool_head, pool_n)
    auto take_from_pool = [&](const unsigned long long idle_mask) {
        const uint32_t rank = uint32_t(__popcll(idle_mask & ((1ull << lane) - 1ull)));
        const uint32_t n_take = min(uint32_t(__popcll(idle_mask)), pool_n);
        if (lvl == IDLE && rank < n_take) {
            const uint32_t e = pool_head + rank;
            slot = lds_pool[0 * WAVE + e];
            const uint32_t root = lds_pool[11 * WAVE + e];
            h = init_hits ? make_hit() : load_hit(hits, slot);
            if (root == POOL_UNPREPARED) { // top-level nodes were pending at its first instance: this lane walks the top level itself
                const float4 a = rays.o_pdf[slot], b = rays.d_cw[slot];
                slow[0 * WAVE] = a.x, slow[1 * WAVE] = a.y, slow[2 * WAVE] = a.z;
                slow[3 * WAVE] = b.x, slow[4 * WAVE] = b.y, slow[5 * WAVE] = b.z;
                world = true;
                begin_round();
            } else {
                o = {uint_as_float(lds_pool[1 * WAVE + e]), uint_as_float(lds_pool[2 * WAVE + e]), uint_as_float(lds_pool[3 * WAVE + e])};
                d = {uint_as_float(lds_pool[4 * WAVE + e]), uint_as_float(lds_pool[5 * WAVE + e]), uint_as_float(lds_pool[6 * WAVE + e])};
                inv_d = {uint_as_float(lds_pool[7 * WAVE + e]), uint_as_float(lds_pool[8 * WAVE + e]), uint_as_float(lds_pool[9 * WAVE + e])};
                mi_index = lds_pool[10 * WAVE + e] & 0xffffffu; // (the ray type in the upper bits was the batch's business)
                world = false;
                // begin_round + the instance entry of part C: two sentinels, nothing pending at the top level
                res = false;
                size = 0;
                st.write_at(size++, BVH4_SENTINEL);
                st.write_at(size++, BVH4_SENTINEL);
                tos = BVH4_SENTINEL;
                cur = root;
                lvl = BLAS;
                leave_blas(); // (a BLAS whose root is the sentinel: nothing to walk)
            }
        }
        pool_head += n_take, pool_n -= n_take;
    };

    uint32_t n_dead = 0; // idle lanes that can no longer be refilled (uniform; only ever non-zero once the pool is empty for good)
    for (;;) {
        // ---- the walk: majority-scheduled node / leaf steps, and swaps
        for (;;) {
            const bool in_blas = (lvl == BLAS);
            const bool at_leaf = in_blas && (cur & BVH2_PRIM_COUNT_BITS) != 0;
            const bool at_node = in_blas && !at_leaf;
            const int n_node = __popcll(__ballot(at_node)), n_leaf = __popcll(__ballot(at_leaf));
            const int n_out = WAVE - n_node - n_leaf - int(n_dead);
#ifdef RT_PROFILE_TRACE
            st_a += n_node, st_b += n_leaf, st_iter += 1;
#endif
            if (n_node + n_leaf == 0) {
                break;
            }
            if (n_out >= MIN_WAIT) {
                // lanes a swap can serve: through with their ray or idle
                const bool through = (lvl == TLAS) && (cur == BVH4_SENTINEL);
                const unsigned long long swap_mask = __ballot(through || lvl == IDLE);
                if (pool_n != 0u && __popcll(swap_mask) >= MIN_WAIT) {
#ifdef RT_PROFILE_TRACE
                    st_swap += 1, st_swap_lanes += uint32_t(__popcll(swap_mask));
#endif
                    if (through) {
                        resolve_prim_index();
                        if (!res || hit_side_is_solid(sc, h)) {
                            store_final_hit();
                        } else {
                            lvl = FIN; // (rare: the transparency round is the full service part's)
                        }
                    }
                    take_from_pool(__ballot(lvl == IDLE));
                    RT_PROF_T(25)
                    continue;
                }
                if (n_out >= RT_POOL_SLOW_MIN || (pool_n == 0u && !exhausted)) {
                    break; // the full service part: refill the pool / serve the lanes a swap cannot
                }
            }
            if (n_node * RT_REFILL_VOTE_DEN >= n_leaf * RT_REFILL_VOTE_NUM) {
                if (at_node) {
                    bvh4_visit(sc.nodes4, o, inv_d, h.t, st, cur, tos, size);
                    leave_blas();
                }
                RT_PROF_T(19)
            } else {
                if (at_leaf) {
                    const int tri_start = int(cur & BVH2_PRIM_INDEX_BITS), tri_end = int(tri_start + ((cur & BVH2_PRIM_COUNT_BITS) >> 29) + 1);
                    const bool hit = intersect_tris_closest(o, d, tri_table(sc), tri_start, tri_end, int(mi_index), h);
                    if (RT_POOL_PREFETCH_INDEX && hit) { // the index indirection of this hit, should it stay the closest: in flight from here on
                        back = h.prim_index < 0;
                        h.prim_index = int(sc.tri_indices[back ? -h.prim_index - 1 : h.prim_index]);
                    }
                    res |= hit;
                    pop();
                    leave_blas();
                }
                RT_PROF_T(26)
            }
        }

        // ---- full service, D: finish rays whose top-level walk is over (the refill kernel's part D)
#ifdef RT_PROFILE_TRACE
        st_serv += 1, st_serv_lanes += uint32_t(__popcll(__ballot(lvl != BLAS))) - n_dead;
#endif
        if (((lvl == TLAS) && (cur == BVH4_SENTINEL)) || lvl == FIN) {
            if (lvl != FIN) {
                resolve_prim_index();
            }
            bool again = false;
            if (res && !hit_side_is_solid(sc, h)) { // tail of the IntersectScene round (rare)
                f3 ro, rd;
                float t_val;
                if (!world) { // a ray from the pool: still at its origin, the round started with the distance it arrived with
                    const float4 a = rays.o_pdf[slot], b = rays.d_cw[slot];
                    ro = {a.x, a.y, a.z};
                    rd = {b.x, b.y, b.z};
                    t_val = init_hits ? MAX_DIST : hits.oi_pi_t_u[slot].z; // (the record in memory is still the one the ray came with)
                } else {
                    ro = slow_ro(), rd = slow_rd();
                    t_val = slow[6 * WAVE];
                }
                const float4 cc = rays.c_cs[slot];
                const uint2 xd = rays.xy_depth[slot];
                Ray r;
                r.c = {cc.x, cc.y, cc.z};
                r.cone_spread = cc.w;
                r.xy = xd.x, r.depth = xd.y;
                const uint32_t xy_virtual = r.xy, layer = xy_layer(xy_virtual, layers);
                TraceParams tpl = tp;
                if (layer != 0) { // a later iteration of the batch: its own sample index / seed, keyed by the real pixel
                    tpl.iteration = tp.iteration + int(layer);
                    tpl.rand_seed = layer_rand_seed(tpl.iteration);
                    r.xy = xy_real(xy_virtual, layers, layer);
                }
                const uint32_t rand_hash = hash_combine(hash(r.xy), tpl.rand_seed);
                uint32_t rand_dim = RAND_DIM_BASE_COUNT + get_total_depth(r.depth) * RAND_DIM_BOUNCE_COUNT;
                const uint32_t depth_in = r.depth;
                const f3 c_in = r.c;
                again = closest_resolve_transparency(sc, tpl, r, h, t_val, rd, ro, rand_dim, rand_hash);
                if (r.depth != depth_in || r.c.x != c_in.x || r.c.y != c_in.y || r.c.z != c_in.z) {
                    rays.c_cs[slot] = mkfloat4(r.c.x, r.c.y, r.c.z, r.cone_spread);
                    uint2 xo;
                    xo.x = xy_virtual, xo.y = r.depth;
                    rays.xy_depth[slot] = xo;
                }
                if (again) { // the next round starts from the advanced origin: this lane walks the top level itself from here on
                    slow[0 * WAVE] = ro.x, slow[1 * WAVE] = ro.y, slow[2 * WAVE] = ro.z;
                    slow[3 * WAVE] = rd.x, slow[4 * WAVE] = rd.y, slow[5 * WAVE] = rd.z;
                    world = true;
                }
            }
            if (again) {
                begin_round();
            } else {
                store_final_hit();
            }
        }
        // ---- full service: idle lanes take the next pool entries; an empty pool is refilled by the whole wavefront
        for (;;) {
            const unsigned long long idle_mask = __ballot(lvl == IDLE);
            if (idle_mask == 0ull) {
                break;
            }
            if (__builtin_amdgcn_readfirstlane(int(pool_n)) == 0) {
                if (exhausted) {
                    break;
                }
                // -- batch prepare: one fresh ray per lane, through the top level, into the pool
                int found = 0;
                uint32_t chunk_slot0 = 0, chunk_live = 0;
                {
                    uint32_t next_chunk;
                    while (!found && walk.next(next_chunk)) { // (uniform)
                        uint32_t stripe, slot0, n_live;
                        found = __builtin_amdgcn_readfirstlane(int(queue.chunk(next_chunk, stripe, slot0, n_live)));
                        if (found) {
                            chunk_slot0 = uint32_t(__builtin_amdgcn_readfirstlane(int(slot0)));
                            chunk_live = uint32_t(__builtin_amdgcn_readfirstlane(int(n_live)));
                        }
                    }
                }
                if (!found) {
                    exhausted = true;
                    break;
                }
                const bool mine = lane < chunk_live;
                const uint32_t ps = chunk_slot0 + lane;
                uint32_t p_root = POOL_UNPREPARED, p_word = 0;
                f3 po = {0.0f, 0.0f, 0.0f}, pd = {0.0f, 0.0f, 1.0f};
                bool entered = false;
                if (mine) {
                    const float4 a = rays.o_pdf[ps], b = rays.d_cw[ps];
                    const uint32_t p_type = get_ray_type(rays.xy_depth[ps].y);
                    const f3 pro = {a.x, a.y, a.z}, prd = {b.x, b.y, b.z};
                    const float p_t = init_hits ? MAX_DIST : hits.oi_pi_t_u[ps].z;
                    const f3 inv = safe_invert(prd);
                    // the top-level walk of the refill kernel's part C, on scratch entries above this lane's own stack top
                    uint32_t p_size = size, p_cur = tp.root_index, p_tos = BVH4_SENTINEL;
                    st.write_at(p_size++, BVH4_SENTINEL);
                    while (p_cur != BVH4_SENTINEL && !entered) {
                        if ((p_cur & BVH2_PRIM_COUNT_BITS) == 0) {
                            const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + p_cur);
                            const float4 d0 = np[0], d1 = np[1], d2 = np[2], links = np[3];
                            const uint32_t left_child = float_as_uint(links.x), right_child = float_as_uint(links.y);
                            const float ch0_min[3] = {d0.x, d0.z, d2.x}, ch0_max[3] = {d0.y, d0.w, d2.y};
                            const float ch1_min[3] = {d1.x, d1.z, d2.z}, ch1_max[3] = {d1.y, d1.w, d2.w};
                            float ch0_dist, ch1_dist;
                            const bool ch0_res = bbox_test(pro, inv, p_t, ch0_min, ch0_max, ch0_dist);
                            const bool ch1_res = bbox_test(pro, inv, p_t, ch1_min, ch1_max, ch1_dist);
                            if (!ch0_res && !ch1_res) {
                                p_cur = p_tos;
                                p_tos = st.read_at(--p_size);
                            } else if (ch0_res && ch1_res) {
                                const bool swap = ch1_dist < ch0_dist;
                                st.write_at(p_size++, p_tos);
                                p_tos = swap ? left_child : right_child;
                                p_cur = swap ? right_child : left_child;
                            } else {
                                p_cur = ch0_res ? left_child : right_child;
                            }
                        } else {
                            const uint32_t mi = (p_cur & BVH2_PRIM_INDEX_BITS);
                            const rayhip_mesh_instance &inst = sc.mesh_instances[mi];
                            if ((inst.ray_visibility & (1u << p_type)) != 0) {
                                entered = true;
                                if (p_tos == BVH4_SENTINEL) { // nothing pending at the top level: the ray can change lanes
                                    po = transform_point(pro, inst.inv_xform);
                                    pd = transform_direction(prd, inst.inv_xform);
                                    p_word = mi | (p_type << 24);
                                    p_root = sc.blas_root4[mi];
                                }
                            } else {
                                p_cur = p_tos;
                                p_tos = st.read_at(--p_size);
                            }
                        }
                    }
                    if (!entered) { // the ray misses the top level: finished here (the refill kernel's part D on a ray without a hit)
                        Hit hm = init_hits ? make_hit() : load_hit(hits, ps);
                        if (hm.prim_index < 0) {
                            hm.prim_index = -int(sc.tri_indices[-hm.prim_index - 1]) - 1;
                        } else {
                            hm.prim_index = int(sc.tri_indices[hm.prim_index]);
                        }
                        store_hit(hits, ps, hm);
                    }
                }
#ifdef RT_PROFILE_TRACE
                st_prep += 1, st_prep_lanes += chunk_live;
#endif
                const unsigned long long ent_mask = __ballot(entered);
                if (entered) {
                    const uint32_t e = uint32_t(__popcll(ent_mask & ((1ull << lane) - 1ull)));
                    const f3 pinv = safe_invert(pd);
                    lds_pool[0 * WAVE + e] = ps;
                    lds_pool[1 * WAVE + e] = float_as_uint(po.x), lds_pool[2 * WAVE + e] = float_as_uint(po.y), lds_pool[3 * WAVE + e] = float_as_uint(po.z);
                    lds_pool[4 * WAVE + e] = float_as_uint(pd.x), lds_pool[5 * WAVE + e] = float_as_uint(pd.y), lds_pool[6 * WAVE + e] = float_as_uint(pd.z);
                    lds_pool[7 * WAVE + e] = float_as_uint(pinv.x), lds_pool[8 * WAVE + e] = float_as_uint(pinv.y), lds_pool[9 * WAVE + e] = float_as_uint(pinv.z);
                    lds_pool[10 * WAVE + e] = p_word;
                    lds_pool[11 * WAVE + e] = p_root;
                }
                __syncthreads(); // (one wavefront per block: orders the pool writes before the reads of other lanes)
                pool_head = 0;
                pool_n = uint32_t(__popcll(ent_mask));
                RT_PROF_T(23)
                continue;
            }
            take_from_pool(idle_mask);
        }
        RT_PROF_T(20)
        // whoever is idle now stays idle (the pool is empty and the queue exhausted)
        n_dead = uint32_t(__builtin_amdgcn_readfirstlane(__popcll(__ballot(lvl == IDLE))));
        if (__builtin_amdgcn_readfirstlane(int(n_dead == uint32_t(WAVE)))) {
            break;
        }

        // ---- top-level steps of the lanes that walk it themselves (the refill kernel's part C)
        for (;;) {
            const bool in_c = (lvl == TLAS) && (cur != BVH4_SENTINEL);
            if (__builtin_amdgcn_readfirstlane(int(__ballot(in_c) == 0ull))) {
                break;
            }
#ifdef RT_PROFILE_TRACE
            st_tlas += 1;
#endif
            if (in_c) {
                const f3 ro = slow_ro(), rd = slow_rd();
                if ((cur & BVH2_PRIM_COUNT_BITS) == 0) { // TLAS node (reference BVH2): near child first, far child pushed
                    const f3 inv = safe_invert(rd);
                    const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + cur);
                    const float4 d0 = np[0], d1 = np[1], d2 = np[2], links = np[3];
                    const uint32_t left_child = float_as_uint(links.x), right_child = float_as_uint(links.y);
                    const float ch0_min[3] = {d0.x, d0.z, d2.x}, ch0_max[3] = {d0.y, d0.w, d2.y};
                    const float ch1_min[3] = {d1.x, d1.z, d2.z}, ch1_max[3] = {d1.y, d1.w, d2.w};
                    float ch0_dist, ch1_dist;
                    const bool ch0_res = bbox_test(ro, inv, h.t, ch0_min, ch0_max, ch0_dist);
                    const bool ch1_res = bbox_test(ro, inv, h.t, ch1_min, ch1_max, ch1_dist);
                    if (!ch0_res && !ch1_res) {
                        pop();
                    } else if (ch0_res && ch1_res) {
                        const bool swap = ch1_dist < ch0_dist;
                        st.write_at(size++, tos);
                        tos = swap ? left_child : right_child;
                        cur = swap ? right_child : left_child;
                    } else {
                        cur = ch0_res ? left_child : right_child;
                    }
                } else { // TLAS leaf: one mesh instance
                    const uint32_t mi = (cur & BVH2_PRIM_INDEX_BITS);
                    const rayhip_mesh_instance &inst = sc.mesh_instances[mi];
                    if ((inst.ray_visibility & (1u << get_ray_type(rays.xy_depth[slot].y))) != 0) { // (slow path: the ray type is re-read, not carried)
                        mi_index = mi;
                        o = transform_point(ro, inst.inv_xform);
                        d = transform_direction(rd, inst.inv_xform);
                        inv_d = safe_invert(d);
                        st.write_at(size++, tos); // the TLAS walk resumes from here
                        tos = BVH4_SENTINEL;
                        cur = sc.blas_root4[mi];
                        lvl = BLAS;
                        leave_blas();
                    } else {
                        pop();
                    }
                }
            }
            RT_PROF_T(24)
        }
    }
#ifdef RT_PROFILE_TRACE
    RT_PROF_T(27)
    if (threadIdx.x < 32 && s_prof_acc[threadIdx.x] != 0) {
        atomicAdd(&g_prof_acc[threadIdx.x], s_prof_acc[threadIdx.x]);
    }
    if (lane == 0) {
        atomicAdd(&g_prof_acc[6], (unsigned long long)st_a), atomicAdd(&g_prof_acc[7], (unsigned long long)st_b);
        atomicAdd(&g_prof_acc[8], (unsigned long long)st_iter);
        atomicAdd(&g_prof_acc[9], (unsigned long long)st_serv), atomicAdd(&g_prof_acc[10], (unsigned long long)st_serv_lanes);
        atomicAdd(&g_prof_acc[11], (unsigned long long)st_tlas);
        atomicAdd(&g_prof_acc[12], (unsigned long long)st_prep), atomicAdd(&g_prof_acc[13], (unsigned long long)st_prep_lanes);
        atomicAdd(&g_prof_acc[14], (unsigned long long)st_swap), atomicAdd(&g_prof_acc[15], (unsigned long long)st_swap_lanes);
    }
#endif
}

// ---- K3 ---------------------------------------------------------------------------------------------------
template <bool COUNT, int WIDE, int MINW = RT_TRACE_MIN_WAVES>
__global__ void __launch_bounds__(WAVE, MINW) k_trace_shadow(const SceneView sc, const TraceParams tp, const ShadowSoA shadow,
                                                      const RayQueue queue, const float limit,
                                                      const int img_w, float4 *__restrict__ temp_buf,
                                                      float4 *__restrict__ out_rc, /* test hook, may be null */
                                                      uint32_t *__restrict__ stack_spill,
                                                      unsigned long long *__restrict__ counters, const Layering layers) {
    __shared__ uint32_t lds_stack[LDS_STACK_DEPTH * WAVE];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_live_chunks = queue.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!queue.chunk(c, stripe, slot0, n_live) || lane >= n_live) {
            continue;
        }
        const uint32_t i = slot0 + lane;
        const ShadowRay r = load_shadow(shadow, i);
        LdsStack st;
        st.lane_base = &lds_stack[lane];
        st.spill_base = stack_spill + size_t(blockIdx.x) * size_t(STACK_SPILL_DEPTH * WAVE) + lane;
        st.size = 0;
        TravCount tc = {0, 0, 0, 0, 0};
        f3 rc;
        const uint32_t layer = xy_layer(r.xy, layers);
        TraceParams tpl = tp;
        tpl.iteration = tp.iteration + int(layer);
        tpl.rand_seed = layer == 0 ? tp.rand_seed : layer_rand_seed(tpl.iteration);
        ShadowRay rl = r;
        rl.xy = xy_real(r.xy, layers, layer);
        rc = intersect_scene_shadow<WIDE>(sc, tpl, rl, st, COUNT ? &tc : nullptr);
        if (out_rc) {
            out_rc[i] = mkfloat4(rc.x, rc.y, rc.z, 0.0f);
        } else {
            add_shadow_pixel(rc, limit, r.xy, img_w, temp_buf);
        }
        if (COUNT) {
            flush_trav_count(counters, tc);
        }
    }
}

// ---- K3, persistent form: the any-hit twin of k_trace_closest_refill ---------------------------------------------------------------------
// Why: the nested walks of k_trace_shadow (top level -> instance -> leaf, each a loop with its own live state) spill in the loop at any
// register budget that keeps five or six wavefronts per SIMD (176-336 bytes of scratch per lane) -- round 3's counters showed 1.29 GB
// written per pass by a kernel whose results are 0.28 GB of pixel updates -- and a wavefront is as slow as its longest ray.  Here the
// walk is the flat state machine of the closest-hit kernel (one loop; BLAS part / finish + refill / top-level steps), with the any-hit
// rules of Ref::IntersectScene(shadow_ray_t) (CoreRef.cpp:3160-3262) and Traverse_*_AnyHit (:2193-2280, :2619-2693):
//   * a leaf step is intersect_tris_any; a hit on a solid side ends the RAY (throughput 0), any other hit only shortens it;
//   * a ray whose top-level walk ends without a solid hit but with a hit crosses that surface: its throughput is multiplied by the
//     surface's transparency and it starts again behind it (the throughput waits in the ray's own c_xy record meanwhile -- nobody
//     reads a shadow ray after this kernel);
//   * per ray the same visits in the same order as the nested form: the same bits (test_gpu_parity.py: the frames of both forms).
#ifndef RT_SHADOW_REFILL_MIN
#define RT_SHADOW_REFILL_MIN 40
#endif
#ifndef RT_SHADOW_REFILL_MIN_WAVES
#define RT_SHADOW_REFILL_MIN_WAVES 6
#endif
template <int MIN_WAIT = RT_SHADOW_REFILL_MIN>
__global__ void __launch_bounds__(WAVE, RT_SHADOW_REFILL_MIN_WAVES) k_trace_shadow_refill(const SceneView sc, const TraceParams tp, const ShadowSoA shadow,
                                                                                     const RayQueue queue, const float limit, const int img_w,
                                                                                     float4 *__restrict__ temp_buf, float4 *__restrict__ out_rc,
                                                                                     uint32_t *__restrict__ stack_spill, const Layering layers) {
    __shared__ uint32_t lds_stack[LDS_STACK_DEPTH * WAVE];
    const uint32_t lane = threadIdx.x;
    LdsStack st;
    st.lane_base = &lds_stack[lane];
    st.spill_base = stack_spill + size_t(blockIdx.x) * size_t(STACK_SPILL_DEPTH * WAVE) + lane;
    st.size = 0;

    enum : uint32_t { IDLE = 0, TLAS = 1, BLAS = 2 };
    // lane state: where the walk stands, the segment being walked (world space: ro, rd, dist left; object space of the instance: o, d,
    // 1 / d), the nearest non-solid hit of this segment, how many surfaces the ray has crossed
    uint32_t lvl = IDLE, slot = 0, cur = BVH4_SENTINEL, tos = BVH4_SENTINEL, size = 0, mi_index = 0, crossed = 0;
    bool solid = false;
    f3 ro = {0.0f, 0.0f, 0.0f}, rd = {0.0f, 0.0f, 1.0f}, o = ro, d = rd, inv_d = rd;
    float dist = 0.0f;
    Hit h = make_hit();
    uint32_t pool_slot = 0, pool_left = 0; // (uniform) the chunk being handed out
    ChunkWalk walk(queue.live_chunks());

    auto begin_segment = [&]() { // loop head of IntersectScene + prologue of the top-level walk
        h = make_hit();
        h.t = dist;
        size = 0;
        st.write_at(size++, BVH4_SENTINEL);
        tos = BVH4_SENTINEL;
        cur = tp.root_index;
        lvl = TLAS;
    };
    auto pop = [&]() {
        cur = tos;
        tos = st.read_at(--size);
    };
    auto leave_blas = [&]() { // the pop that ends a BLAS walk hands back the sentinel and restores the top-level `tos`
        if (lvl == BLAS && cur == BVH4_SENTINEL) {
            lvl = TLAS;
            pop();
        }
    };
    auto deliver = [&](const f3 rc, const uint32_t xy_virtual) {
        if (out_rc) {
            out_rc[slot] = mkfloat4(rc.x, rc.y, rc.z, 0.0f);
        } else {
            add_shadow_pixel(rc, limit, xy_virtual, img_w, temp_buf);
        }
    };

    uint32_t n_dead = 0; // idle lanes that can no longer be refilled (uniform)
    for (;;) {
        // ---- BLAS part: majority-scheduled node / leaf steps over the lanes inside an instance
        for (;;) {
            const bool in_blas = (lvl == BLAS);
            const bool at_leaf = in_blas && (cur & BVH2_PRIM_COUNT_BITS) != 0;
            const bool at_node = in_blas && !at_leaf;
            const int n_node = __popcll(__ballot(at_node)), n_leaf = __popcll(__ballot(at_leaf));
            const int n_out = WAVE - n_node - n_leaf - int(n_dead);
            if (n_node + n_leaf == 0 || n_out >= MIN_WAIT) {
                break;
            }
            if (n_node >= n_leaf) {
                if (at_node) {
                    bvh4_visit(sc.nodes4, o, inv_d, h.t, st, cur, tos, size);
                    leave_blas();
                }
            } else if (at_leaf) {
                const int tri_start = int(cur & BVH2_PRIM_INDEX_BITS), tri_end = int(tri_start + ((cur & BVH2_PRIM_COUNT_BITS) >> 29) + 1);
                bool stop = false;
                if (intersect_tris_any(o, d, tri_table(sc), sc.tri_materials, sc.tri_indices, tri_start, tri_end, int(mi_index), h)) {
                    // (blas leaf of traverse_any: the side that was hit)
                    const bool is_backfacing = h.prim_index < 0;
                    const uint32_t prim = is_backfacing ? uint32_t(-h.prim_index - 1) : uint32_t(h.prim_index);
                    if (sc.all_solid != 0u) {
                        stop = true;
                    } else {
                        const rayhip_tri_mat_data md = sc.tri_materials[sc.tri_indices[prim]];
                        stop = (!is_backfacing && (md.front_mi & MATERIAL_SOLID_BIT)) || (is_backfacing && (md.back_mi & MATERIAL_SOLID_BIT));
                    }
                }
                if (stop) { // a solid occluder: the ray is over
                    solid = true;
                    lvl = TLAS;
                    cur = BVH4_SENTINEL;
                } else {
                    pop();
                    leave_blas();
                }
            }
        }

        // ---- service part, D: rays whose segment is through; idle lanes take their next rays
        {
            const bool in_fin = (lvl == TLAS) && (cur == BVH4_SENTINEL);
            if (in_fin) {
                const uint32_t depth_word = float_as_uint(shadow.o_depth[slot].w);
                const float4 cx = shadow.c_xy[slot];
                const uint32_t xy_virtual = float_as_uint(cx.w);
                f3 rc = {cx.x, cx.y, cx.z};
                const bool over = (get_transp_depth(depth_word) + int(crossed)) > tp.max_transp_depth;
                if (solid || over) {
                    rc = {0.0f, 0.0f, 0.0f};
                }
                bool again = false;
                if (!solid && !over && h.v >= 0.0f) { // the segment ended on a surface that lets light through (rare)
                    if (h.prim_index < 0) { // (tail of traverse_any: the index indirection)
                        h.prim_index = -int(sc.tri_indices[-h.prim_index - 1]) - 1;
                    } else {
                        h.prim_index = int(sc.tri_indices[h.prim_index]);
                    }
                    const uint32_t layer = xy_layer(xy_virtual, layers);
                    TraceParams tpl = tp;
                    tpl.iteration = tp.iteration + int(layer);
                    tpl.rand_seed = layer == 0 ? tp.rand_seed : layer_rand_seed(tpl.iteration);
                    const uint32_t rand_hash = hash_combine(hash(xy_real(xy_virtual, layers, layer)), tpl.rand_seed);
                    const uint32_t rand_dim = RAND_DIM_BASE_COUNT + (get_total_depth(depth_word) + crossed) * RAND_DIM_BOUNCE_COUNT;
                    rc *= shadow_surface_throughput(sc, tpl, h, rand_dim, rand_hash);
                    if (!(lum(rc) < FLT_EPS_)) {
                        const float t = h.t + HIT_BIAS;
                        ro += rd * t;
                        dist -= t;
                        ++crossed;
                        again = dist > HIT_BIAS;
                    }
                    if (again) {
                        shadow.c_xy[slot] = mkfloat4(rc.x, rc.y, rc.z, cx.w);
                    }
                }
                if (again) {
                    begin_segment();
                } else {
                    deliver(rc, xy_virtual);
                    lvl = IDLE;
                }
            }
            for (;;) {
                const unsigned long long idle_mask = __ballot(lvl == IDLE);
                if (idle_mask == 0ull) {
                    break;
                }
                if (pool_left == 0) {
                    int found = 0;
                    uint32_t next_chunk;
                    while (!found && walk.next(next_chunk)) { // (uniform)
                        uint32_t stripe, slot0, n_live;
                        found = __builtin_amdgcn_readfirstlane(int(queue.chunk(next_chunk, stripe, slot0, n_live)));
                        if (found) {
                            pool_slot = uint32_t(__builtin_amdgcn_readfirstlane(int(slot0)));
                            pool_left = uint32_t(__builtin_amdgcn_readfirstlane(int(n_live)));
                        }
                    }
                    if (!found) {
                        break;
                    }
                }
                const uint32_t rank = uint32_t(__popcll(idle_mask & ((1ull << lane) - 1ull)));
                const uint32_t n_take = min(uint32_t(__popcll(idle_mask)), pool_left);
                if (lvl == IDLE && rank < n_take) {
                    slot = pool_slot + rank;
                    const float4 a = shadow.o_depth[slot], b = shadow.d_dist[slot];
                    ro = {a.x, a.y, a.z};
                    rd = {b.x, b.y, b.z};
                    dist = b.w > 0.0f ? b.w : MAX_DIST;
                    crossed = 0, solid = false;
                    if (dist > HIT_BIAS) {
                        begin_segment();
                    } else { // (no segment to walk: the throughput arrives as it is; the lane stays idle and is served again)
                        const float4 cx = shadow.c_xy[slot];
                        deliver(f3{cx.x, cx.y, cx.z}, float_as_uint(cx.w));
                    }
                }
                pool_slot += n_take, pool_left -= n_take;
            }
        }
        n_dead = uint32_t(__builtin_amdgcn_readfirstlane(__popcll(__ballot(lvl == IDLE))));
        if (__builtin_amdgcn_readfirstlane(int(n_dead == uint32_t(WAVE)))) {
            break;
        }

        // ---- service part, C: top-level steps until every lane is inside an instance or through with its segment
        for (;;) {
            const bool in_c = (lvl == TLAS) && (cur != BVH4_SENTINEL);
            if (__builtin_amdgcn_readfirstlane(int(__ballot(in_c) == 0ull))) {
                break;
            }
            if (in_c) {
                if ((cur & BVH2_PRIM_COUNT_BITS) == 0) { // reference BVH2 node: near child first, far child pushed (bvh2_node_step)
                    const f3 inv = safe_invert(rd);
                    const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + cur);
                    const float4 d0 = np[0], d1 = np[1], d2 = np[2], links = np[3];
                    const uint32_t left_child = float_as_uint(links.x), right_child = float_as_uint(links.y);
                    const float ch0_min[3] = {d0.x, d0.z, d2.x}, ch0_max[3] = {d0.y, d0.w, d2.y};
                    const float ch1_min[3] = {d1.x, d1.z, d2.z}, ch1_max[3] = {d1.y, d1.w, d2.w};
                    float ch0_dist, ch1_dist;
                    const bool ch0_res = bbox_test(ro, inv, h.t, ch0_min, ch0_max, ch0_dist);
                    const bool ch1_res = bbox_test(ro, inv, h.t, ch1_min, ch1_max, ch1_dist);
                    if (!ch0_res && !ch1_res) {
                        pop();
                    } else if (ch0_res && ch1_res) {
                        const bool swap = ch1_dist < ch0_dist;
                        st.write_at(size++, tos);
                        tos = swap ? left_child : right_child;
                        cur = swap ? right_child : left_child;
                    } else {
                        cur = ch0_res ? left_child : right_child;
                    }
                } else { // one mesh instance
                    const uint32_t mi = (cur & BVH2_PRIM_INDEX_BITS);
                    const rayhip_mesh_instance &inst = sc.mesh_instances[mi];
                    if ((inst.ray_visibility & (1u << RAY_TYPE_SHADOW)) != 0) {
                        mi_index = mi;
                        o = transform_point(ro, inst.inv_xform);
                        d = transform_direction(rd, inst.inv_xform);
                        inv_d = safe_invert(d);
                        st.write_at(size++, tos); // the top-level walk resumes from here
                        tos = BVH4_SENTINEL;
                        cur = sc.blas_root4[mi];
                        lvl = BLAS;
                        leave_blas();
                    } else {
                        pop();
                    }
                }
            }
        }
    }
}

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
                                                       const uint32_t *__restrict__ vtx_indices, const uint32_t n_tris, float4 *__restrict__ tri_verts,
                                                       float4 *__restrict__ tri_bitangents) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tris) {
        float4 rows[TRI_VERTS_STRIDE], brows[TRI_BITANGENTS_STRIDE];
        fill_tri_verts(vertices, vertices_count, vtx_indices, t, rows, brows);
        for (int k = 0; k < TRI_VERTS_STRIDE; ++k) {
            tri_verts[size_t(t) * TRI_VERTS_STRIDE + k] = rows[k];
        }
        for (int k = 0; k < TRI_BITANGENTS_STRIDE; ++k) {
            tri_bitangents[size_t(t) * TRI_BITANGENTS_STRIDE + k] = brows[k];
        }
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
