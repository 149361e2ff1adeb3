// wavefront.hip.h -- what every wavefront kernel of librayhip shares: the wave-level slot allocator, the striped ray
// queues, the SoA views of deferred emitter hits and of the pixel buffers.  Included by kernels.hip.h (ray generation,
// traversal, accumulate, ...) and by shade_kernels.hip (the shade stage, which is its own translation unit).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_pixel.h"
#include "rt_rng.h"
#include "rt_types.h"

namespace rt {

constexpr int WAVE = 64;

// Reserve one output slot per lane with `pred` set: one atomicAdd per wavefront.
__device__ __forceinline__ uint32_t wave_alloc(uint32_t *counter, const bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) {
        return 0u;
    }
    const uint32_t lane = __lane_id();
    const uint32_t prefix = uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (int(lane) == leader) {
        base = atomicAdd(counter, uint32_t(__popcll(mask)));
    }
    base = uint32_t(__shfl(int(base), leader));
    return base + prefix;
}

// Striped ray queue.  The slots of a wavefront-state buffer are split into `stripes` equal segments, each with its
// own fill counter on its own 256-byte line.  A 64-ray chunk read from stripe s writes its survivors to stripe s of
// the output queue, so (a) a stripe can never overflow -- it receives at most what it held, and the ray generator
// deals pixel chunks round-robin -- and (b) the one-atomic-per-wavefront slot allocation is spread over `stripes`
// addresses.  Measured on MI355X (tools/atomic_bench.hip): 11.5 ns per atomic on one counter, 0.37 ns on 64; with
// one counter the 250 k allocations of a 1080p frame were half of the shade kernels' time.  Rays stay densely packed
// inside each stripe, so wavefronts stay full; stripes == 1 is the plain dense queue (kernel-level test hooks, ray
// sort).
constexpr uint32_t QUEUE_COUNTER_STRIDE = 64; // uint32 words between stripe counters
constexpr uint32_t QUEUE_MAX_STRIPES = 64;
struct RayQueue {
    uint32_t *counts; // counts[s * QUEUE_COUNTER_STRIDE] = rays in stripe s
    uint32_t stripes;
    uint32_t chunks_per_stripe; // stripe capacity / 64

    __device__ __forceinline__ uint32_t total_chunks() const { return stripes * chunks_per_stripe; }
    // Chunk indices that can hold rays: chunks are numbered stripe-minor, so nothing lives beyond the fullest stripe's
    // last chunk.  A consumer that walks [0, live_chunks()) instead of [0, total_chunks()) does not poll the empty tail of
    // the queue (late bounces fill a few per cent of it; with a 16x oversubscribed grid the polling was 16 % of the shade
    // kernel's wave time).  Wavefront-collective: call with all 64 lanes active.
    __device__ __forceinline__ uint32_t live_chunks() const {
        const uint32_t lane = __lane_id();
        uint32_t fill = lane < stripes ? counts[lane * QUEUE_COUNTER_STRIDE] : 0u;
        for (int m = 32; m >= 1; m >>= 1) {
            fill = max(fill, uint32_t(__shfl_xor(int(fill), m)));
        }
        return uint32_t(__builtin_amdgcn_readfirstlane(int(min(total_chunks(), stripes * ((fill + WAVE - 1) / WAVE)))));
    }
    // chunk c (wave-uniform) -> its stripe, first slot and number of live lanes; false if the chunk is empty.
    // Chunks are numbered stripe-minor so that consecutive wavefronts work on different stripes.
    __device__ __forceinline__ bool chunk(const uint32_t c, uint32_t &stripe, uint32_t &slot0, uint32_t &n_live) const {
        stripe = c % stripes;
        const uint32_t j = c / stripes;
        const uint32_t n = counts[stripe * QUEUE_COUNTER_STRIDE];
        if (j * WAVE >= n) {
            return false;
        }
        slot0 = (stripe * chunks_per_stripe + j) * WAVE;
        n_live = n - j * WAVE < uint32_t(WAVE) ? n - j * WAVE : uint32_t(WAVE);
        return true;
    }
    // the same with the fill counts held in registers (lane s: the count of stripe s, read once when the kernel starts -- a consumer's input queue
    // does not change while it runs): no scalar load, and no wait for one, per chunk
    __device__ __forceinline__ uint32_t fill_counts() const { return __lane_id() < stripes ? counts[__lane_id() * QUEUE_COUNTER_STRIDE] : 0u; }
    __device__ __forceinline__ bool chunk(const uint32_t c, const uint32_t fills, uint32_t &stripe, uint32_t &slot0, uint32_t &n_live) const {
        stripe = c % stripes;
        const uint32_t j = c / stripes;
        const uint32_t n = uint32_t(__builtin_amdgcn_readlane(int(fills), int(stripe)));
        if (j * WAVE >= n) {
            return false;
        }
        slot0 = (stripe * chunks_per_stripe + j) * WAVE;
        n_live = n - j * WAVE < uint32_t(WAVE) ? n - j * WAVE : uint32_t(WAVE);
        return true;
    }
    __device__ __forceinline__ uint32_t alloc(const uint32_t stripe, const bool pred) const {
        return stripe * chunks_per_stripe * WAVE + wave_alloc(counts + stripe * QUEUE_COUNTER_STRIDE, pred);
    }
};

// Which chunks a block takes.  Plain striding (c = blockIdx, += gridDim) hands consecutive chunks -- neighbouring 8x8 pixel
// tiles at the primary level, their survivors later -- to consecutive blocks, and consecutive blocks sit on different XCDs
// (block b runs on XCD b % 8: observed placement, MI355X_MICROARCH.md; used for speed only, any placement is correct), so
// each of the eight L2s sees rays from everywhere.  With RT_XCD_CHUNKS=1 every sweep of gridDim chunks is cut into eight
// contiguous parts and the blocks of one XCD take one part: an XCD's L2 then serves a compact piece of the frame (the
// sub-trees, triangle rows and materials behind it) instead of an eighth of every piece.
// MEASURED (MI355X, all traversal and shade kernels, bit-identical frames): no gain -- Bistro-class 64 spp 508.6 vs 507.9
// Msamples/s, 20 spp 488.9 vs 485.6, Sponza-class 567.5 vs 561.6 (plain vs XCD-aware; profiles/r02/experiments/
// variants_xcd.txt).  The per-XCD L2 hit rate is not what limits these kernels (the 256 MB last-level cache behind the
// L2s holds the whole tree either way); the plain order stays the default.
#ifndef RT_XCD_CHUNKS
#define RT_XCD_CHUNKS 0
#endif
// Round 5 -- chunks handed out DYNAMICALLY (the persistent kernels with lane refill: K2, K3, the light pick).  With `work` set, a block takes
// chunk blockIdx.x as before and every further one from a shared counter (one atomic per run of `run` chunks, issued by lane 0) instead of
// a fixed share.  Why: a wavefront that refills its lanes idles through the tail of its longest ray ONCE PER BLOCK, when its share is used up;
// with a fixed share per block the grid has to be many times what the device holds to even out the end of the launch (16 x: a block of a rank
// of 8 then owns three or four live chunks -- three or four rays per lane -- and drains after each).  Dynamically, a grid of resident blocks
// ends when the LAST CHUNK is done, and every wavefront drains once.
// What it costs, and why only those kernels use it (profiles/r05/experiments/dynamic_chunks.txt): read-modify-writes on ONE address retire
// every ~11.5 ns chip-wide (tools/atomic_bench.hip) -- enough for K2's secondary bounces (a chunk per wavefront every ~100 us, ~35 M fetches
// per second), not for a kernel that streams (the primary K2 launch: 2 M chunks in 16 ms) -- and the wait for the fetched index is a wait for
// ALL of the wavefront's outstanding memory operations (vmcnt counts loads and stores alike), which a streaming kernel cannot afford: its
// stores of chunk k overlap its loads of chunk k + 1.  First form tried: every chunk fetched, a second counter to re-arm the first -- 2 x 6 k
// serialized atomics = 140 us in every launch, empty or not.  Hence: the first chunk of a block costs nothing (a launch with no more chunks
// than blocks issues no atomic at all), and the host clears the counters of a pass together with its queue counters.
// The order in which chunks are taken has no influence on results (every ray's pixel, slot stripe and arithmetic are its own).
// `work` == nullptr is the static walk.  Wavefront-collective: all 64 lanes active.
constexpr uint32_t WORK_COUNTER_STRIDE = 64; // uint32 words between the counters of consecutive launches (a 256-byte line each)
struct ChunkWalk {
    uint32_t base, n, grid, x, local;
    uint32_t *work;
    uint32_t run, run_next, run_left;
    // `walker` of `walkers`: who takes chunks -- a block by default (one wavefront per block); a kernel with several wavefronts per block passes
    // its wavefront's number among all of the grid
    __device__ __forceinline__ explicit ChunkWalk(const uint32_t n_chunks, uint32_t *work_counter = nullptr, const uint32_t run_length = 1u,
                                                  const uint32_t walker = blockIdx.x, const uint32_t walkers = gridDim.x)
        : base(0), n(n_chunks), work(work_counter), run(run_length), run_next(0), run_left(0), me(walker), all(walkers) {
        const bool remap = RT_XCD_CHUNKS != 0 && all >= 8u && work_counter == nullptr;
        grid = remap ? (all & ~7u) : all; // (up to seven surplus blocks of a grid that is not a multiple of 8 idle)
        x = remap ? (me & 7u) : 0u;
        local = remap ? (me >> 3) : me;
        if (remap && me >= grid) {
            base = n;
        }
        parts = remap ? 8u : 1u;
    }
    uint32_t parts, me, all;
    // the next chunk of this block (wave-uniform); false when there is none left
    __device__ __forceinline__ bool next(uint32_t &c) {
        if (work != nullptr) {
            if (base == 0u) { // the first chunk of a block is its own
                base = 1u;
                c = me;
                if (c >= n || all >= n) { // (nothing, or nothing beyond the first chunks: no atomic from this block)
                    base = 2u;
                }
                return c < n;
            }
            if (run_left != 0u) {
                c = run_next++;
                --run_left;
                return true;
            }
            if (base == 2u) {
                return false;
            }
            uint32_t v = 0u;
            if (__lane_id() == 0u) {
                v = atomicAdd(work, run);
            }
            c = all + uint32_t(__builtin_amdgcn_readfirstlane(int(v)));
            if (c >= n) {
                base = 2u;
                return false;
            }
            run_next = c + 1u, run_left = min(run, n - c) - 1u;
            return true;
        }
        while (base < n) {
            const uint32_t width = min(grid, n - base), per = (width + parts - 1u) / parts;
            const uint32_t off = x * per + local;
            const bool mine = local < per && off < width;
            c = base + off;
            base += grid;
            if (mine) {
                return true;
            }
        }
        return false;
    }
};

// hits of importance-sampled emitters whose MIS weight is evaluated by k_shade_emissive
struct DeferredSoA {
    float4 *a; // ray slot, tri_index, material index (bits), mix_weight
    float4 *b; // base_color.rgb
};

struct PixelBuffers {
    float4 *temp; // [layers][h][w]: radiance of the iteration(s) in flight
    float4 *full, *half, *raw, *final_, *base_color, *depth_normals;
    uint16_t *required_samples;
    float4 *aux_base_layers, *aux_dn_layers; // [layers][h][w], batched passes only (rt_pixel.h)
    float4 *variance; // [h][w]: the variance estimate of the last accumulate (the reference leaves it in its temp buffer,
                      // RendererCPU.h:641-645; DenoiseImage reads it from there)
};

// per-layer part of AccumParams for a batched pass
struct AccumLayer {
    int iteration;
    float mix_factor, half_mix_factor;
    int is_class_a;
    float variance_threshold;
};
#ifndef RT_MAX_BATCH
#define RT_MAX_BATCH 64
#endif
constexpr int MAX_BATCH = RT_MAX_BATCH;
struct AccumLayers {
    AccumLayer l[MAX_BATCH];
};

__device__ __forceinline__ uint32_t layer_rand_seed(const int iteration) { return hash(uint32_t((iteration - 1) / RAND_SAMPLES_COUNT)); }

} // namespace rt
