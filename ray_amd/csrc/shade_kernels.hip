// shade_kernels.hip -- K5, the shade stage of the wavefront path tracer, as its own translation unit.
//
// Why a separate unit: the shade kernels are two thirds of the device code of the library (k_scatter alone is ~20 k
// instructions); compiled next to the traversal kernels in parallel they halve the build time, and build-time experiments
// on one side do not touch the other.
// One such experiment is recorded here because its result shapes the code: compiling this unit with
// -fno-hip-fp32-correctly-rounded-divide-sqrt (division = v_rcp_f32 + multiply, <= 2.5 ulp; square root = v_sqrt_f32) removes
// a quarter of the instructions -- the stage is bound by VALU issue, k_scatter holds 464 correctly rounded divisions at ~10
// instructions each -- and makes it 0.3 ms per 1080p iteration faster (1.94 -> 1.66 ms, MI355X, Bistro-class scene).  It also
// moves 0.47 % of the pixels of a 1-spp frame out of the stated tolerance (PSNR 37.8 dB against the oracle instead of 133 dB):
// the solid angle of a triangle emitter is computed as alpha + beta + gamma - pi (Arvo), a difference of O(1) angles that is
// O(1e-3) for the light triangles of that scene, so 2-ulp operations upstream become 1e-3 relative errors in the density.
// The shade stage therefore keeps IEEE division and square root (only the light-tree importance heuristic, which steers
// probabilities the estimator divides out again, uses the hardware reciprocal / square root: shade_lights.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#ifdef RT_PROFILE_SHADE
// tuning build (tools/variants.py "+shade:-DRT_PROFILE_SHADE"): which branches of the scatter stage run, and with how many lanes
namespace rt {
__device__ unsigned long long g_prof_shade[64];
}
#define RT_PROF_SHADE_LANES(k)                                                                                          \
    {                                                                                                                   \
        const unsigned long long m_ = __ballot(1);                                                                      \
        if (int(__lane_id()) == __ffsll((long long)m_) - 1) {                                                           \
            atomicAdd(&::rt::g_prof_shade[k], (unsigned long long)__popcll(m_));                                        \
            atomicAdd(&::rt::g_prof_shade[(k) + 1], 1ull);                                                              \
        }                                                                                                               \
    }
#endif

#include "shade_launch.h"

namespace rt {
namespace shade {

// ---- K5: shade, as three stages with a queue of shade points between them ------------------------------------------------
// (reference kernel: shade.comp.glsl, one thread per ray doing everything; here cut where the working set changes --
// shade_point.h has the rationale)
//   k_surface      (ray, hit) -> pixel radiance for paths that end (miss, emitter hit, culled back face, emissive surface)
//                  or a ShadePoint (7 float4 planes) appended to the point queue of its stripe; first-hit feature images
//   k_light_pick   light-tree descent for every point: reads 16 B, writes 16 B per point, nothing else live
//   k_scatter      samples the picked light, evaluates the material towards it (-> shadow ray), draws the continuation
//                  (-> secondary ray), Russian roulette.  <NEE, CONTINUE> lets the two halves run as one launch or two.
// Occupancy hints (waves per SIMD) are per stage; the register budget of the old one-kernel form was set by the sum of
// all three working sets (168 VGPRs, 3 waves).
#ifndef RT_SURFACE_MIN_WAVES
#define RT_SURFACE_MIN_WAVES 4
#endif
#ifndef RT_PICK_MIN_WAVES
#define RT_PICK_MIN_WAVES 8
#endif
#ifndef RT_SCATTER_MIN_WAVES
#define RT_SCATTER_MIN_WAVES 4
#endif
// k_surface_scatter holds the ray across the surface stage and the point across the continuation: ~175 registers at its peak.  Three
// waves (168 VGPRs, no spills) against four (128 + 130 bytes of scratch) / five: 1.24 / 1.50 / 2.21 ms per iteration (Bistro-class,
// profiles/r06/experiments/shade_fused_variants.txt)
#ifndef RT_FUSED_MIN_WAVES
#define RT_FUSED_MIN_WAVES 3
#endif
// the two halves of the split form hold less than the combined kernel: their own budgets
#ifndef RT_SCATTER_NEE_MIN_WAVES
#define RT_SCATTER_NEE_MIN_WAVES 4
#endif
#ifndef RT_SCATTER_CONT_MIN_WAVES
#define RT_SCATTER_CONT_MIN_WAVES 4
#endif

__device__ __forceinline__ void store_point(const PointSoA &s, const uint32_t i, const ShadePoint &pt, const uint32_t ray_slot) {
    s.p_slot[i] = mkfloat4(pt.P.x, pt.P.y, pt.P.z, uint_as_float(ray_slot));
    s.n_gx[i] = mkfloat4(pt.N.x, pt.N.y, pt.N.z, pt.plane_N.x);
    s.b_gy[i] = mkfloat4(pt.B.x, pt.B.y, pt.B.z, pt.plane_N.y);
    s.base_gz[i] = mkfloat4(pt.base.x, pt.base.y, pt.base.z, pt.plane_N.z);
    s.scalars[i] = mkfloat4(pt.roughness, pt.metallic, pt.specular, pt.mix_weight);
    s.misc[i] = mkfloat4(pt.mix_pick, uint_as_float(pt.material | (pt.backfacing ? 0x80000000u : 0u)), pt.cone_width, 0.0f);
}
__device__ __forceinline__ ShadePoint load_point(const PointSoA &s, const uint32_t i, uint32_t &ray_slot) {
    const float4 a = s.p_slot[i], b = s.n_gx[i], c = s.b_gy[i], d = s.base_gz[i], e = s.scalars[i], f = s.misc[i];
    ShadePoint pt;
    pt.P = {a.x, a.y, a.z}, ray_slot = float_as_uint(a.w);
    pt.N = {b.x, b.y, b.z}, pt.B = {c.x, c.y, c.z}, pt.plane_N = {b.w, c.w, d.w};
    pt.base = {d.x, d.y, d.z};
    pt.roughness = e.x, pt.metallic = e.y, pt.specular = e.z, pt.mix_weight = e.w;
    pt.mix_pick = f.x;
    const uint32_t m = float_as_uint(f.y);
    pt.material = m & 0x7fffffffu, pt.backfacing = (m >> 31) != 0;
    pt.cone_width = f.z;
    return pt;
}
__device__ __forceinline__ void store_pick(const PointSoA &s, const uint32_t i, const LightPick &k) {
    s.light[i] = mkfloat4(uint_as_float(k.light), k.inv_prob, k.u_left, 0.0f);
}
__device__ __forceinline__ LightPick load_pick(const PointSoA &s, const uint32_t i) {
    const float4 v = s.light[i];
    return LightPick{float_as_uint(v.x), v.y, v.z};
}

// the parameters of the layer a (virtual) pixel belongs to: later iterations of a batched pass carry their own sample
// index and seed, and their random numbers are keyed by the REAL pixel (rt_base.h: Layering)
__device__ __forceinline__ ShadeParams layer_params(const ShadeParams &sp, const uint32_t layer) {
    ShadeParams p = sp;
    if (layer != 0) {
        p.iteration = sp.iteration + int(layer);
        p.rand_seed = layer_rand_seed(p.iteration);
    }
    return p;
}

template <bool PRIMARY, bool PICK, bool SKY = false>
__global__ void __launch_bounds__(WAVE, RT_SURFACE_MIN_WAVES) k_surface(const SceneView sc, const ShadeParams sp, const RaySoA rays_in, const HitSoA hits,
                                                                       const RayQueue in, const PointSoA points, const RayQueue out_points,
                                                                       const DeferredSoA deferred_out, const RayQueue out_deferred,
                                                                       const PixelBuffers px, const int img_w, const float mix_factor,
                                                                       const Layering layers, uint32_t *__restrict__ sky_index, const RayQueue out_sky) {
    const uint32_t n_live_chunks = in.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!in.chunk(c, stripe, slot0, n_live)) {
            continue;
        }
        const uint32_t i = slot0 + threadIdx.x; // (the whole wavefront stays in the body for the ballots)
        const bool active = threadIdx.x < n_live;
        bool continues = false, defer = false, sky = false;
        ShadePoint pt;
        SurfaceOut so;
        LightPick pick = no_light_pick();
        if (active) {
            Ray ray = load_ray(rays_in, i, sp.plain_ior == 0u); // (plain_ior: the ior plane is not written -- the mix nodes' Fresnel term reads the stack)
            const Hit hit = load_hit(hits, i);
            const uint32_t xy = ray.xy; // virtual (layered) pixel: where the pixel writes go
            const uint32_t layer = xy_layer(xy, layers);
            const ShadeParams spl = layer_params(sp, layer);
            ray.xy = xy_real(xy, layers, layer);
            continues = surface_stage<true, SKY>(sc, spl, hit, ray, pt, so); // (emitter MIS weights: k_shade_emissive)
            defer = so.deferred_emitter;
            sky = SKY && so.deferred_sky;
            if (PRIMARY) {
                // the pixel of a continuing path starts at (0, 0, 0, 1); the scatter stage adds what shadow-less lights give
                ShadeResult res;
                res.col = continues ? f4{0.0f, 0.0f, 0.0f, 1.0f} : so.radiance;
                res.base_color = so.base_color, res.depth_normal = so.normal_depth;
                if (layers.count > 1) {
                    write_primary_pixel_layered(res, xy, img_w, px.temp, px.aux_base_layers, px.aux_dn_layers);
                } else {
                    write_primary_pixel(res, xy, img_w, mix_factor, px.temp, px.base_color, px.depth_normals);
                }
            } else if (!continues) {
                ShadeResult res;
                res.col = so.radiance;
                add_secondary_pixel(res, xy, img_w, px.temp);
            }
            if (PICK && continues && sc.light_cwnodes_count != 0) {
                pick = pick_light(sc, pt.P, light_pick_random(sc, spl, ray.xy, ray.depth));
            }
        }
        // survivors go to the stripe they came from (RayQueue)
        const uint32_t p_slot = out_points.alloc(stripe, continues);
        if (continues) {
            store_point(points, p_slot, pt, i);
            if (PICK) {
                store_pick(points, p_slot, pick);
            }
        }
        if (SKY) { // the physical sky: paths that ended in it wait for k_shade_sky (their pixel got zeros above)
            const uint32_t s_slot = out_sky.alloc(stripe, sky);
            if (sky) {
                sky_index[s_slot] = i;
            }
        }
        if (__any(defer)) { // rare
            const uint32_t d_slot = out_deferred.alloc(stripe, defer);
            if (defer) {
                deferred_out.a[d_slot] = mkfloat4(uint_as_float(i), uint_as_float(so.emitter_triangle), uint_as_float(pt.material), so.emitter_mix_weight);
                deferred_out.b[d_slot] = mkfloat4(pt.base.x, pt.base.y, pt.base.z, 0.0f);
            }
        }
    }
}

// COMPACT: the points whose descent ended at a light (inv_prob != 0: the only ones the next-event estimation has work for -- 18 % of
// the points of the Bistro-class scene) are listed densely in the `nee` queue of their stripe, for k_scatter<true, false, true>
template <bool COMPACT>
__global__ void __launch_bounds__(WAVE, RT_PICK_MIN_WAVES) k_light_pick(const SceneView sc, const ShadeParams sp, const RaySoA rays_in,
                                                                       const PointSoA points, const RayQueue queue, const RayQueue nee,
                                                                       const Layering layers) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_live_chunks = queue.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!queue.chunk(c, stripe, slot0, n_live)) {
            continue;
        }
        const uint32_t i = slot0 + lane;
        bool usable = false;
        if (lane < n_live) {
            const float4 ps = points.p_slot[i];
            const uint2 xd = rays_in.xy_depth[float_as_uint(ps.w)];
            const uint32_t layer = xy_layer(xd.x, layers);
            const ShadeParams spl = layer_params(sp, layer);
            const LightPick pk = pick_light(sc, f3{ps.x, ps.y, ps.z}, light_pick_random(sc, spl, xy_real(xd.x, layers, layer), xd.y));
            store_pick(points, i, pk);
            usable = pk.inv_prob != 0.0f;
        }
        if (COMPACT) {
            const uint32_t slot = nee.alloc(stripe, usable);
            if (usable) {
                points.nee_index[slot] = i;
            }
        }
    }
}

// The same pick with lanes that take the next point as soon as theirs is through (round 4).  A descent ends where a subtree cannot light
// the point -- at the root for some points, five levels down for others: in the chunk-at-a-time kernel above 58 % of the lanes were busy
// (profiles/r03/pmc_sq_tcc_summary.txt).  A lane's whole state is eight registers (position, random number, probability, node), so a
// finished lane is refilled for the price of two loads and one Owen-scrambled sample, once RT_PICK_REFILL_MIN lanes wait for it.  Per
// point the descent is pick_light's, statement for statement: the same pick in the same slot; only the order of the `nee` list differs.
#ifndef RT_PICK_REFILL_MIN
#define RT_PICK_REFILL_MIN 16
#endif
// chunks per fetch of the dynamic hand-out (wavefront.hip.h: ChunkWalk): this kernel takes a chunk every ~4.5 ns chip-wide, one counter hands out
// a run every ~11.5 ns -- with 1 the launch took 10 ms instead of 4 (profiles/r05/experiments/dynamic_chunks.txt)
#ifndef RT_PICK_RUN
#define RT_PICK_RUN 8
#endif
// DYN: chunks from the work counter (wavefront.hip.h; opt-in) -- a template parameter, not a null test, so that the default kernel is the code it
// was before the counter existed (as a run-time pointer it cost this kernel nine more spilled registers and 14 % of its time: 64 VGPRs at 8 waves)
template <bool COMPACT, bool DYN = false>
__global__ void __launch_bounds__(WAVE, RT_PICK_MIN_WAVES) k_light_pick_refill(const SceneView sc, const ShadeParams sp, const RaySoA rays_in,
                                                                              const PointSoA points, const RayQueue queue, const RayQueue nee,
                                                                              const Layering layers, uint32_t *__restrict__ work /* dynamic chunk hand-out, may be null */) {
    const uint32_t lane = threadIdx.x;
    ChunkWalk walk(queue.live_chunks(), DYN ? work : nullptr, RT_PICK_RUN);
    uint32_t pool_slot = 0, pool_left = 0, pool_stripe = 0; // (uniform) the chunk being handed out
    bool exhausted = false;                                  // (uniform) no chunk left to hand out
    // lane state
    bool busy = false;
    uint32_t i = 0, stripe = 0, cur = 0;
    f3 P = {0.0f, 0.0f, 0.0f};
    float u = 0.0f, u_first = 0.0f, prob = 1.0f;
    for (;;) {
        const int n_idle = __popcll(__ballot(!busy));
        if (!exhausted && n_idle >= RT_PICK_REFILL_MIN) {
            for (;;) {
                const unsigned long long idle_mask = __ballot(!busy);
                if (idle_mask == 0ull) {
                    break;
                }
                if (pool_left == 0) {
                    int found = 0;
                    uint32_t next_chunk;
                    while (!found && walk.next(next_chunk)) { // (uniform)
                        uint32_t s, slot0, n_live;
                        found = __builtin_amdgcn_readfirstlane(int(queue.chunk(next_chunk, s, slot0, n_live)));
                        if (found) {
                            pool_slot = uint32_t(__builtin_amdgcn_readfirstlane(int(slot0)));
                            pool_left = uint32_t(__builtin_amdgcn_readfirstlane(int(n_live)));
                            pool_stripe = uint32_t(__builtin_amdgcn_readfirstlane(int(s)));
                        }
                    }
                    if (!found) {
                        exhausted = true;
                        break;
                    }
                }
                const uint32_t rank = uint32_t(__popcll(idle_mask & ((1ull << lane) - 1ull)));
                const uint32_t n_take = min(uint32_t(__popcll(idle_mask)), pool_left);
                if (!busy && rank < n_take) {
                    i = pool_slot + rank, stripe = pool_stripe;
                    const float4 ps = points.p_slot[i];
                    const uint2 xd = rays_in.xy_depth[float_as_uint(ps.w)];
                    const uint32_t layer = xy_layer(xd.x, layers);
                    const ShadeParams spl = layer_params(sp, layer);
                    P = {ps.x, ps.y, ps.z};
                    u = u_first = light_pick_random(sc, spl, xy_real(xd.x, layers, layer), xd.y);
                    prob = 1.0f, cur = 0;
                    busy = true;
                }
                pool_slot += n_take, pool_left -= n_take;
            }
        }
        if (__ballot(busy) == 0ull) {
            if (exhausted) {
                break;
            }
            continue; // (every lane idle: the next round refills)
        }
        // one level of the descent for the lanes that hold a point (pick_light, shade_lights.h)
        bool lit = false;
        if (busy) {
            LightPick pk;
            pk.light = 0, pk.inv_prob = 0.0f, pk.u_left = u_first;
            float imp[8];
            light_node_importances(sc, cur, P, imp);
            int chosen;
            bool done = !light_level_choice(imp, u, prob, chosen); // false: nothing in this subtree can light P
            if (!done) {
                cur = light_child_link(sc, cur, chosen);
                if ((cur & LEAF_NODE_BIT) != 0) {
                    pk.light = (cur & PRIM_INDEX_BITS), pk.inv_prob = 1.0f / prob, pk.u_left = u;
                    done = true;
                }
            }
            if (done) {
                store_pick(points, i, pk);
                lit = pk.inv_prob != 0.0f;
                busy = false;
            }
        }
        if (COMPACT) { // the points that got a light join the `nee` list of their stripe (the lanes of a wavefront hold at most a few stripes)
            unsigned long long todo = __ballot(lit);
            while (todo != 0ull) {
                const uint32_t s = uint32_t(__shfl(int(stripe), __ffsll((long long)todo) - 1));
                const bool mine = lit && stripe == s;
                const uint32_t slot = nee.alloc(uint32_t(__builtin_amdgcn_readfirstlane(int(s))), mine);
                if (mine) {
                    points.nee_index[slot] = i;
                }
                todo &= ~__ballot(mine);
            }
        }
    }
}

// Which form of stage 3 runs is decided on the device, from the two fill counts k_light_pick left behind, by every wavefront of
// the three candidate launches in the same way (wavefront-collective: all 64 lanes active): the split form when fewer than half
// of the points have a light to sample -- otherwise the work both launches repeat (frame, lobe set-up, loads) costs more than
// the idle lanes (03_principled, every point lit: 2.70 against 2.47 ms).  The launches of the form that lost return at once.
__device__ __forceinline__ uint32_t queue_fill(const RayQueue &q) {
    const uint32_t lane = __lane_id();
    uint32_t n = lane < q.stripes ? q.counts[lane * QUEUE_COUNTER_STRIDE] : 0u;
    for (int m = 32; m >= 1; m >>= 1) {
        n += uint32_t(__shfl_xor(int(n), m));
    }
    return uint32_t(__builtin_amdgcn_readfirstlane(int(n)));
}
__device__ __forceinline__ bool lit_points_are_sparse(const RayQueue &pts, const RayQueue &nee) { return 2u * queue_fill(nee) < queue_fill(pts); }

// INDEXED: `in` is the queue of points that got a light (k_light_pick<true>); its slots name the point slots.
// MODE: 0 = runs unconditionally; 1 = only if the lit points are sparse (the split form); 2 = only if they are not
template <bool NEE, bool CONTINUE, bool INDEXED = false, int MODE = 0>
__global__ void __launch_bounds__(WAVE, (NEE && CONTINUE) ? RT_SCATTER_MIN_WAVES : (NEE ? RT_SCATTER_NEE_MIN_WAVES : RT_SCATTER_CONT_MIN_WAVES)) k_scatter(const SceneView sc, const ShadeParams sp, const RaySoA rays_in,
                                                                       const PointSoA points, const RayQueue in, const RaySoA rays_out,
                                                                       const RayQueue out_rays, const ShadowSoA shadow_out, const RayQueue out_shadow,
                                                                       const PixelBuffers px, const int img_w, const Layering layers,
                                                                       const RayQueue all_points, const RayQueue lit_points) {
    if (MODE != 0 && lit_points_are_sparse(all_points, lit_points) != (MODE == 1)) {
        return;
    }
    const uint32_t n_live_chunks = in.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!in.chunk(c, stripe, slot0, n_live)) {
            continue;
        }
        const bool active = threadIdx.x < n_live;
        Scatter sct;
        sct.has_next = sct.has_shadow = false;
        uint32_t xy = 0;
        if (active) {
            uint32_t ray_slot;
            const uint32_t point_slot = INDEXED ? points.nee_index[slot0 + threadIdx.x] : slot0 + threadIdx.x;
            const ShadePoint pt = load_point(points, point_slot, ray_slot);
            const LightPick pick = (NEE && sc.light_cwnodes_count != 0) ? load_pick(points, point_slot) : no_light_pick();
            Ray ray;
            {
                const float4 d = rays_in.d_cw[ray_slot], cc = rays_in.c_cs[ray_slot];
                const float4 io = sp.plain_ior != 0u ? mkfloat4(-1.0f, -1.0f, -1.0f, -1.0f) : rays_in.ior[ray_slot]; // (plain_ior: the camera's stack)
                const uint2 xd = rays_in.xy_depth[ray_slot];
                ray.o = pt.P, ray.pdf = 0.0f; // (not read by the scatter stage)
                ray.d = {d.x, d.y, d.z}, ray.cone_width = d.w;
                ray.c = {cc.x, cc.y, cc.z}, ray.cone_spread = cc.w;
                ray.ior[0] = io.x, ray.ior[1] = io.y, ray.ior[2] = io.z, ray.ior[3] = io.w;
                ray.xy = xd.x, ray.depth = xd.y;
            }
            xy = ray.xy;
            const uint32_t layer = xy_layer(xy, layers);
            const ShadeParams spl = layer_params(sp, layer);
            ray.xy = xy_real(xy, layers, layer);
            scatter_stage<NEE, CONTINUE>(sc, spl, ray, pt, pick, sct);
            sct.next.xy = xy, sct.shadow.xy = xy;
            if (NEE) {
                const f3 col = direct_radiance(spl, sct, ray.c);
                if (col.x != 0.0f || col.y != 0.0f || col.z != 0.0f) { // (x + 0 == x: the add is skipped when nothing is booked)
                    ShadeResult res;
                    res.col = mk4(col, 1.0f);
                    add_secondary_pixel(res, xy, img_w, px.temp);
                }
            }
        }
        if (NEE) {
            const uint32_t sh_slot = out_shadow.alloc(stripe, sct.has_shadow);
            if (sct.has_shadow) {
                store_shadow(shadow_out, sh_slot, sct.shadow);
            }
        }
        if (CONTINUE) {
            const uint32_t ray_slot = out_rays.alloc(stripe, sct.has_next);
            if (sct.has_next) {
                store_ray(rays_out, ray_slot, sct.next, sp.plain_ior == 0u);
            }
        }
    }
}

// ---- round 6: the stage without the queue of shade points ---------------------------------------------------------------------------
// What the three-kernel form moves per path vertex (Bistro-class, profiles/r05/kernel_hbm_bistro.txt): k_surface writes the point (96 B),
// the pick reads 24 B of it and writes a 16-byte pick into a 64-byte line, the next-event kernel gathers the 18 % of the points that got a
// light through an index (16-byte reads out of 64-byte lines: 124 GB fetched for ~30 used), the continuation reads the point again with
// 40 B of the ray -- ~880 B, and the continuation alone already runs at the rate a plain copy reaches (4.3 TB/s on bounce 0).  The stage
// is bound by those bytes, not by its arithmetic (profiles/r06/experiments: contraction changes nothing, dropping the microfacet draws
// 17 % of one kernel).  So the order of the parts changes:
//   k_light_pick_first   the light pick BEFORE the surface stage.  The descent needs the position only, and that is ray.o + hit.t * ray.d
//                        (shade_point.h: pt.P) -- the same two IEEE operations here.  Persistent, lanes refilled as in k_light_pick_refill.
//                        A pick is written only for a ray that GOT a light (18 %), stamped with the tag of this launch: the other slots of
//                        the plane keep whatever they held, with an older tag.
//   k_surface_scatter    surface stage and continuation in one kernel: the point never leaves the registers.  A lane whose slot of the pick
//                        plane carries the launch's tag appends what the next-event estimation needs -- point, pick, the ray's direction,
//                        throughput, pixel, depth -- to a dense list (`nee` queue, coalesced stores).
//   k_scatter<NEE only>  over that list: coalesced reads, full wavefronts.
// ~380 B per vertex instead of ~880.  Per path vertex every stage function gets the arguments it got before (surface_stage, pick_light's
// levels, scatter_stage<false, true>, scatter_stage<true, false>): frames are bit-identical to the three-kernel form
// (test_shade_forms_agree_bit_for_bit); only the order of the lists differs, which no pixel sees.
// LDS_TREE: the wavefronts of a block (PICK_BLOCK_WAVES) share a copy of the light table (`lds_nodes` = all of its nodes) in LDS.  A level of the descent reads
// 24 float4 rows per lane (eight children x 48 bytes): 2.07 M wavefronts x ~2.5 levels x 24 loads x 16 cycles of the CU's vector-memory path is
// what the kernel's time was made of (profiles/r06/experiments/pick_first_counters.txt: 53 % of the wave cycles waiting, the vector ALU a third
// busy); the table of a scene with a few hundred lights is a few kilobytes.  A table that does not fit is read from memory as before (the
// launcher picks the form; a kernel with both paths spills).
// (7 waves: 72 registers, no spills; 8 spill 3-8 dwords: 16.4 against 16.8 ms per frame)
#ifndef RT_PICK_LDS_MIN_WAVES
#define RT_PICK_LDS_MIN_WAVES 7
#endif
#ifndef RT_PICK_BLOCK_WAVES
#define RT_PICK_BLOCK_WAVES 8
#endif
constexpr int PICK_BLOCK_WAVES = RT_PICK_BLOCK_WAVES;
// eight wavefronts per block: four blocks (32 wavefronts, 8 per SIMD) per CU share the 160 KB -- 93 nodes x 416 B = 38 688 B per block (the
// Bistro-class atrium has 41 nodes for 96 emitters, the asset street 73 for 200)
constexpr uint32_t PICK_LDS_MAX_NODES = (160u * 1024u / (32u / uint32_t(PICK_BLOCK_WAVES)) - 2048u) / (uint32_t(LIGHT_CHILDREN_STRIDE) * 16u);
template <bool DYN, bool LDS_TREE = false>
__global__ void __launch_bounds__(LDS_TREE ? PICK_BLOCK_WAVES *WAVE : WAVE, LDS_TREE ? RT_PICK_LDS_MIN_WAVES : RT_PICK_MIN_WAVES) k_light_pick_first(const SceneView sc, const ShadeParams sp, const RaySoA rays_in, const HitSoA hits,
                                                                             const RayQueue queue, float4 *__restrict__ picks, const uint32_t tag,
                                                                             const Layering layers, uint32_t *__restrict__ work, const uint32_t lds_nodes) {
    extern __shared__ float4 s_tree[];
    if (LDS_TREE) {
        const uint32_t n_rows = lds_nodes * uint32_t(LIGHT_CHILDREN_STRIDE);
        for (uint32_t r = threadIdx.x; r < n_rows; r += blockDim.x) {
            s_tree[r] = sc.light_children[r];
        }
        __syncthreads();
    }
    const uint32_t lane = LDS_TREE ? (threadIdx.x & uint32_t(WAVE - 1)) : threadIdx.x;
    const uint32_t waves = LDS_TREE ? uint32_t(PICK_BLOCK_WAVES) : 1u;
    ChunkWalk walk(queue.live_chunks(), DYN ? work : nullptr, RT_PICK_RUN, blockIdx.x * waves + (LDS_TREE ? threadIdx.x / WAVE : 0u), gridDim.x * waves);
    uint32_t pool_slot = 0, pool_left = 0; // (uniform) the chunk being handed out
    bool exhausted = false;                // (uniform) no chunk left to hand out
    bool busy = false;
    uint32_t i = 0, cur = 0;
    f3 P = {0.0f, 0.0f, 0.0f};
    float u = 0.0f, prob = 1.0f;
    for (;;) {
        const int n_idle = __popcll(__ballot(!busy));
        if (!exhausted && n_idle >= RT_PICK_REFILL_MIN) {
            for (;;) {
                const unsigned long long idle_mask = __ballot(!busy);
                if (idle_mask == 0ull) {
                    break;
                }
                if (pool_left == 0) {
                    int found = 0;
                    uint32_t next_chunk;
                    while (!found && walk.next(next_chunk)) { // (uniform)
                        uint32_t s, slot0, n_live;
                        found = __builtin_amdgcn_readfirstlane(int(queue.chunk(next_chunk, s, slot0, n_live)));
                        if (found) {
                            pool_slot = uint32_t(__builtin_amdgcn_readfirstlane(int(slot0)));
                            pool_left = uint32_t(__builtin_amdgcn_readfirstlane(int(n_live)));
                        }
                    }
                    if (!found) {
                        exhausted = true;
                        break;
                    }
                }
                const uint32_t rank = uint32_t(__popcll(idle_mask & ((1ull << lane) - 1ull)));
                const uint32_t n_take = min(uint32_t(__popcll(idle_mask)), pool_left);
                if (!busy && rank < n_take) {
                    i = pool_slot + rank;
                    const float4 h = hits.oi_pi_t_u[i];
                    // a triangle was hit (misses carry v < 0, analytic emitters a negative object): everything else ends in k_surface_scatter
                    // without asking for a light.  (A triangle whose path ends there too -- culled back face, emissive material -- gets a
                    // pick nobody reads.)
                    if (hits.v[i] >= 0.0f && float_as_int(h.x) >= 0) {
                        const float4 o = rays_in.o_pdf[i], d = rays_in.d_cw[i];
                        const uint2 xd = rays_in.xy_depth[i];
                        const uint32_t layer = xy_layer(xd.x, layers);
                        const ShadeParams spl = layer_params(sp, layer);
                        P = f3{o.x, o.y, o.z} + h.z * f3{d.x, d.y, d.z}; // == ShadePoint::P (shade_point.h)
                        u = light_pick_random(sc, spl, xy_real(xd.x, layers, layer), xd.y);
                        prob = 1.0f, cur = 0;
                        busy = true;
                    }
                }
                pool_slot += n_take, pool_left -= n_take;
            }
        }
        if (__ballot(busy) == 0ull) {
            if (exhausted) {
                break;
            }
            continue; // (every lane idle: the next round refills)
        }
        if (busy) { // one level of the descent (pick_light, shade_lights.h)
            float imp[8];
            if (LDS_TREE) {
                light_node_importances_rows(s_tree + cur * uint32_t(LIGHT_CHILDREN_STRIDE), P, imp);
            } else {
                light_node_importances(sc, cur, P, imp);
            }
            int chosen;
            if (!light_level_choice(imp, u, prob, chosen)) {
                busy = false; // nothing in this subtree can light P: no pick is written
            } else {
                cur = LDS_TREE ? light_child_link_rows(s_tree + cur * uint32_t(LIGHT_CHILDREN_STRIDE), chosen) : light_child_link(sc, cur, chosen);
                if ((cur & LEAF_NODE_BIT) != 0) {
                    picks[i] = mkfloat4(uint_as_float(cur & PRIM_INDEX_BITS), 1.0f / prob, u, uint_as_float(tag));
                    busy = false;
                }
            }
        }
    }
}

// (Round 6 also tried the pick with the NEXT rays staged in registers -- lane L holds element L of a chunk whose loads are in flight and of one
// whose positions and random numbers are computed, a finished lane takes the next ready element through a lane permutation, no memory access
// on the refill path: 16.8 ms per frame against 16.3, at 80-100 registers.  The refill rounds are not what the kernel waits for; with the table
// in LDS it is bound by the eight importance evaluations of a level -- nine quarter-rate reciprocals / square roots each.
// profiles/r06/experiments/pick_lds_staged.txt; the kernel was removed.)

// TEX = false: the scene holds no texture (ShadeLaunch::no_textures): the surface stage without its lookups, RT_FUSED_NOTEX_MIN_WAVES
#ifndef RT_FUSED_NOTEX_MIN_WAVES
#define RT_FUSED_NOTEX_MIN_WAVES 3
#endif
template <bool PRIMARY, bool SKY, bool TEX = true>
__global__ void __launch_bounds__(WAVE, TEX ? RT_FUSED_MIN_WAVES : RT_FUSED_NOTEX_MIN_WAVES) k_surface_scatter(const SceneView sc, const ShadeParams sp, const RaySoA rays_in, const HitSoA hits,
                                                                               const RayQueue in, const float4 *__restrict__ picks, const uint32_t tag,
                                                                               const PointSoA records, const RaySoA record_rays, const RayQueue out_records,
                                                                               const RaySoA rays_out, const RayQueue out_rays,
                                                                               const DeferredSoA deferred_out, const RayQueue out_deferred,
                                                                               const PixelBuffers px, const int img_w, const float mix_factor,
                                                                               const Layering layers, uint32_t *__restrict__ sky_index, const RayQueue out_sky) {
    const uint32_t n_live_chunks = in.live_chunks();
    const uint32_t fills = in.fill_counts(); // (the input queue's fill counts, in registers: no scalar load per chunk)
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!in.chunk(c, fills, stripe, slot0, n_live)) {
            continue;
        }
        const uint32_t i = slot0 + threadIdx.x; // (the whole wavefront stays in the body for the ballots)
        const bool active = threadIdx.x < n_live;
        bool continues = false, defer = false, sky = false, lit = false;
        ShadePoint pt;
        SurfaceOut so;
        Ray ray;
        ShadeParams spl = sp;
        uint32_t xy = 0;
        float4 pick = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
        VertexRandoms ahead = {};
        if (active) {
            ray = load_ray(rays_in, i, sp.plain_ior == 0u);
            const Hit hit = load_hit(hits, i);
            pick = picks[i]; // (with the ray, not after the surface stage: one round trip less on the wavefront's critical path)
            xy = ray.xy; // virtual (layered) pixel: where the pixel writes go
            const uint32_t layer = xy_layer(xy, layers);
            spl = layer_params(sp, layer);
            ray.xy = xy_real(xy, layers, layer);
            ahead = vertex_randoms(path_random(sc, spl, ray.xy, ray.depth));
            continues = surface_stage<true, SKY, TEX>(sc, spl, hit, ray, pt, so, &ahead);
            defer = so.deferred_emitter;
            sky = SKY && so.deferred_sky;
            if (PRIMARY) {
                ShadeResult res;
                res.col = continues ? f4{0.0f, 0.0f, 0.0f, 1.0f} : so.radiance;
                res.base_color = so.base_color, res.depth_normal = so.normal_depth;
                if (layers.count > 1) {
                    write_primary_pixel_layered(res, xy, img_w, px.temp, px.aux_base_layers, px.aux_dn_layers);
                } else {
                    write_primary_pixel(res, xy, img_w, mix_factor, px.temp, px.base_color, px.depth_normals);
                }
            } else if (!continues) {
                ShadeResult res;
                res.col = so.radiance;
                add_secondary_pixel(res, xy, img_w, px.temp);
            }
            lit = continues && sc.light_cwnodes_count != 0 && float_as_uint(pick.w) == tag;
        }
        // what the next-event estimation needs of a lit point, densely in the `nee` queue of its stripe
        const uint32_t r_slot = out_records.alloc(stripe, lit);
        if (lit) {
            store_point(records, r_slot, pt, r_slot); // (its "ray slot" is the record's own: record_rays holds the ray's part)
            records.light[r_slot] = pick;
            record_rays.d_cw[r_slot] = mkfloat4(ray.d.x, ray.d.y, ray.d.z, ray.cone_width);
            record_rays.c_cs[r_slot] = mkfloat4(ray.c.x, ray.c.y, ray.c.z, ray.cone_spread);
            if (sp.plain_ior == 0u) {
                record_rays.ior[r_slot] = mkfloat4(ray.ior[0], ray.ior[1], ray.ior[2], ray.ior[3]);
            }
            uint2 xd;
            xd.x = xy, xd.y = ray.depth;
            record_rays.xy_depth[r_slot] = xd;
        }
        if (SKY) { // the physical sky: paths that ended in it wait for k_shade_sky (their pixel got zeros above)
            const uint32_t s_slot = out_sky.alloc(stripe, sky);
            if (sky) {
                sky_index[s_slot] = i;
            }
        }
        if (__any(defer)) { // rare
            const uint32_t d_slot = out_deferred.alloc(stripe, defer);
            if (defer) {
                deferred_out.a[d_slot] = mkfloat4(uint_as_float(i), uint_as_float(so.emitter_triangle), uint_as_float(pt.material), so.emitter_mix_weight);
                deferred_out.b[d_slot] = mkfloat4(pt.base.x, pt.base.y, pt.base.z, 0.0f);
            }
        }
        // the continuation, from the registers
        Scatter sct;
        sct.has_next = sct.has_shadow = false;
        if (continues) {
            ray.o = pt.P, ray.pdf = 0.0f; // (what k_scatter hands the stage: neither is read by it)
            scatter_stage<false, true>(sc, spl, ray, pt, no_light_pick(), sct, &ahead);
            sct.next.xy = xy;
        }
        const uint32_t n_slot = out_rays.alloc(stripe, sct.has_next);
        if (sct.has_next) {
            store_ray(rays_out, n_slot, sct.next, sp.plain_ior == 0u);
        }
    }
}

// MIS-weighted radiance of importance-sampled emitters that secondary rays hit (shade_point.h: emissive_hit_mis_weight): a
// light-tree walk + a spherical-triangle density per hit -- rare, but every wavefront containing one used to pay for it.
// Runs after k_surface of the same bounce on the ray / hit buffers that kernel read; such a path ends there (k_surface booked
// zero radiance for it), so this is the only contribution of its pixel in this bounce and the per-pixel addition order of
// the reference is kept.  (First bounce: only camera rays that crossed a transparent surface can get here.)
__global__ void __launch_bounds__(WAVE) k_shade_emissive(const SceneView sc, const ShadeParams sp, const RaySoA rays_in, const HitSoA hits,
                                                        const DeferredSoA deferred, const RayQueue queue, const PixelBuffers px,
                                                        const int img_w) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_live_chunks = queue.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!queue.chunk(c, stripe, slot0, n_live) || lane >= n_live) {
            continue;
        }
        const float4 a = deferred.a[slot0 + lane], b = deferred.b[slot0 + lane];
        const uint32_t i = float_as_uint(a.x), tri = float_as_uint(a.y), material = float_as_uint(a.z);
        const float4 o = rays_in.o_pdf[i], d = rays_in.d_cw[i], throughput = rays_in.c_cs[i];
        const uint32_t xy = rays_in.xy_depth[i].x;
        const Hit hit = load_hit(hits, i);
        const f3 origin = {o.x, o.y, o.z}, dir = {d.x, d.y, d.z};
        const float mis = emissive_hit_mis_weight(sc, origin, dir, origin + hit.t * dir, hit.t, o.w, tri, &sc.mesh_instances[hit.obj_index]);
        ShadeResult res;
        res.col = emissive_hit_radiance(sp, a.w, mis, sc.materials[material].tangent_rotation_or_strength, f3{b.x, b.y, b.z},
                                        f3{throughput.x, throughput.y, throughput.z});
        add_secondary_pixel(res, xy, img_w, px.temp);
    }
}

// The physical sky for the paths k_surface deferred (ShadeSkyPrimary / ShadeSkySecondary, RendererCPU.h:484-486, 555-557): the analytic
// integrator of rt_sky.h per ray -- air below / inside / above the cloud layer, 48 cloud steps with a 24-step shadow march each, cirrus,
// sun disk, stars, moon: thousands of texture taps per ray, three orders of magnitude above any other per-ray work of the stage, which
// is why it has its own launch over its own queue.  One ray per pixel and layer: the pixel update needs no atomics.
__global__ void __launch_bounds__(WAVE) k_shade_sky(const SceneView sc, const ShadeParams sp, const RaySoA rays_in, const HitSoA hits,
                                                   const uint32_t *__restrict__ sky_index, const RayQueue queue, const PixelBuffers px, const int img_w,
                                                   const Layering layers) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_live_chunks = queue.live_chunks();
    ChunkWalk walk(n_live_chunks);
    for (uint32_t c; walk.next(c);) {
        uint32_t stripe, slot0, n_live;
        if (!queue.chunk(c, stripe, slot0, n_live) || lane >= n_live) {
            continue;
        }
        const uint32_t i = sky_index[slot0 + lane];
        Ray ray = load_ray(rays_in, i, sp.plain_ior == 0u);
        const Hit hit = load_hit(hits, i);
        const uint32_t xy = ray.xy, layer = xy_layer(xy, layers);
        const ShadeParams spl = layer_params(sp, layer);
        ray.xy = xy_real(xy, layers, layer);
        add_sky_pixel(shade_sky_ray(sc, ray, hit, spl.iteration, int(spl.ps.max_total_depth), spl.limits[0]), xy, img_w, px.temp);
    }
}

// (Round 3 also tried the pick with EIGHT lanes per shade point -- lane j fetches and evaluates child j, the group exchanges the
// importances and every lane runs light_level_choice; the table is row-type major for it, 128 consecutive bytes per group and
// load.  Bit-identical picks, an eighth of the L1 traffic -- and 0.73 instead of 0.34 ms per iteration: only the importance
// (40 of the ~110 instructions of a level) is shared out, the selection, the exchange and the per-point set-up run once per
// eight points instead of once per sixty-four.  profiles/r03/experiments/variants_pick_*.txt; the kernel was removed.)

// (Round 4 tried the continuation CLASS BY CLASS: k_light_pick filed every point under the branch its continuation takes -- principled /
// diffuse lobe, principled / specular lobes, Diffuse material, other -- in four index lists, and the continuation walked the lists, so that
// no wavefront ran the GGX draw for 9 of its 64 points.  Bit-identical frames, and slower: continuation 114 -> 152 ms, pick 86 -> 128 ms
// per four 64-spp frames.  The stage moves ~80 GB per frame at 3 TB/s; an index list turns every 16-byte plane read of the rarer classes
// into a 64-byte line, and that costs more than the idle lanes did.  profiles/r04/experiments/shade_by_class.txt; code: commit e0e002d.)

// ---- launcher ---------------------------------------------------------------------------------------------------------------
// blocks of `kernel` the device holds at once (one wavefront per block), per device and kernel; 0 if the runtime will not say
static int resident_blocks(const void *kernel) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, kernel);
    const auto it = cache.find(key);
    if (it != cache.end()) {
        return it->second;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE, 0) != hipSuccess || per_cu <= 0 ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        per_cu = 0;
    }
    return cache[key] = per_cu * cus;
}

void launch(const ShadeLaunch &a) {
    hipStream_t s = a.stream;
    const int g = a.grid;
    // Grid of one launch of the stage.  Round 5: sized from what its queue is EXPECTED to hold (a.expect: live chunks, from the counts the
    // previous pass left behind; 0 = unknown -> the pass's full grid) -- a.chunks_per_block live chunks per block, but at least one block
    // per wave slot of the device while there are chunks for them.  The kernels walk their queue with any grid; before, every launch
    // dispatched the full 32 k blocks of the pass (~20 us even when the queue holds a few hundred rays: the late bounces, a rank of 8).
    auto sized = [&](auto kernel, const uint32_t expect, const int cap) {
        int grid = std::min(g, cap);
        if (expect != 0u && a.chunks_per_block > 0) {
            const int resident = std::max(1, resident_blocks(reinterpret_cast<const void *>(kernel)));
            const uint32_t want = std::max(std::min(expect, uint32_t(resident)), expect / uint32_t(a.chunks_per_block));
            grid = int(std::min<uint32_t>(uint32_t(grid), std::max(want, 1u)));
        }
        return grid;
    };
    const int all = 1 << 30;
    if ((a.split & 16) != 0) { // round 6: pick first, surface + continuation in one kernel, next-event estimation over dense records
        const bool lights = a.sc.light_cwnodes_count != 0;
        if (lights) {
            const uint32_t lds_nodes = (a.pick_lds && a.sc.light_cwnodes_count <= PICK_LDS_MAX_NODES) ? a.sc.light_cwnodes_count : 0u;
            auto go = [&](auto kernel, auto kernel_wave) {
                // (grids in wavefronts, from the one-wavefront form: the LDS form launches a quarter as many blocks of four)
                int grid = sized(kernel_wave, a.expect[EXPECT_RAYS], all);
                if (a.work) {
                    const int resident = std::max(1, resident_blocks(reinterpret_cast<const void *>(kernel_wave))) * std::max(1, a.dyn_mult);
                    const uint32_t bound = a.expect[EXPECT_RAYS] != 0u ? a.expect[EXPECT_RAYS] : a.chunks;
                    grid = int(std::max<uint32_t>(1u, std::min<uint32_t>({uint32_t(g), uint32_t(resident), std::max(bound, 1u)})));
                }
                if (lds_nodes != 0u) {
                    kernel<<<(grid + PICK_BLOCK_WAVES - 1) / PICK_BLOCK_WAVES, PICK_BLOCK_WAVES * WAVE, size_t(lds_nodes) * LIGHT_CHILDREN_STRIDE * sizeof(float4), s>>>(
                        a.sc, a.sp, a.rays_in, a.hits, a.in, a.picks, a.tag, a.layers, a.work, lds_nodes);
                } else {
                    kernel_wave<<<grid, WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.hits, a.in, a.picks, a.tag, a.layers, a.work, 0u);
                }
            };
            a.work ? go(k_light_pick_first<true, true>, k_light_pick_first<true, false>) : go(k_light_pick_first<false, true>, k_light_pick_first<false, false>);
        }
#define RT_FUSED(...) k_surface_scatter<__VA_ARGS__><<<sized(k_surface_scatter<__VA_ARGS__>, a.expect[EXPECT_RAYS], all), WAVE, 0, s>>>( \
        a.sc, a.sp, a.rays_in, a.hits, a.in, a.picks, a.tag, a.points, a.record_rays, a.nee, a.rays_out, a.out_rays, a.deferred, a.out_deferred, a.px, a.vw, \
        a.mix_factor, a.layers, a.sky_index, a.out_sky)
        const bool sky_scene = a.sc.sky.desc != nullptr;
        if (a.bounce == 0) {
            if (sky_scene) {
                RT_FUSED(true, true);
            } else if (a.no_textures) {
                RT_FUSED(true, false, false);
            } else {
                RT_FUSED(true, false);
            }
        } else if (sky_scene) {
            RT_FUSED(false, true);
        } else if (a.no_textures) {
            RT_FUSED(false, false, false);
        } else {
            RT_FUSED(false, false);
        }
#undef RT_FUSED
        if (sky_scene) {
            k_shade_sky<<<sized(k_shade_sky, a.expect[EXPECT_SKY], 4096), WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.hits, a.sky_index, a.out_sky, a.px, a.vw, a.layers);
        }
        k_shade_emissive<<<sized(k_shade_emissive, a.expect[EXPECT_DEFERRED], 2048), WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.hits, a.deferred, a.out_deferred, a.px, a.vw);
        if (lights) { // the records name themselves as their ray slot: the ray's part comes from record_rays
            k_scatter<true, false><<<sized(k_scatter<true, false>, a.expect[EXPECT_LIT], all), WAVE, 0, s>>>(
                a.sc, a.sp, a.record_rays, a.points, a.nee, a.rays_out, a.out_rays, a.shadow, a.out_shadow, a.px, a.vw, a.layers, a.pts, a.nee);
        }
        return;
    }
    const bool pick_apart = (a.split & 1) != 0 && a.sc.light_cwnodes_count != 0;
    // stage 1: what was hit (SKY: the environment is the physical sky -- narrow rays that leave the scene are queued for k_shade_sky)
#define RT_SURFACE_ARGS a.sc, a.sp, a.rays_in, a.hits, a.in, a.points, a.pts, a.deferred, a.out_deferred, a.px, a.vw, a.mix_factor, a.layers, a.sky_index, a.out_sky
#define RT_SURFACE(...) k_surface<__VA_ARGS__><<<sized(k_surface<__VA_ARGS__>, a.expect[EXPECT_RAYS], all), WAVE, 0, s>>>(RT_SURFACE_ARGS)
    const bool sky = a.sc.sky.desc != nullptr;
    if (a.bounce == 0) {
        if (sky) {
            if (pick_apart) {
                RT_SURFACE(true, false, true);
            } else {
                RT_SURFACE(true, true, true);
            }
        } else if (pick_apart) {
            RT_SURFACE(true, false);
        } else {
            RT_SURFACE(true, true);
        }
    } else {
        if (sky) {
            if (pick_apart) {
                RT_SURFACE(false, false, true);
            } else {
                RT_SURFACE(false, true, true);
            }
        } else if (pick_apart) {
            RT_SURFACE(false, false);
        } else {
            RT_SURFACE(false, true);
        }
    }
#undef RT_SURFACE
#undef RT_SURFACE_ARGS
    if (a.sc.sky.desc != nullptr) { // paths that ended in the physical sky: diffuse bounces and wide cones never defer to it, so the queue is
                                    // often empty -- the grid is capped like the emissive kernel's
        k_shade_sky<<<sized(k_shade_sky, a.expect[EXPECT_SKY], 4096), WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.hits, a.sky_index, a.out_sky, a.px, a.vw, a.layers);
    }
    // emitter hits whose MIS weight was deferred; an empty queue costs a few microseconds
    k_shade_emissive<<<sized(k_shade_emissive, a.expect[EXPECT_DEFERRED], 2048), WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.hits, a.deferred, a.out_deferred, a.px, a.vw);
    // stage 2: which light
    const bool nee_compact = pick_apart && (a.split & 4) != 0;
    const bool pick_refill = (a.split & 8) != 0;
    if (pick_apart && pick_refill) {
        // the persistent pick: with a work counter its chunks are handed out dynamically (wavefront.hip.h: ChunkWalk) and the grid is what the
        // device holds at once
        auto go = [&](auto kernel) {
            int grid = sized(kernel, a.expect[EXPECT_POINTS], all);
            if (a.work) {
                const int resident = std::max(1, resident_blocks(reinterpret_cast<const void *>(kernel))) * std::max(1, a.dyn_mult);
                const uint32_t bound = a.expect[EXPECT_POINTS] != 0u ? a.expect[EXPECT_POINTS] : a.chunks;
                grid = int(std::max<uint32_t>(1u, std::min<uint32_t>({uint32_t(g), uint32_t(resident), std::max(bound, 1u)})));
            }
            kernel<<<grid, WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.points, a.pts, a.nee, a.layers, a.work);
        };
        if (a.work) {
            nee_compact ? go(k_light_pick_refill<true, true>) : go(k_light_pick_refill<false, true>);
        } else {
            nee_compact ? go(k_light_pick_refill<true, false>) : go(k_light_pick_refill<false, false>);
        }
    } else if (nee_compact) {
        k_light_pick<true><<<sized(k_light_pick<true>, a.expect[EXPECT_POINTS], all), WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.points, a.pts, a.nee, a.layers);
    } else if (pick_apart) {
        k_light_pick<false><<<sized(k_light_pick<false>, a.expect[EXPECT_POINTS], all), WAVE, 0, s>>>(a.sc, a.sp, a.rays_in, a.points, a.pts, a.nee, a.layers);
    }
    // stage 3: shadow ray + continuation
#define RT_SCATTER_ARGS(queue) a.sc, a.sp, a.rays_in, a.points, queue, a.rays_out, a.out_rays, a.shadow, a.out_shadow, a.px, a.vw, a.layers, a.pts, a.nee
#define RT_SCATTER(queue, expect, ...) k_scatter<__VA_ARGS__><<<sized(k_scatter<__VA_ARGS__>, expect, all), WAVE, 0, s>>>(RT_SCATTER_ARGS(queue))
    if (nee_compact) {
        // the next-event estimation runs over the points that have a light to sample -- full wavefronts instead of the 11 of 64
        // lanes that take that branch in the combined kernel (Bistro-class scene) -- the continuation over all points; or, when
        // most points are lit, the combined kernel (lit_points_are_sparse)
        RT_SCATTER(a.nee, a.expect[EXPECT_LIT], true, false, true, 1);
        RT_SCATTER(a.pts, a.expect[EXPECT_POINTS], false, true, false, 1);
        RT_SCATTER(a.pts, a.expect[EXPECT_POINTS], true, true, false, 2);
    } else if ((a.split & 2) != 0) {
        RT_SCATTER(a.pts, a.expect[EXPECT_POINTS], true, false);
        RT_SCATTER(a.pts, a.expect[EXPECT_POINTS], false, true);
    } else {
        RT_SCATTER(a.pts, a.expect[EXPECT_POINTS], true, true);
    }
#undef RT_SCATTER
#undef RT_SCATTER_ARGS
}

} // namespace shade
} // namespace rt

#ifdef RT_PROFILE_SHADE
extern "C" __attribute__((visibility("default"))) int rayhip_tuning_read_shade_profile(unsigned long long out[64], int reset) {
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_prof_shade), 64 * sizeof(unsigned long long)) != hipSuccess) {
        return 1;
    }
    if (reset) {
        unsigned long long z[64] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(rt::g_prof_shade), z, sizeof(z)) != hipSuccess;
    }
    return 0;
}
#endif
