// shade_launch.h -- the interface between the host launcher (rayhip.hip) and the shade-stage translation unit
// (shade_kernels.hip).
#pragma once

#include "rt_shade.h"
#include "wavefront.hip.h"

namespace rt {

// ShadePoint in memory: SoA float4 planes, one store / load instruction per plane and wavefront (1 KiB each)
struct PointSoA {
    float4 *p_slot;  // P.xyz | slot of the ray that produced the point
    float4 *n_gx;    // N.xyz | plane_N.x
    float4 *b_gy;    // B.xyz | plane_N.y
    float4 *base_gz; // base.rgb | plane_N.z
    float4 *scalars; // roughness, metallic, specular, mix_weight
    float4 *misc;    // mix_pick, material | backfacing << 31, cone_width, -
    float4 *light;   // written by k_light_pick: light index, 1 / pick probability, rest of the random number, -
    uint32_t *nee_index; // written by k_light_pick: the points that got a light, densely (slots of the `nee` queue -> point slots)
};

// one bounce of K5: shade the rays of queue `in` (ray buffer rays_in + hits) -> shade points (pts) -> secondary rays into
// out_rays of rays_out, shadow rays into out_shadow, radiance into the per-iteration pixel buffer
struct ShadeLaunch {
    SceneView sc;
    ShadeParams sp;
    RaySoA rays_in, rays_out;
    HitSoA hits;
    ShadowSoA shadow;
    DeferredSoA deferred;
    PointSoA points;
    RayQueue in, pts, out_rays, out_shadow, out_deferred, nee, out_sky;
    uint32_t *sky_index; // ray slots of the paths that ended in the physical sky (k_surface -> k_shade_sky), densely per stripe
    PixelBuffers px;
    Layering layers;
    int vw;           // row pitch of the per-iteration pixel buffers (virtual frame width)
    float mix_factor; // 1 / iteration: blend of the first-hit feature images (single-layer passes)
    int bounce, grid;
    // round 5: grids sized from what the queues are expected to hold (live chunks of 64, from the counts of the previous pass; 0: unknown)
    uint32_t expect[5] = {0, 0, 0, 0, 0}; // EXPECT_*
    int chunks_per_block = 0;             // live chunks a block of a streaming kernel should find (0: the pass's full grid, as before round 5)
    uint32_t *work = nullptr;             // the persistent light pick: zeroed counter of the dynamic chunk hand-out (wavefront.hip.h), or null
    uint32_t chunks = 0;                  // upper bound of the chunks a queue of the pass holds (slots / 64)
    int dyn_mult = 1;                     // with `work`: blocks per resident wave slot
    // round 6 (split bit 4): the light pick runs first and leaves its picks in a plane indexed by RAY slot, stamped with `tag` (unique per launch
    // of the stage, never 0; the plane starts zeroed); k_surface_scatter lists what the next-event estimation needs of a lit point in the point
    // planes + `record_rays` (direction | cone width, throughput | cone spread, ior stack, pixel | depth: the planes of a RaySoA, o_pdf unused)
    bool no_textures = false; // the uploaded scene holds no texture at all: k_surface_scatter without the lookups (shade_point.h: TEX)
    bool pick_lds = true; // k_light_pick_first keeps the top of the light table in LDS (RAYHIP_PICK_LDS=0: every row from memory)
    float4 *picks = nullptr;
    uint32_t tag = 0;
    RaySoA record_rays = {};
    int split;        // bit 4: pick -> surface + continuation in one kernel -> next-event estimation over dense records (shade_kernels.hip, round 6);
                      // otherwise  bit 0: the light pick as its own kernel; bit 1: next-event estimation and continuation as two launches;
                      // bit 2 (with bit 0): next-event estimation as its own launch over the points that GOT a light, densely packed;
                      // bit 3 (with bit 0): the light pick as a persistent kernel whose lanes take the next point when theirs is through
    hipStream_t stream;
};
enum { EXPECT_RAYS = 0, EXPECT_POINTS, EXPECT_LIT, EXPECT_DEFERRED, EXPECT_SKY };
namespace shade {
void launch(const ShadeLaunch &a);
}

} // namespace rt
