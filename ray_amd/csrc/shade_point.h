// shade_point.h -- stage 1 of shading: from a (ray, hit) pair to either a finished radiance value or a SHADE POINT.
//
// The shade stage of the path (reference: ShadeSurface, internal/ShadeRef.cpp:1174-1652) is cut where its working set
// changes (kernels.hip.h runs the cuts as separate kernels, tests/hostsim as consecutive calls):
//
//   1 surface   resolve what was hit.  Misses, emitter hits, culled back faces and emissive materials end here with a
//               radiance value.  Everything else becomes a ShadePoint: the world-space shading frame, the material after its
//               mix nodes were resolved stochastically, the texture-modulated parameters.  Needs: the triangle's vertices,
//               the instance transform, the material chain, textures.
//   2 pick      (shade_lights.h) light-tree descent for the point's position.
//   3 scatter   (shade_lobes.h) sample the picked light and evaluate the material towards it (shadow ray), draw the
//               continuation direction from the material (secondary ray), Russian roulette.
//
// A ShadePoint is 112 bytes in seven SoA planes -- what stage 3 cannot recompute from the ray it re-reads: position, two
// frame vectors + the geometric normal (the tangent is N x B), base colour, three scalars, the mix bookkeeping, the
// material index.
//
// Order-of-operations source for the arithmetic (bit-exact parity on the host build): ShadeRef.cpp:1030-1172 (environment
// and emitter radiance), :1174-1470 (surface set-up), :1500-1525 (emissive hit MIS).
#pragma once

#include "shade_lights.h"
#include "shade_math.h"

namespace rt {

struct PassLimits {
    int max_diff_depth, max_spec_depth, max_refr_depth, max_transp_depth, max_total_depth;
    int min_total_depth, min_transp_depth;
    float regularize_alpha;
};
struct ShadeParams {
    PassLimits ps;
    float limits[2]; // {direct, indirect} clamp on the rgb SUM (3 x clamp) or FLT_MAX
    uint32_t rand_seed;
    int iteration;
    // 1: no surface of the scene can change a ray's stack of refractive indices (no Refractive node, no Principled transmission) and the
    // rays of this pass come from the camera (stack = four times -1, rt_raygen.h): the scatter stage takes that constant instead of
    // reading the ior plane of the ray and does not write the plane of the ray it spawns -- 48 of the ~450 bytes a path vertex moves
    // through the stage.  Set by render_pass only (kernel-level hooks are handed rays by the caller: 0).
    uint32_t plain_ior;
};

// the per-path random stream: dimensions of this bounce, keyed by pixel and iteration
struct PathRandom {
    uint32_t hash, dim;
    int sample;
    const uint32_t *table;
    RT_HD f2 get(const int which) const { return get_scrambled_2d_rand(dim + uint32_t(which), hash, sample, table); }
};
RT_HD PathRandom path_random(const SceneView &sc, const ShadeParams &sp, const uint32_t xy, const uint32_t depth) {
    return PathRandom{hash_combine(hash(xy), sp.rand_seed), uint32_t(RAND_DIM_BASE_COUNT + get_total_depth(depth) * RAND_DIM_BOUNCE_COUNT),
                      sp.iteration - 1, sc.pmj};
}

// random pairs of a path vertex a caller fetched ahead of the stages (k_surface_scatter: the two table reads per pair are then on their way while
// the triangle rows are, instead of behind the material chain); the values are what the stages would compute themselves
struct VertexRandoms {
    f2 bsdf_pick, bsdf;
};
RT_HD VertexRandoms vertex_randoms(const PathRandom &rnd) { return VertexRandoms{rnd.get(RAND_DIM_BSDF_PICK), rnd.get(RAND_DIM_BSDF)}; }

struct ShadePoint {
    f3 P;              // world position
    f3 N, B, plane_N;  // shading normal, bitangent, geometric normal (unit, world space, flipped towards the ray)
    f3 base;           // base colour x base texture
    float roughness, metallic, specular; // x their textures (metallic / specular: Principled only)
    float mix_weight;  // product of the mix_add compensations on the way down the mix chain
    float mix_pick;    // what is left of the mix random number: picks the Principled lobe
    uint32_t material; // index of the resolved (non-mix) material
    bool backfacing;
    float cone_width;  // ray-cone footprint at the hit
};
RT_HD f3 tangent_of(const ShadePoint &pt) { return cross(pt.N, pt.B); }

// what stage 1 hands to the pixel: radiance when the path ends here, the first-hit feature images always
struct SurfaceOut {
    f4 radiance;       // rgb + coverage; meaningful when the stage returns false
    f3 base_color;     // first-hit albedo / normal + depth (primary stage: the denoiser's guides)
    f4 normal_depth;
    // an importance-sampled emitter hit by a BSDF-sampled ray: `radiance` does not hold its MIS-weighted value; the caller runs
    // emissive_hit_radiance later (its own kernel on the device: rare, but a light-tree walk per hit)
    bool deferred_emitter;
    uint32_t emitter_triangle;
    float emitter_mix_weight;
    // the ray left the scene into a physical sky and is narrower than the baked map resolves: `radiance` is all zero, the caller runs
    // shade_sky_ray (rt_sky.h) on it later and adds the result to the pixel (ShadeRef.cpp:1192-1196 -> ShadeSky, AtmosphereRef.cpp:928)
    bool deferred_sky;
};

// `tri_verts` table: what the surface stage needs of a triangle, gathered through vtx_indices[] once per scene.  One row
// of 8 float4 = 128 bytes = exactly two 64-byte lines per triangle, fetched in ONE round trip (the reference follows three
// indices to three 44-byte vertices: 3 + 33 dword loads, two dependent trips):
//   [2k]   vertex k: position, normal.x        [2k+1]  vertex k: normal.yz, uv
//   [6]    object-space geometric normal (unit) and the length of the edge cross product (twice the area)
//   [7]    twice the area of the triangle in uv space, the triangle's material indices (front | back << 16: tri_materials[], which would be one
//          more gather of a 4-byte entry in its own line), -, -
// Rows 6-7 are values every shade point of the triangle would compute from the corners with the same operations (same
// IEEE results on host and device); the vertex bitangents, which only normal-mapped materials read, live in their own
// table (`tri_bitangents`, 4 float4 per triangle: b0, b1, b2, -).  Pure data movement otherwise.
constexpr int TRI_VERTS_STRIDE = 8, TRI_BITANGENTS_STRIDE = 4;
struct Corner {
    f3 p, n;
    f2 uv;
};
RT_HD void fill_tri_verts(const rayhip_vertex *vertices, const uint32_t vertices_count, const uint32_t *vtx_indices, const uint32_t tri,
                          const rayhip_tri_mat_data *tri_materials, const uint32_t tri_materials_count,
                          float4 *out /* [TRI_VERTS_STRIDE] */, float4 *out_bitangents /* [TRI_BITANGENTS_STRIDE] */) {
    for (int k = 0; k < TRI_VERTS_STRIDE; ++k) {
        out[k] = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (tri < tri_materials_count) {
        out[7].y = uint_as_float(uint32_t(tri_materials[tri].front_mi) | (uint32_t(tri_materials[tri].back_mi) << 16));
    }
    for (int k = 0; k < TRI_BITANGENTS_STRIDE; ++k) {
        out_bitangents[k] = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // (vtx_indices is a sparse pool: slots no mesh owns hold anything -- those rows stay zero, nothing reads them)
    for (int k = 0; k < 3; ++k) {
        if (vtx_indices[tri * 3 + k] >= vertices_count) {
            return;
        }
    }
    const rayhip_vertex *v[3] = {&vertices[vtx_indices[tri * 3 + 0]], &vertices[vtx_indices[tri * 3 + 1]], &vertices[vtx_indices[tri * 3 + 2]]};
    for (int k = 0; k < 3; ++k) {
        out[2 * k + 0] = mkfloat4(v[k]->p[0], v[k]->p[1], v[k]->p[2], v[k]->n[0]);
        out[2 * k + 1] = mkfloat4(v[k]->n[1], v[k]->n[2], v[k]->t[0], v[k]->t[1]);
        out_bitangents[k] = mkfloat4(v[k]->b[0], v[k]->b[1], v[k]->b[2], 0.0f);
    }
    float twice_area;
    const f3 ng = normalize_len(cross(mk3(v[1]->p) - mk3(v[0]->p), mk3(v[2]->p) - mk3(v[0]->p)), twice_area);
    out[6] = mkfloat4(ng.x, ng.y, ng.z, twice_area);
    const float uv_area = fabsf((v[1]->t[0] - v[0]->t[0]) * (v[2]->t[1] - v[0]->t[1]) - (v[2]->t[0] - v[0]->t[0]) * (v[1]->t[1] - v[0]->t[1]));
    out[7].x = uv_area;
}
RT_HD Corner load_corner(const float4 *t) {
    const float4 a = t[0], b = t[1];
    return Corner{f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f2{b.z, b.w}};
}

// hue-preserving clamp on the rgb sum (ShadeRef.cpp:1646-1649)
RT_HD f3 clamp_radiance_sum(f3 c, const float limit) {
    const float sum = hsum(mk4(c, 0.0f));
    if (sum > limit) {
        c *= (limit / sum);
    }
    return c;
}

// ---- terminal: the ray left the scene (ShadeRef.cpp:1030-1066) ----------------------------------------------------------------
// `inv_pick_prob` < 0: no MIS against next-event estimation (the path could not have continued)
// TEX = false: the scene has no texture at all (no environment map either) -- the lookups are compiled out (see surface_stage)
template <bool TEX = true, class Jitter>
RT_HD f4 environment_radiance(const SceneView &sc, const Ray &ray, const float inv_pick_prob, Jitter &&jitter) {
    const rayhip_environment &env = sc.env;
    const bool indirect = is_indirect(ray.depth);
    const uint32_t map = indirect ? env.env_map : env.back_map;
    const float rotation = indirect ? env.env_map_rotation : env.back_map_rotation;
    f4 c = {1.0f, 1.0f, 1.0f, 1.0f};
    if (TEX && map != 0xffffffff) {
        c = mk4(latlong_rgbe(sc, map, ray.d, rotation, jitter()), 1.0f);
    }
    if (env.light_index != 0xffffffff && inv_pick_prob >= 0.0f && indirect) {
        const float light_pdf = env.qtree_levels ? safe_div_pos(env_quadtree_pdf(sc, rotation, ray.d), inv_pick_prob)
                                                 : safe_div_pos(0.5f, PI * inv_pick_prob);
        c *= power_heuristic(ray.pdf, light_pdf);
    }
    c *= indirect ? mk4(env.env_col[0], env.env_col[1], env.env_col[2], 1.0f) : mk4(env.back_col[0], env.back_col[1], env.back_col[2], 1.0f);
    c.w = 1.0f;
    return c;
}

// ---- terminal: the ray hit a visible analytic emitter (ShadeRef.cpp:1068-1172) ------------------------------------------------------
// hit.u carries the pick probability of that emitter as seen from the ray origin (rt_arealights.h); the emission is weighted
// against the density next-event estimation has for the same direction
template <class Jitter>
RT_HD f3 emitter_radiance(const SceneView &sc, const Ray &ray, const Hit &hit, Jitter &&jitter) {
    const rayhip_light &l = sc.lights[-hit.obj_index - 1];
    const float inv_pick_prob = (1.0f / hit.u);
    f3 c = mk3(l.col);
    if (light_sky_portal(l)) {
        c *= env_radiance_towards(sc, ray.d, jitter());
    }
    float nee_pdf = 0.0f;
    bool weighted = true;
    switch (light_type(l)) {
    case LIGHT_TYPE_SPHERE: {
        const f3 centre = mk3(&l.params[0]);
        const float radius = l.params[7];
        float d;
        const f3 towards = normalize_len(centre - ray.o, d);
        weighted = false;
        if (d > radius) {
            const float tangent_len = sqrtf(d * d - radius * radius);
            const float disk_radius = (tangent_len * radius) / d;
            float disk_dist = dot(ray.o, towards) - dot(centre, towards);
            const float disk_area = PI * disk_radius * disk_radius;
            const float cos_theta = dot(ray.d, towards);
            disk_dist /= cos_theta;
            nee_pdf = (disk_dist * disk_dist) / (disk_area * cos_theta * inv_pick_prob);
            c *= power_heuristic(ray.pdf, nee_pdf);
            const float spot = l.params[8], blend = l.params[9];
            if (spot > 0.0f && blend > 0.0f) {
                const float angle = acosf(saturatef(-dot(ray.d, mk3(&l.params[4]))));
                c *= saturatef((spot - angle) / blend);
            }
        }
    } break;
    case LIGHT_TYPE_DIR: {
        const float radius = l.params[4];
        const float cone_area = PI * radius * radius;
        nee_pdf = 1.0f / (cone_area * dot(ray.d, mk3(&l.params[0])) * inv_pick_prob);
    } break;
    case LIGHT_TYPE_RECT: {
        const f3 side_u = mk3(&l.params[4]), side_v = mk3(&l.params[8]);
        nee_pdf = solid_angle_rect(ray.o, mk3(&l.params[0]), side_u, side_v, f2{0.0f, 0.0f}, nullptr) / inv_pick_prob;
        if (nee_pdf == 0.0f) {
            const float cos_theta = dot(ray.d, normalize(cross(side_u, side_v)));
            nee_pdf = (hit.t * hit.t) / (l.params[3] * cos_theta * inv_pick_prob);
        }
    } break;
    case LIGHT_TYPE_DISK: {
        const float cos_theta = dot(ray.d, normalize(cross(mk3(&l.params[4]), mk3(&l.params[8]))));
        nee_pdf = (hit.t * hit.t) / (l.params[3] * cos_theta * inv_pick_prob);
    } break;
    case LIGHT_TYPE_LINE: {
        const float cos_theta = 1.0f - fabsf(dot(ray.d, mk3(&l.params[8])));
        nee_pdf = (hit.t * hit.t) / (l.params[3] * cos_theta * inv_pick_prob);
    } break;
    default:
        weighted = false;
        break;
    }
    if (weighted) {
        c *= power_heuristic(ray.pdf, nee_pdf);
    }
    return c;
}

// ---- terminal: an importance-sampled emissive triangle hit by a BSDF-sampled ray (ShadeRef.cpp:1500-1525) --------------------------------
// MIS weight of the hit against the density with which next-event estimation samples that very triangle from the ray origin
RT_HD float emissive_hit_mis_weight(const SceneView &sc, const f3 origin, const f3 dir, const f3 P, const float t, const float bsdf_pdf,
                                    const uint32_t tri, const rayhip_mesh_instance *inst) {
    const float inv_pick_prob = triangle_light_inv_pick_prob(sc, P, origin, tri);
    const f3 p1 = mk3(sc.vertices[sc.vtx_indices[tri * 3 + 0]].p), p2 = mk3(sc.vertices[sc.vtx_indices[tri * 3 + 1]].p),
             p3 = mk3(sc.vertices[sc.vtx_indices[tri * 3 + 2]].p);
    float twice_area;
    const f3 facing = normalize_len(transform_direction(cross(p2 - p1, p3 - p1), inst->xform), twice_area);
    const float cos_theta = fabsf(dot(dir, facing)); // emissive triangles emit from both sides
    if (!(cos_theta > 0.0f)) {
        return 1.0f;
    }
    float nee_pdf = solid_angle_triangle(transform_point(origin, inst->inv_xform), p1, p2, p3, f2{0.0f, 0.0f}, nullptr) / inv_pick_prob;
    if (nee_pdf == 0.0f) {
        nee_pdf = (t * t) / (0.5f * twice_area * cos_theta * inv_pick_prob);
    }
    return power_heuristic(bsdf_pdf, nee_pdf);
}
// emission * weights * path throughput, clamped like every indirect contribution
RT_HD f4 emissive_hit_radiance(const ShadeParams &sp, const float mix_weight, const float mis_weight, const float strength, const f3 base,
                               const f3 throughput) {
    f3 c = {0.0f, 0.0f, 0.0f};
    c += mix_weight * mis_weight * strength * base;
    c *= throughput;
    return mk4(clamp_radiance_sum(c, sp.limits[1]), 1.0f);
}

// ---- the surface stage -----------------------------------------------------------------------------------------------------------
// Returns true when `pt` was filled (the path continues through stages 2 and 3), false when the path ends with out.radiance.
// DEFER_EMITTERS: leave the MIS weight of emitter hits to the caller (see SurfaceOut).
// SKY: the environment may be the physical sky (false: the test is compiled out -- the device picks the kernel per scene).
// TEX (round 6): false when the uploaded scene holds NO texture (SceneView::tex_flags / textures_count: decided at upload, exact) -- every texture
// lookup of the stage is compiled out, which is worth 20 registers to k_surface_scatter (175 -> 153 at its peak).
template <bool DEFER_EMITTERS, bool SKY = true, bool TEX = true>
RT_HD bool surface_stage(const SceneView &sc, const ShadeParams &sp, const Hit &hit, const Ray &ray, ShadePoint &pt, SurfaceOut &out,
                         const VertexRandoms *ahead = nullptr) {
    out.radiance = f4{0.0f, 0.0f, 0.0f, 0.0f};
    out.base_color = f3{0.0f, 0.0f, 0.0f};
    out.normal_depth = f4{0.0f, 0.0f, 0.0f, 0.0f};
    out.deferred_emitter = false;
    out.deferred_sky = false;

    const PathRandom rnd = path_random(sc, sp, ray.xy, ray.depth);
    // the texture-lookup random pair of this path vertex: computed where a lookup happens (an integer hash chain + two
    // table reads that untextured surfaces never need); every use gets the same pair
    struct {
        const PathRandom &rnd;
        bool have;
        f2 v;
        RT_HD f2 operator()() {
            if (!have) {
                v = rnd.get(RAND_DIM_TEX), have = true;
            }
            return v;
        }
    } tex_jitter = {rnd, false, f2{0.0f, 0.0f}};

    if (hit.v < 0.0f) { // nothing hit
        if (SKY && sc.sky.desc != nullptr && ray.cone_spread < sc.env.sky_map_spread_angle) {
            out.deferred_sky = true;
            return false;
        }
        const float inv_pick_prob = (get_total_depth(ray.depth) < sp.ps.max_total_depth) ? safe_div_pos(1.0f, hit.u) : -1.0f;
        f4 c = environment_radiance<TEX>(sc, ray, inv_pick_prob, tex_jitter);
        c *= mk4(ray.c.x, ray.c.y, ray.c.z, 0.0f);
        const float sum = hsum(c);
        if (sum > sp.limits[0]) {
            c *= (sp.limits[0] / sum);
        }
        out.radiance = c;
        return false;
    }
    if (hit.obj_index < 0) { // a visible analytic emitter
        f3 c = emitter_radiance(sc, ray, hit, tex_jitter);
        c *= ray.c;
        out.radiance = mk4(clamp_radiance_sum(c, sp.limits[0]), 1.0f);
        return false;
    }

    // ---- the triangle: interpolate, orient, take to world space ----
    const f3 view = ray.d; // points INTO the surface
    pt.P = ray.o + hit.t * view;
    pt.backfacing = (hit.prim_index < 0);
    const uint32_t tri = pt.backfacing ? uint32_t(-hit.prim_index - 1) : uint32_t(hit.prim_index);
    const rayhip_mesh_instance *inst = &sc.mesh_instances[hit.obj_index];
    const float4 *rows = sc.tri_verts + size_t(tri) * TRI_VERTS_STRIDE;
    const Corner c1 = load_corner(rows), c2 = load_corner(rows + 2), c3 = load_corner(rows + 4);
    const float4 plane = rows[6];
    const float4 row7 = rows[7];
    const float uv_area = row7.x;
    const rayhip_tri_mat_data sides = {uint16_t(float_as_uint(row7.y) & 0xffffu), uint16_t(float_as_uint(row7.y) >> 16)}; // == sc.tri_materials[tri]

    const float w1 = 1.0f - hit.u - hit.v;
    const f3 N_obj = normalize(c1.n * w1 + c2.n * hit.u + c3.n * hit.v);
    const f2 uv = c1.uv * w1 + c2.uv * hit.u + c3.uv * hit.v;
    const float twice_area_obj = plane.w;
    f3 Ng = {plane.x, plane.y, plane.z}, N = N_obj;

    const rayhip_material *mat = &sc.materials[sides.front_mi & MATERIAL_INDEX_BITS];
    if (pt.backfacing) {
        if (sides.back_mi == 0xffff) {
            return false; // single-sided: nothing there from this side (radiance and coverage stay zero)
        }
        mat = &sc.materials[sides.back_mi & MATERIAL_INDEX_BITS];
        Ng = -Ng, N = -N;
    }
    Ng = safe_normalize(transform_normal(Ng, inst->inv_xform));
    N = safe_normalize(transform_normal(N, inst->inv_xform));

    // texture level of detail from the ray-cone footprint: uv area over surface area, times the cone width
    pt.cone_width = ray.cone_width + ray.cone_spread * hit.t;
    float lod_lambda = 0.5f * fast_log2(uv_area / twice_area_obj);
    lod_lambda += fast_log2(pt.cone_width);

    const float outside_ior = peek_ior_stack(ray.ior, pt.backfacing);

    // ---- mix nodes: one random number walks down the chain, re-stretched at every node ----
    const f2 pick = ahead ? ahead->bsdf_pick : rnd.get(RAND_DIM_BSDF_PICK);
    float mix_u = pick.x;
    pt.mix_weight = 1.0f;
    while (mat->type == NODE_MIX) {
        float k = mat->tangent_rotation_or_strength;
        if (TEX && mat->textures[BASE_TEXTURE] != 0xffffffff) {
            k *= sample_color(sc, mat->textures[BASE_TEXTURE], uv, 0, tex_jitter()).x;
        }
        const float eta = pt.backfacing ? safe_div_pos(outside_ior, mat->ior) : safe_div_pos(mat->ior, outside_ior);
        const float fresnel = mat->ior != 0.0f ? fresnel_dielectric(dot(view, N), eta) : 1.0f;
        k *= saturatef(fresnel);
        const bool additive = (mat->flags & MAT_FLAG_MIX_ADD) != 0;
        if (mix_u > k) {
            pt.mix_weight *= additive ? 1.0f / (1.0f - k) : 1.0f;
            mat = &sc.materials[mat->textures[MIX_MAT1]];
            mix_u = safe_div_pos(mix_u - k, 1.0f - k);
        } else {
            pt.mix_weight *= additive ? 1.0f / k : 1.0f;
            mat = &sc.materials[mat->textures[MIX_MAT2]];
            mix_u = safe_div_pos(mix_u, k);
        }
    }
    pt.mix_pick = mix_u;
    pt.material = uint32_t(mat - sc.materials);

    // ---- normal map, bent back above the horizon of the view direction ----
    if (TEX && mat->textures[NORMALS_TEXTURE] != 0xffffffff) {
        f4 nm = sample_bilinear(sc, mat->textures[NORMALS_TEXTURE], uv, 0, tex_jitter());
        nm = nm * 2.0f;
        nm = {nm.x - 1.0f, nm.y - 1.0f, nm.z - 1.0f, nm.w - 1.0f};
        nm.z = 1.0f;
        if (mat->textures[NORMALS_TEXTURE] & TEX_RECONSTRUCT_Z_BIT) {
            nm.z = safe_sqrt(1.0f - nm.x * nm.x - nm.y * nm.y);
        }
        // the vertex tangent frame is only needed here: interpolated bitangent, tangent = B x N in object space, both
        // oriented and taken to world space like the normals
        const float4 *brow = sc.tri_bitangents + size_t(tri) * TRI_BITANGENTS_STRIDE;
        const float4 b1 = brow[0], b2 = brow[1], b3 = brow[2];
        f3 Bt = f3{b1.x, b1.y, b1.z} * w1 + f3{b2.x, b2.y, b2.z} * hit.u + f3{b3.x, b3.y, b3.z} * hit.v;
        f3 Tg = cross(Bt, N_obj);
        if (pt.backfacing) {
            Bt = -Bt, Tg = -Tg;
        }
        Bt = safe_normalize(transform_normal(Bt, inst->inv_xform));
        Tg = safe_normalize(transform_normal(Tg, inst->inv_xform));
        const f3 smooth = N;
        N = normalize(nm.x * Tg + nm.z * N + nm.y * Bt);
        if (mat->normal_map_strength_unorm != 0xffff) {
            N = normalize(smooth + (N - smooth) * (float(mat->normal_map_strength_unorm) / 65535.0f));
        }
        N = keep_reflection_above_surface(Ng, -view, N);
    }

    // ---- anisotropy frame: the radial direction around the object's Y axis, optionally rotated about N ----
    const f3 P_obj = c1.p * w1 + c2.p * hit.u + c3.p * hit.v;
    f3 radial = transform_normal(f3{-P_obj.z, 0.0f, P_obj.x}, inst->inv_xform);
    if (length2(cross(radial, N)) == 0.0f) {
        radial = transform_normal(P_obj, inst->inv_xform);
    }
    if (mat->tangent_rotation_or_strength != 0.0f) {
        radial = rotate_about_axis(radial, N, mat->tangent_rotation_or_strength);
    }
    pt.N = N;
    pt.plane_N = Ng;
    pt.B = safe_normalize(cross(radial, N));

    // ---- texture-modulated parameters ----
    pt.base = mk3(mat->base_color);
    if (TEX && mat->textures[BASE_TEXTURE] != 0xffffffff) {
        const uint32_t tex = mat->textures[BASE_TEXTURE];
        pt.base *= xyz(sample_color(sc, tex, uv, int(get_texture_lod(sc, tex, lod_lambda)), tex_jitter()));
    }
    out.base_color = pt.base;
    out.normal_depth = mk4(N, hit.t);

    pt.roughness = float(mat->roughness_unorm) / 65535.0f;
    if (TEX && mat->textures[ROUGH_TEXTURE] != 0xffffffff) {
        const uint32_t tex = mat->textures[ROUGH_TEXTURE];
        const float r = sample_bilinear(sc, tex, uv, int(get_texture_lod(sc, tex, lod_lambda)), tex_jitter()).x;
        f4 splat = {r, r, r, r};
        if (tex & TEX_SRGB_BIT) {
            splat = srgb_to_linear(splat);
        }
        pt.roughness *= splat.x;
    }
    pt.metallic = pt.specular = 0.0f;
    if (mat->type == NODE_PRINCIPLED) {
        pt.metallic = float(mat->metallic_unorm) / 65535.0f;
        if (TEX && mat->textures[METALLIC_TEXTURE] != 0xffffffff) {
            const uint32_t tex = mat->textures[METALLIC_TEXTURE];
            pt.metallic *= sample_bilinear(sc, tex, uv, int(get_texture_lod(sc, tex, lod_lambda)), tex_jitter()).x;
        }
        pt.specular = float(mat->specular_unorm) / 65535.0f;
        if (TEX && mat->textures[SPECULAR_TEXTURE] != 0xffffffff) {
            const uint32_t tex = mat->textures[SPECULAR_TEXTURE];
            f4 s = sample_bilinear(sc, tex, uv, int(get_texture_lod(sc, tex, lod_lambda)), tex_jitter());
            if (tex & TEX_SRGB_BIT) {
                s = srgb_to_linear(s);
            }
            pt.specular *= s.x;
        }
    }

    if (mat->type != NODE_EMISSIVE) {
        return true;
    }
    // ---- an emissive surface ends the path ----
    float mis = 1.0f;
    if ((ray.depth & 0x00ffffff) != 0 && (mat->flags & MAT_FLAG_IMP_SAMPLE)) { // reached by a sampled direction: weigh against NEE
        if (DEFER_EMITTERS) {
            out.deferred_emitter = true;
            out.emitter_triangle = tri;
            out.emitter_mix_weight = pt.mix_weight;
            mis = 0.0f;
        } else {
            mis = emissive_hit_mis_weight(sc, ray.o, view, pt.P, hit.t, ray.pdf, tri, inst);
        }
    }
    out.radiance = emissive_hit_radiance(sp, pt.mix_weight, mis, mat->tangent_rotation_or_strength, pt.base, ray.c);
    return false;
}

} // namespace rt
