// rt_shade.h -- surface shading for one (ray, hit) pair: emits at most one secondary ray and one shadow ray.
// Restates Ref::ShadeSurface, reference internal/ShadeRef.cpp:1174-1652, with Evaluate_EnvColor (:1030-1066) and
// Evaluate_LightColor (:1068-1172).  (GLSL twin: shaders/shade.comp.glsl:1978-2467.)
//
// Not carried over (out of scope): the spatial radiance cache branches
// (cache_mode != None), deferred physical-sky shading (out_def_sky), env-map quadtree importance sampling.
#pragma once

#include "rt_bsdf.h"

namespace rt {

struct ShadeParams {
    PassLimits ps;
    float limits[2]; // {direct, indirect} clamp on the rgb SUM (3*clamp) or FLT_MAX, ShadeRef.cpp:1661-1662,1710-1711
    uint32_t rand_seed;
    int iteration;
};

struct ShadeResult {
    f4 col;          // radiance to assign (primary) / add (secondary) into the pixel
    f3 base_color;   // aux outputs (only meaningful when has_aux)
    f4 depth_normal; // N.xyz, t
    bool emit_secondary, emit_shadow;
    // DEFER_EMISSIVE builds only: the hit is an importance-sampled emitter reached by a secondary ray; its MIS weight
    // (a light-tree walk + a spherical-triangle pdf) is left to k_shade_emissive and `col` does not contain it yet
    bool defer_emissive;
    uint32_t def_tri_index, def_mat_index;
    float def_mix_weight;
    f3 def_base_color;
};

// `tri_verts` table: the three vertices of every triangle, gathered through vtx_indices[] once per scene and laid out
// as 3 x 3 float4 -- (p, n.x) (n.yz, b.xy) (b.z, t, -) per vertex.  A shade point then needs ONE round trip of nine
// 16-byte loads instead of three index loads followed by 33 dword loads (rayhip_vertex is 44 bytes, 4-byte aligned).
// Costs 144 B per triangle of HBM (0.43 GB for the Bistro-class scene).  Pure data movement: same values.
constexpr int TRI_VERTS_STRIDE = 9;
// (vtx_indices is a sparse pool on the host: slots no mesh owns hold anything -- those entries stay zero, nothing reads them)
RT_HD void fill_tri_verts(const rayhip_vertex *vertices, const uint32_t vertices_count, const uint32_t *vtx_indices, const uint32_t tri,
                          float4 *out /* [9] */) {
    for (int k = 0; k < 9; ++k) {
        out[k] = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (vtx_indices[tri * 3 + 0] >= vertices_count || vtx_indices[tri * 3 + 1] >= vertices_count || vtx_indices[tri * 3 + 2] >= vertices_count) {
        return;
    }
    for (int k = 0; k < 3; ++k) {
        const rayhip_vertex &v = vertices[vtx_indices[tri * 3 + k]];
        out[3 * k + 0] = mkfloat4(v.p[0], v.p[1], v.p[2], v.n[0]);
        out[3 * k + 1] = mkfloat4(v.n[1], v.n[2], v.b[0], v.b[1]);
        out[3 * k + 2] = mkfloat4(v.b[2], v.t[0], v.t[1], 0.0f);
    }
}
RT_HD rayhip_vertex load_tri_vert(const float4 *t /* the vertex's three float4 */) {
    const float4 a = t[0], b = t[1], c = t[2];
    rayhip_vertex v;
    v.p[0] = a.x, v.p[1] = a.y, v.p[2] = a.z, v.n[0] = a.w;
    v.n[1] = b.x, v.n[2] = b.y, v.b[0] = b.z, v.b[1] = b.w;
    v.b[2] = c.x, v.t[0] = c.y, v.t[1] = c.z;
    return v;
}

// MIS weight of an emissive triangle hit by a BSDF-sampled ray, ShadeRef.cpp:1500-1525 (the NODE_EMISSIVE branch of
// ShadeSurface).  Factored out so that the device can run it in a kernel of its own (k_shade_emissive): hits of
// emitters are rare but every wavefront containing one used to pay for this path -- 15 % of the shade kernels' time.
RT_HD float emissive_hit_mis_weight(const SceneView &sc, const f3 ro, const f3 I, const f3 P, const float inter_t, const float ray_pdf,
                                    const uint32_t tri_index, const rayhip_mesh_instance *mi) {
    float mis_weight = 1.0f;
    const float pdf_factor = eval_tri_light_factor(sc, P, ro, tri_index);

    const rayhip_vertex &v1 = sc.vertices[sc.vtx_indices[tri_index * 3 + 0]];
    const rayhip_vertex &v2 = sc.vertices[sc.vtx_indices[tri_index * 3 + 1]];
    const rayhip_vertex &v3 = sc.vertices[sc.vtx_indices[tri_index * 3 + 2]];
    const f3 p1 = mk3(v1.p), p2 = mk3(v2.p), p3 = mk3(v3.p);

    float light_forward_len;
    const f3 light_forward = normalize_len(transform_direction(cross(p2 - p1, p3 - p1), mi->xform), light_forward_len);
    const float tri_area = 0.5f * light_forward_len;

    const float cos_theta = fabsf(dot(I, light_forward)); // abs for doublesided light
    if (cos_theta > 0.0f) {
        const f3 P_ls = transform_point(ro, mi->inv_xform);
        float light_pdf = sample_spherical_triangle(P_ls, p1, p2, p3, f2{0.0f, 0.0f}, nullptr) / pdf_factor;
        if (light_pdf == 0.0f) {
            light_pdf = (inter_t * inter_t) / (tri_area * cos_theta * pdf_factor);
        }
        const float bsdf_pdf = ray_pdf;
        mis_weight = power_heuristic(bsdf_pdf, light_pdf);
    }
    return mis_weight;
}

// radiance of an emissive hit and its clamp, exactly the tail ShadeSurface applies to `col` (ShadeRef.cpp:1646-1651)
RT_HD f4 emissive_hit_radiance(const ShadeParams &sp, const float mix_weight, const float mis_weight, const float strength,
                               const f3 base_color, const f3 ray_c) {
    f3 col = {0.0f, 0.0f, 0.0f};
    col += mix_weight * mis_weight * strength * base_color;
    col *= ray_c;
    const float sum = hsum(mk4(col, 0.0f));
    if (sum > sp.limits[1]) {
        col *= (sp.limits[1] / sum);
    }
    return mk4(col, 1.0f);
}

// ShadeRef.cpp:1030-1066
RT_HD f4 Evaluate_EnvColor(const SceneView &sc, const Ray &ray, const float pdf_factor, const f2 rnd) {
    const rayhip_environment &env = sc.env;
    const f3 I = ray.d;
    f4 env_col = {1.0f, 1.0f, 1.0f, 1.0f};

    const uint32_t env_map = is_indirect(ray.depth) ? env.env_map : env.back_map;
    const float env_map_rotation = is_indirect(ray.depth) ? env.env_map_rotation : env.back_map_rotation;
    if (env_map != 0xffffffff) {
        env_col = mk4(sample_latlong_rgbe(sc, env_map, I, env_map_rotation, rnd), 1.0f);
    }

    if (env.light_index != 0xffffffff && pdf_factor >= 0.0f && is_indirect(ray.depth)) {
        if (env.qtree_levels) {
            const float light_pdf = safe_div_pos(evaluate_env_qtree(sc, env_map_rotation, I), pdf_factor);
            const float bsdf_pdf = ray.pdf;

            const float mis_weight = power_heuristic(bsdf_pdf, light_pdf);
            env_col *= mis_weight;
        } else {
            const float light_pdf = safe_div_pos(0.5f, PI * pdf_factor);
            const float bsdf_pdf = ray.pdf;

            const float mis_weight = power_heuristic(bsdf_pdf, light_pdf);
            env_col *= mis_weight;
        }
    }

    env_col *= is_indirect(ray.depth) ? mk4(env.env_col[0], env.env_col[1], env.env_col[2], 1.0f)
                                      : mk4(env.back_col[0], env.back_col[1], env.back_col[2], 1.0f);
    env_col.w = 1.0f;
    return env_col;
}

// ShadeRef.cpp:1068-1172
RT_HD f3 Evaluate_LightColor(const SceneView &sc, const Ray &ray, const Hit &inter, const f2 rnd) {
    const f3 ro = ray.o, I = ray.d;

    const rayhip_light &l = sc.lights[-inter.obj_index - 1];
    const float pdf_factor = (1.0f / inter.u);
    const uint32_t ltype = light_type(l);

    f3 lcol = mk3(l.col);
    if (light_sky_portal(l)) {
        f3 env_col = mk3(sc.env.env_col);
        if (sc.env.env_map != 0xffffffff) {
            env_col *= sample_latlong_rgbe(sc, sc.env.env_map, I, sc.env.env_map_rotation, rnd);
        }
        lcol *= env_col;
    }
    if (ltype == LIGHT_TYPE_SPHERE) {
        const f3 light_pos = mk3(&l.params[0]);
        const float radius = l.params[7];

        float d;
        const f3 disk_normal = normalize_len(light_pos - ro, d);

        if (d > radius) {
            const float temp = sqrtf(d * d - radius * radius);
            const float disk_radius = (temp * radius) / d;
            float disk_dist = dot(ro, disk_normal) - dot(light_pos, disk_normal);

            const float sampled_area = PI * disk_radius * disk_radius;
            const float cos_theta = dot(I, disk_normal);
            disk_dist /= cos_theta;

            const float light_pdf = (disk_dist * disk_dist) / (sampled_area * cos_theta * pdf_factor);
            const float bsdf_pdf = ray.pdf;

            const float mis_weight = power_heuristic(bsdf_pdf, light_pdf);
            lcol *= mis_weight;

            const float spot = l.params[8], blend = l.params[9];
            if (spot > 0.0f && blend > 0.0f) {
                const float _dot = -dot(I, mk3(&l.params[4]));
                const float _angle = acosf(saturatef(_dot));
                if (blend > 0.0f) {
                    lcol *= saturatef((spot - _angle) / blend);
                }
            }
        }
    } else if (ltype == LIGHT_TYPE_DIR) {
        const float radius = l.params[4]; // tan_angle
        const float light_area = PI * radius * radius;

        const float cos_theta = dot(I, mk3(&l.params[0]));

        const float light_pdf = 1.0f / (light_area * cos_theta * pdf_factor);
        const float bsdf_pdf = ray.pdf;

        const float mis_weight = power_heuristic(bsdf_pdf, light_pdf);
        lcol *= mis_weight;
    } else if (ltype == LIGHT_TYPE_RECT) {
        const f3 light_pos = mk3(&l.params[0]);
        const f3 light_u = mk3(&l.params[4]), light_v = mk3(&l.params[8]);

        float light_pdf = sample_spherical_rectangle(ro, light_pos, light_u, light_v, f2{0.0f, 0.0f}, nullptr) / pdf_factor;
        if (light_pdf == 0.0f) {
            const f3 light_forward = normalize(cross(light_u, light_v));
            const float light_area = l.params[3];
            const float cos_theta = dot(I, light_forward);
            light_pdf = (inter.t * inter.t) / (light_area * cos_theta * pdf_factor);
        }

        const float bsdf_pdf = ray.pdf;
        const float mis_weight = power_heuristic(bsdf_pdf, light_pdf);
        lcol *= mis_weight;
    } else if (ltype == LIGHT_TYPE_DISK) {
        const f3 light_u = mk3(&l.params[4]), light_v = mk3(&l.params[8]);

        const f3 light_forward = normalize(cross(light_u, light_v));
        const float light_area = l.params[3];

        const float cos_theta = dot(I, light_forward);

        const float light_pdf = (inter.t * inter.t) / (light_area * cos_theta * pdf_factor);
        const float bsdf_pdf = ray.pdf;

        const float mis_weight = power_heuristic(bsdf_pdf, light_pdf);
        lcol *= mis_weight;
    } else if (ltype == LIGHT_TYPE_LINE) {
        const f3 light_dir = mk3(&l.params[8]);
        const float light_area = l.params[3];

        const float cos_theta = 1.0f - fabsf(dot(I, light_dir));

        const float light_pdf = (inter.t * inter.t) / (light_area * cos_theta * pdf_factor);
        const float bsdf_pdf = ray.pdf;

        const float mis_weight = power_heuristic(bsdf_pdf, light_pdf);
        lcol *= mis_weight;
    }
    return lcol;
}

// Ref::ShadeSurface.  new_ray / sh_r are fully written only when the corresponding emit flag is set.
// DEFER_EMISSIVE: see ShadeResult::defer_emissive.
template <bool DEFER_EMISSIVE = false>
RT_HD ShadeResult shade_surface(const SceneView &sc, const ShadeParams &sp, const Hit &inter, const Ray &ray, Ray &new_ray,
                                ShadowRay &sh_r) {
    const PassLimits &ps = sp.ps;
    ShadeResult res;
    res.col = {0.0f, 0.0f, 0.0f, 0.0f};
    res.base_color = {0.0f, 0.0f, 0.0f};
    res.depth_normal = {0.0f, 0.0f, 0.0f, 0.0f};
    res.emit_secondary = res.emit_shadow = false;
    res.defer_emissive = false;

    const f3 I = ray.d;
    const f3 ro = ray.o;

    // used to randomize random sequence among pixels
    const uint32_t px_hash = hash(ray.xy);
    const uint32_t rand_hash = hash_combine(px_hash, sp.rand_seed);
    const uint32_t rand_dim = RAND_DIM_BASE_COUNT + get_total_depth(ray.depth) * RAND_DIM_BOUNCE_COUNT;
    const int sample = sp.iteration - 1;

    const f2 tex_rand = get_scrambled_2d_rand(rand_dim + RAND_DIM_TEX, rand_hash, sample, sc.pmj);

    if (inter.v < 0.0f) {
        const float pdf_factor = (get_total_depth(ray.depth) < ps.max_total_depth) ? safe_div_pos(1.0f, inter.u) : -1.0f;

        f4 env_col = Evaluate_EnvColor(sc, ray, pdf_factor, tex_rand);
        env_col *= mk4(ray.c.x, ray.c.y, ray.c.z, 0.0f);
        const float sum = hsum(env_col);
        if (sum > sp.limits[0]) {
            env_col *= (sp.limits[0] / sum);
        }
        res.col = env_col;
        RT_PROF(1)
        return res;
    }

    Surface surf;
    surf.P = ro + inter.t * I;

    if (inter.obj_index < 0) { // Area light intersection
        f3 lcol = Evaluate_LightColor(sc, ray, inter, tex_rand);
        lcol *= ray.c;
        const float sum = hsum(mk4(lcol, 0.0f));
        if (sum > sp.limits[0]) {
            lcol *= (sp.limits[0] / sum);
        }
        res.col = mk4(lcol, 1.0f);
        RT_PROF(2)
        return res;
    }

    const bool is_backfacing = (inter.prim_index < 0);
    const uint32_t tri_index = is_backfacing ? uint32_t(-inter.prim_index - 1) : uint32_t(inter.prim_index);

    const rayhip_tri_mat_data tmd = sc.tri_materials[tri_index];
    const rayhip_material *mat = &sc.materials[tmd.front_mi & MATERIAL_INDEX_BITS];
    const rayhip_mesh_instance *mi = &sc.mesh_instances[inter.obj_index];

    const float4 *tv = sc.tri_verts + size_t(tri_index) * TRI_VERTS_STRIDE; // (fill_tri_verts)
    const rayhip_vertex v1 = load_tri_vert(tv), v2 = load_tri_vert(tv + 3), v3 = load_tri_vert(tv + 6);

    const float w = 1.0f - inter.u - inter.v;
    surf.N = normalize(mk3(v1.n) * w + mk3(v2.n) * inter.u + mk3(v3.n) * inter.v);
    surf.uvs = mk2(v1.t[0], v1.t[1]) * w + mk2(v2.t[0], v2.t[1]) * inter.u + mk2(v3.t[0], v3.t[1]) * inter.v;

    float pa;
    surf.plane_N = normalize_len(cross(mk3(v2.p) - mk3(v1.p), mk3(v3.p) - mk3(v1.p)), pa);

    surf.B = mk3(v1.b) * w + mk3(v2.b) * inter.u + mk3(v3.b) * inter.v;
    surf.T = cross(surf.B, surf.N);

    if (is_backfacing) {
        if (tmd.back_mi == 0xffff) {
            res.col = {0.0f, 0.0f, 0.0f, 0.0f};
            return res;
        } else {
            mat = &sc.materials[tmd.back_mi & MATERIAL_INDEX_BITS];
            surf.plane_N = -surf.plane_N;
            surf.N = -surf.N;
            surf.B = -surf.B;
            surf.T = -surf.T;
        }
    }

    surf.plane_N = transform_normal(surf.plane_N, mi->inv_xform);
    surf.N = transform_normal(surf.N, mi->inv_xform);
    surf.B = transform_normal(surf.B, mi->inv_xform);
    surf.T = transform_normal(surf.T, mi->inv_xform);

    // normalize vectors (scaling might have been applied)
    surf.plane_N = safe_normalize(surf.plane_N);
    surf.N = safe_normalize(surf.N);
    surf.B = safe_normalize(surf.B);
    surf.T = safe_normalize(surf.T);

    const float ta = fabsf((v2.t[0] - v1.t[0]) * (v3.t[1] - v1.t[1]) - (v3.t[0] - v1.t[0]) * (v2.t[1] - v1.t[1]));

    const float cone_width = ray.cone_width + ray.cone_spread * inter.t;

    float lambda = 0.5f * fast_log2(ta / pa);
    lambda += fast_log2(cone_width);

    const float ext_ior = peek_ior_stack(ray.ior, is_backfacing);
    RT_PROF(3)

    f3 col = {0.0f, 0.0f, 0.0f};

    const int diff_depth = get_diff_depth(ray.depth), spec_depth = get_spec_depth(ray.depth),
              refr_depth = get_refr_depth(ray.depth);
    // NOTE: transparency depth is not accounted here
    const int total_depth = diff_depth + spec_depth + refr_depth;

    const f2 mix_term_rand = get_scrambled_2d_rand(rand_dim + RAND_DIM_BSDF_PICK, rand_hash, sample, sc.pmj);

    float mix_rand = mix_term_rand.x;
    float mix_weight = 1.0f;

    // resolve mix material
    while (mat->type == NODE_MIX) {
        float mix_val = mat->tangent_rotation_or_strength;
        const uint32_t base_texture = mat->textures[BASE_TEXTURE];
        if (base_texture != 0xffffffff) {
            const f4 tex_color = sample_color(sc, base_texture, surf.uvs, 0, tex_rand);
            mix_val *= tex_color.x;
        }

        const float eta = is_backfacing ? safe_div_pos(ext_ior, mat->ior) : safe_div_pos(mat->ior, ext_ior);
        const float RR = mat->ior != 0.0f ? fresnel_dielectric_cos(dot(I, surf.N), eta) : 1.0f;

        mix_val *= saturatef(RR);

        if (mix_rand > mix_val) {
            mix_weight *= (mat->flags & MAT_FLAG_MIX_ADD) ? 1.0f / (1.0f - mix_val) : 1.0f;

            mat = &sc.materials[mat->textures[MIX_MAT1]];
            mix_rand = safe_div_pos(mix_rand - mix_val, 1.0f - mix_val);
        } else {
            mix_weight *= (mat->flags & MAT_FLAG_MIX_ADD) ? 1.0f / mix_val : 1.0f;

            mat = &sc.materials[mat->textures[MIX_MAT2]];
            mix_rand = safe_div_pos(mix_rand, mix_val);
        }
    }

    // apply normal map
    if (mat->textures[NORMALS_TEXTURE] != 0xffffffff) {
        f4 normals = sample_bilinear(sc, mat->textures[NORMALS_TEXTURE], surf.uvs, 0, tex_rand);
        normals = normals * 2.0f;
        normals = {normals.x - 1.0f, normals.y - 1.0f, normals.z - 1.0f, normals.w - 1.0f};
        normals.z = 1.0f;
        if (mat->textures[NORMALS_TEXTURE] & TEX_RECONSTRUCT_Z_BIT) {
            normals.z = safe_sqrt(1.0f - normals.x * normals.x - normals.y * normals.y);
        }
        const f3 in_normal = surf.N;
        surf.N = normalize(normals.x * surf.T + normals.z * surf.N + normals.y * surf.B);
        if (mat->normal_map_strength_unorm != 0xffff) {
            surf.N = normalize(in_normal + (surf.N - in_normal) * (float(mat->normal_map_strength_unorm) / 65535.0f));
        }
        surf.N = ensure_valid_reflection(surf.plane_N, -I, surf.N);
    }

    // Find radial tangent in local space
    const f3 P_ls = mk3(v1.p) * w + mk3(v2.p) * inter.u + mk3(v3.p) * inter.v;
    // rotate around Y axis by 90 degrees in 2d
    f3 tangent = {-P_ls.z, 0.0f, P_ls.x};
    tangent = transform_normal(tangent, mi->inv_xform);
    if (length2(cross(tangent, surf.N)) == 0.0f) {
        tangent = transform_normal(P_ls, mi->inv_xform);
    }
    if (mat->tangent_rotation_or_strength != 0.0f) {
        tangent = rotate_around_axis(tangent, surf.N, mat->tangent_rotation_or_strength);
    }

    surf.B = safe_normalize(cross(tangent, surf.N));
    surf.T = cross(surf.N, surf.B);
    RT_PROF(4)

    LightSample ls = make_light_sample();
    if (sc.light_cwnodes_count != 0 && mat->type != NODE_EMISSIVE) {
        const float rand_pick_light = get_scrambled_2d_rand(rand_dim + RAND_DIM_LIGHT_PICK, rand_hash, sample, sc.pmj).x;
        const f2 rand_light_uv = get_scrambled_2d_rand(rand_dim + RAND_DIM_LIGHT, rand_hash, sample, sc.pmj);

        sample_light_source(sc, surf.P, surf.T, surf.B, surf.N, rand_pick_light, rand_light_uv, tex_rand, ls);
    }
    RT_PROF(5)
    const float N_dot_L = dot(surf.N, ls.L);

    // sample base texture
    f3 base_color = mk3(mat->base_color);
    if (mat->textures[BASE_TEXTURE] != 0xffffffff) {
        const uint32_t base_texture = mat->textures[BASE_TEXTURE];
        const float base_lod = get_texture_lod(sc, base_texture, lambda);
        const f4 tex_color = sample_color(sc, base_texture, surf.uvs, int(base_lod), tex_rand);
        base_color *= xyz(tex_color);
    }

    res.base_color = base_color;
    res.depth_normal = mk4(surf.N, inter.t);

    f3 tint_color = {0.0f, 0.0f, 0.0f};

    const float base_color_lum = lum(base_color);
    if (base_color_lum > 0.0f) {
        tint_color = base_color / base_color_lum;
    }

    float roughness = float(mat->roughness_unorm) / 65535.0f;
    if (mat->textures[ROUGH_TEXTURE] != 0xffffffff) {
        const uint32_t roughness_tex = mat->textures[ROUGH_TEXTURE];
        const float roughness_lod = get_texture_lod(sc, roughness_tex, lambda);
        // fvec4 roughness_color = SampleBilinear(...).get<0>()  (splat of the red channel)
        const float r0 = sample_bilinear(sc, roughness_tex, surf.uvs, int(roughness_lod), tex_rand).x;
        f4 roughness_color = {r0, r0, r0, r0};
        if (roughness_tex & TEX_SRGB_BIT) {
            roughness_color = srgb_to_linear(roughness_color);
        }
        roughness *= roughness_color.x;
    }

    const f2 rand_bsdf = get_scrambled_2d_rand(rand_dim + RAND_DIM_BSDF, rand_hash, sample, sc.pmj);

    new_ray.ior[0] = ray.ior[0], new_ray.ior[1] = ray.ior[1], new_ray.ior[2] = ray.ior[2], new_ray.ior[3] = ray.ior[3];
    new_ray.cone_width = cone_width;
    new_ray.cone_spread = ray.cone_spread;
    new_ray.xy = ray.xy;
    new_ray.pdf = 0.0f;
    // (the reference leaves o/d/c/depth of an unsampled slot stale; such a slot is never emitted because pdf == 0)
    new_ray.o = new_ray.d = new_ray.c = {0.0f, 0.0f, 0.0f};
    new_ray.depth = 0;

    sh_r.c = {0.0f, 0.0f, 0.0f};
    sh_r.depth = ray.depth;
    sh_r.xy = ray.xy;
    sh_r.o = sh_r.d = {0.0f, 0.0f, 0.0f};
    sh_r.dist = 0.0f;

    const float regularize_alpha = (get_diff_depth(ray.depth) > 0) ? ps.regularize_alpha : 0.0f;
    RT_PROF(6)

    // Sample materials
    if (mat->type == NODE_DIFFUSE) {
        if (ls.pdf > 0.0f && (ls.ray_flags & RAY_TYPE_DIFFUSE_BIT) != 0 && N_dot_L > 0.0f) {
            col += Evaluate_DiffuseNode(ls, ray, surf, base_color, roughness, mix_weight, (total_depth < ps.max_total_depth), sh_r);
        }
        if (diff_depth < ps.max_diff_depth && total_depth < ps.max_total_depth) {
            Sample_DiffuseNode(ray, surf, base_color, roughness, rand_bsdf, mix_weight, new_ray);
        }
        RT_PROF(7)
    } else if (mat->type == NODE_GLOSSY) {
        const float specular = 0.5f;
        const float spec_ior = (2.0f / (1.0f - sqrtf(0.08f * specular))) - 1.0f;
        const float spec_F0 = fresnel_dielectric_cos(1.0f, spec_ior);
        if (ls.pdf > 0.0f && (ls.ray_flags & RAY_TYPE_SPECULAR_BIT) != 0 && N_dot_L > 0.0f) {
            col += Evaluate_GlossyNode(ls, ray, surf, base_color, roughness, regularize_alpha, spec_ior, spec_F0, mix_weight,
                                       (total_depth < ps.max_total_depth), sh_r);
        }
        if (spec_depth < ps.max_spec_depth && total_depth < ps.max_total_depth) {
            Sample_GlossyNode(ray, surf, base_color, roughness, regularize_alpha, spec_ior, spec_F0, rand_bsdf, mix_weight,
                              new_ray);
        }
        RT_PROF(8)
    } else if (mat->type == NODE_REFRACTIVE) {
        if (ls.pdf > 0.0f && (ls.ray_flags & RAY_TYPE_REFR_BIT) != 0 && N_dot_L < 0.0f) {
            const float eta = is_backfacing ? (mat->ior / ext_ior) : (ext_ior / mat->ior);
            col += Evaluate_RefractiveNode(ls, ray, surf, base_color, roughness, regularize_alpha, eta, mix_weight,
                                           (total_depth < ps.max_total_depth), sh_r);
        }
        if (refr_depth < ps.max_refr_depth && total_depth < ps.max_total_depth) {
            Sample_RefractiveNode(ray, surf, base_color, roughness, regularize_alpha, is_backfacing, mat->ior, ext_ior,
                                  rand_bsdf, mix_weight, new_ray);
        }
        RT_PROF(9)
    } else if (mat->type == NODE_EMISSIVE) {
        float mis_weight = 1.0f;
        if ((ray.depth & 0x00ffffff) != 0 && (mat->flags & MAT_FLAG_IMP_SAMPLE)) {
            if (DEFER_EMISSIVE) {
                res.defer_emissive = true;
                res.def_tri_index = tri_index;
                res.def_mat_index = uint32_t(mat - sc.materials);
                res.def_mix_weight = mix_weight;
                res.def_base_color = base_color;
                mis_weight = 0.0f; // nothing is added here; k_shade_emissive adds the weighted radiance
            } else {
                mis_weight = emissive_hit_mis_weight(sc, ro, I, surf.P, inter.t, ray.pdf, tri_index, mi);
            }
        }
        col += mix_weight * mis_weight * mat->tangent_rotation_or_strength * base_color;
        RT_PROF(10)
    } else if (mat->type == NODE_PRINCIPLED) {
        float metallic = float(mat->metallic_unorm) / 65535.0f;
        if (mat->textures[METALLIC_TEXTURE] != 0xffffffff) {
            const uint32_t metallic_tex = mat->textures[METALLIC_TEXTURE];
            const float metallic_lod = get_texture_lod(sc, metallic_tex, lambda);
            metallic *= sample_bilinear(sc, metallic_tex, surf.uvs, int(metallic_lod), tex_rand).x;
        }

        float specular = float(mat->specular_unorm) / 65535.0f;
        if (mat->textures[SPECULAR_TEXTURE] != 0xffffffff) {
            const uint32_t specular_tex = mat->textures[SPECULAR_TEXTURE];
            const float specular_lod = get_texture_lod(sc, specular_tex, lambda);
            f4 specular_color = sample_bilinear(sc, specular_tex, surf.uvs, int(specular_lod), tex_rand);
            if (specular_tex & TEX_SRGB_BIT) {
                specular_color = srgb_to_linear(specular_color);
            }
            specular *= specular_color.x;
        }

        const float specular_tint = float(mat->specular_tint_unorm) / 65535.0f;
        const float transmission = float(mat->transmission_unorm) / 65535.0f;
        const float clearcoat = float(mat->clearcoat_unorm) / 65535.0f;
        const float clearcoat_roughness = float(mat->clearcoat_roughness_unorm) / 65535.0f;
        const float sheen = 2.0f * (float(mat->sheen_unorm) / 65535.0f);
        const float sheen_tint = float(mat->sheen_tint_unorm) / 65535.0f;

        DiffParams diff;
        diff.base_color = base_color;
        diff.sheen_color = sheen * mix3(splat3(1.0f), tint_color, sheen_tint);
        diff.roughness = roughness;

        SpecParams spec;
        spec.tmp_col = mix3(splat3(1.0f), tint_color, specular_tint);
        spec.tmp_col = mix3(specular * 0.08f * spec.tmp_col, base_color, metallic);
        spec.roughness = roughness;
        spec.ior = (2.0f / (1.0f - sqrtf(0.08f * specular))) - 1.0f;
        spec.F0 = fresnel_dielectric_cos(1.0f, spec.ior);
        spec.anisotropy = float(mat->anisotropic_unorm) / 65535.0f;

        CoatParams coat;
        coat.roughness = clearcoat_roughness;
        coat.ior = (2.0f / (1.0f - sqrtf(0.08f * clearcoat))) - 1.0f;
        coat.F0 = fresnel_dielectric_cos(1.0f, coat.ior);

        TransParams trans;
        trans.roughness = 1.0f - (1.0f - roughness) * (1.0f - float(mat->transmission_roughness_unorm) / 65535.0f);
        trans.int_ior = mat->ior;
        trans.eta = is_backfacing ? (mat->ior / ext_ior) : (ext_ior / mat->ior);
        trans.fresnel = fresnel_dielectric_cos(dot(I, surf.N), 1.0f / trans.eta);
        trans.backfacing = is_backfacing;

        // Approximation of FH (using shading normal)
        const float FN = (fresnel_dielectric_cos(dot(I, surf.N), spec.ior) - spec.F0) / (1.0f - spec.F0);

        const f3 approx_spec_col = mix3(spec.tmp_col, splat3(1.0f), FN);
        const float spec_color_lum = lum(approx_spec_col);

        const LobeWeights lobe_weights =
            get_lobe_weights(mixf(base_color_lum, 1.0f, sheen), spec_color_lum, specular, metallic, transmission, clearcoat);
        RT_PROF(11)

        if (ls.pdf > 0.0f) {
            col += Evaluate_PrincipledNode(ls, ray, surf, lobe_weights, diff, spec, coat, trans, metallic, transmission,
                                           N_dot_L, mix_weight, (total_depth < ps.max_total_depth), regularize_alpha, sh_r);
        }
        RT_PROF(12)
        Sample_PrincipledNode(ps, ray, surf, lobe_weights, diff, spec, coat, trans, metallic, transmission, rand_bsdf,
                              mix_rand, mix_weight, regularize_alpha, new_ray);
        RT_PROF(13)
    }

    const bool can_terminate_path = total_depth > ps.min_total_depth;

    new_ray.c *= ray.c;
    const float lum_ = fmaxf(new_ray.c.x, fmaxf(new_ray.c.y, new_ray.c.z));
    const float p = mix_term_rand.y;
    const float q = can_terminate_path ? fmaxf(0.05f, 1.0f - lum_) : 0.0f;
    if (p >= q && lum_ > 0.0f && new_ray.pdf > 0.0f) {
        new_ray.pdf = fminf(new_ray.pdf, 1e6f);
        new_ray.c.x /= (1.0f - q);
        new_ray.c.y /= (1.0f - q);
        new_ray.c.z /= (1.0f - q);
        res.emit_secondary = true;
    }

    {
        sh_r.c *= ray.c;
        const float sh_lum = fmaxf(sh_r.c.x, fmaxf(sh_r.c.y, sh_r.c.z));
        if (sh_lum > 0.0f) {
            // actual ray direction accouning for bias from both ends
            const f3 to_light = normalize_len(ls.lp - sh_r.o, sh_r.dist);
            sh_r.d = to_light;
            sh_r.dist *= ls.dist_mul;
            if (ls.from_env) {
                // NOTE: hacky way to identify env ray
                sh_r.dist = -sh_r.dist;
            }
            res.emit_shadow = true;
        }
    }
    col *= ray.c;
    const float sum = hsum(mk4(col, 0.0f));
    if (sum > sp.limits[1]) {
        col *= (sp.limits[1] / sum);
    }
    res.col = mk4(col, 1.0f);
    RT_PROF(14)
    return res;
}

} // namespace rt
