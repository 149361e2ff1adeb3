// rt_traverse.h -- BVH2 traversal with an explicit stack and the reference's precomputed 3-plane triangle test.
//
// Restates, operation for operation:
//   IntersectTri                         reference internal/CoreRef.cpp:24-50
//   bbox_test                            CoreRef.cpp:171-210
//   IntersectTris_ClosestHit / _AnyHit   CoreRef.cpp:1798-1817 / 1840-1863
//   Traverse_TLAS/BLAS_WithStack_ClosestHit (bvh2)   CoreRef.cpp:1943-2025 / 2428-2493
//   Traverse_TLAS/BLAS_WithStack_AnyHit (bvh2)       CoreRef.cpp:2193-2280 / 2619-2693
//   IntersectScene (closest, transparency loop)      CoreRef.cpp:3041-3158
//   IntersectScene (shadow_ray_t)                    CoreRef.cpp:3160-3262
//
// The Stack policy is what differs between builds: kernels.hip passes a per-wavefront LDS stack
// (depth-major, 64 lanes wide -> bank = lane, conflict free at any mix of depths), tests/hostsim a plain
// array.  BLAS traversal runs on the same stack above the TLAS entries (the GLSL shader does the same,
// reference shaders/intersect_scene.comp.glsl:89-94,215); `base` plays the role of the reference's separate
// `stack_size == 0`.
#pragma once

#include "rt_bvh4.h"
#include "rt_bvh8.h"
#include "rt_isect.h"

namespace rt {

// One visit of an inner node (body of the inner loop, CoreRef.cpp:1961-1993): fetch the 64-byte node as four
// 16-byte loads issued together -- child links included, so the node costs ONE memory round trip -- slab-test both
// children against the current t, descend into the nearer one and push the farther one.
template <class Stack>
RT_HD void bvh2_node_step(const rayhip_bvh2_node *nodes, const f3 ro, const f3 inv_d, const float t, uint32_t &cur,
                          Stack &st, TravCount *cnt) {
    const float4 *np = reinterpret_cast<const float4 *>(nodes + cur);
    const float4 d0 = np[0], d1 = np[1], d2 = np[2], links = np[3];
    const uint32_t left_child = float_as_uint(links.x), right_child = float_as_uint(links.y);
    if (cnt) {
        ++cnt->nodes;
        cnt->max_stack = st.size > cnt->max_stack ? st.size : cnt->max_stack;
    }
    const float ch0_min[3] = {d0.x, d0.z, d2.x};
    const float ch0_max[3] = {d0.y, d0.w, d2.y};
    const float ch1_min[3] = {d1.x, d1.z, d2.z};
    const float ch1_max[3] = {d1.y, d1.w, d2.w};

    float ch0_dist, ch1_dist;
    const bool ch0_res = bbox_test(ro, inv_d, t, ch0_min, ch0_max, ch0_dist);
    const bool ch1_res = bbox_test(ro, inv_d, t, ch1_min, ch1_max, ch1_dist);

    // `cur = ch0_res ? children[0] : children[1]; if both: nearer first, push the farther` as selects
    uint32_t next = ch0_res ? left_child : right_child;
    uint32_t far_child = right_child;
    if (ch0_res && ch1_res && ch1_dist < ch0_dist) {
        far_child = next;
        next = right_child;
    }
    if (!ch0_res && !ch1_res) {
        cur = st.pop();
    } else {
        cur = next;
        if (ch0_res && ch1_res) {
            st.push(far_child);
        }
    }
}

// Ordered DFS over the bvh2 pool.  `leaf(word)` handles one leaf word and returns true to stop the whole
// walk (any-hit early out).  `t_ref` is read at every node so that hits found in earlier leaves prune.
// Loop structure == CoreRef.cpp:1958-2014 / 2439-2490.
template <class Stack, class LeafFn>
RT_HD bool walk_bvh2(const rayhip_bvh2_node *nodes, const uint32_t root_index, const f3 ro, const f3 inv_d,
                     const float &t_ref, Stack &st, TravCount *cnt, LeafFn &&leaf) {
    const uint32_t base = st.size;
    st.push(0x1fffffffu);

    uint32_t cur = root_index;
    while (st.size > base) {
        uint32_t leaf_node = 0;
        while (st.size > base && (cur & BVH2_PRIM_COUNT_BITS) == 0) {
            bvh2_node_step(nodes, ro, inv_d, t_ref, cur, st, cnt);
            if ((cur & BVH2_PRIM_COUNT_BITS) != 0 && (leaf_node & BVH2_PRIM_COUNT_BITS) == 0) {
                leaf_node = cur;
                cur = st.pop();
            }
            if ((leaf_node & BVH2_PRIM_COUNT_BITS) != 0) {
                break;
            }
        }

        while ((leaf_node & BVH2_PRIM_COUNT_BITS) != 0) {
            if (leaf(leaf_node)) {
                st.size = base;
                return true;
            }
            leaf_node = cur;
            if ((cur & BVH2_PRIM_COUNT_BITS) != 0) {
                cur = st.pop();
            }
        }
    }
    st.size = base; // (already true; keeps the invariant explicit)
    return false;
}

RT_HD TriTable tri_table(const SceneView &sc) { return TriTable{reinterpret_cast<const float4 *>(sc.tris), sc.tri_pitch}; }

// CoreRef.cpp:1798-1817
RT_HD bool intersect_tris_closest(const f3 ro, const f3 rd, const TriTable tris, const int tri_start,
                                  const int tri_end, const int obj_index, Hit &out_inter) {
    Hit inter;
    inter.obj_index = obj_index;
    inter.prim_index = 0;
    inter.t = out_inter.t;
    inter.u = 0.0f;
    inter.v = -1.0f;
    // a leaf word carries count - 1 in bits that are non-zero for a leaf (rt_base.h), i.e. a leaf has at least TWO records (the
    // reference's builder and lbvh.h duplicate a lone triangle): both are requested at once and tested back to back -- one round
    // trip, and none of the register shuffling of a rotating "next triangle" (26 v_mov per triangle in round 2's loop); the
    // refined trees have nothing but such leaves.  Longer leaves go on one record at a time.
    RT_PROF_T(20)
    const TriData tri0 = load_tri(tris, uint32_t(tri_start)), tri1 = load_tri(tris, uint32_t(tri_start + 1));
    RT_PROF_WAIT(tri0.n, tri0.u, tri1.n, tri1.v)
    RT_PROF_T(21)
    RT_PROF_LANES(2)
    intersect_tri(ro, rd, tri0, uint32_t(tri_start), inter);
    RT_PROF_LANES(2)
    intersect_tri(ro, rd, tri1, uint32_t(tri_start + 1), inter);
    for (int i = tri_start + 2; i < tri_end; ++i) {
        RT_PROF_LANES(2)
        const TriData tri = load_tri(tris, uint32_t(i));
        intersect_tri(ro, rd, tri, uint32_t(i), inter);
    }
    RT_PROF_T(22)
    const bool hit = inter.v >= 0.0f;
    out_inter.obj_index = hit ? inter.obj_index : out_inter.obj_index;
    out_inter.prim_index = hit ? inter.prim_index : out_inter.prim_index;
    out_inter.t = inter.t; // already contains min value
    out_inter.u = hit ? inter.u : out_inter.u;
    out_inter.v = hit ? inter.v : out_inter.v;
    return hit;
}

// CoreRef.cpp:1840-1863
RT_HD bool intersect_tris_any(const f3 ro, const f3 rd, const TriTable tris,
                              const rayhip_tri_mat_data *materials, const uint32_t *indices, const int tri_start,
                              const int tri_end, const int obj_index, Hit &out_inter) {
    Hit inter;
    inter.obj_index = obj_index;
    inter.prim_index = 0;
    inter.t = out_inter.t;
    inter.u = 0.0f;
    inter.v = -1.0f;
    // (the reference's loop, CoreRef.cpp:1846-1856, with the first two records -- a leaf has at least two, see above -- requested together)
    const auto solid_hit = [&](const int i) {
        return inter.v >= 0.0f && ((inter.prim_index > 0 && (materials[indices[i]].front_mi & MATERIAL_SOLID_BIT)) ||
                                   (inter.prim_index < 0 && (materials[indices[i]].back_mi & MATERIAL_SOLID_BIT)));
    };
    const TriData tri0 = load_tri(tris, uint32_t(tri_start)), tri1 = load_tri(tris, uint32_t(tri_start + 1));
    intersect_tri(ro, rd, tri0, uint32_t(tri_start), inter);
    if (!solid_hit(tri_start)) {
        intersect_tri(ro, rd, tri1, uint32_t(tri_start + 1), inter);
        if (!solid_hit(tri_start + 1)) {
            for (int i = tri_start + 2; i < tri_end; ++i) {
                const TriData tri = load_tri(tris, uint32_t(i));
                intersect_tri(ro, rd, tri, uint32_t(i), inter);
                if (solid_hit(i)) {
                    break;
                }
            }
        }
    }
    const bool hit = inter.v >= 0.0f;
    out_inter.obj_index = hit ? inter.obj_index : out_inter.obj_index;
    out_inter.prim_index = hit ? inter.prim_index : out_inter.prim_index;
    out_inter.t = inter.t;
    out_inter.u = hit ? inter.u : out_inter.u;
    out_inter.v = hit ? inter.v : out_inter.v;
    return hit;
}

// Traverse_TLAS_WithStack_ClosestHit(bvh2), CoreRef.cpp:1943-2025 (+ BLAS :2428-2493)
// WIDE (0, 4 or 8): the BLAS level walks the 4-wide (rt_bvh4.h) or the 8-wide (rt_bvh8.h) quantised tree instead of the
// reference's BVH2 -- same leaves, same triangle tests, conservative culling.  Counters: `nodes` = BVH2 nodes (TLAS level in
// the wide walks), `nodes4` = wide nodes, so the product kernel's own algorithmic bytes can be stated next to the reference
// algorithm's.
template <int WIDE = 0, class Stack>
RT_HD bool traverse_closest(const SceneView &sc, const f3 ro, const f3 rd, const uint32_t ray_flags,
                            const uint32_t root_index, Hit &inter, Stack &st, TravCount *cnt) {
    bool res = false;
    const f3 inv_d = safe_invert(rd);

    walk_bvh2(sc.nodes, root_index, ro, inv_d, inter.t, st, cnt, [&](const uint32_t leaf_node) {
        const uint32_t mi_index = (leaf_node & BVH2_PRIM_INDEX_BITS);
        const rayhip_mesh_instance &mi = sc.mesh_instances[mi_index];
        if ((mi.ray_visibility & ray_flags) != 0) {
            if (cnt) {
                ++cnt->instances;
            }
            const f3 _ro = transform_point(ro, mi.inv_xform);
            const f3 _rd = transform_direction(rd, mi.inv_xform);
            const f3 _inv_d = safe_invert(_rd);
            RT_PROF_T(23)
            RT_PROF_LANES(4)
            auto blas_leaf_fn = [&](const uint32_t blas_leaf) {
                const int tri_start = int(blas_leaf & BVH2_PRIM_INDEX_BITS),
                          tri_end = int(tri_start + ((blas_leaf & BVH2_PRIM_COUNT_BITS) >> 29) + 1);
                if (cnt) {
                    cnt->tris += uint32_t(tri_end - tri_start);
                }
                res |= intersect_tris_closest(_ro, _rd, tri_table(sc), tri_start, tri_end, int(mi_index), inter);
                return false;
            };
            if (WIDE == 8) {
                walk_bvh8(sc.nodes8, sc.blas_root4[mi_index], _ro, _inv_d, inter.t, st, blas_leaf_fn, cnt);
            } else if (WIDE == 4) {
                walk_bvh4(sc.nodes4, sc.blas_root4[mi_index], _ro, _inv_d, inter.t, st, blas_leaf_fn, cnt);
            } else {
                walk_bvh2(sc.nodes, mi.node_index, _ro, _inv_d, inter.t, st, cnt, blas_leaf_fn);
            }
        }
        return false;
    });

    // resolve primitive index indirection (note: runs on misses too, exactly like the reference)
    if (inter.prim_index < 0) {
        inter.prim_index = -int(sc.tri_indices[-inter.prim_index - 1]) - 1;
    } else {
        inter.prim_index = int(sc.tri_indices[inter.prim_index]);
    }
    return res;
}

// Traverse_TLAS_WithStack_AnyHit(bvh2), CoreRef.cpp:2193-2280 (+ BLAS :2619-2693); returns "solid hit found"
template <int WIDE = 0, class Stack>
RT_HD bool traverse_any(const SceneView &sc, const f3 ro, const f3 rd, const int ray_type, const uint32_t root_index,
                        Hit &inter, Stack &st, TravCount *cnt) {
    const uint32_t ray_vismask = (1u << ray_type);
    const f3 inv_d = safe_invert(rd);

    const bool solid = walk_bvh2(sc.nodes, root_index, ro, inv_d, inter.t, st, cnt, [&](const uint32_t leaf_node) {
        const uint32_t mi_index = (leaf_node & BVH2_PRIM_INDEX_BITS);
        const rayhip_mesh_instance &mi = sc.mesh_instances[mi_index];
        if ((mi.ray_visibility & ray_vismask) != 0) {
            if (cnt) {
                ++cnt->instances;
            }
            const f3 _ro = transform_point(ro, mi.inv_xform);
            const f3 _rd = transform_direction(rd, mi.inv_xform);
            const f3 _inv_d = safe_invert(_rd);
            auto blas_leaf_fn = [&](const uint32_t blas_leaf) {
                const int tri_start = int(blas_leaf & BVH2_PRIM_INDEX_BITS),
                          tri_end = int(tri_start + ((blas_leaf & BVH2_PRIM_COUNT_BITS) >> 29) + 1);
                if (cnt) {
                    cnt->tris += uint32_t(tri_end - tri_start);
                }
                const bool hit_found = intersect_tris_any(_ro, _rd, tri_table(sc), sc.tri_materials, sc.tri_indices,
                                                          tri_start, tri_end, int(mi_index), inter);
                if (hit_found) {
                    const bool is_backfacing = inter.prim_index < 0;
                    const uint32_t prim_index = is_backfacing ? uint32_t(-inter.prim_index - 1) : uint32_t(inter.prim_index);
                    const rayhip_tri_mat_data md = sc.tri_materials[sc.tri_indices[prim_index]];
                    if ((!is_backfacing && (md.front_mi & MATERIAL_SOLID_BIT)) ||
                        (is_backfacing && (md.back_mi & MATERIAL_SOLID_BIT))) {
                        return true;
                    }
                }
                return false;
            };
            if (WIDE == 8) {
                return walk_bvh8(sc.nodes8, sc.blas_root4[mi_index], _ro, _inv_d, inter.t, st, blas_leaf_fn, cnt);
            }
            if (WIDE == 4) {
                return walk_bvh4(sc.nodes4, sc.blas_root4[mi_index], _ro, _inv_d, inter.t, st, blas_leaf_fn, cnt);
            }
            return walk_bvh2(sc.nodes, mi.node_index, _ro, _inv_d, inter.t, st, cnt, blas_leaf_fn);
        }
        return false;
    });
    if (solid) {
        return true; // reference returns before the index indirection (CoreRef.cpp:2258-2260)
    }
    if (inter.prim_index < 0) {
        inter.prim_index = -int(sc.tri_indices[-inter.prim_index - 1]) - 1;
    } else {
        inter.prim_index = int(sc.tri_indices[inter.prim_index]);
    }
    return false;
}

struct TraceParams {
    int min_transp_depth, max_transp_depth;
    uint32_t rand_seed;
    int iteration;
    uint32_t root_index;
};

// Tail of one round of Ref::IntersectScene's closest-hit loop (CoreRef.cpp:3071-3152): decides what a found hit
// means.  Returns true when the ray crossed a transparent surface and must be traced again from the advanced
// origin `ro` (inter, r.c, r.depth and rand_dim are updated for the next round); false when the hit is final
// (solid surface, opaque material after mix resolve, or the path was terminated -> r.c = 0).
RT_HD_RARE bool closest_resolve_transparency(const SceneView &sc, const TraceParams &tp, Ray &r, Hit &inter, const float t_val,
                                        const f3 rd, f3 &ro, uint32_t &rand_dim, const uint32_t rand_hash) {
    const bool is_backfacing = (inter.prim_index < 0);
    const uint32_t tri_index = is_backfacing ? uint32_t(-inter.prim_index - 1) : uint32_t(inter.prim_index);

    const rayhip_tri_mat_data md = sc.tri_materials[tri_index];
    if ((!is_backfacing && (md.front_mi & MATERIAL_SOLID_BIT)) || (is_backfacing && (md.back_mi & MATERIAL_SOLID_BIT))) {
        return false; // solid hit found
    }

    const rayhip_material *mat =
        is_backfacing ? &sc.materials[md.back_mi & MATERIAL_INDEX_BITS] : &sc.materials[md.front_mi & MATERIAL_INDEX_BITS];

    const rayhip_vertex &v1 = sc.vertices[sc.vtx_indices[tri_index * 3 + 0]];
    const rayhip_vertex &v2 = sc.vertices[sc.vtx_indices[tri_index * 3 + 1]];
    const rayhip_vertex &v3 = sc.vertices[sc.vtx_indices[tri_index * 3 + 2]];

    const float w = 1.0f - inter.u - inter.v;
    const f2 uvs = mk2(v1.t[0], v1.t[1]) * w + mk2(v2.t[0], v2.t[1]) * inter.u + mk2(v3.t[0], v3.t[1]) * inter.v;

    const f2 mix_term_rand = get_scrambled_2d_rand(rand_dim + RAND_DIM_BSDF_PICK, rand_hash, tp.iteration - 1, sc.pmj);
    const f2 tex_rand = get_scrambled_2d_rand(rand_dim + RAND_DIM_TEX, rand_hash, tp.iteration - 1, sc.pmj);

    float trans_r = mix_term_rand.x;

    // resolve mix material
    while (mat->type == NODE_MIX) {
        float mix_val = mat->tangent_rotation_or_strength;
        const uint32_t base_texture = mat->textures[BASE_TEXTURE];
        if (base_texture != 0xffffffff) {
            const f4 tex_color = sample_color(sc, base_texture, uvs, 0, tex_rand);
            mix_val *= tex_color.x;
        }
        if (trans_r > mix_val) {
            mat = &sc.materials[mat->textures[MIX_MAT1]];
            trans_r = safe_div_pos(trans_r - mix_val, 1.0f - mix_val);
        } else {
            mat = &sc.materials[mat->textures[MIX_MAT2]];
            trans_r = safe_div_pos(trans_r, mix_val);
        }
    }

    if (mat->type != NODE_TRANSPARENT) {
        return false;
    }

    const bool can_terminate_path = get_transp_depth(r.depth) > tp.min_transp_depth;

    const float lum_ = fmaxf(r.c.x, fmaxf(r.c.y, r.c.z));
    const float p = mix_term_rand.y;
    const float q = can_terminate_path ? fmaxf(0.05f, 1.0f - lum_) : 0.0f;
    if (p < q || lum_ == 0.0f || get_transp_depth(r.depth) + 1 >= tp.max_transp_depth) {
        // terminate ray
        r.c = {0.0f, 0.0f, 0.0f};
        return false;
    }

    r.c.x *= mat->base_color[0] / (1.0f - q);
    r.c.y *= mat->base_color[1] / (1.0f - q);
    r.c.z *= mat->base_color[2] / (1.0f - q);

    const float t = inter.t + HIT_BIAS;
    ro += rd * t;

    // discard current intersection
    inter.v = -1.0f;
    inter.t = t_val - inter.t;

    r.depth += pack_ray_depth(0, 0, 0, 1);
    rand_dim += RAND_DIM_BOUNCE_COUNT;
    return true;
}

// is the surface the hit landed on opaque for the transparency loop?  (the first test of closest_resolve_transparency,
// CoreRef.cpp:3071-3079 -- true for almost every hit, and it needs nothing of the ray)
RT_HD bool hit_side_is_solid(const SceneView &sc, const Hit &inter) {
    if (sc.all_solid != 0u) { // (scene-wide flag set at upload: no per-hit fetch of the material word)
        return true;
    }
    const bool is_backfacing = (inter.prim_index < 0);
    const uint32_t tri_index = is_backfacing ? uint32_t(-inter.prim_index - 1) : uint32_t(inter.prim_index);
    const rayhip_tri_mat_data md = sc.tri_materials[tri_index];
    return (!is_backfacing && (md.front_mi & MATERIAL_SOLID_BIT)) || (is_backfacing && (md.back_mi & MATERIAL_SOLID_BIT));
}

// Ref::IntersectScene, closest hit + transparency/mix resolve loop.  CoreRef.cpp:3041-3158.
// In: r (o, d and the ray type in depth), inter (t preset by the caller: clip range for primary rays, MAX_DIST otherwise).
// Out: inter; r.c and r.depth are updated when transparent surfaces are crossed.
// `tail(r, tp)`: called once, right before the first non-solid hit is resolved, to complete what the walk itself never
// touches -- throughput, pixel, depth counters, and the per-layer parameters that depend on the pixel.  The device keeps
// those in memory instead of in registers across the walk (k_trace_closest); the default does nothing (ray complete).
struct RayTailComplete {
    RT_HD void operator()(Ray &, TraceParams &) const {}
};
template <int WIDE = 0, class Stack, class Tail = RayTailComplete>
RT_HD void intersect_scene_closest(const SceneView &sc, const TraceParams &tp_in, Ray &r, Hit &inter, Stack &st, TravCount *cnt,
                                   Tail &&tail = Tail()) {
    const f3 rd = r.d;
    f3 ro = r.o;
    const uint32_t ray_flags = (1u << get_ray_type(r.depth));

    TraceParams tp = tp_in;
    bool resolving = false;
    uint32_t rand_hash = 0, rand_dim = 0;
    while (true) {
        const float t_val = inter.t;

        const bool hit_found = traverse_closest<WIDE>(sc, ro, rd, ray_flags, tp.root_index, inter, st, cnt);
        if (!hit_found || hit_side_is_solid(sc, inter)) {
            break;
        }
        if (!resolving) {
            tail(r, tp);
            rand_hash = hash_combine(hash(r.xy), tp.rand_seed);
            rand_dim = RAND_DIM_BASE_COUNT + get_total_depth(r.depth) * RAND_DIM_BOUNCE_COUNT;
            resolving = true;
        }
        if (!closest_resolve_transparency(sc, tp, r, inter, t_val, rd, ro, rand_dim, rand_hash)) {
            break;
        }
    }

    inter.t += length(r.o - ro);
}

// What a non-solid surface lets through towards the light: the weights of the Transparent leaves of its material tree
// (CoreRef.cpp:3202-3246).  Runs for the few shadow rays that cross such a surface.
RT_HD_RARE f3 shadow_surface_throughput(const SceneView &sc, const TraceParams &tp, const Hit &inter, const uint32_t rand_dim, const uint32_t rand_hash) {
    const bool is_backfacing = (inter.prim_index < 0);
    const uint32_t tri_index = is_backfacing ? uint32_t(-inter.prim_index - 1) : uint32_t(inter.prim_index);

    const uint32_t mat_index = is_backfacing ? (sc.tri_materials[tri_index].back_mi & MATERIAL_INDEX_BITS)
                                             : (sc.tri_materials[tri_index].front_mi & MATERIAL_INDEX_BITS);

    const rayhip_vertex &v1 = sc.vertices[sc.vtx_indices[tri_index * 3 + 0]];
    const rayhip_vertex &v2 = sc.vertices[sc.vtx_indices[tri_index * 3 + 1]];
    const rayhip_vertex &v3 = sc.vertices[sc.vtx_indices[tri_index * 3 + 2]];

    const float w = 1.0f - inter.u - inter.v;
    const f2 sh_uvs = mk2(v1.t[0], v1.t[1]) * w + mk2(v2.t[0], v2.t[1]) * inter.u + mk2(v3.t[0], v3.t[1]) * inter.v;

    const f2 tex_rand = get_scrambled_2d_rand(rand_dim + RAND_DIM_TEX, rand_hash, tp.iteration - 1, sc.pmj);

    struct {
        uint32_t index;
        float weight;
    } stack[16];
    int stack_size = 0;

    stack[stack_size].index = mat_index;
    stack[stack_size++].weight = 1.0f;

    f3 throughput = {0.0f, 0.0f, 0.0f};

    while (stack_size--) {
        const rayhip_material *mat = &sc.materials[stack[stack_size].index];
        const float weight = stack[stack_size].weight;

        // resolve mix material
        if (mat->type == NODE_MIX) {
            float mix_val = mat->tangent_rotation_or_strength;
            const uint32_t base_texture = mat->textures[BASE_TEXTURE];
            if (base_texture != 0xffffffff) {
                const f4 tex_color = sample_color(sc, base_texture, sh_uvs, 0, tex_rand);
                mix_val *= tex_color.x;
            }
            stack[stack_size].index = mat->textures[MIX_MAT1];
            stack[stack_size++].weight = weight * (1.0f - mix_val);
            stack[stack_size].index = mat->textures[MIX_MAT2];
            stack[stack_size++].weight = weight * mix_val;
        } else if (mat->type == NODE_TRANSPARENT) {
            throughput += weight * mk3(mat->base_color);
        }
    }
    return throughput;
}

// Ref::IntersectScene(shadow_ray_t): visibility * throughput towards the light.  CoreRef.cpp:3160-3262.
template <int WIDE = 0, class Stack>
RT_HD f3 intersect_scene_shadow(const SceneView &sc, const TraceParams &tp, const ShadowRay &r, Stack &st,
                                TravCount *cnt) {
    const f3 rd = r.d;
    f3 ro = r.o;
    f3 rc = r.c;
    int depth = get_transp_depth(r.depth);

    const uint32_t px_hash = hash(r.xy);
    const uint32_t rand_hash = hash_combine(px_hash, tp.rand_seed);

    uint32_t rand_dim = RAND_DIM_BASE_COUNT + get_total_depth(r.depth) * RAND_DIM_BOUNCE_COUNT;

    float dist = r.dist > 0.0f ? r.dist : MAX_DIST;
    while (dist > HIT_BIAS) {
        Hit inter = make_hit();
        inter.t = dist;

        const bool solid_hit = traverse_any<WIDE>(sc, ro, rd, RAY_TYPE_SHADOW, tp.root_index, inter, st, cnt);

        if (solid_hit || depth > tp.max_transp_depth) {
            rc = {0.0f, 0.0f, 0.0f};
        }
        if (solid_hit || depth > tp.max_transp_depth || inter.v < 0.0f) {
            break;
        }

        const f3 throughput = shadow_surface_throughput(sc, tp, inter, rand_dim, rand_hash);

        rc *= throughput;
        if (lum(rc) < FLT_EPS_) {
            break;
        }

        const float t = inter.t + HIT_BIAS;
        ro += rd * t;
        dist -= t;

        ++depth;
        rand_dim += RAND_DIM_BOUNCE_COUNT;
    }

    return rc;
}

} // namespace rt
