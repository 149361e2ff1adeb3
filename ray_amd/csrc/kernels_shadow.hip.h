// kernels_shadow.hip.h -- K3, the any-hit kernels (included by kernels.hip.h): the nested form (every width, instrumented, test hooks) and
// the flat persistent form the passes use over the 4-wide tree.
#pragma once

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
// chunks per fetch of the dynamic hand-out (wavefront.hip.h: ChunkWalk): a shadow launch takes a chunk every ~11 ns chip-wide -- the rate one
// counter can hand out runs at -- so four
#ifndef RT_SHADOW_RUN
#define RT_SHADOW_RUN 4
#endif
template <int MIN_WAIT = RT_SHADOW_REFILL_MIN>
__global__ void __launch_bounds__(WAVE, RT_SHADOW_REFILL_MIN_WAVES) k_trace_shadow_refill(const SceneView sc, const TraceParams tp, const ShadowSoA shadow,
                                                                                     const RayQueue queue, const float limit, const int img_w,
                                                                                     float4 *__restrict__ temp_buf, float4 *__restrict__ out_rc,
                                                                                     uint32_t *__restrict__ stack_spill, const Layering layers,
                                                                                     uint32_t *__restrict__ work /* dynamic chunk hand-out, may be null */) {
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
    ChunkWalk walk(queue.live_chunks(), work, RT_SHADOW_RUN);

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
