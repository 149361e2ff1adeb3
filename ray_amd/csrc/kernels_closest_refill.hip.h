// kernels_closest_refill.hip.h -- K2, the persistent closest-hit kernel (included by kernels.hip.h, after the plain form): the product's
// closest-hit kernel for every bounce -- secondary rays lane by lane, primary rays in whole chunks.
#pragma once

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
// chunks per fetch of the dynamic hand-out: one (64 rays) -- a fetch per 75 us and wavefront, ~35 M atomics per second chip-wide
#ifndef RT_REFILL_RUN
#define RT_REFILL_RUN 1
#endif
template <int WIDE, int MIN_WAIT = RT_REFILL_MIN>
__global__ void __launch_bounds__(WAVE, RT_REFILL_MIN_WAVES) k_trace_closest_refill(const SceneView sc, const TraceParams tp, const RaySoA rays,
                                                               const HitSoA hits, const RayQueue queue, const int init_hits,
                                                               uint32_t *__restrict__ stack_spill, const Layering layers,
                                                               uint32_t *__restrict__ work /* dynamic chunk hand-out (wavefront.hip.h), may be null */) {
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
    ChunkWalk walk(queue.live_chunks(), work, RT_REFILL_RUN);

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
