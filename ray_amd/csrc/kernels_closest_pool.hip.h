// kernels_closest_pool.hip.h -- K2, the pooled form of the persistent kernel (included by kernels.hip.h): round 4's experiment with prepared
// rays in LDS.  Opt-in (RAYHIP_REFILL=4), bit-identical, measured slower than the refill kernel: DESIGN.md section 3a.
#pragma once

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
