// rt_travmachine.h -- the closest-hit traversal of rt_traverse.h unrolled into a resumable per-ray state machine.
//
// Why: with one ray per lane traced to completion, a wavefront is as slow as its longest ray and the measured
// VALU lane utilisation of the traversal kernel was 28 % (profiles/r01).  A state machine lets the persistent
// kernel in kernels.hip.h hand a finished lane its next ray immediately ("ray refill", Aila & Laine 2009) while
// the other lanes keep walking.
//
// The machine performs, for each ray, EXACTLY the sequence of node tests / leaf visits / triangle tests of
// walk_bvh2 + traverse_closest + intersect_scene_closest (reference CoreRef.cpp:1943-2025, 2428-2493, 3041-3158):
// the control flow of those loops is kept, only the program counter is made explicit:
//
//   NODE   = body of the inner `while (stack_size && cur is inner)` loop
//   LEAF   = body of the `while (leaf_node is a leaf)` loop (TLAS level: enter the instance; BLAS: test triangles)
//   FINISH = code after the TLAS loop (index indirection) + the transparency round of IntersectScene
//
// so hits AND the work counters are identical to the straight-line version (tests compare both).
#pragma once

#include "rt_traverse.h"

namespace rt {

enum : uint32_t { TM_IDLE = 0, TM_NODE = 1, TM_LEAF = 2, TM_FINISH = 3 };

struct ClosestMachine {
    // current-level ray (world space at TLAS level, object space inside an instance)
    f3 o, d, inv_d;
    // world-space origin of the current transparency segment and the world direction
    f3 ro, rd;
    Hit h;
    float t_val;         // inter.t at the start of the current IntersectScene round
    uint32_t cur;        // node the walk is at
    uint32_t leaf_node;  // pending leaf word (0 = none)
    uint32_t cur_tlas;   // TLAS `cur` saved while inside an instance
    uint32_t base;       // stack size below the current level's sentinel
    uint32_t mi_index;   // instance being traversed (BLAS level)
    uint32_t ray_flags;
    uint32_t rand_dim;
    uint32_t state;      // TM_*
    bool in_blas;
    bool res;            // any triangle hit in this round
};

RT_HD bool tm_is_leaf(uint32_t w) { return (w & BVH2_PRIM_COUNT_BITS) != 0; }

template <class Stack> RT_HD void tm_begin_round(ClosestMachine &m, const TraceParams &tp, Stack &st) {
    m.t_val = m.h.t;
    m.res = false;
    m.o = m.ro;
    m.d = m.rd;
    m.inv_d = safe_invert(m.rd);
    m.in_blas = false;
    m.cur_tlas = 0;
    m.mi_index = 0;
    // walk_bvh2 prologue at TLAS level
    st.size = 0;
    m.base = 0;
    st.push(0x1fffffffu);
    m.cur = tp.root_index;
    m.leaf_node = 0;
    // `while (size > base)` + inner loop condition
    m.state = (!tm_is_leaf(m.cur)) ? TM_NODE : TM_FINISH; // (a leaf root never terminates in the reference either)
}

// start a ray: r.o/r.d/r.depth/r.xy must be loaded; h preset by the caller
template <class Stack> RT_HD void tm_start(ClosestMachine &m, const TraceParams &tp, const Ray &r, const Hit &h, Stack &st) {
    m.ro = r.o;
    m.rd = r.d;
    m.h = h;
    m.ray_flags = (1u << get_ray_type(r.depth));
    m.rand_dim = RAND_DIM_BASE_COUNT + get_total_depth(r.depth) * RAND_DIM_BOUNCE_COUNT;
    tm_begin_round(m, tp, st);
}

// Control transfer after a leaf-loop iteration (`advance` = true: run `leaf_node = cur; if (cur is leaf) cur = pop();`
// first) or after the inner loop ended with no pending leaf (`advance` = false).  Decides the next state; when a
// BLAS level completes it restores the TLAS walk and continues the TLAS leaf loop (at most two levels, so the
// loop below runs at most twice -- written as a loop instead of recursion).
template <class Stack> RT_HD void tm_resume(ClosestMachine &m, Stack &st, bool advance) {
    for (int level = 0; level < 2; ++level) {
        if (advance) {
            m.leaf_node = m.cur;
            if (tm_is_leaf(m.cur)) {
                m.cur = st.pop();
            }
            if (tm_is_leaf(m.leaf_node)) {
                m.state = TM_LEAF;
                return;
            }
        }
        // leaf loop over: outer `while (size > base)` + inner loop condition
        m.leaf_node = 0;
        if (st.size > m.base && !tm_is_leaf(m.cur)) {
            m.state = TM_NODE;
            return;
        }
        // level finished
        st.size = m.base;
        if (!m.in_blas) {
            m.state = TM_FINISH;
            return;
        }
        // back to the TLAS leaf loop: restore the world-space ray and the TLAS walk position
        m.in_blas = false;
        m.base = 0;
        m.o = m.ro;
        m.d = m.rd;
        m.inv_d = safe_invert(m.rd);
        m.cur = m.cur_tlas;
        advance = true;
    }
}

// one iteration of the inner node loop, CoreRef.cpp:1961-2010
template <class Stack> RT_HD void tm_node_step(ClosestMachine &m, const SceneView &sc, Stack &st, TravCount *cnt) {
    bvh2_node_step(sc.nodes, m.o, m.inv_d, m.h.t, m.cur, st, cnt);
    if (tm_is_leaf(m.cur) && !tm_is_leaf(m.leaf_node)) {
        m.leaf_node = m.cur;
        m.cur = st.pop();
    }
    if (tm_is_leaf(m.leaf_node)) {
        m.state = TM_LEAF;
    } else if (!(st.size > m.base && !tm_is_leaf(m.cur))) {
        // inner loop ends without a pending leaf -> the leaf loop is skipped -> outer loop re-check
        tm_resume(m, st, false);
    }
}

// one iteration of the leaf loop
template <class Stack> RT_HD void tm_leaf_step(ClosestMachine &m, const SceneView &sc, Stack &st, TravCount *cnt) {
    if (!m.in_blas) {
        // TLAS leaf: CoreRef.cpp:1996-2014
        const uint32_t mi_index = (m.leaf_node & BVH2_PRIM_INDEX_BITS);
        const rayhip_mesh_instance &mi = sc.mesh_instances[mi_index];
        if ((mi.ray_visibility & m.ray_flags) != 0) {
            if (cnt) {
                ++cnt->instances;
            }
            m.cur_tlas = m.cur;
            m.mi_index = mi_index;
            m.o = transform_point(m.ro, mi.inv_xform);
            m.d = transform_direction(m.rd, mi.inv_xform);
            m.inv_d = safe_invert(m.d);
            m.in_blas = true;
            // walk_bvh2 prologue at BLAS level
            m.base = st.size;
            st.push(0x1fffffffu);
            m.cur = mi.node_index;
            m.leaf_node = 0;
            if (!tm_is_leaf(m.cur)) {
                m.state = TM_NODE;
            } else {
                tm_resume(m, st, false); // degenerate: leaf root (the reference would spin)
            }
            return;
        }
        tm_resume(m, st, true);
    } else {
        // BLAS leaf: CoreRef.cpp:2478-2483
        const int tri_start = int(m.leaf_node & BVH2_PRIM_INDEX_BITS),
                  tri_end = int(tri_start + ((m.leaf_node & BVH2_PRIM_COUNT_BITS) >> 29) + 1);
        if (cnt) {
            cnt->tris += uint32_t(tri_end - tri_start);
        }
        m.res |= intersect_tris_closest(m.o, m.d, sc.tris, tri_start, tri_end, int(m.mi_index), m.h);
        tm_resume(m, st, true);
    }
}

// After the TLAS loop.  Returns true when the ray is complete (m.h is final, r.c / r.depth may have changed);
// false when another round was started (transparent surface crossed).
template <class Stack>
RT_HD bool tm_finish(ClosestMachine &m, const SceneView &sc, const TraceParams &tp, Ray &r, Stack &st) {
    // resolve primitive index indirection, CoreRef.cpp:2017-2022
    if (m.h.prim_index < 0) {
        m.h.prim_index = -int(sc.tri_indices[-m.h.prim_index - 1]) - 1;
    } else {
        m.h.prim_index = int(sc.tri_indices[m.h.prim_index]);
    }
    if (m.res) {
        const uint32_t rand_hash = hash_combine(hash(r.xy), tp.rand_seed);
        if (closest_resolve_transparency(sc, tp, r, m.h, m.t_val, m.rd, m.ro, m.rand_dim, rand_hash)) {
            tm_begin_round(m, tp, st);
            return false;
        }
    }
    m.h.t += length(r.o - m.ro);
    m.state = TM_IDLE;
    return true;
}

// straight-line driver (host simulation / validation of the machine against intersect_scene_closest)
template <class Stack>
RT_HD void intersect_scene_closest_machine(const SceneView &sc, const TraceParams &tp, Ray &r, Hit &inter, Stack &st,
                                           TravCount *cnt) {
    ClosestMachine m;
    tm_start(m, tp, r, inter, st);
    for (;;) {
        if (m.state == TM_NODE) {
            tm_node_step(m, sc, st, cnt);
        } else if (m.state == TM_LEAF) {
            tm_leaf_step(m, sc, st, cnt);
        } else {
            if (tm_finish(m, sc, tp, r, st)) {
                break;
            }
        }
    }
    inter = m.h;
}

} // namespace rt
