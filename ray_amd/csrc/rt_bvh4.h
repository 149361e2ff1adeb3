// rt_bvh4.h -- 4-wide BLAS with 8-bit quantised child boxes: the acceleration structure the non-instrumented
// traversal kernels walk on MI355X.
//
// Why not the reference's BVH2 as it is: a bvh2_node_t visit is four 16-byte gathers for two boxes, and the texture
// addresser of a CU retires about one lane-address per cycle for 128-bit loads (tools/gather_bench.hip, TA_BUSY 60 %
// in profiles/r01), so the closest-hit kernel is bound by the NUMBER of node fetches and by the length of the
// dependent fetch chain, not by bytes or flops.  Collapsing two BVH2 levels into one node halves both: the same 64
// bytes / four gathers now carry four boxes.
//
// What makes it safe (results must be the reference's): the BVH only culls.  A child box here is the reference's
// fp32 box rounded OUTWARDS onto a 256-step grid spanning the node (lo: floor, hi: ceil; checked after rounding with
// the very expression the device evaluates, and widened until it contains the fp32 box), and the slab test is the
// reference's own bbox_test on the de-quantised box.  Floating-point subtraction and multiplication are monotonic, so
// whenever the reference's test accepts a child with its exact box, this test accepts it with the wider box: every
// triangle the reference tests with a chance to hit is tested here, IntersectTri is unchanged, and the closest hit
// (t, u, v, triangle) is the same -- only exact-t ties between different triangles could resolve differently, as they
// already do between the reference's own BVH2 and wide-BVH back-ends.  Any-hit (shadow) rays have one more order
// dependence, also the reference's own: IntersectScene(shadow_ray_t) stops a ray whose throughput fell below FLT_EPS after
// a transparent surface and returns that (~1e-8) throughput, while a walk that happens to reach a solid occluder first
// returns 0 -- which of the two a ray gets depends on the leaf order, i.e. on the tree.  Found by the scene fuzzer
// (tests/test_hostsim_parity.py: random_cornell seed 215 in wide mode: 8 pixels differ by <= 4e-7); the committed wide-BVH
// parity scenes are bit-exact.  The instrumented kernels
// (RAYHIP_FLAG_COUNT_TRAVERSAL) keep walking the BVH2, so the algorithmic-bytes counters stay those of the reference
// algorithm.  The TLAS stays BVH2 (a handful of nodes).
//
// Node, 64 bytes, 16-byte aligned, fetched as 4 x global_load_dwordx4:
//   [0]  org.xyz                float  grid origin = node box min
//        step.x                 float  grid step of the x axis, a power of two (steps as floats since round 3: the kernel is bound by
//                                      instruction issue, and three exponent bytes cost a shift and a mask each to unpack)
//   [1]  child[4]               inner: bvh4 node index | leaf: (count-1)<<29 | first entry in tris[] (reference leaf word)
//   [2]  qlo[x][4] qlo[y][4] qlo[z][4] qhi[x][4]     u8, [axis][child]
//   [3]  qhi[y][4] qhi[z][4] step.y step.z            (empty slot: child = BVH4_EMPTY)
#pragma once

#include "rt_isect.h"

namespace rt {


#ifndef RT_BVH4_PINNED_FETCH
#define RT_BVH4_PINNED_FETCH 1
#endif

struct alignas(16) Bvh4Node {
    float org[3];
    float step_x; // grid steps: powers of two
    uint32_t child[4];
    uint32_t qlo[3]; // byte c of qlo[a] = child c, axis a
    uint32_t qhi[3];
    float step_y, step_z;
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node must be one 64-byte fetch unit");

constexpr uint32_t BVH4_SENTINEL = 0x1fffffffu; // same stack sentinel as the BVH2 walk (never a node index)
constexpr uint32_t BVH4_EMPTY = 0xffffffffu;    // unused child slot

// one de-quantised box coordinate: a single fused multiply-add, the same operation on host (builder check) and device
RT_HD float bvh4_dequant(const uint32_t q, const float scale, const float org) { return __builtin_fmaf(float(q), scale, org); }

// One visit of a 4-wide node: test the four children against [0, t] and return them sorted by entry distance
// (ref[0] nearest); n_hit = number of children hit.
//
// The slab test runs in the ray's parameter space (the compressed-wide-BVH formulation of Ylitie et al. 2017): plane q of
// axis a of this node's grid is crossed at  t = q * (step_a * inv_d_a) + (org_a - o_a) * inv_d_a,  so a child costs one
// byte-to-float conversion and one fma per plane instead of de-quantise, subtract, multiply, and the sign of inv_d_a says
// which of (qlo, qhi) is the entry plane -- no min / max per axis.  The box test only CULLS, so it does not have to be the
// reference's arithmetic, it has to be CONSERVATIVE against it: whenever the reference's bbox_test (rt_isect.h) accepts the
// exact fp32 child box, this test accepts the quantised box that contains it.  Error budget (eps = 2^-24, M_a = |base_a| + 255 |k_a|:
// every plane of the node's grid is crossed at a |t| <= M_a):
//   * base_a = fl(fl(org_a - o_a) * inv_d_a): two roundings;  step_a * inv_d_a: exact (step is a power of two);  the fma: one;  the
//     padding folded into the addend: one more  ->  |computed - real plane parameter| <= 4 eps M_a;
//   * the reference's own value is within 3 eps M_a of the real one for ITS plane, and it stretches tmax by 1 + 2^-22 (4 eps)
//     (rt_isect.h: bbox_test);
//   => entry planes moved back / exit planes moved forward by E_a = 2^-20 M_a = 16 eps M_a cover both sides (7 and 11 eps) with
//      room to spare; no relative slack on tmin / tmax afterwards (rounds 1-2 used 2^-22 M_a plus a slack of 2^-21: two fma and two
//      compares more per child);
//   * the quantised box contains the child box in REAL arithmetic (bvh4_build.h checks org + q * step in double).
// Checked by the bit-exact frame / hit tests of the wide walk against the BVH2 walk and the oracle.
RT_HD void bvh4_test_node(const Bvh4Node *nodes4, const uint32_t cur, const f3 ro, const f3 inv_d, const float t, uint32_t ref[4],
                          uint32_t &n_hit, float *out_dist = nullptr) {
    RT_PROF_T(16)
    RT_PROF_LANES(0)
    const float4 *np = reinterpret_cast<const float4 *>(nodes4 + cur);
#if defined(__HIP_DEVICE_COMPILE__) && RT_BVH4_PINNED_FETCH
    // the four loads of a node issued back to back, one wait.  Left to the scheduler, the folded test below came out as
    // "load, load, load, WAIT for the first, compare, load the fourth, wait": two memory round trips per visit and K2 9 % slower
    // with 10 % fewer instructions (round 3; the fourth load's destination overlapped the address registers, so it went last,
    // and a compare on the first load's result was placed in front of it)
    float4 w0, w1, w2, w3;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dwordx4 %2, %4, off offset:32\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                 : "v"(np)
                 : "memory");
#else
    const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
#endif
    RT_PROF_WAIT(w0, w1, w2, w3)
    RT_PROF_T(17)
    const float step[3] = {w0.w, w3.z, w3.w};
    const uint32_t child[4] = {float_as_uint(w1.x), float_as_uint(w1.y), float_as_uint(w1.z), float_as_uint(w1.w)};

    // the grid in ray-parameter space, per axis: t(q) = q * k + base, padded by the error bound on the side that matters
    const float org[3] = {w0.x, w0.y, w0.z}, o[3] = {ro.x, ro.y, ro.z}, id[3] = {inv_d.x, inv_d.y, inv_d.z};
    const uint32_t qlo_w[3] = {float_as_uint(w2.x), float_as_uint(w2.y), float_as_uint(w2.z)};
    const uint32_t qhi_w[3] = {float_as_uint(w2.w), float_as_uint(w3.x), float_as_uint(w3.y)};
    float k[3], base_in[3], base_out[3];
    uint32_t q_in[3], q_out[3]; // the four children's entry / exit plane indices of this axis, one byte each
    for (int a = 0; a < 3; ++a) {
        k[a] = step[a] * id[a];
        const float base = (org[a] - o[a]) * id[a];
        const float err = __builtin_fmaf(255.0f, fabsf(k[a]), fabsf(base)) * 9.5367431640625e-07f; // 2^-20: covers both sides' roundings (rt_bvh8.h)
        base_in[a] = base - err, base_out[a] = base + err;
        const bool forward = id[a] >= 0.0f;
        q_in[a] = forward ? qlo_w[a] : qhi_w[a], q_out[a] = forward ? qhi_w[a] : qlo_w[a];
    }

    const float none = 3.402823466e+38f;
    float dist[4];
    n_hit = 0;
    for (int c = 0; c < 4; ++c) {
        const int sh = 8 * c;
        float t_in[3], t_out[3];
        for (int a = 0; a < 3; ++a) {
            t_in[a] = __builtin_fmaf(float((q_in[a] >> sh) & 0xffu), k[a], base_in[a]);
            t_out[a] = __builtin_fmaf(float((q_out[a] >> sh) & 0xffu), k[a], base_out[a]);
        }
        // the whole error budget sits in the padding of the planes (2^-20 M_a per plane, derivation in rt_bvh8.h: the same grid, the
        // same arithmetic): no relative slack afterwards, and  max(tmin, 0) <= min(tmax, t)  -- implied by the reference's
        // tmin <= tmax && tmin <= t && tmax > 0 -- is two 3-operand min / max and one compare (rounds 1-2: a relative slack of
        // 2^-21 on tmin / tmax, two fma and three compares per child)
        const float tmin = fmaxf(fmaxf(fmaxf(t_in[0], t_in[1]), t_in[2]), 0.0f), tmax = fminf(fminf(fminf(t_out[0], t_out[1]), t_out[2]), t);
        const bool hit = tmin <= tmax && child[c] != BVH4_EMPTY;
        dist[c] = hit ? tmin : none;
        ref[c] = child[c];
        n_hit += hit ? 1u : 0u;
    }
    // sorting network on (dist, ref), ascending: (0,1) (2,3) (0,2) (1,3) (1,2); children that were not hit sort last
#define RT_CSWAP(a, b)                                                                                                  \
    {                                                                                                                   \
        const bool sw = dist[b] < dist[a];                                                                              \
        const float da = sw ? dist[b] : dist[a], db = sw ? dist[a] : dist[b];                                           \
        const uint32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];                                            \
        dist[a] = da, dist[b] = db, ref[a] = ra, ref[b] = rb;                                                           \
    }
    RT_CSWAP(0, 1)
    RT_CSWAP(2, 3)
    RT_CSWAP(0, 2)
    RT_CSWAP(1, 3)
    RT_CSWAP(1, 2)
#undef RT_CSWAP
    if (out_dist) { // (entry distances of the sorted children, conservative: never larger than the real ones)
        for (int c = 0; c < 4; ++c) {
            out_dist[c] = dist[c];
        }
    }
    RT_PROF_T(18)
}

// One node visit of the ordered walk with the stack update (see "stack discipline" below): tests the four children of
// `cur`, continues with the nearest one that was hit and pushes the others (farthest first), or pops when none was hit.
template <class Stack>
RT_HD void bvh4_visit(const Bvh4Node *nodes4, const f3 ro, const f3 inv_d, const float t, Stack &st, uint32_t &cur, uint32_t &tos,
                      uint32_t &size, TravCount *cnt = nullptr) {
    uint32_t ref[4], n_hit;
    if (cnt) {
        ++cnt->nodes4;
        cnt->max_stack = size > cnt->max_stack ? size : cnt->max_stack;
    }
    bvh4_test_node(nodes4, cur, ro, inv_d, t, ref, n_hit);
    if (n_hit == 0) {
        cur = tos;
        tos = st.read_at(--size);
    } else {
        cur = ref[0];
        // bottom -> top: old tos, farthest ... third nearest; the second nearest becomes the new tos
        const uint32_t s1 = (n_hit == 4) ? ref[3] : ref[2];
        if (st.fast_range(size + 3)) {
            st.write3_fast(size, tos, s1, ref[2]);
        } else if (n_hit > 1) {
            st.write_at(size, tos);
            if (n_hit > 2) {
                st.write_at(size + 1, s1);
            }
            if (n_hit > 3) {
                st.write_at(size + 2, ref[2]);
            }
        }
        size += n_hit - 1;
        tos = (n_hit > 1) ? ref[1] : tos;
    }
}

// Ordered walk over a 4-wide BLAS.  `leaf(word)` gets a reference leaf word and returns true to stop (any-hit early
// out).  `t_ref` is re-read at every node so that hits found in earlier leaves prune.
//
// Stack discipline (measured: with a plain LDS push/pop per child the stack code was 35 % of the kernel's wave time):
//   * the top of the stack lives in a register (`tos`).  A pop hands out the register and starts the LDS read of
//     the next entry, whose latency then hides behind the next node fetch instead of sitting in front of it;
//   * a node visit pushes up to three children with three UNCONDITIONAL stores to consecutive slots (one address,
//     immediate offsets) and then advances `size` by the number that were real -- stale values above `size` are
//     never read -- instead of three compare/branch/store sequences;
//   * only when the three slots would leave the LDS part of the stack does it fall back to the generic path that
//     can spill to HBM.
template <class Stack, class LeafFn>
RT_HD bool walk_bvh4(const Bvh4Node *nodes4, const uint32_t root, const f3 ro, const f3 inv_d, const float &t_ref, Stack &st,
                     LeafFn &&leaf, TravCount *cnt = nullptr) {
    const uint32_t base = st.size;
    uint32_t size = base;
    st.write_at(size++, BVH4_SENTINEL); // second sentinel, so that the read-ahead of a pop never leaves this level
    uint32_t tos = BVH4_SENTINEL;
    uint32_t cur = root;
    bool early_out = false;
    (void)early_out;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_WALK_NO_MAJORITY)
    // Majority scheduling of the two phases: instead of walking inner nodes until EVERY lane holds a leaf (the last
    // lanes still descending keep the others waiting) and then testing leaves until every lane is back at a node, each
    // iteration runs the one phase most of the wavefront's lanes are waiting for.  Per ray the sequence of node visits
    // and leaf tests is unchanged, so are the results.  Measured (Bistro-class, 1080p): lanes busy in a node visit
    // 28 % -> 48 %, closest-hit kernel 3.21 -> 2.83 ms.
    for (;;) {
        const bool at_node = (cur & BVH2_PRIM_COUNT_BITS) == 0 && cur != BVH4_SENTINEL;
        const bool at_leaf = (cur & BVH2_PRIM_COUNT_BITS) != 0;
        const int n_node = __popcll(__ballot(at_node)), n_leaf = __popcll(__ballot(at_leaf));
        if (n_node == 0 && n_leaf == 0) {
            break;
        }
#ifdef RT_PROFILE_TRACE
        if (int(__lane_id()) == __ffsll((long long)__ballot(1)) - 1) { // lanes waiting for a node / a leaf, loop iterations
            ::rt::s_prof_acc[6] += (unsigned long long)n_node, ::rt::s_prof_acc[7] += (unsigned long long)n_leaf, ::rt::s_prof_acc[8] += 1ull;
        }
#endif
        if (n_node >= n_leaf) { // weights 2:1, 3:2, 2:3, 1:2 measured: all within 1 % or slower
            if (at_node) {
                bvh4_visit(nodes4, ro, inv_d, t_ref, st, cur, tos, size, cnt);
            }
        } else if (at_leaf) {
            if (leaf(cur)) {
                // any-hit early out: this lane is done; the others keep going (they cannot return from here without
                // leaving the wave-level loop, so the lane parks on the sentinel)
                cur = BVH4_SENTINEL;
                early_out = true;
            } else {
                cur = tos;
                tos = st.read_at(--size);
            }
        }
    }
    st.size = base;
    return early_out;
#else // one ray at a time (host build of the same walk; tests/hostsim)
#ifdef RT_EXPERIMENT_DISTCULL
    // experiment (host build only, -DRT_EXPERIMENT_DISTCULL): entry distances travel with the stack entries, a popped entry whose
    // box starts behind the current hit is dropped without being fetched.  Measured on the Sponza-class atrium (bounces 0-8,
    // leaves refined to 2): 19.63 -> 18.31 node visits and 5.12 -> 4.86 triangle tests per closest-hit ray (1.46 entries
    // culled per ray), shadow rays unchanged, frames identical.  Not in the device walk: the second LDS array it needs (16-bit
    // distances) halves the stack depth that fits at 5 waves per SIMD.
    {
        float dstack[4 * MAX_STACK_SIZE];
        float cur_d = -1.0f, tos_d = -1.0f;
        for (;;) {
            while (cur != BVH4_SENTINEL && cur_d > t_ref) { // dead on arrival: pop
                if (cnt) {
                    ++cnt->instances; // (re-used as "entries culled by distance" in this experiment)
                }
                cur = tos, cur_d = tos_d;
                --size;
                tos = st.read_at(size), tos_d = dstack[size];
            }
            if (cur == BVH4_SENTINEL) {
                break;
            }
            if ((cur & BVH2_PRIM_COUNT_BITS) == 0) {
                uint32_t ref[4], n_hit;
                float d4[4];
                if (cnt) {
                    ++cnt->nodes4;
                }
                bvh4_test_node(nodes4, cur, ro, inv_d, t_ref, ref, n_hit, d4);
                if (n_hit == 0) {
                    cur = tos, cur_d = tos_d;
                    --size;
                    tos = st.read_at(size), tos_d = dstack[size];
                } else {
                    // push farthest first: old tos, then ref[n_hit-1] .. ref[2]; ref[1] becomes tos
                    for (uint32_t k = n_hit; k-- > 1;) {
                        st.write_at(size, tos), dstack[size] = tos_d;
                        ++size;
                        tos = ref[k], tos_d = d4[k];
                    }
                    cur = ref[0], cur_d = d4[0];
                }
            } else {
                if (leaf(cur)) {
                    st.size = base;
                    return true;
                }
                cur = tos, cur_d = tos_d;
                --size;
                tos = st.read_at(size), tos_d = dstack[size];
            }
        }
        st.size = base;
        return false;
    }
#endif
    for (;;) {
        while ((cur & BVH2_PRIM_COUNT_BITS) == 0 && cur != BVH4_SENTINEL) {
            bvh4_visit(nodes4, ro, inv_d, t_ref, st, cur, tos, size, cnt);
            RT_PROF_T(19)
        }
        if (cur == BVH4_SENTINEL) {
            break;
        }
        if (leaf(cur)) {
            st.size = base;
            return true;
        }
        cur = tos;
        tos = st.read_at(--size);
    }
    st.size = base;
    return false;
#endif
}

} // namespace rt
