// rt_bvh8.h -- 8-wide BLAS with 8-bit quantised child boxes in one 128-byte line: what the product traversal kernels walk
// on MI355X (round 3; the 4-wide form of rt_bvh4.h stays as the A/B alternative, RAYHIP_BVH_WIDTH=4).
//
// Why: the closest-hit kernel is bound by the number of DEPENDENT fetches a ray performs and by the instructions between
// them (profiles/r02: waves parked on memory 52 % of their cycles at 5 waves per SIMD, VALU at half its issue rate, HBM at
// 0.3 of its peak).  An 8-wide node halves the levels of the tree: fewer round trips per ray, and per round trip one
// 128-byte line instead of 64 bytes -- bytes are what this kernel has to spare.  The layout follows the compressed wide
// BVH of Ylitie, Karras and Laine (HPG 2017), re-cut for wave64 and this code's leaf conventions:
//   * the children of a node sit in slots ordered by OCTANT: slot s holds the child lying towards corner s of the node
//     (bit a of s = 1: the + side of axis a; greedy assignment in bvh8_build.h).  A ray with direction signs `oct` visits
//     the slots in the order of s ^ oct -- front to back along every axis -- so NO distance sort happens per visit: the
//     hit mask, permuted by three conditional bit swaps, is the traversal order;
//   * inner children are stored CONSECUTIVELY (child_base + number of inner slots below s) and the triangles of all leaf
//     children consecutively behind tri_base, so one stack entry describes every pending child of a node: (child_base,
//     pending mask | inner mask) -- 8 bytes per LEVEL instead of 4 bytes per pending CHILD;
//   * a leaf child is a range of the reference's own triangle records, handed to the same leaf functions as in the other
//     walks (IntersectTris_ClosestHit / _AnyHit restated in rt_traverse.h), i.e. per leaf the reference's loop.
//
// What makes it safe is what makes rt_bvh4.h safe: the tree only culls, every child box is the reference's fp32 box
// rounded outwards onto the node's 256-step grid (bvh8_build.h checks containment in real arithmetic), the slab test is
// conservative against the reference's bbox_test (error budget below), and the triangle test is the reference's.  The
// visiting ORDER differs from the reference's walk (octant order instead of nearest child first, all leaf children of a
// node before its inner children), which changes nothing but the winner of an exact-distance tie between two triangles --
// as between the reference's own BVH2 and wide back-ends -- and, for shadow rays through transparent surfaces, which of
// two order-dependent outcomes the reference's any-hit loop produces (rt_bvh4.h, head comment).
//
// Node: 80 bytes used, 128-byte stride and alignment (one L2 line, two 64-byte HBM sectors), fetched as 5 x dwordx4:
//   [0] org.xyz                 float   grid origin = node box min
//       exps | imask            3 x u8 biased exponents of the per-axis grid step, u8 mask of the slots holding inner nodes
//   [1] child_base, tri_base    u32     first inner child (index into nodes8), first triangle record of the leaf children
//       meta[8]                 u8      leaf slot: (count - 1) << 5 | (offset + 1), offset <= 30 relative to tri_base;
//                                       0: inner or empty slot
//   [2] qlo.x[8] qlo.y[8]       u8      [axis][slot]; an empty slot holds the inverted box (255, 0)
//   [3] qlo.z[8] qhi.x[8]
//   [4] qhi.y[8] qhi.z[8]
#pragma once

#include "rt_isect.h"

namespace rt {

struct alignas(16) Bvh8Node {
    float org[3];
    uint32_t exps_imask; // bytes 0..2: biased exponent of the x, y, z grid step; byte 3: inner-slot mask
    uint32_t child_base, tri_base;
    uint32_t meta[2];   // byte c of meta[h] = slot 4 h + c
    uint32_t qlo[3][2]; // byte c of qlo[a][h] = slot 4 h + c, axis a
    uint32_t qhi[3][2];
    uint32_t _pad[12];
};
static_assert(sizeof(Bvh8Node) == 128, "Bvh8Node must be one 128-byte line");

constexpr uint32_t BVH8_SENTINEL = 0xffffffffu; // child_base of the stack sentinel (its pending mask is 0)
constexpr uint32_t BVH8_MAX_LEAF_OFFSET = 30;

// 8-bit mask with bit s moved to bit s ^ x (x: 3 bits): three butterfly stages
RT_HD uint32_t bvh8_permute(uint32_t m, const uint32_t x) {
    m = (x & 1u) ? (((m & 0x55u) << 1) | ((m >> 1) & 0x55u)) : m;
    m = (x & 2u) ? (((m & 0x33u) << 2) | ((m >> 2) & 0x33u)) : m;
    m = (x & 4u) ? (((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu)) : m;
    return m;
}
// sum of the four bytes of v, plus acc (v_sad_u8 against zero on the device)
RT_HD uint32_t bvh8_byte_sum(const uint32_t v, const uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sad_u8(v, 0u, acc);
#else
    return (v & 0xffu) + ((v >> 8) & 0xffu) + ((v >> 16) & 0xffu) + (v >> 24) + acc;
#endif
}
// direction signs of a ray as the XOR constant of the slot order: 7 ^ oct, oct bit a = (d_a < 0)
RT_HD uint32_t bvh8_oct_inv(const f3 inv_d) {
    return (inv_d.x >= 0.0f ? 1u : 0u) | (inv_d.y >= 0.0f ? 2u : 0u) | (inv_d.z >= 0.0f ? 4u : 0u);
}

// what one node visit leaves behind
struct Bvh8Visit {
    uint32_t child_base; // the node's inner children ...
    uint32_t bits;       // ... pending ones in traversal order in bits 8..15 (highest first), inner-slot mask in bits 0..7
    uint32_t tri_base;   // the node's leaf children ...
    uint32_t leaf[2];    // ... meta bytes of the ones that were hit (others zeroed)
};

// One visit of an 8-wide node: slab-test the eight children against [0, t].
//
// As in rt_bvh4.h the test runs in the ray's parameter space: plane q of axis a is crossed at t = q * k_a + base_a with
// k_a = step_a * inv_d_a (exact: a power of two times inv_d) and base_a = (org_a - o_a) * inv_d_a; the sign of inv_d_a says
// which of (qlo, qhi) is the entry plane.  It must be CONSERVATIVE against the reference's bbox_test on the exact fp32 child
// box (which the quantised box contains in real arithmetic).  With eps = 2^-24 and M_a = |base_a| + 255 |k_a| (every plane of
// the node's grid is crossed at a |t| <= M_a):
//   * computed plane parameter vs the real one of the QUANTISED plane: two roundings in base_a, one in the fma, one when
//     the padding is folded into the addend: <= 4 eps M_a;
//   * the reference's own value vs the real one of ITS plane: <= 3 eps M_a, and it stretches tmax by 1 + 2^-22 (4 eps);
//   => entry planes moved back / exit planes moved forward by E_a = 2^-20 M_a = 16 eps M_a cover both sides (7 and 11 eps
//      needed) with room to spare, and no relative slack on tmin / tmax is needed afterwards (rt_bvh4.h spends two fma per
//      child on one).  The padding is negligible next to the 8-bit grid itself (1 / 255 of the node's extent).
//   * accept iff max(tmin, 0) <= min(tmax, t): implied by the reference's  tmin <= tmax && tmin <= t && tmax > 0.
// Empty slots carry an inverted box and meta 0 and are outside imask: even if a degenerate node let one pass the slab test
// it contributes neither a leaf byte nor an inner bit.
RT_HD void bvh8_test_node(const Bvh8Node *nodes8, const uint32_t node, const f3 ro, const f3 inv_d, const float t, const uint32_t oct_inv,
                          Bvh8Visit &out) {
    RT_PROF_T(16)
    RT_PROF_LANES(0)
    const float4 *np = reinterpret_cast<const float4 *>(nodes8 + node);
#if defined(__HIP_DEVICE_COMPILE__) && RT_BVH4_PINNED_FETCH
    float4 w0, w1, w2, w3, w4; // (the five loads back to back, one wait: rt_bvh4.h)
    asm volatile("global_load_dwordx4 %0, %5, off\n\tglobal_load_dwordx4 %1, %5, off offset:16\n\tglobal_load_dwordx4 %2, %5, off offset:32\n\t"
                 "global_load_dwordx4 %3, %5, off offset:48\n\tglobal_load_dwordx4 %4, %5, off offset:64\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4)
                 : "v"(np)
                 : "memory");
#else
    const float4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3], w4 = np[4];
#endif
    RT_PROF_WAIT(w0, w1, w2, w4)
    RT_PROF_T(17)
    const uint32_t exps = float_as_uint(w0.w);
    const float org[3] = {w0.x, w0.y, w0.z}, o[3] = {ro.x, ro.y, ro.z}, id[3] = {inv_d.x, inv_d.y, inv_d.z};
    const uint32_t qlo_w[3][2] = {{float_as_uint(w2.x), float_as_uint(w2.y)}, {float_as_uint(w2.z), float_as_uint(w2.w)}, {float_as_uint(w3.x), float_as_uint(w3.y)}};
    const uint32_t qhi_w[3][2] = {{float_as_uint(w3.z), float_as_uint(w3.w)}, {float_as_uint(w4.x), float_as_uint(w4.y)}, {float_as_uint(w4.z), float_as_uint(w4.w)}};
    float k[3], base_in[3], base_out[3];
    uint32_t q_in[3][2], q_out[3][2];
    for (int a = 0; a < 3; ++a) {
        k[a] = uint_as_float(((exps >> (8 * a)) & 0xffu) << 23) * id[a];
        const float base = (org[a] - o[a]) * id[a];
        const float err = __builtin_fmaf(255.0f, fabsf(k[a]), fabsf(base)) * 9.5367431640625e-07f; // 2^-20
        base_in[a] = base - err, base_out[a] = base + err;
        const bool forward = id[a] >= 0.0f;
        for (int h = 0; h < 2; ++h) {
            q_in[a][h] = forward ? qlo_w[a][h] : qhi_w[a][h], q_out[a][h] = forward ? qhi_w[a][h] : qlo_w[a][h];
        }
    }
    uint32_t hit_bytes[2];
    for (int h = 0; h < 2; ++h) {
        uint32_t bm = 0;
        for (int c = 0; c < 4; ++c) {
            const int sh = 8 * c;
            float t_in[3], t_out[3];
            for (int a = 0; a < 3; ++a) {
                t_in[a] = __builtin_fmaf(float((q_in[a][h] >> sh) & 0xffu), k[a], base_in[a]);
                t_out[a] = __builtin_fmaf(float((q_out[a][h] >> sh) & 0xffu), k[a], base_out[a]);
            }
            const float tmin = fmaxf(fmaxf(fmaxf(t_in[0], t_in[1]), t_in[2]), 0.0f), tmax = fminf(fminf(fminf(t_out[0], t_out[1]), t_out[2]), t);
            bm |= (tmin <= tmax) ? (0xffu << sh) : 0u;
        }
        hit_bytes[h] = bm;
    }
    out.child_base = float_as_uint(w1.x), out.tri_base = float_as_uint(w1.y);
    out.leaf[0] = float_as_uint(w1.z) & hit_bytes[0], out.leaf[1] = float_as_uint(w1.w) & hit_bytes[1];
    // byte mask -> bit mask: each byte keeps the bit of its slot, the byte sum gathers them
    const uint32_t hits8 = bvh8_byte_sum(hit_bytes[1] & 0x80402010u, bvh8_byte_sum(hit_bytes[0] & 0x08040201u, 0u));
    const uint32_t imask = exps >> 24;
    out.bits = (bvh8_permute(hits8 & imask, oct_inv) << 8) | imask;
    RT_PROF_T(18)
}

// pending child of a group with the highest priority: returns its node index and removes it from the group
RT_HD uint32_t bvh8_take_child(const uint32_t child_base, uint32_t &bits, const uint32_t oct_inv) {
    const uint32_t p = 31u - uint32_t(__builtin_clz(bits)); // 8 .. 15
    bits ^= 1u << p;
    const uint32_t slot = (p - 8u) ^ oct_inv;
    return child_base + uint32_t(__builtin_popcount(bits & ((1u << slot) - 1u))); // (the mask only reaches the inner-slot byte)
}
// first pending leaf child of a visit as a reference leaf word ((count - 1) << 29 | first record); removes it
RT_HD uint32_t bvh8_take_leaf(const uint32_t tri_base, uint32_t &l0, uint32_t &l1) {
    const bool low = l0 != 0u;
    const uint32_t m = low ? l0 : l1;
    const uint32_t sh = uint32_t(__builtin_ctz(m)) & ~7u;
    const uint32_t meta = (m >> sh) & 0xffu;
    const uint32_t rest = m & ~(0xffu << sh);
    l0 = low ? rest : l0, l1 = low ? l1 : rest;
    return ((meta >> 5) << 29) | (tri_base + (meta & 31u) - 1u);
}

// Ordered walk over an 8-wide BLAS.  `leaf(word)` gets a reference leaf word and returns true to stop (any-hit early out);
// `t_ref` is re-read at every node so that hits found in earlier leaves prune.
//
// Lane state: the CURRENT group (inner children of the node visited last that are still pending), the pending LEAF
// children of that node, and the stack of older groups, whose top lives in registers (as in rt_bvh4.h: a pop hands out the
// registers and starts the LDS read of the next entry, which then hides behind the next node fetch).  Per visit at most
// one entry is pushed (the group the child was taken from, if others remain in it) -- the stack holds one 8-byte entry per
// LEVEL with pending siblings.  Leaves first: all hit leaf children of a node are tested before one of its inner children is
// entered, so a found hit prunes the inner children's subtrees at their first node.
template <class Stack, class LeafFn>
RT_HD bool walk_bvh8(const Bvh8Node *nodes8, const uint32_t root, const f3 ro, const f3 inv_d, const float &t_ref, Stack &st, LeafFn &&leaf,
                     TravCount *cnt = nullptr) {
    const uint32_t base = st.size;
    const uint32_t oct_inv = bvh8_oct_inv(inv_d);
    uint32_t size = base;
    st.write2_at(size, BVH8_SENTINEL, 0u); // second sentinel, so that the read-ahead of a pop never leaves this level
    size += 2;
    uint32_t tos_base = BVH8_SENTINEL, tos_bits = 0u;
    // a virtual group that holds the root in slot 0
    uint32_t cur_base = root, cur_bits = (1u << (8u + oct_inv)) | 1u;
    uint32_t tri_base = 0u, l0 = 0u, l1 = 0u;
    bool early_out = false;
    (void)early_out;
    auto pop = [&]() {
        cur_base = tos_base, cur_bits = tos_bits;
        size -= 2;
        st.read2_at(size, tos_base, tos_bits);
    };
    auto node_step = [&]() {
        const uint32_t node = bvh8_take_child(cur_base, cur_bits, oct_inv);
        if ((cur_bits >> 8) != 0u) { // siblings remain: the group goes onto the stack
            st.write2_at(size, tos_base, tos_bits);
            size += 2;
            tos_base = cur_base, tos_bits = cur_bits;
        }
        if (cnt) {
            ++cnt->nodes4;
            cnt->max_stack = size > cnt->max_stack ? size : cnt->max_stack;
        }
        Bvh8Visit v;
        bvh8_test_node(nodes8, node, ro, inv_d, t_ref, oct_inv, v);
        cur_base = v.child_base, cur_bits = v.bits, tri_base = v.tri_base, l0 = v.leaf[0], l1 = v.leaf[1];
        if ((cur_bits >> 8) == 0u && (l0 | l1) == 0u) {
            pop();
        }
    };
    auto leaf_step = [&]() -> bool {
        if (leaf(bvh8_take_leaf(tri_base, l0, l1))) {
            return true;
        }
        if ((l0 | l1) == 0u && (cur_bits >> 8) == 0u) {
            pop();
        }
        return false;
    };
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_WALK_NO_MAJORITY)
    // majority scheduling of the two phases (rt_bvh4.h): each iteration runs the phase most lanes are waiting for
    for (;;) {
        const bool at_leaf = (l0 | l1) != 0u;
        const bool at_node = !at_leaf && (cur_bits >> 8) != 0u;
        const int n_node = __popcll(__ballot(at_node)), n_leaf = __popcll(__ballot(at_leaf));
        if (n_node == 0 && n_leaf == 0) {
            break;
        }
        if (n_node >= n_leaf) {
            if (at_node) {
                node_step();
            }
        } else if (at_leaf) {
            if (leaf_step()) { // any-hit early out: this lane parks on the sentinel, the others keep going
                cur_base = BVH8_SENTINEL, cur_bits = 0u, l0 = l1 = 0u;
                early_out = true;
            }
        }
    }
    st.size = base;
    return early_out;
#else // one ray at a time (host build of the same walk; tests/hostsim)
    for (;;) {
        if ((l0 | l1) != 0u) {
            if (leaf_step()) {
                st.size = base;
                return true;
            }
        } else if ((cur_bits >> 8) != 0u) {
            node_step();
            RT_PROF_T(19)
        } else {
            break;
        }
    }
    st.size = base;
    return false;
#endif
}

} // namespace rt
