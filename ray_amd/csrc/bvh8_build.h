// bvh8_build.h -- host-side collapse of the BVH2 BLAS trees into the 8-wide quantised form of rt_bvh8.h, together with the
// triangle order that form needs.  Runs once per scene upload (rayhip.hip) and in the host build of the kernels
// (tests/hostsim).
//
// The 8-wide node addresses its leaf children as offsets from ONE base into tris[], so the triangle records of all leaf
// children of a node must be consecutive.  The builder therefore also decides the order of tris[] / tri_indices[] (both
// are unobservable: results name triangles through tri_indices[]) and re-bases the leaf words of the BVH2 it was given, so
// that the BVH2 walk (instrumented kernels), the 4-wide collapse built afterwards and the 8-wide nodes all describe the
// same arrays.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "bvh4_build.h"
#include "rt_bvh8.h"

namespace rayhip_bvh8 {

using rayhip_bvh4::Box;
using rayhip_bvh4::half_area;
using rayhip_bvh4::is_leaf;
using rayhip_bvh4::real_plane;

struct Slot {
    Box box;
    uint32_t ref;             // BVH2 child word: inner node index or leaf word
    uint32_t src_node, which; // where that word is stored (node, 0 = left_child / 1 = right_child): leaf words get re-based
};

inline void children_of(const rayhip_bvh2_node &n, const uint32_t index, Slot out[2]) {
    rayhip_bvh4::Slot two[2];
    rayhip_bvh4::children_of(n, two);
    for (int k = 0; k < 2; ++k) {
        out[k].box = two[k].box, out[k].ref = two[k].ref, out[k].src_node = index, out[k].which = uint32_t(k);
    }
}

inline uint32_t leaf_count(const uint32_t w) { return ((w & rt::BVH2_PRIM_COUNT_BITS) >> 29) + 1u; }

// Quantise the boxes of the occupied slots (slot_of[c] = slot of child c) onto the node grid; false if a box cannot be
// represented conservatively (non-finite coordinates).  Same construction and the same containment check in real
// arithmetic as rayhip_bvh4::quantise.
inline bool quantise(const Box *ch, const int *slot_of, const int n, rt::Bvh8Node &out) {
    float lo[3], hi[3], scale[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = ch[0].lo[a], hi[a] = ch[0].hi[a];
        for (int c = 1; c < n; ++c) {
            lo[a] = std::fmin(lo[a], ch[c].lo[a]);
            hi[a] = std::fmax(hi[a], ch[c].hi[a]);
        }
        if (!std::isfinite(lo[a]) || !std::isfinite(hi[a]) || hi[a] < lo[a]) {
            return false;
        }
    }
    uint32_t exps = 0;
    for (int a = 0; a < 3; ++a) {
        out.org[a] = lo[a];
        int e = 1;
        const float need = (hi[a] - lo[a]) / 255.0f;
        if (need > 0.0f) {
            int ex;
            std::frexp(need, &ex);
            e = ex + 127;
            if (e < 1) {
                e = 1;
            }
        }
        while (e < 254 && real_plane(255u, rt::uint_as_float(uint32_t(e) << 23), lo[a]) < double(hi[a])) {
            ++e;
        }
        if (e >= 254) {
            return false;
        }
        scale[a] = rt::uint_as_float(uint32_t(e) << 23);
        exps |= uint32_t(e) << (8 * a);
    }
    out.exps_imask = (out.exps_imask & 0xff000000u) | exps;
    for (int a = 0; a < 3; ++a) { // empty slots: inverted box
        out.qlo[a][0] = out.qlo[a][1] = 0xffffffffu;
        out.qhi[a][0] = out.qhi[a][1] = 0u;
    }
    for (int c = 0; c < n; ++c) {
        const int s = slot_of[c], h = s >> 2, sh = 8 * (s & 3);
        for (int a = 0; a < 3; ++a) {
            const float flo = std::floor((ch[c].lo[a] - lo[a]) / scale[a]);
            const float fhi = std::ceil((ch[c].hi[a] - lo[a]) / scale[a]);
            int qlo = int(std::fmin(std::fmax(flo, 0.0f), 255.0f)), qhi = int(std::fmin(std::fmax(fhi, 0.0f), 255.0f));
            while (qlo > 0 && real_plane(uint32_t(qlo), scale[a], lo[a]) > double(ch[c].lo[a])) {
                --qlo;
            }
            while (qhi < 255 && real_plane(uint32_t(qhi), scale[a], lo[a]) < double(ch[c].hi[a])) {
                ++qhi;
            }
            if (real_plane(uint32_t(qlo), scale[a], lo[a]) > double(ch[c].lo[a]) ||
                real_plane(uint32_t(qhi), scale[a], lo[a]) < double(ch[c].hi[a])) {
                return false;
            }
            out.qlo[a][h] = (out.qlo[a][h] & ~(0xffu << sh)) | (uint32_t(qlo) << sh);
            out.qhi[a][h] = (out.qhi[a][h] & ~(0xffu << sh)) | (uint32_t(qhi) << sh);
        }
    }
    return true;
}

// Octant slots (Ylitie et al. 2017, section 4.2): child c goes to the slot whose corner direction (+-1, +-1, +-1) agrees
// best with the offset of the child's centre from the node's centre; greedy over the (child, slot) pairs.
inline void assign_slots(const Box *ch, const int n, int *slot_of) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = ch[0].lo[a], hi[a] = ch[0].hi[a];
        for (int c = 1; c < n; ++c) {
            lo[a] = std::fmin(lo[a], ch[c].lo[a]), hi[a] = std::fmax(hi[a], ch[c].hi[a]);
        }
    }
    float cost[8][8];
    for (int c = 0; c < n; ++c) {
        for (int s = 0; s < 8; ++s) {
            float v = 0.0f;
            for (int a = 0; a < 3; ++a) {
                const float off = 0.5f * (ch[c].lo[a] + ch[c].hi[a]) - 0.5f * (lo[a] + hi[a]);
                v += ((s >> a) & 1) ? off : -off;
            }
            cost[c][s] = v;
        }
    }
    bool child_done[8] = {}, slot_done[8] = {};
    for (int k = 0; k < n; ++k) {
        int bc = -1, bs = -1;
        for (int c = 0; c < n; ++c) {
            for (int s = 0; s < 8 && !child_done[c]; ++s) {
                if (!slot_done[s] && (bc < 0 || cost[c][s] > cost[bc][bs])) {
                    bc = c, bs = s;
                }
            }
        }
        slot_of[bc] = bs;
        child_done[bc] = slot_done[bs] = true;
    }
}

// ---- which BVH2 subtrees become the children of an 8-wide node -------------------------------------------------------------
// Greedy opening of the largest child (what bvh4_build.h does) leaves the bottom of an 8-wide tree half empty: measured on the
// Sponza-class scene 4.25 children per node.  Instead the collapse minimises the surface-area cost of the WIDE tree by
// dynamic programming over the BVH2 (Ylitie, Karras, Laine 2017, section 3):
//   C(n, i)  = cheapest way to represent the subtree of n by at most i roots (wide nodes or leaves), i = 1 .. 7
//   C(n, 1)  = min( A(n) * P(n) * c_prim   if P(n) <= p_max        -- all triangles below n as ONE leaf child
//                 , A(n) * c_node + D(n, 8) )                      -- n becomes a wide node over at most 8 roots
//   C(n, i)  = min( D(n, i), C(n, i - 1) ),   D(n, j) = min over 0 < k < j of C(left, k) + C(right, j - k)
// A = half surface area of the subtree's box, P = its triangle count.  A leaf child made of several BVH2 leaves is only a
// different GROUPING of the same triangle records (their ranges are laid out next to each other); what a grouping changes
// is what the leaf refinement of the upload already changes (rt_bvh4.h, head comment).
struct CostModel {
    float c_node = 1.0f, c_prim = 0.4f; // a node visit (80-byte fetch, 8 slab tests) vs one triangle test (48-byte fetch)
    uint32_t p_max = 3;                 // most triangles in a leaf child that groups several BVH2 leaves
};

struct Dp {
    struct Entry {
        float c[7];       // C(n, 1 .. 7)
        uint8_t split[9]; // D(n, j), j = 2 .. 8: roots given to the left child
        uint8_t from[7];  // C(n, i) = D(n, from[i - 1]) when from >= 2; from == 1: the decision of C(n, 1)
        uint8_t leaf;     // C(n, 1): all triangles below n as one leaf child
    };
    std::vector<Entry> e;        // per BVH2 node
    std::vector<float> area;     // per BVH2 node: half area of its box
    std::vector<uint32_t> prims; // per BVH2 node: triangles below
};

// cost table of one child word (leaf words cost the same for every budget)
inline void child_costs(const Dp &dp, const uint32_t word, const float area, const CostModel &cm, float c[7], uint32_t &prims) {
    if (is_leaf(word)) {
        prims = leaf_count(word);
        for (int i = 0; i < 7; ++i) {
            c[i] = area * float(prims) * cm.c_prim;
        }
    } else {
        prims = dp.prims[word];
        memcpy(c, dp.e[word].c, sizeof(float) * 7);
    }
}

// fills dp for the BLAS rooted at `root`; returns false if the links do not form a tree inside [0, n_nodes).  `max_leaf`: the
// largest leaf of the tree.
inline bool dp_blas(const rayhip_bvh2_node *nodes2, const uint32_t n_nodes, const uint32_t root, const CostModel &cm, Dp &dp,
                    std::vector<uint32_t> &order, uint32_t &max_leaf) {
    order.clear();
    max_leaf = 0;
    { // areas top-down (a node's box lives in its parent), DFS order
        Slot two[2];
        children_of(nodes2[root], root, two);
        Box rb = two[0].box;
        for (int a = 0; a < 3; ++a) {
            rb.lo[a] = std::fmin(rb.lo[a], two[1].box.lo[a]), rb.hi[a] = std::fmax(rb.hi[a], two[1].box.hi[a]);
        }
        dp.area[root] = half_area(rb);
        std::vector<uint32_t> stack = {root};
        while (!stack.empty()) {
            const uint32_t n = stack.back();
            stack.pop_back();
            if (order.size() > n_nodes) {
                return false;
            }
            order.push_back(n);
            children_of(nodes2[n], n, two);
            for (int k = 0; k < 2; ++k) {
                if (is_leaf(two[k].ref)) {
                    max_leaf = std::max(max_leaf, leaf_count(two[k].ref));
                } else {
                    if (two[k].ref >= n_nodes) {
                        return false;
                    }
                    dp.area[two[k].ref] = half_area(two[k].box);
                    stack.push_back(two[k].ref);
                }
            }
        }
    }
    for (size_t oi = order.size(); oi-- > 0;) { // children before parents
        const uint32_t n = order[oi];
        Slot two[2];
        children_of(nodes2[n], n, two);
        float cl[7], cr[7];
        uint32_t pl, pr;
        child_costs(dp, two[0].ref, half_area(two[0].box), cm, cl, pl);
        child_costs(dp, two[1].ref, half_area(two[1].box), cm, cr, pr);
        Dp::Entry &e = dp.e[n];
        dp.prims[n] = pl + pr;
        float d[9];
        for (int j = 2; j <= 8; ++j) {
            float best = 3.0e38f;
            int bk = 1;
            for (int k = 1; k < j; ++k) {
                if (k > 7 || j - k > 7) {
                    continue;
                }
                const float v = cl[k - 1] + cr[j - k - 1];
                if (v < best) {
                    best = v, bk = k;
                }
            }
            d[j] = best, e.split[j] = uint8_t(bk);
        }
        const float c_leaf = dp.prims[n] <= cm.p_max ? dp.area[n] * float(dp.prims[n]) * cm.c_prim : 3.0e38f;
        const float c_int = dp.area[n] * cm.c_node + d[8];
        e.leaf = c_leaf <= c_int;
        e.c[0] = e.leaf ? c_leaf : c_int;
        e.from[0] = 1;
        for (int i = 2; i <= 7; ++i) {
            if (d[i] < e.c[i - 2]) {
                e.c[i - 1] = d[i], e.from[i - 1] = uint8_t(i);
            } else {
                e.c[i - 1] = e.c[i - 2], e.from[i - 1] = e.from[i - 2];
            }
        }
    }
    return true;
}

struct LeafRef {
    uint32_t src_node, which, word; // where a BVH2 leaf word is stored, and the word
};
struct Child {
    Box box;
    bool inner;
    uint32_t node;                 // inner: the BVH2 node that becomes a wide node
    uint32_t first_leaf, n_leaves; // leaf child: its BVH2 leaves, a range of the scratch list
    uint32_t count;                // ... and their triangles
};

// all BVH2 leaves below `s` (left first) as ONE leaf child
inline void group_leaves(const rayhip_bvh2_node *nodes2, const Slot &s, std::vector<LeafRef> &leaves, Child &out) {
    out.box = s.box, out.inner = false, out.node = 0, out.first_leaf = uint32_t(leaves.size()), out.count = 0;
    std::vector<Slot> stack = {s};
    while (!stack.empty()) {
        const Slot cur = stack.back();
        stack.pop_back();
        if (is_leaf(cur.ref)) {
            leaves.push_back(LeafRef{cur.src_node, cur.which, cur.ref});
            out.count += leaf_count(cur.ref);
        } else {
            Slot two[2];
            children_of(nodes2[cur.ref], cur.ref, two);
            stack.push_back(two[1]), stack.push_back(two[0]);
        }
    }
    out.n_leaves = uint32_t(leaves.size()) - out.first_leaf;
}

// the roots that represent subtree `s` under a budget of `budget` (1 .. 7), by the decisions stored in dp
inline void gather(const rayhip_bvh2_node *nodes2, const Dp &dp, const Slot &s, const int budget, std::vector<LeafRef> &leaves, Child *out, int &n) {
    if (is_leaf(s.ref)) {
        group_leaves(nodes2, s, leaves, out[n++]);
        return;
    }
    const Dp::Entry &e = dp.e[s.ref];
    const int from = e.from[budget - 1];
    if (from == 1) {
        if (e.leaf) {
            group_leaves(nodes2, s, leaves, out[n++]);
        } else {
            Child &c = out[n++];
            c.box = s.box, c.inner = true, c.node = s.ref, c.first_leaf = c.n_leaves = c.count = 0;
        }
        return;
    }
    Slot two[2];
    children_of(nodes2[s.ref], s.ref, two);
    gather(nodes2, dp, two[0], e.split[from], leaves, out, n);
    gather(nodes2, dp, two[1], from - e.split[from], leaves, out, n);
}

struct Result {
    std::vector<rt::Bvh8Node> nodes;
    std::vector<uint32_t> blas_root8; // per mesh instance: root of its 8-wide BLAS (0xffffffff: not referenced)
    std::vector<rayhip_tri_accel> tris;
    std::vector<uint32_t> tri_indices;
    bool ok = false;
    const char *why_not = "";
};

// nodes2 / mis as they will be uploaded (after bvh_layout); leaf words of nodes2 are re-based IN PLACE onto out.tris when the
// build succeeds (and only then).  Only instances referenced by TLAS leaves are followed.
inline Result build(rayhip_bvh2_node *nodes2, const uint32_t n_nodes, const rayhip_mesh_instance *mis, const uint32_t n_mis, const uint32_t tlas_root,
                    const rayhip_tri_accel *tris, const uint32_t *tri_indices, const uint32_t n_tris, const CostModel cm = CostModel()) {
    Result out;
    out.blas_root8.assign(n_mis, 0xffffffffu);
    if (tlas_root == 0xffffffffu || tlas_root >= n_nodes) {
        out.why_not = "no top level";
        return out;
    }
    std::vector<uint32_t> inst;
    {
        std::vector<uint32_t> stack = {tlas_root};
        size_t visited = 0;
        while (!stack.empty()) {
            const uint32_t n = stack.back();
            stack.pop_back();
            if (n >= n_nodes || ++visited > n_nodes) {
                out.why_not = "top level is not a tree";
                return out;
            }
            const uint32_t ch[2] = {nodes2[n].left_child, nodes2[n].right_child};
            for (int k = 0; k < 2; ++k) {
                if (is_leaf(ch[k])) {
                    const uint32_t mi = ch[k] & rt::BVH2_PRIM_INDEX_BITS;
                    if (mi >= n_mis) {
                        out.why_not = "instance index out of range";
                        return out;
                    }
                    inst.push_back(mi);
                } else {
                    stack.push_back(ch[k]);
                }
            }
        }
    }
    struct Patch {
        uint32_t node, which, word;
    };
    std::vector<Patch> patches; // applied at the very end: the input stays untouched on failure
    std::vector<uint8_t> tri_used(n_tris, 0);
    std::vector<uint32_t> root8_of_bvh2(n_nodes, 0xffffffffu);
    Dp dp;
    dp.e.resize(n_nodes), dp.area.assign(n_nodes, 0.0f), dp.prims.assign(n_nodes, 0);
    std::vector<uint32_t> order;
    std::vector<LeafRef> leaves;
    struct Work {
        uint32_t bvh2_node, out_index;
    };
    out.tris.reserve(n_tris), out.tri_indices.reserve(n_tris);
    for (const uint32_t mi : inst) {
        const uint32_t root2 = mis[mi].node_index;
        if (root2 >= n_nodes) {
            out.why_not = "BLAS root out of range";
            return out;
        }
        if (root8_of_bvh2[root2] != 0xffffffffu) {
            out.blas_root8[mi] = root8_of_bvh2[root2];
            continue;
        }
        uint32_t max_leaf = 0;
        if (!dp_blas(nodes2, n_nodes, root2, cm, dp, order, max_leaf)) {
            out.why_not = "a BLAS is not a tree";
            return out;
        }
        // leaf offsets inside a node are 5 bits: with leaves of at most three triangles eight children always fit and the
        // optimal collapse is used; fatter leaves (refinement switched off) take the greedy one, which checks as it goes
        const bool optimal = max_leaf <= 3 && cm.p_max <= 3;
        const uint32_t root8 = uint32_t(out.nodes.size());
        out.nodes.emplace_back();
        std::vector<Work> stack = {Work{root2, root8}};
        size_t made = 0;
        while (!stack.empty()) {
            const Work w = stack.back();
            stack.pop_back();
            if (++made > size_t(n_nodes) + 1) {
                out.why_not = "a BLAS is not a tree";
                return out;
            }
            Child ch[8];
            int n = 0;
            leaves.clear();
            Slot two[2];
            children_of(nodes2[w.bvh2_node], w.bvh2_node, two);
            if (optimal) {
                const int k = dp.e[w.bvh2_node].split[8];
                gather(nodes2, dp, two[0], k, leaves, ch, n);
                gather(nodes2, dp, two[1], 8 - k, leaves, ch, n);
            } else {
                Slot sl[8] = {two[0], two[1]};
                int ns = 2;
                bool closed[8] = {}; // inner children that must stay closed (opening them would overflow the leaf offsets)
                auto leaf_tris = [&](const Slot *s, int cnt) {
                    uint32_t total = 0;
                    for (int c = 0; c < cnt; ++c) {
                        total += is_leaf(s[c].ref) ? leaf_count(s[c].ref) : 0u;
                    }
                    return total;
                };
                while (ns < 8) {
                    int best = -1;
                    float best_area = -1.0f;
                    for (int c = 0; c < ns; ++c) {
                        if (!is_leaf(sl[c].ref) && !closed[c] && half_area(sl[c].box) > best_area) {
                            best_area = half_area(sl[c].box), best = c;
                        }
                    }
                    if (best < 0) {
                        break;
                    }
                    Slot open[2];
                    children_of(nodes2[sl[best].ref], sl[best].ref, open);
                    if (leaf_tris(sl, ns) + leaf_tris(open, 2) > rt::BVH8_MAX_LEAF_OFFSET + 1u) {
                        closed[best] = true;
                        continue;
                    }
                    sl[best] = open[0], closed[best] = false;
                    sl[ns] = open[1], closed[ns] = false;
                    ++ns;
                }
                for (int c = 0; c < ns; ++c) {
                    if (is_leaf(sl[c].ref)) {
                        group_leaves(nodes2, sl[c], leaves, ch[n++]);
                    } else {
                        Child &cc = ch[n++];
                        cc.box = sl[c].box, cc.inner = true, cc.node = sl[c].ref, cc.first_leaf = cc.n_leaves = cc.count = 0;
                    }
                }
            }
            if (n < 1 || n > 8) {
                out.why_not = "internal: a wide node with no or too many children";
                return out;
            }
            Box boxes[8];
            for (int c = 0; c < n; ++c) {
                boxes[c] = ch[c].box;
            }
            int slot_of[8];
            assign_slots(boxes, n, slot_of);
            rt::Bvh8Node node;
            memset(&node, 0, sizeof(node));
            if (!quantise(boxes, slot_of, n, node)) {
                out.why_not = "a child box cannot be quantised (non-finite?)";
                return out;
            }
            int child_in_slot[8];
            for (int s = 0; s < 8; ++s) {
                child_in_slot[s] = -1;
            }
            for (int c = 0; c < n; ++c) {
                child_in_slot[slot_of[c]] = c;
            }
            const uint32_t first_child = uint32_t(out.nodes.size());
            uint32_t n_inner = 0, imask = 0;
            node.tri_base = uint32_t(out.tris.size());
            uint32_t inner_ref[8], inner_index[8];
            for (int s = 0; s < 8; ++s) {
                const int c = child_in_slot[s];
                if (c < 0) {
                    continue;
                }
                if (!ch[c].inner) {
                    const uint32_t offset = uint32_t(out.tris.size()) - node.tri_base;
                    if (offset > rt::BVH8_MAX_LEAF_OFFSET || ch[c].count < 1 || ch[c].count > 8) {
                        out.why_not = "internal: leaf child does not fit the node";
                        return out;
                    }
                    for (uint32_t l = 0; l < ch[c].n_leaves; ++l) {
                        const LeafRef &lr = leaves[ch[c].first_leaf + l];
                        const uint32_t start = lr.word & rt::BVH2_PRIM_INDEX_BITS, count = leaf_count(lr.word);
                        if (uint64_t(start) + count > n_tris) {
                            out.why_not = "leaf range outside the triangle array";
                            return out;
                        }
                        if (out.tris.size() + count > size_t(rt::BVH2_PRIM_INDEX_BITS)) {
                            out.why_not = "triangle array too large";
                            return out;
                        }
                        patches.push_back(Patch{lr.src_node, lr.which, (lr.word & rt::BVH2_PRIM_COUNT_BITS) | uint32_t(out.tris.size())});
                        for (uint32_t k = 0; k < count; ++k) {
                            if (tri_used[start + k]) {
                                out.why_not = "overlapping leaf ranges";
                                return out;
                            }
                            tri_used[start + k] = 1;
                            out.tris.push_back(tris[start + k]);
                            out.tri_indices.push_back(tri_indices[start + k]);
                        }
                    }
                    node.meta[s >> 2] |= (((ch[c].count - 1u) << 5) | (offset + 1u)) << (8 * (s & 3));
                } else {
                    imask |= 1u << s;
                    inner_ref[n_inner] = ch[c].node, inner_index[n_inner] = first_child + n_inner;
                    ++n_inner;
                }
            }
            node.child_base = first_child;
            node.exps_imask |= imask << 24;
            if (uint64_t(first_child) + n_inner >= 0xfffffff0ull) {
                out.why_not = "too many nodes";
                return out;
            }
            out.nodes.resize(size_t(first_child) + n_inner);
            out.nodes[w.out_index] = node;
            for (int k = int(n_inner) - 1; k >= 0; --k) { // the first inner child is processed next (depth-first layout)
                stack.push_back(Work{inner_ref[k], inner_index[k]});
            }
        }
        root8_of_bvh2[root2] = root8;
        out.blas_root8[mi] = root8;
    }
    for (const Patch &p : patches) {
        (p.which ? nodes2[p.node].right_child : nodes2[p.node].left_child) = p.word;
    }
    out.ok = true;
    return out;
}

} // namespace rayhip_bvh8
