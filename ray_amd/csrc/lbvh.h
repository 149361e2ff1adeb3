// lbvh.h -- a linear BVH builder whose every step is one independent function per element, so that the same source is a
// set of device kernels (lbvh.hip.h: rocPRIM sort + scan between them) and a plain host loop (tests/hostsim, the CPU tests).
//
// Widening row N1 of SURVEY.md section 8f: the reference builds its trees on the host (PreprocessPrims_SAH /
// PreprocessPrims_HLBVH + EmitLBVH, internal/Core.cpp:492-720; Scene::RebuildTLAS_nolock, internal/SceneCPU.cpp:928-1015).
// This is the Karras 2012 construction ("Maximizing parallelism in the construction of BVHs, octrees and k-d trees") over
// 64-bit keys  group << 32 | 30-bit Morton code  -- one sort and one hierarchy pass build the trees of ALL groups (= meshes
// for the bottom level, one group for the top level) at once: a group's primitives share a key prefix no other group has,
// so the radix tree contains, for every group, a node that spans exactly that group; the nodes above those are never emitted.
//
// Output is the reference's own node format (bvh2_node_t: a node holds the boxes and links of its two children; leaf word
// = (count - 1) << 29 | first entry, count >= 2; internal/Core.h:107-115, Constants.inl:24-25), so everything downstream --
// the layout pass, the 4-wide collapse, both traversal kernels, the validator -- consumes it unchanged.  A BVH only culls:
// any correct tree over the same triangle records yields the reference's hits (exact-t ties between triangles aside, as
// between the reference's own tree flavours).
//
// Steps (N = primitives, sorted by key):
//   1 bounds      per primitive: box + centroid; per group: box of the centroids                  (element fn: prim_bounds)
//   2 keys        Morton code of the centroid inside its group's box                              (morton_key)
//   3 sort        (key, primitive) pairs, stable
//   4 hierarchy   internal node i in [0, N-1): its key range [first, last] and split -> two children    (karras_node)
//   5 fit         bottom-up boxes; every internal node is finished by the second child to arrive    (host: reverse loop;
//                 device: one atomic flag per node)
//   6 cut         a node whose range holds <= leaf_max primitives becomes a leaf; a range of one primitive is padded to two
//                 entries (the leaf word cannot say "1")                                             (node_kind)
//   7 emit        scan the emitted nodes / leaf entries into dense arrays, write reference-format nodes   (emit_node)
#pragma once

#include "host_parallel.h"

#include <stdint.h>

#include <algorithm>
#include <vector>

#include "rt_types.h"

namespace rayhip_lbvh {

constexpr uint32_t NONE = 0xffffffffu;

struct Box {
    float lo[3], hi[3];
};
RT_HD Box empty_box() { return Box{{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f}}; }
RT_HD void grow(Box &b, const Box &o) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(b.lo[a], o.lo[a]), b.hi[a] = fmaxf(b.hi[a], o.hi[a]);
    }
}
RT_HD void grow_point(Box &b, const float p[3]) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(b.lo[a], p[a]), b.hi[a] = fmaxf(b.hi[a], p[a]);
    }
}

// ---- 2: keys -----------------------------------------------------------------------------------------------------------
RT_HD uint32_t spread3(uint32_t v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
RT_HD uint64_t morton_key(const uint32_t group, const float c[3], const Box &group_centroids) {
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = group_centroids.hi[a] - group_centroids.lo[a];
        const float u = ext > 0.0f ? (c[a] - group_centroids.lo[a]) / ext : 0.0f;
        q[a] = uint32_t(fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f));
    }
    const uint32_t m = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    return (uint64_t(group) << 32) | uint64_t(m);
}

// ---- 4: hierarchy (Karras 2012, section 4) ---------------------------------------------------------------------------------
// length of the common prefix of keys i and j, ties broken by the index (so all keys are distinct); -1 outside [0, n)
RT_HD int common_prefix(const uint64_t *keys, const int n, const int i, const int j) {
    if (j < 0 || j >= n) {
        return -1;
    }
    const uint64_t x = keys[i] ^ keys[j];
    if (x != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __clzll((long long)x);
#else
        return __builtin_clzll(x);
#endif
    }
    const uint32_t y = uint32_t(i) ^ uint32_t(j);
#if defined(__HIP_DEVICE_COMPILE__)
    return 64 + __clz(int(y));
#else
    return 64 + __builtin_clz(y);
#endif
}
struct RadixNode {
    uint32_t first, last;   // key range the node spans
    uint32_t left, right;   // child: internal node index, or LEAF_FLAG | sorted position
    uint32_t parent;
};
constexpr uint32_t LEAF_FLAG = 0x80000000u;
// internal node i of a tree over n >= 2 keys
RT_HD RadixNode karras_node(const uint64_t *keys, const int n, const int i) {
    const int d = (common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1)) > 0 ? 1 : -1;
    const int min_prefix = common_prefix(keys, n, i, i - d);
    int span = 2;
    while (common_prefix(keys, n, i, i + span * d) > min_prefix) {
        span *= 2;
    }
    int len = 0;
    for (int t = span / 2; t >= 1; t /= 2) {
        if (common_prefix(keys, n, i, i + (len + t) * d) > min_prefix) {
            len += t;
        }
    }
    const int j = i + len * d;
    const int node_prefix = common_prefix(keys, n, i, j);
    int s = 0;
    for (int t = (len + 1) / 2;; t = (t + 1) / 2) {
        if (common_prefix(keys, n, i, i + (s + t) * d) > node_prefix) {
            s += t;
        }
        if (t == 1) {
            break;
        }
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    RadixNode r;
    r.first = uint32_t(i < j ? i : j), r.last = uint32_t(i < j ? j : i);
    r.left = (int(r.first) == gamma) ? (LEAF_FLAG | uint32_t(gamma)) : uint32_t(gamma);
    r.right = (int(r.last) == gamma + 1) ? (LEAF_FLAG | uint32_t(gamma + 1)) : uint32_t(gamma + 1);
    r.parent = NONE;
    return r;
}

// ---- 6: cut ----------------------------------------------------------------------------------------------------------------
// what a radix-tree node (or a single sorted primitive) turns into
enum Kind : uint32_t {
    KIND_ABOVE = 0, // spans more than one group: part of the forest's top, never emitted
    KIND_NODE = 1,  // emitted as a bvh2 node
    KIND_LEAF = 2,  // the top of a leaf: its whole range becomes one leaf word
    KIND_INSIDE = 3 // below a leaf top
};
RT_HD uint32_t group_of(const uint64_t key) { return uint32_t(key >> 32); }
// count of primitives in [first, last] if the range lies inside one group, else 0
RT_HD uint32_t single_group_count(const uint64_t *keys, const uint32_t first, const uint32_t last) {
    return group_of(keys[first]) == group_of(keys[last]) ? last - first + 1 : 0u;
}
// a range is a leaf when it holds at most leaf_max primitives -- except, with roots_are_nodes, a group's root, which then has
// to be a node when the group has two or more primitives (a walk starts at a node)
RT_HD bool spans_a_group(const uint64_t *keys, const uint32_t n, const uint32_t first, const uint32_t last) {
    const bool group_starts = first == 0 || group_of(keys[first - 1]) != group_of(keys[first]);
    const bool group_ends = last + 1 == n || group_of(keys[last + 1]) != group_of(keys[last]);
    return group_starts && group_ends && group_of(keys[first]) == group_of(keys[last]);
}
RT_HD bool range_is_leaf(const uint64_t *keys, const uint32_t n, const uint32_t first, const uint32_t last, const uint32_t leaf_max,
                         const bool roots_are_nodes) {
    const uint32_t count = single_group_count(keys, first, last);
    if (count == 0 || count > leaf_max) {
        return false;
    }
    return !(roots_are_nodes && count >= 2 && spans_a_group(keys, n, first, last));
}

// leaf entries: a leaf of `count` primitives occupies max(count, 2) consecutive output entries (a lone primitive is written twice)
RT_HD uint32_t leaf_word(const uint32_t first_entry, const uint32_t count) { return ((count < 2 ? 2u : count) - 1u) << 29 | first_entry; }

// ---- host driver -------------------------------------------------------------------------------------------------------------
struct Input {
    const Box *prim_box;       // N boxes (bottom level: object space of their mesh; top level: world space)
    const uint32_t *prim_group; // N group ids, < n_groups
    const Box *group_centroids; // per group: box of its primitives' centroids (the Morton grid); null = computed here
    uint32_t n_prims, n_groups;
    uint32_t leaf_max;          // most primitives per leaf, 1 .. 8
    // top level: a leaf is ONE primitive and its word carries the primitive itself -- 1 << 29 | primitive, the reference's
    // TLAS leaf (CoreRef.cpp:1998-1999) -- instead of a range of output entries
    bool leaf_is_primitive;
    // true: every group's link is a NODE (a traversal starts there: mesh trees, the top level); false: a group small enough
    // to be one leaf is linked as a leaf word (leaf refinement: the group replaces a leaf of an existing tree)
    bool roots_are_nodes;
};
struct Output {
    std::vector<rayhip_bvh2_node> nodes; // dense, reference format; links are indices into this array
    std::vector<uint32_t> group_root;    // per group: the link to it -- a node index, or (roots_are_nodes == false) possibly a leaf
                                         // word; NONE for an empty group
    std::vector<uint32_t> entries;       // leaf entries in output order: the primitive each one holds (lone primitives twice)
    Box bounds;                          // of everything
};

inline void write_child(rayhip_bvh2_node &n, const int k, const Box &b, const uint32_t link) {
    if (k == 0) {
        n.ch_data0[0] = b.lo[0], n.ch_data0[1] = b.hi[0], n.ch_data0[2] = b.lo[1], n.ch_data0[3] = b.hi[1];
        n.ch_data2[0] = b.lo[2], n.ch_data2[1] = b.hi[2];
        n.left_child = link;
    } else {
        n.ch_data1[0] = b.lo[0], n.ch_data1[1] = b.hi[0], n.ch_data1[2] = b.lo[1], n.ch_data1[3] = b.hi[1];
        n.ch_data2[2] = b.lo[2], n.ch_data2[3] = b.hi[2];
        n.right_child = link;
    }
}

RT_HD void centroid_of(const Box &b, float c[3]) {
    for (int a = 0; a < 3; ++a) {
        c[a] = 0.5f * (b.lo[a] + b.hi[a]);
    }
}
inline std::vector<Box> centroid_boxes(const Input &in) {
    if (in.group_centroids) {
        return std::vector<Box>(in.group_centroids, in.group_centroids + in.n_groups);
    }
    std::vector<Box> cbox(in.n_groups, empty_box());
    // when the primitives come group after group (the leaf refinement hands them over that way) the groups are independent
    // ranges and the loop is shared out over the host cores; min / max do not depend on the order, so the boxes are the same
    bool grouped = true;
    for (uint32_t p = 1; p < in.n_prims && grouped; ++p) {
        grouped = in.prim_group[p] >= in.prim_group[p - 1];
    }
    if (grouped && in.n_prims > (1u << 16)) {
        rayhip_host::parallel_blocks(in.n_prims, 1 << 16, [&](size_t b, size_t e) {
            // a block takes the groups that START inside it (a group that straddles the block's end is finished by this block)
            while (b < e && b > 0 && in.prim_group[b] == in.prim_group[b - 1]) {
                ++b;
            }
            if (b >= e) {
                return; // (the whole block belongs to a group that started earlier)
            }
            for (size_t p = b; p < in.n_prims && (p < e || in.prim_group[p] == in.prim_group[p - 1]); ++p) {
                float c[3];
                centroid_of(in.prim_box[p], c);
                grow_point(cbox[in.prim_group[p]], c);
            }
        });
        return cbox;
    }
    for (uint32_t p = 0; p < in.n_prims; ++p) {
        float c[3];
        centroid_of(in.prim_box[p], c);
        grow_point(cbox[in.prim_group[p]], c);
    }
    return cbox;
}

// The whole pipeline as plain loops (same element functions the device kernels call).
inline Output build_host(const Input &in) {
    Output out;
    out.group_root.assign(in.n_groups, NONE);
    out.bounds = empty_box();
    const uint32_t n = in.n_prims;
    if (n == 0) {
        return out;
    }
    // 1-2: centroid boxes per group, keys
    std::vector<Box> cbox = centroid_boxes(in);
    std::vector<float> cent(size_t(n) * 3);
    for (uint32_t p = 0; p < n; ++p) {
        centroid_of(in.prim_box[p], &cent[size_t(p) * 3]);
        grow(out.bounds, in.prim_box[p]);
    }
    std::vector<uint64_t> keys(n);
    std::vector<uint32_t> order(n);
    for (uint32_t p = 0; p < n; ++p) {
        keys[p] = morton_key(in.prim_group[p], &cent[size_t(p) * 3], cbox[in.prim_group[p]]);
        order[p] = p;
    }
    // 3: stable sort of (key, primitive)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    std::vector<uint64_t> skeys(n);
    for (uint32_t i = 0; i < n; ++i) {
        skeys[i] = keys[order[i]];
    }
    // 4: hierarchy
    const uint32_t n_int = n - 1;
    std::vector<RadixNode> rn(n_int);
    std::vector<uint32_t> leaf_parent(n, NONE);
    for (uint32_t i = 0; i < n_int; ++i) {
        rn[i] = karras_node(skeys.data(), int(n), int(i));
    }
    for (uint32_t i = 0; i < n_int; ++i) {
        for (const uint32_t c : {rn[i].left, rn[i].right}) {
            if (c & LEAF_FLAG) {
                leaf_parent[c & ~LEAF_FLAG] = i;
            } else {
                rn[c].parent = i;
            }
        }
    }
    // 5: fit (children before parents: process internal nodes by decreasing range length would do; a post-order walk from the
    // root is simplest on the host)
    std::vector<Box> nbox(n_int, empty_box());
    auto child_box = [&](const uint32_t c) -> Box { return (c & LEAF_FLAG) ? in.prim_box[order[c & ~LEAF_FLAG]] : nbox[c]; };
    if (n_int > 0) {
        std::vector<uint32_t> stack = {0};
        std::vector<uint8_t> expanded(n_int, 0);
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            if (!expanded[i]) {
                expanded[i] = 1;
                if (!(rn[i].left & LEAF_FLAG)) {
                    stack.push_back(rn[i].left);
                }
                if (!(rn[i].right & LEAF_FLAG)) {
                    stack.push_back(rn[i].right);
                }
            } else {
                stack.pop_back();
                nbox[i] = child_box(rn[i].left);
                grow(nbox[i], child_box(rn[i].right));
            }
        }
    }
    // 6: kinds
    auto range_of = [&](const uint32_t c, uint32_t &first, uint32_t &last) {
        if (c & LEAF_FLAG) {
            first = last = (c & ~LEAF_FLAG);
        } else {
            first = rn[c].first, last = rn[c].last;
        }
    };
    std::vector<uint32_t> kind(n_int, KIND_ABOVE);
    for (uint32_t i = 0; i < n_int; ++i) {
        const uint32_t count = single_group_count(skeys.data(), rn[i].first, rn[i].last);
        if (count == 0) {
            kind[i] = KIND_ABOVE;
        } else if (!range_is_leaf(skeys.data(), n, rn[i].first, rn[i].last, in.leaf_max, in.roots_are_nodes)) {
            kind[i] = KIND_NODE;
        } else {
            const uint32_t par = rn[i].parent;
            const bool parent_is_leaf = par != NONE && range_is_leaf(skeys.data(), n, rn[par].first, rn[par].last, in.leaf_max, in.roots_are_nodes);
            kind[i] = parent_is_leaf ? KIND_INSIDE : KIND_LEAF;
        }
    }
    // a sorted position starts a leaf when it is the first primitive of a leaf top (an internal KIND_LEAF node, or a lone
    // primitive whose parent is a node / above)
    std::vector<uint32_t> entry_at(n + 1, 0); // output entry of sorted position i
    {
        std::vector<uint8_t> lone(n, 0);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t par = leaf_parent[i];
            const bool parent_is_leaf = par != NONE && range_is_leaf(skeys.data(), n, rn[par].first, rn[par].last, in.leaf_max, in.roots_are_nodes);
            lone[i] = parent_is_leaf ? 0 : 1; // a primitive that is a leaf of its own: written twice
        }
        uint32_t e = 0;
        for (uint32_t i = 0; i < n; ++i) {
            entry_at[i] = e;
            e += lone[i] ? 2u : 1u;
        }
        entry_at[n] = e;
        out.entries.resize(e);
        for (uint32_t i = 0; i < n; ++i) {
            out.entries[entry_at[i]] = order[i];
            if (lone[i]) {
                out.entries[entry_at[i] + 1] = order[i];
            }
        }
    }
    // 7: emit
    std::vector<uint32_t> new_index(n_int, NONE);
    uint32_t n_out = 0;
    for (uint32_t i = 0; i < n_int; ++i) {
        if (kind[i] == KIND_NODE) {
            new_index[i] = n_out++;
        }
    }
    // (a group with a single primitive has no node of its own in the radix tree: it is linked as a leaf, or -- when a node is
    // demanded -- gets one whose second child is an empty box)
    std::vector<uint32_t> single_groups;
    for (uint32_t i = 0; i < n; ++i) {
        if (spans_a_group(skeys.data(), n, i, i)) {
            if (in.roots_are_nodes) {
                single_groups.push_back(i);
            } else {
                out.group_root[group_of(skeys[i])] = in.leaf_is_primitive ? ((1u << 29) | order[i]) : leaf_word(entry_at[i], 1);
            }
        }
    }
    for (uint32_t i = 0; i < n_int; ++i) { // groups that became one leaf
        if (kind[i] == KIND_LEAF && spans_a_group(skeys.data(), n, rn[i].first, rn[i].last)) {
            out.group_root[group_of(skeys[rn[i].first])] = leaf_word(entry_at[rn[i].first], rn[i].last - rn[i].first + 1);
        }
    }
    out.nodes.assign(size_t(n_out) + single_groups.size(), rayhip_bvh2_node{});
    auto link_of = [&](const uint32_t c, Box &b) -> uint32_t {
        uint32_t first, last;
        range_of(c, first, last);
        b = child_box(c);
        if (!(c & LEAF_FLAG) && kind[c] == KIND_NODE) {
            return new_index[c];
        }
        if (in.leaf_is_primitive) {
            return (1u << 29) | order[first];
        }
        return leaf_word(entry_at[first], last - first + 1);
    };
    for (uint32_t i = 0; i < n_int; ++i) {
        if (kind[i] != KIND_NODE) {
            continue;
        }
        rayhip_bvh2_node &o = out.nodes[new_index[i]];
        Box b;
        const uint32_t l = link_of(rn[i].left, b);
        write_child(o, 0, b, l);
        const uint32_t r = link_of(rn[i].right, b);
        write_child(o, 1, b, r);
        if (spans_a_group(skeys.data(), n, rn[i].first, rn[i].last)) {
            out.group_root[group_of(skeys[rn[i].first])] = new_index[i];
        }
    }
    for (size_t k = 0; k < single_groups.size(); ++k) {
        const uint32_t i = single_groups[k];
        rayhip_bvh2_node &o = out.nodes[n_out + k];
        const Box b = in.prim_box[order[i]];
        const uint32_t w = in.leaf_is_primitive ? ((1u << 29) | order[i]) : leaf_word(entry_at[i], 1);
        write_child(o, 0, b, w);
        if (in.leaf_is_primitive) {
            // top level: the second child is a point at infinity, which no ray reaches (an inverted box would not do: the slab
            // test orders each axis' two planes itself)
            const float far_ = 3.402823466e+38f;
            write_child(o, 1, Box{{far_, far_, far_}, {far_, far_, far_}}, w);
        } else {
            write_child(o, 1, b, w); // a one-triangle mesh: the triangle on both sides (tested twice, found once)
        }
        out.group_root[group_of(skeys[i])] = uint32_t(n_out + k);
    }
    return out;
}

} // namespace rayhip_lbvh
