// bvh_layout.h -- host-side re-layout of the reference's BVH2 pool and triangle array for HBM / cache-line locality.
//
// The C ABI hands over the scene in the reference's own layouts (internal/Core.h: bvh2_node_t, tri_accel_t) and in
// the reference builder's ORDER, which is poor for a GPU: measured on the Sponza-class scene only 41 % of sibling
// nodes are neighbours and a child sits on average 32 000 nodes away from its parent, so nearly every node visit is
// its own 128-byte line.  Node order and triangle order are not observable (results reference triangles through
// tri_indices[], nodes through child links), so librayhip is free to permute both when it copies the scene to HBM:
//
//   * nodes: every tree (the TLAS and each BLAS) is laid out depth-first, and the two children of a node are always
//     placed together in one aligned 128-byte line (2 x 64 B): the far child pushed on the stack is already in
//     L1/L2 when it is popped, and a parent's line is followed by its left subtree;
//   * triangles: leaf ranges are emitted in the same depth-first order, each placed so that it touches the minimum
//     number of 128-byte lines (48-byte entries); tri_indices[] is permuted alike, leaf words are re-based.
//
// Traversal order per ray, hits, and the visit counters are unchanged: the tree is the same tree.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rayhip.h"

namespace rayhip_layout {

// rayhip_bvh2_node / rayhip_tri_accel are declared 16-byte aligned (the kernels fetch them with 128-bit loads), but the
// C ABI takes the reference's own arrays, and its SparseStorage only guarantees alignof(T) = 4.  Host code that reads
// the structs by value (the layout pass, the 4-wide collapse) must not see a misaligned pointer: this view re-homes
// the two arrays when needed.
struct AlignedDesc {
    rayhip_scene_desc d;
    std::vector<rayhip_bvh2_node> nodes;
    std::vector<rayhip_tri_accel> tris;
    explicit AlignedDesc(const rayhip_scene_desc &in) : d(in) {
        if ((reinterpret_cast<uintptr_t>(in.nodes) & 15u) != 0 && in.nodes_count) {
            nodes.resize(in.nodes_count);
            memcpy(static_cast<void *>(nodes.data()), static_cast<const void *>(in.nodes), size_t(in.nodes_count) * sizeof(rayhip_bvh2_node));
            d.nodes = nodes.data();
        }
        if ((reinterpret_cast<uintptr_t>(in.tris) & 15u) != 0 && in.tris_count) {
            tris.resize(in.tris_count);
            memcpy(static_cast<void *>(tris.data()), static_cast<const void *>(in.tris), size_t(in.tris_count) * sizeof(rayhip_tri_accel));
            d.tris = tris.data();
        }
    }
};

constexpr uint32_t PRIM_COUNT_BITS = 7u << 29;
constexpr uint32_t PRIM_INDEX_BITS = ~PRIM_COUNT_BITS;
constexpr uint32_t NONE = 0xffffffffu;

inline bool is_leaf(uint32_t w) { return (w & PRIM_COUNT_BITS) != 0; }

struct Result {
    std::vector<rayhip_bvh2_node> nodes;
    std::vector<rayhip_tri_accel> tris;
    std::vector<uint32_t> tri_indices;
    std::vector<rayhip_mesh_instance> mesh_instances;
    uint32_t tlas_root = NONE;
    bool applied = false;
    const char *why_not = ""; // reason when applied == false
};

// lines of 128 B touched by `count` 48-byte entries starting at entry `first`
inline uint32_t lines_touched(uint64_t first, uint32_t count) {
    const uint64_t b0 = first * 48u, b1 = (first + count) * 48u - 1u;
    return uint32_t(b1 / 128u - b0 / 128u + 1u);
}

// Returns applied == false (and leaves everything else empty) when the input does not look like what the reference
// produces (indices out of range, overlapping leaf ranges, cycles): the caller then uploads the arrays as they are.
inline Result optimize(const rayhip_scene_desc &d) {
    Result out;
    const uint32_t n_nodes = d.nodes_count, n_tris = d.tris_count;
    if (n_nodes == 0 || d.tlas_root == NONE || d.tlas_root >= n_nodes || d.tri_indices_count != n_tris) {
        out.why_not = "empty scene or tri_indices / tris size mismatch";
        return out;
    }
    std::vector<uint32_t> remap(n_nodes, NONE);
    std::vector<rayhip_bvh2_node> nodes;
    nodes.reserve(n_nodes + n_nodes / 8);
    struct LeafRef {
        uint32_t node;  // new node index holding the leaf word
        uint32_t which; // 0 = left_child, 1 = right_child
    };
    std::vector<LeafRef> blas_leaves;    // in depth-first order
    std::vector<uint32_t> tlas_instances; // mesh instances referenced by TLAS leaves (the instance array is a sparse
                                          // pool: unreferenced slots hold garbage)
    rayhip_bvh2_node pad_node;
    memset(&pad_node, 0, sizeof(pad_node));

    bool ok = true;
    auto place_tree = [&](const uint32_t root, const bool is_blas) {
        if (root >= n_nodes) {
            ok = false;
            return;
        }
        if (remap[root] != NONE) {
            return; // BLAS shared by several instances
        }
        if (nodes.size() & 1u) {
            nodes.push_back(pad_node); // roots start a line
        }
        remap[root] = uint32_t(nodes.size());
        nodes.push_back(d.nodes[root]);
        std::vector<uint32_t> stack = {root};
        while (!stack.empty() && ok) {
            const uint32_t n = stack.back();
            stack.pop_back();
            const rayhip_bvh2_node &nd = d.nodes[n];
            const uint32_t ch[2] = {nd.left_child, nd.right_child};
            const bool inner[2] = {!is_leaf(ch[0]), !is_leaf(ch[1])};
            for (int k = 0; k < 2; ++k) {
                if (inner[k] && (ch[k] >= n_nodes || remap[ch[k]] != NONE)) {
                    ok = false; // out of range, or a node with two parents / a cycle
                    return;
                }
            }
            if (inner[0] && inner[1] && (nodes.size() & 1u)) {
                nodes.push_back(pad_node); // sibling pair = one aligned 128-byte line
            }
            for (int k = 0; k < 2; ++k) {
                if (inner[k]) {
                    remap[ch[k]] = uint32_t(nodes.size());
                    nodes.push_back(d.nodes[ch[k]]);
                } else if (is_blas) {
                    blas_leaves.push_back(LeafRef{remap[n], uint32_t(k)});
                } else {
                    tlas_instances.push_back(ch[k] & PRIM_INDEX_BITS);
                }
            }
            // left subtree is laid out first
            if (inner[1]) {
                stack.push_back(ch[1]);
            }
            if (inner[0]) {
                stack.push_back(ch[0]);
            }
        }
    };

    place_tree(d.tlas_root, false);
    for (size_t i = 0; i < tlas_instances.size() && ok; ++i) {
        if (tlas_instances[i] >= d.mesh_instances_count) {
            ok = false;
            break;
        }
        place_tree(d.mesh_instances[tlas_instances[i]].node_index, true);
    }
    if (!ok) {
        out.why_not = "node links out of range or not a tree";
        return out;
    }
    // child links
    for (size_t i = 0; i < nodes.size(); ++i) {
        if (!is_leaf(nodes[i].left_child) && nodes[i].left_child < n_nodes && remap[nodes[i].left_child] != NONE) {
            nodes[i].left_child = remap[nodes[i].left_child];
        }
        if (!is_leaf(nodes[i].right_child) && nodes[i].right_child < n_nodes && remap[nodes[i].right_child] != NONE) {
            nodes[i].right_child = remap[nodes[i].right_child];
        }
    }
    // (pad nodes have child words 0 == "inner node 0": never reached, no link points at them)

    // triangles: leaf ranges in depth-first order; ranges must be disjoint (a shared range is emitted once)
    std::vector<uint32_t> new_start(n_tris, NONE); // by old range start
    std::vector<uint8_t> used(n_tris, 0);
    std::vector<rayhip_tri_accel> tris;
    std::vector<uint32_t> tri_indices;
    tris.reserve(n_tris);
    tri_indices.reserve(n_tris);
    for (const LeafRef &lr : blas_leaves) {
        uint32_t &word = lr.which ? nodes[lr.node].right_child : nodes[lr.node].left_child;
        const uint32_t start = word & PRIM_INDEX_BITS, count = ((word & PRIM_COUNT_BITS) >> 29) + 1u;
        if (uint64_t(start) + count > n_tris) {
            out.why_not = "leaf range outside the triangle array";
            return out;
        }
        if (new_start[start] == NONE) {
            for (uint32_t k = 0; k < count; ++k) {
                if (used[start + k]) {
                    out.why_not = "overlapping leaf ranges";
                    return out;
                }
                used[start + k] = 1;
            }
            // smallest padding that makes the range touch the minimum number of lines
            const uint32_t best = (count * 48u + 127u) / 128u;
            while (lines_touched(tris.size(), count) > best) {
                tris.push_back(d.tris[start]); // harmless filler, never referenced
                tri_indices.push_back(d.tri_indices[start]);
            }
            new_start[start] = uint32_t(tris.size());
            for (uint32_t k = 0; k < count; ++k) {
                tris.push_back(d.tris[start + k]);
                tri_indices.push_back(d.tri_indices[start + k]);
            }
        } else if (!used[start + count - 1]) {
            out.why_not = "two leaf ranges with the same start and different lengths";
            return out;
        }
        if (new_start[start] > PRIM_INDEX_BITS) {
            out.why_not = "triangle array too large";
            return out;
        }
        word = (word & PRIM_COUNT_BITS) | new_start[start];
    }

    out.mesh_instances.assign(d.mesh_instances, d.mesh_instances + d.mesh_instances_count);
    for (const uint32_t i : tlas_instances) {
        const uint32_t old_root = d.mesh_instances[i].node_index;
        out.mesh_instances[i].node_index = remap[old_root];
    }
    out.tlas_root = remap[d.tlas_root];
    out.nodes.swap(nodes);
    out.tris.swap(tris);
    out.tri_indices.swap(tri_indices);
    out.applied = true;
    return out;
}

} // namespace rayhip_layout
