// scene_update.h -- the host-side planning of an instance / light / environment update (rayhip_scene_update_instances):
// which instance slots are alive and with which boxes, the instance array as the kernels follow it, and the new top-level
// tree moved to its place behind the uploaded nodes.  Shared by librayhip (device builder, lbvh.hip.h) and the host build of
// tests/hostsim (host builder over the same element functions), so the logic is tested against the oracle without a GPU.
//
// Reference: SceneCPU.cpp:1004-1094 (mutators), 928-1015 (RebuildTLAS_nolock).
#pragma once

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "scene_rebuild.h"

namespace rayhip_update {

using rayhip_lbvh::Box;

// per mesh (key: mesh_instance_t::mesh_index) the roots of its bottom-level trees as they were uploaded
struct MeshRef {
    uint32_t node_index, root4;
};
using MeshRefs = std::unordered_map<uint32_t, MeshRef>;

// the meshes in use after a full upload: `nodes` / `instances` are the arrays as uploaded (after refinement and layout)
inline void collect_mesh_refs(const rayhip_bvh2_node *nodes, const uint32_t nodes_count, const rayhip_mesh_instance *instances,
                              const uint32_t instances_count, const uint32_t tlas_root, const uint32_t *blas_root4 /* may be null */,
                              MeshRefs &out) {
    out.clear();
    std::vector<std::pair<uint32_t, uint32_t>> top;
    if (tlas_root != 0xffffffffu && rayhip_rebuild::collect_leaf_ranges(nodes, nodes_count, tlas_root, top)) {
        for (const auto &leaf : top) {
            if (leaf.first < instances_count && instances[leaf.first].node_index < nodes_count) {
                out[instances[leaf.first].mesh_index] = MeshRef{instances[leaf.first].node_index, blas_root4 ? blas_root4[leaf.first] : 0u};
            }
        }
    }
}

struct Plan {
    std::vector<uint32_t> live;                    // instance slots the host's top level names, ascending
    std::vector<Box> boxes;                        // their world-space boxes (as the host stored them)
    std::vector<rayhip_mesh_instance> instances;   // the instance array with device-side tree roots
    std::vector<uint32_t> root4;                   // per slot: root of the 4-wide tree
};

// 0 = ok, 1 = malformed input (why), 2 = needs a full upload (why)
//
// Live instances are the leaves of the host's top level, WITH the boxes the host gave them.  (Not recomputed from the
// transforms: after a RemoveMeshInstance the reference numbers its top-level leaves by the position of an instance among the
// live ones, not by its slot (SceneCPU.cpp:945-951 walks the sparse array, :1000-1008 writes that position into the leaf), so
// a leaf may name another slot than the one its box was made from.  The reference's renderers follow the leaf as written; so
// do we -- taking slot and box as a pair from the host tree keeps every frame identical to theirs.)
inline int plan(const rayhip_scene_desc &d, const MeshRefs &refs, Plan &out, std::string &why) {
    out = Plan();
    if (d.tlas_root != 0xffffffffu) {
        std::vector<std::pair<uint32_t, Box>> leaves;
        if (!rayhip_rebuild::collect_leaf_boxes(d, d.tlas_root, leaves)) {
            why = "top-level tree is malformed";
            return 1;
        }
        std::sort(leaves.begin(), leaves.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        for (const auto &l : leaves) {
            if (l.first >= d.mesh_instances_count) {
                why = "top-level leaf names instance " + std::to_string(l.first) + " of " + std::to_string(d.mesh_instances_count);
                return 1;
            }
            const bool real = l.second.lo[0] <= l.second.hi[0] && l.second.lo[1] <= l.second.hi[1] && l.second.lo[2] <= l.second.hi[2];
            if (!out.live.empty() && out.live.back() == l.first) {
                if (real) {
                    rayhip_lbvh::grow(out.boxes.back(), l.second); // (a lone instance is stored as both children of the root)
                }
            } else {
                out.live.push_back(l.first);
                out.boxes.push_back(real ? l.second : rayhip_lbvh::empty_box());
            }
        }
    }
    out.instances.assign(d.mesh_instances, d.mesh_instances + d.mesh_instances_count);
    out.root4.assign(d.mesh_instances_count, 0);
    for (const uint32_t mi : out.live) {
        const auto it = refs.find(out.instances[mi].mesh_index);
        if (it == refs.end()) {
            why = "instance " + std::to_string(mi) + " uses mesh " + std::to_string(out.instances[mi].mesh_index) + ", which is not on the device";
            return 2;
        }
        out.instances[mi].node_index = it->second.node_index;
        out.root4[mi] = it->second.root4;
    }
    return 0;
}

// input of the linear builder for the top level of `p` (one group, one instance per leaf, the leaf word names the primitive)
inline rayhip_lbvh::Input top_level_input(const Plan &p, const std::vector<uint32_t> &group_of_zeroes) {
    rayhip_lbvh::Input ti;
    ti.prim_box = p.boxes.data(), ti.prim_group = group_of_zeroes.data(), ti.group_centroids = nullptr;
    ti.n_prims = uint32_t(p.boxes.size()), ti.n_groups = 1, ti.leaf_max = 1, ti.leaf_is_primitive = true, ti.roots_are_nodes = true;
    return ti;
}

// the builder's output moved to node slots [base, base + n): inner links shift, leaves name instance slots; returns the root
inline uint32_t relocate_top_level(rayhip_lbvh::Output &tlas, const Plan &p, const uint32_t base) {
    constexpr uint32_t COUNT_BITS = 7u << 29, INDEX_BITS = ~COUNT_BITS;
    for (rayhip_bvh2_node &n : tlas.nodes) {
        for (uint32_t *link : {&n.left_child, &n.right_child}) {
            *link = (*link & COUNT_BITS) == 0 ? *link + base : ((1u << 29) | p.live[*link & INDEX_BITS]);
        }
    }
    return base + tlas.group_root[0];
}

} // namespace rayhip_update
